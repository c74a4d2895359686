"""Thin host-side wrapper over the C ABI: owns one car_ctx, feeds it reference-named weights,
hands it data_ptr()s of PyTorch-ROCm tensors on the current stream.  PyTorch here is plumbing
(device memory + streams); all arithmetic of the path runs in libcontrolar_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib as L
from .config import PathConfig


def _stream_ptr() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.CAR_DT_F32
    if t.dtype == torch.bfloat16:
        return L.CAR_DT_BF16
    raise TypeError(f"unsupported tensor dtype {t.dtype} (float32 or bfloat16)")


def first_valid_position(emb_masks: torch.Tensor) -> int:
    """First attendable prompt position over a batch of LEFT-padded caption masks [B, T] (sample_t2i.py:146-160): min over the rows of the index of the first
    non-zero entry; a row without any valid token counts as T.  This is what `car_sampling.first_valid_hint - 1` carries: the prefill window of car_generate
    starts at the multiple of 16 at or below it.  One reduction — free on a host mask, one sync on a device mask (do it once, outside a timed loop)."""
    B, T = emb_masks.shape
    nz = emb_masks != 0
    first = torch.where(nz.any(dim=1), nz.to(torch.int64).argmax(dim=1), torch.full((B,), T, device=emb_masks.device))
    return int(first.min())


class Engine:
    """One context = one set of weights in one arithmetic mode ('fp32' exact | 'bf16' fast)."""

    def __init__(self, cfg: PathConfig, precision: str = "bf16", device: Optional[torch.device] = None, stream_priority: int = 0,
                 weights_fp8: bool = False, kv_fp8: bool = False, dev: Optional[bool] = None):
        # dev=True: the development build of the library (the CAR_* A/B switches exist only there; default: CONTROLAR_DEV_LIB=1 in the environment)
        self.lib = L.load(dev)
        if not torch.cuda.is_available():
            raise RuntimeError("controlar_amd needs a HIP device (no CPU fallback)")
        dev = torch.device(device if device is not None else "cuda")
        if dev.type != "cuda":
            raise RuntimeError(f"controlar_amd runs on a HIP device only, got {dev}")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        self.cfg = cfg
        self.precision = precision
        self.mode = {"fp32": L.CAR_F32, "none": L.CAR_F32, "bf16": L.CAR_BF16}[precision]
        self.dtype = torch.float32 if self.mode == L.CAR_F32 else torch.bfloat16
        g, v, q = cfg.gpt, cfg.vit, cfg.vq
        cc = L.CarConfig()
        cc.abi_version, cc.mode = L.CAR_ABI_VERSION, self.mode
        cc.dim, cc.n_layer, cc.n_head, cc.ffn_hidden, cc.vocab_size = g.dim, g.n_layer, g.n_head, g.ffn_hidden, g.vocab_size
        cc.cls_token_num, cc.block_size, cc.caption_dim = g.cls_token_num, g.block_size, g.caption_dim
        cc.norm_eps, cc.rope_base = g.norm_eps, g.rope_base
        cc.vit_hidden, cc.vit_layers, cc.vit_heads, cc.vit_mlp = v.hidden, v.layers, v.heads, v.mlp
        cc.vit_patch, cc.vit_pos_grid, cc.vit_ln_eps = v.patch, v.pos_grid, v.ln_eps
        # dinov2_adapter.py:19-23: nearest for canny/seg, bicubic(align_corners=True) otherwise; the c2i ViT adapter
        # (vit_adapter.py:13-15) does not resize: nearest onto the same grid is the identity
        vit16 = getattr(v, "variant", "dinov2") == "vit"
        cc.resize_mode = L.CAR_RESIZE_NEAREST if (vit16 or g.condition_type in ("canny", "seg")) else L.CAR_RESIZE_BICUBIC_AC
        cc.vit_variant = 1 if vit16 else 0
        cc.model_type = 1 if g.model_type == "c2i" else 0
        cc.num_classes = g.num_classes
        cc.stream_priority = int(stream_priority)
        # BASELINE config 5; bf16 mode only.  True / 1 = weight-only e4m3 (bf16 MFMA); 'mfma' / 2 = W8A8 on the fp8 MFMA
        cc.decode_weight_fp8 = 2 if weights_fp8 in ("mfma", 2) else int(bool(weights_fp8))
        # opt-in e4m3 KV cache (bf16 mode only; bf16 KV is the default and the parity path): halves the KV stream, doubles the sequences that fit
        cc.kv_cache_fp8 = int(bool(kv_fp8))
        cc.codebook_size, cc.codebook_dim, cc.z_channels, cc.vq_ch = q.codebook_size, q.codebook_embed_dim, q.z_channels, q.ch
        cc.vq_num_res_blocks, cc.vq_n_mult, cc.gn_eps = q.num_res_blocks, len(q.ch_mult), q.gn_eps
        for i, m in enumerate(q.ch_mult):
            cc.vq_ch_mult[i] = m
        self._cc = cc
        h = C.c_void_p()
        rc = self.lib.car_create(C.byref(h), C.byref(cc))
        if rc != 0:
            raise RuntimeError("car_create: " + self.lib.car_last_error(None).decode())
        self._h = h

    # ------------------------------------------------------------------ errors / lifecycle
    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.lib.car_last_error(self._h).decode()}")

    def close(self):
        """Destroys the context.  A device-side error that no entry could report synchronously (wrong first_valid hint, out-of-range device label) is raised here
        at the latest — after the context is gone, so that close() always releases the memory."""
        if getattr(self, "_h", None):
            h, self._h = self._h, None
            rc, msg = 0, ""
            try:
                rc = self.lib.car_check_errors(h)
                if rc:
                    msg = (self.lib.car_last_error(h) or b"").decode(errors="replace")
            finally:
                self.lib.car_destroy(h)
            if rc:
                raise RuntimeError(f"car_check_errors at close(): {msg}")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Dict[str, torch.Tensor], finalize: bool = False):
        """Accepts the reference's state_dict names (gpt_t2i.Transformer incl. adapter.model.*, VQModel)."""
        for name, t in sd.items():
            if not torch.is_floating_point(t):
                continue
            t = t.detach()
            if t.dtype not in (torch.float32, torch.bfloat16):
                t = t.float()
            t = t.contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            self._check(self.lib.car_load_tensor(self._h, name.encode(), C.c_void_p(t.data_ptr()), shape, t.dim(), _dt(t)),
                        f"car_load_tensor({name})")
        if finalize:
            self.finalize()

    def finalize(self):
        self._check(self.lib.car_finalize_weights(self._h), "car_finalize_weights")

    # ------------------------------------------------------------------ path stages
    def encode_control(self, img: torch.Tensor, want_output: bool = False) -> Optional[torch.Tensor]:
        """model.adapter_mlp(model.adapter(img))  (generate.py:136-138).  img [B,3,H,W] in [-1,1]."""
        assert img.dim() == 4 and img.shape[1] == 3
        img = img.to(self.device)
        if img.dtype not in (torch.float32, torch.bfloat16):
            img = img.float()
        img = img.contiguous()
        B, _, H, W = img.shape
        out = None
        if want_output:
            out = torch.empty(B, (H // 16) * (W // 16), self.cfg.gpt.dim, dtype=self.dtype, device=self.device)
        self._check(self.lib.car_encode_control(self._h, C.c_void_p(img.data_ptr()), _dt(img), B, H, W,
                                                C.c_void_p(out.data_ptr() if out is not None else 0), C.c_void_p(_stream_ptr())),
                    "car_encode_control")
        return out

    def generate(self, cond: torch.Tensor, max_new_tokens: int, emb_masks: Optional[torch.Tensor] = None,
                 cfg_scale: float = 1.0, cfg_interval: int = -1, use_control: bool = True, control_strength: float = 1.0,
                 temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0, sample_logits: bool = False, seed: int = 0,
                 forced_tokens: Optional[torch.Tensor] = None, return_logits: bool = False, first_valid: Optional[int] = None):
        """`first_valid`: a lower bound of the first valid (unpadded) prompt position over the batch, if the caller knows it — car_generate then sizes its prefill
        window without reading the device mask back (no host wait).  A mask that is still on the host supplies it for free, and a caller-supplied value is
        checked against such a mask here.  With a DEVICE mask a value that is too large cannot be seen without the host wait the hint exists to avoid: the
        device raises a sticky error flag, and it surfaces at the NEXT entry into the context, at stats(), check_errors() or close() — the tokens of the
        offending call are invalid."""
        c2i = self.cfg.gpt.model_type == "c2i"
        if c2i and cond.device.type == "cpu" and cond.numel():
            # labels that are still on the host are checked for free; device-resident labels are checked on the device (sticky flag, stats())
            lo, hi = int(cond.min()), int(cond.max())
            if lo < 0 or hi > self.cfg.gpt.num_classes:
                raise RuntimeError(f"car_generate_c2i: class label out of range [0,{self.cfg.gpt.num_classes}] (got {lo}..{hi})")
        cond = cond.to(self.device)
        if c2i:
            cond = cond.to(torch.int64).contiguous().view(-1)          # class labels [B]
            B, T = cond.shape[0], 1
        else:
            if cond.dtype not in (torch.float32, torch.bfloat16):
                cond = cond.float()
            cond = cond.contiguous()
            B, T, cap = cond.shape
            assert T == self.cfg.gpt.cls_token_num and cap == self.cfg.gpt.caption_dim
        mask_t = None
        if emb_masks is not None:
            assert emb_masks.shape[0] == B and emb_masks.shape[-1] == T          # generate.py:185-186
            if emb_masks.device.type == "cpu" and emb_masks.numel():
                true_first = first_valid_position(emb_masks.reshape(B, T))
                if first_valid is not None and int(first_valid) > true_first:
                    raise RuntimeError(f"generate: first_valid={int(first_valid)} is beyond the first valid prompt position of the batch ({true_first}): valid rows would be dropped")
                if first_valid is None:
                    first_valid = true_first
            mask_t = emb_masks.to(device=self.device, dtype=torch.int64).contiguous()
        sp = L.CarSampling()
        sp.first_valid_hint = 0 if (first_valid is None or emb_masks is None) else max(0, min(int(first_valid), T)) + 1
        sp.cfg_scale, sp.cfg_interval, sp.temperature, sp.top_k = float(cfg_scale), int(cfg_interval), float(temperature), int(top_k or 0)
        sp.top_p, sp.sample_logits, sp.seed, sp.control_strength = float(top_p), int(bool(sample_logits)), int(seed), float(control_strength)
        out = torch.empty(B, max_new_tokens, dtype=torch.int32, device=self.device)
        forced = None
        if forced_tokens is not None:
            forced = forced_tokens.to(device=self.device, dtype=torch.int32).contiguous()
        logits = torch.empty(B, max_new_tokens, self.cfg.gpt.vocab_size, dtype=torch.float32, device=self.device) if return_logits else None
        if c2i:
            self._check(self.lib.car_generate_c2i(self._h, C.c_void_p(cond.data_ptr()), B, int(max_new_tokens), int(bool(use_control)),
                                                  C.byref(sp), C.c_void_p(out.data_ptr()),
                                                  C.c_void_p(forced.data_ptr() if forced is not None else 0),
                                                  C.c_void_p(logits.data_ptr() if logits is not None else 0), C.c_void_p(_stream_ptr())),
                        "car_generate_c2i")
        else:
            self._check(self.lib.car_generate(self._h, C.c_void_p(cond.data_ptr()), _dt(cond),
                                              C.c_void_p(mask_t.data_ptr() if mask_t is not None else 0), B, int(max_new_tokens),
                                              int(bool(use_control)), C.byref(sp), C.c_void_p(out.data_ptr()),
                                              C.c_void_p(forced.data_ptr() if forced is not None else 0),
                                              C.c_void_p(logits.data_ptr() if logits is not None else 0), C.c_void_p(_stream_ptr())),
                        "car_generate")
        return (out, logits) if return_logits else out

    def sample(self, logits: torch.Tensor, cfg_scale: float = 1.0, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0,
               sample_logits: bool = True, seed: int = 0, step: int = 0) -> torch.Tensor:
        """generate.py:59-74 sample() on caller-provided fp32 logits [rows, V] (rows = 2B under CFG) -> int32 [B]."""
        logits = logits.to(device=self.device, dtype=torch.float32).contiguous()
        rows, V = logits.shape
        B = rows // 2 if cfg_scale > 1.0 else rows
        sp = L.CarSampling()
        sp.cfg_scale, sp.cfg_interval, sp.temperature, sp.top_k = float(cfg_scale), -1, float(temperature), int(top_k or 0)
        sp.top_p, sp.sample_logits, sp.seed, sp.control_strength = float(top_p), int(bool(sample_logits)), int(seed), 1.0
        out = torch.empty(B, dtype=torch.int32, device=self.device)
        self._check(self.lib.car_sample_logits(self._h, C.c_void_p(logits.data_ptr()), B, V, C.byref(sp), int(step),
                                               C.c_void_p(out.data_ptr()), C.c_void_p(_stream_ptr())), "car_sample_logits")
        return out

    def vq_decode(self, tokens: torch.Tensor, h: int, w: int) -> torch.Tensor:
        """VQModel.decode_code(tokens, [B,C,h,w]) -> fp32 [B,3,16h,16w]  (vq_model.py:53-56)."""
        tokens = tokens.to(device=self.device, dtype=torch.int32).contiguous().view(-1, h * w)
        B = tokens.shape[0]
        up = 2 ** (len(self.cfg.vq.ch_mult) - 1)
        out = torch.empty(B, 3, h * up, w * up, dtype=torch.float32, device=self.device)
        self._check(self.lib.car_vq_decode(self._h, C.c_void_p(tokens.data_ptr()), B, h, w, C.c_void_p(out.data_ptr()),
                                           C.c_void_p(_stream_ptr())), "car_vq_decode")
        return out

    def vq_encode(self, img: torch.Tensor) -> torch.Tensor:
        """VQModel.encode(img) -> min_encoding_indices int32 [B, (H/16)(W/16)]  (vq_model.py:41-46)."""
        img = img.to(device=self.device, dtype=torch.float32).contiguous()
        B, _, H, W = img.shape
        down = 2 ** (len(self.cfg.vq.ch_mult) - 1)
        out = torch.empty(B, (H // down) * (W // down), dtype=torch.int32, device=self.device)
        self._check(self.lib.car_vq_encode(self._h, C.c_void_p(img.data_ptr()), B, H, W, C.c_void_p(out.data_ptr()), C.c_void_p(_stream_ptr())),
                    "car_vq_encode")
        return out

    def canny(self, img: torch.Tensor, low_threshold: float = 100, high_threshold: float = 200, want_control: bool = False):
        """cv2.Canny(img, low, high) of condition/canny.py:6-14 on the GPU.  img uint8 [B,H,W,3] (or [H,W,3]) -> uint8 [B,H,W] in
        {0,255}; with want_control also the control tensor [B,3,H,W] = 2*(edges/255-0.5) (sample_t2i.py:125,141)."""
        single = img.dim() == 3
        x = (img[None] if single else img).to(device=self.device, dtype=torch.uint8).contiguous()
        B, H, W, ch = x.shape
        assert ch == 3, "expected an RGB image [.., H, W, 3]"
        edges = torch.empty(B, H, W, dtype=torch.uint8, device=self.device)
        ctrl = torch.empty(B, 3, H, W, dtype=self.dtype, device=self.device) if want_control else None
        self._check(self.lib.car_canny(self._h, C.c_void_p(x.data_ptr()), B, H, W, float(low_threshold), float(high_threshold),
                                       C.c_void_p(edges.data_ptr()), C.c_void_p(ctrl.data_ptr() if ctrl is not None else 0), C.c_void_p(_stream_ptr())),
                    "car_canny")
        edges = edges[0] if single else edges
        return (edges, ctrl) if want_control else edges

    # ------------------------------------------------------------------ caption encoder (language/t5.py)
    def t5_configure(self, t5cfg):
        tc = L.CarT5Config()
        tc.vocab_size, tc.d_model, tc.d_kv, tc.num_heads = t5cfg.vocab_size, t5cfg.d_model, t5cfg.d_kv, t5cfg.num_heads
        tc.d_ff, tc.num_layers = t5cfg.d_ff, t5cfg.num_layers
        tc.rel_buckets, tc.rel_max_distance = t5cfg.relative_attention_num_buckets, t5cfg.relative_attention_max_distance
        tc.ln_eps = t5cfg.layer_norm_epsilon
        self._check(self.lib.car_t5_configure(self._h, C.byref(tc)), "car_t5_configure")
        self.t5cfg = t5cfg

    def load_t5_state_dict(self, sd: Dict[str, torch.Tensor], finalize: bool = True):
        """HF T5EncoderModel / T5ForConditionalGeneration state-dict names; the C ABI namespaces them under 't5.'."""
        self.load_state_dict({"t5." + k: v for k, v in sd.items()}, finalize=finalize)

    def t5_encode(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """T5EncoderModel(input_ids, attention_mask)['last_hidden_state'] (language/t5.py:194-199) -> [B,T,d_model] in self.dtype."""
        assert input_ids.dim() == 2
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        mk = None if attention_mask is None else attention_mask.to(device=self.device, dtype=torch.int64).contiguous()
        B, T = ids.shape
        out = torch.empty(B, T, self.t5cfg.d_model, dtype=self.dtype, device=self.device)
        self._check(self.lib.car_t5_encode(self._h, C.c_void_p(ids.data_ptr()), C.c_void_p(mk.data_ptr() if mk is not None else 0), B, T,
                                           C.c_void_p(out.data_ptr()), C.c_void_p(_stream_ptr())), "car_t5_encode")
        return out

    def check_errors(self):
        """Waits for the work enqueued on this context and raises if device code reported an error no entry could fail on synchronously
        (today: a c2i class label outside [0, num_classes] in a device-resident label tensor — include/controlar_hip.h, car_generate_c2i)."""
        self._check(self.lib.car_check_errors(self._h), "car_check_errors")

    def stats(self) -> dict:
        s = L.CarStats()
        self._check(self.lib.car_get_stats(self._h, C.byref(s)), "car_get_stats")
        self._check(self.lib.car_check_errors(self._h), "car_check_errors")          # car_get_stats waited for the stream: a sticky device error is visible now
        return dict(decode_ms=s.decode_ms, prefill_ms=s.prefill_ms, decode_steps=s.decode_steps,
                    decode_algo_bytes=s.decode_algo_bytes, decode_kernels_per_step=s.decode_kernels_per_step,
                    graph_used=bool(s.graph_used), dev_knobs_active=int(s.dev_knobs_active))

    def control_tokens(self, k: int, b: int, n_tok: int) -> torch.Tensor:
        n = b * n_tok * self.cfg.gpt.dim
        buf = torch.empty(n, dtype=torch.float32)
        self._check(self.lib.car_debug_control_tokens(self._h, k, C.c_void_p(buf.data_ptr()), n), "car_debug_control_tokens")
        return buf.view(b, n_tok, self.cfg.gpt.dim)
