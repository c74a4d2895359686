#!/usr/bin/env python3
"""Mint golden vectors from the UNMODIFIED reference modules imported from /root/reference.

Runs only in the build container (the GPU box has no /root/reference).  The reference has
no golden vectors of its own for this path ("parity unpinned"), so the
pin is: reference code on CPU + the deterministic synthetic weights/inputs of
controlar_amd/synth.py -> committed .npz fixtures.  The CPU oracle (oracle/) is checked
against these fixtures by tests/test_oracle_golden.py; the HIP path by the -m gpu tests.

Shims: AutoModel.from_pretrained is patched to build a random-init
Dinov2Model of the configured size (no checkpoints offline); weights are then overwritten
with the synthetic state dict; greedy decode is requested with sample_logits=False.

Usage:  python tests/golden/make_golden.py [case ...]     (default: all small cases)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("CONTROLAR_REFERENCE", "/root/reference")
OUT = os.environ.get("CONTROLAR_GOLDEN_OUT", HERE)       # where fixtures are written (tests regenerate into a temp dir)
sys.path.insert(0, REF)

import transformers  # noqa: E402
from transformers import Dinov2Config, Dinov2Model  # noqa: E402

from controlar_amd import config as C  # noqa: E402
from controlar_amd import synth  # noqa: E402

_VIT_CFG = {}


def _fake_from_pretrained(name, *a, **k):
    v = _VIT_CFG["cfg"]
    return Dinov2Model(Dinov2Config(hidden_size=v.hidden, num_attention_heads=v.heads,
                                    num_hidden_layers=v.layers, mlp_ratio=v.mlp_ratio,
                                    image_size=v.image_size, patch_size=v.patch))


transformers.AutoModel.from_pretrained = staticmethod(_fake_from_pretrained)

from autoregressive.models import gpt_t2i as ref_gpt  # noqa: E402
from autoregressive.models import generate as ref_gen  # noqa: E402
from tokenizer.tokenizer_image import vq_model as ref_vq  # noqa: E402


def build_ref_gpt(cfg: C.PathConfig, sd, dtype):
    _VIT_CFG["cfg"] = cfg.vit
    g = cfg.gpt
    m = ref_gpt.Transformer(ref_gpt.ModelArgs(
        dim=g.dim, n_layer=g.n_layer, n_head=g.n_head, vocab_size=g.vocab_size,
        block_size=g.block_size, cls_token_num=g.cls_token_num, caption_dim=g.caption_dim,
        model_type="t2i", condition_type=g.condition_type, adapter_size=g.adapter_size,
        multiple_of=g.multiple_of))
    if cfg.vit.hidden not in (384, 768):     # tiny test encoder: adapter_mlp input follows it
        m.adapter_mlp = ref_gpt.MLP(cfg.vit.hidden, g.dim, g.dim)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    allowed = ("condition_embeddings.weight", "condition_mlp.uncond_embedding")
    assert all(k in allowed for k in missing), missing
    assert not unexpected, unexpected
    return m.to(dtype).eval()


def build_ref_vq(cfg: C.VQConfig, sd):
    m = ref_vq.VQModel(ref_vq.ModelArgs(codebook_size=cfg.codebook_size,
                                        codebook_embed_dim=cfg.codebook_embed_dim,
                                        encoder_ch_mult=list(cfg.ch_mult), decoder_ch_mult=list(cfg.ch_mult),      # VQ_16 / VQ_8 set both (vq_model.py:415-420)
                                        z_channels=cfg.z_channels))
    if cfg.ch != 128:
        # tiny test architecture: the reference's ModelArgs has no `ch` knob, so both halves are rebuilt at the test width
        # (synth.vq_state_dict emits encoder.* / quant_conv.* since car_vq_encode exists; strict=False only forgives the buffers)
        m.decoder = ref_vq.Decoder(ch_mult=list(cfg.ch_mult), z_channels=cfg.z_channels, ch=cfg.ch,
                                   num_res_blocks=cfg.num_res_blocks)
        m.encoder = ref_vq.Encoder(ch_mult=list(cfg.ch_mult), z_channels=cfg.z_channels, ch=cfg.ch,
                                   num_res_blocks=cfg.num_res_blocks)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert all(k.startswith(("encoder.", "quant_conv.", "quantize.codebook_used")) for k in missing), missing
    assert not unexpected
    return m.eval()


class _Tap:
    """Records the logits handed to generate.sample() (after the CFG mix, before temperature)."""

    def __init__(self):
        self.rows = []
        self._orig = ref_gen.sample

    def __enter__(self):
        def tapped(logits, **kw):
            self.rows.append(logits[:, -1, :].detach().float().clone())
            return self._orig(logits, **kw)
        ref_gen.sample = tapped
        return self

    def __exit__(self, *a):
        ref_gen.sample = self._orig


def run_case(name, cfg: C.PathConfig, B, H, W, cfg_scale, control_strength=1.0, dtype=torch.float32,
             control="canny", seed=0, threads=8, vq=True, keep_logits="all", cfg_interval=-1, lengths=None, no_mask=False):
    torch.set_num_threads(threads)
    t0 = time.time()
    gsd, vsd = synth.path_state_dicts(cfg, seed=seed)
    model = build_ref_gpt(cfg, gsd, dtype)
    g = cfg.gpt
    img = (synth.canny_like_control(B, H, W) if control == "canny" else synth.smooth_control(B, H, W))
    emb, mask = (synth.text_embeddings(B, g.cls_token_num, g.caption_dim) if lengths is None
                 else synth.text_embeddings_with_lengths(lengths, g.cls_token_num, g.caption_dim))
    if no_mask:
        mask = None
    n_new = (H // 16) * (W // 16)
    with torch.no_grad():
        ad = model.adapter(img.to(dtype))
        adm = model.adapter_mlp(ad)
    with _Tap() as tap, torch.no_grad():
        toks = ref_gen.generate(model, emb.to(dtype), n_new, mask, cfg_scale=cfg_scale,
                                cfg_interval=cfg_interval, condition=img.to(dtype),
                                control_strength=control_strength, temperature=1.0, top_k=0, top_p=1.0,
                                sample_logits=False)
    logits = torch.stack(tap.rows, dim=1)          # [B, n_new, V]
    top2 = logits.topk(2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1])
    out = dict(tokens=toks.numpy().astype(np.int32), margin=margin.numpy().astype(np.float32),
               adapter_out=ad.float().numpy()[:, ::7, ::5].copy(),
               adapter_mlp_out=adm.float().numpy()[:, ::7, ::5].copy(),
               ctrl0=model.condition_token[0].float().numpy()[:B, ::7, ::5].copy(),
               ctrl2=model.condition_token[2].float().numpy()[:B, ::7, ::5].copy(),
               meta=np.array([B, H, W, seed, threads], dtype=np.int64),
               cfg_scale=np.float32(cfg_scale), control_strength=np.float32(control_strength))
    if keep_logits == "all":
        out["logits"] = logits.numpy().astype(np.float32)
    else:   # strided subset keeps big fixtures small
        out["logits_steps"] = np.arange(0, n_new, keep_logits, dtype=np.int64)
        out["logits"] = logits[:, ::keep_logits, ::4].numpy().astype(np.float32)
    if vq:
        vqm = build_ref_vq(cfg.vq, vsd)
        with torch.no_grad():
            px = vqm.decode_code(toks, [B, cfg.vq.codebook_embed_dim, H // 16, W // 16])
        out["pixels"] = px.numpy().astype(np.float32)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {time.time()-t0:.1f}s distinct={len(np.unique(out['tokens']))} "
          f"min_margin={margin.min():.4g} med_margin={margin.median():.3g} -> {os.path.getsize(path)/1e3:.0f} KB")


def case_vq16_real(name="vq16_real_8x8"):
    """The real VQ-16 decoder architecture (ch=128, z=256, 16384x8 codebook) on an 8x8 token grid."""
    cfg = C.VQConfig()
    sd = synth.vq_state_dict(cfg, seed=2)
    m = build_ref_vq(cfg, sd)
    g = torch.Generator().manual_seed(7)
    toks = torch.randint(0, cfg.codebook_size, (2, 64), generator=g, dtype=torch.int32)
    with torch.no_grad():
        px = m.decode_code(toks, [2, 8, 8, 8])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), tokens=toks.numpy(), pixels=px.numpy())
    print(name, px.shape, float(px.abs().max()), float(px.abs().mean()))


def case_vq8_real(name="vq8_real_8x8"):
    """The VQ-8 variant (vq_model.py:415-417: ch_mult (1, 2, 2, 4), three upsampling levels) at its real width on an 8x8 token grid -> 64x64 pixels."""
    cfg = C.VQConfig(ch_mult=(1, 2, 2, 4))
    sd = synth.vq_state_dict(cfg, seed=3)
    m = build_ref_vq(cfg, sd)
    g = torch.Generator().manual_seed(8)
    toks = torch.randint(0, cfg.codebook_size, (2, 64), generator=g, dtype=torch.int32)
    with torch.no_grad():
        px = m.decode_code(toks, [2, 8, 8, 8])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), tokens=toks.numpy(), pixels=px.numpy())
    print(name, px.shape, float(px.abs().max()), float(px.abs().mean()))


def case_vq16_real_512(name="vq16_real_32x32"):
    """The real VQ-16 decoder at the bench's size: 32x32 tokens -> 512x512 pixels (vq_model.py:53-56,174-195).  The fixture
    keeps a strided pixel lattice plus two dense patches (image corner and centre) so that tiling/chunking errors show."""
    cfg = C.VQConfig()
    sd = synth.vq_state_dict(cfg, seed=2)
    m = build_ref_vq(cfg, sd)
    g = torch.Generator().manual_seed(9)
    toks = torch.randint(0, cfg.codebook_size, (2, 1024), generator=g, dtype=torch.int32)
    with torch.no_grad():
        px = m.decode_code(toks, [2, 8, 32, 32])
    px = px.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), tokens=toks.numpy(), lattice=px[:, :, ::8, ::8].copy(),
                        corner=px[:, :, :24, :24].copy(), centre=px[:, :, 244:268, 244:268].copy(),
                        stats=np.array([np.abs(px).max(), np.abs(px).mean()], dtype=np.float32))
    print(name, px.shape, float(np.abs(px).max()), float(np.abs(px).mean()))


CASES = {
    # tiny graph, canny (nearest resize), cfg=1
    "tiny_canny_cfg1": lambda: run_case("tiny_canny_cfg1", C.tiny_t2i(64, "canny"), 2, 128, 128, 1.0),
    # tiny graph, depth (bicubic align_corners resize), cfg=4, control_strength 0.6
    "tiny_depth_cfg4": lambda: run_case("tiny_depth_cfg4", C.tiny_t2i(64, "depth"), 2, 128, 128, 4.0,
                                        control_strength=0.6, control="smooth"),
    # non-square MR: rope grid 12 (block 144), token grid 12 rows x 8 cols (linear-index quirk)
    "tiny_mr_192x128": lambda: run_case("tiny_mr_192x128", C.tiny_t2i(144, "canny"), 1, 192, 128, 1.5),
    "tiny_mr_128x192": lambda: run_case("tiny_mr_128x192", C.tiny_t2i(144, "canny"), 1, 128, 192, 1.5),
    # cfg_interval switch
    "tiny_cfg_interval": lambda: run_case("tiny_cfg_interval", C.tiny_t2i(64, "canny"), 2, 128, 128, 3.0,
                                          cfg_interval=20, vq=False),
    # the reference's own bf16 default, for tolerance calibration of the fast mode
    "tiny_canny_cfg1_bf16": lambda: run_case("tiny_canny_cfg1_bf16", C.tiny_t2i(64, "canny"), 2, 128, 128, 1.0,
                                             dtype=torch.bfloat16, vq=False),
    # 'base'-shaped encoder (heads x 64), bicubic resize, CFG + control_strength
    "tiny_hed_base_cfg1p5": lambda: run_case("tiny_hed_base_cfg1p5", C.tiny_t2i_base(64, "hed"), 2, 128, 128, 1.5,
                                             control_strength=0.6, control="smooth", vq=False),
    # ragged / extreme text-pad masks: one real token, no padding at all, typical; and emb_masks=None
    "tiny_mask_edges": lambda: run_case("tiny_mask_edges", C.tiny_t2i(64, "canny"), 3, 128, 128, 1.5, vq=False, lengths=[1, 120, 40]),
    "tiny_no_mask": lambda: run_case("tiny_no_mask", C.tiny_t2i(64, "canny"), 2, 128, 128, 1.0, vq=False, no_mask=True),
    "vq16_real_8x8": case_vq16_real,
    "vq8_real_8x8": case_vq8_real,
    "vq16_real_32x32": case_vq16_real_512,
    # GPT-B sized, 256 tokens
    "b_canny_256_cfg4": lambda: run_case("b_canny_256_cfg4", C.b_t2i(256, "small", "canny"), 1, 256, 256, 4.0,
                                         vq=False, keep_logits=16),
    # the REAL DINOv2-base control encoder (768 / 12 heads / 12 layers) with the bicubic resize, CFG and control_strength, GPT-B sized (BASELINE
    # configs 3 / 5 use this encoder under GPT-XL; the GPU test of config 3 checks the adapter against the oracle, which this case pins)
    "b_depth_base_256_cfg1p5": lambda: run_case("b_depth_base_256_cfg1p5", C.b_t2i(256, "base", "depth"), 1, 256, 256, 1.5,
                                                control_strength=0.6, control="smooth", vq=False, keep_logits=16),
    # BASELINE config 4's geometry at full size (sample_t2i_MR.py:73-78: 768x512 -> 48 x 32 tokens, rope grid 48, block_size 2304, S_max 1656; the real
    # DINOv2-small at 672x448 with interpolated position embeddings), GPT-B sized so that the CPU reference finishes in a minute
    "b_mr_768x512_cfg4": lambda: run_case("b_mr_768x512_cfg4", C.b_t2i(2304, "small", "canny"), 1, 768, 512, 4.0, vq=False, keep_logits=64),
    # BASELINE config 2 at full size (GPT-XL, 512x512, 1024 tokens), B=1; ~2-4 min of CPU
    "xl_canny_512_cfg1": lambda: run_case("xl_canny_512_cfg1", C.xl_t2i(1024, "small", "canny"), 1, 512, 512, 1.0,
                                          vq=False, keep_logits=64),
    # BASELINE config 2 with the reference's default guidance (sample_t2i.py:207: cfg_scale 4; 2B = 2 rows per step), GPT-XL, 512x512, B=1; ~5 min of CPU
    "xl_canny_512_cfg4": lambda: run_case("xl_canny_512_cfg4", C.xl_t2i(1024, "small", "canny"), 1, 512, 512, 4.0,
                                          vq=False, keep_logits=64),
    # BASELINE config 4 at full size AND full width (sample_t2i_MR.py:73-78,182-185: GPT-XL, 768x512 -> 48 x 32 = 1536 tokens on a rope grid of 48,
    # block_size 2304, S_max 1656, DINOv2-small at 672x448), cfg 4; ~10 min of CPU
    "xl_mr_768x512_cfg4": lambda: run_case("xl_mr_768x512_cfg4", C.xl_t2i(2304, "small", "canny"), 1, 768, 512, 4.0,
                                           vq=False, keep_logits=64),
}
DEFAULT = ["tiny_canny_cfg1", "tiny_depth_cfg4", "tiny_mr_192x128", "tiny_mr_128x192", "tiny_cfg_interval",
           "tiny_canny_cfg1_bf16", "vq16_real_8x8"]



# --------------------------------------------------------------------------------------------------
# Calibration of the fast-mode tolerance at full size: the reference's OWN bf16 path, teacher-forced on
# the fp32 golden tokens (SURVEY.md Appendix G protocol), against the fp32 golden logits.
def teacher_forced_reference(model, cond, emb_masks, condition, forced, dtype):
    """generate.py:134-204 unrolled with decode_one_token fed the forced tokens (cfg_scale = 1)."""
    T = cond.shape[1]
    n_new = forced.shape[1]
    with torch.no_grad():
        cnd = model.adapter_mlp(model.adapter(condition.to(dtype)))
        model.setup_caches(max_batch_size=cond.shape[0], max_seq_length=T + n_new, dtype=model.tok_embeddings.weight.dtype)
        model.causal_mask[:, :, :T] = model.causal_mask[:, :, :T] * emb_masks.unsqueeze(1)
        eye = torch.eye(model.causal_mask.size(1), model.causal_mask.size(2))
        model.causal_mask[:] = model.causal_mask * (1 - eye) + eye
        rows = []
        with _Tap() as tap:
            ref_gen.prefill(model, cond.to(dtype), torch.arange(0, T), 1.0, cnd, 1, temperature=1.0, top_k=0, top_p=1.0, sample_logits=False)
            input_pos = torch.tensor([T], dtype=torch.int)
            for i in range(n_new - 1):
                ref_gen.decode_one_token(model, forced[:, i:i + 1].long(), input_pos, 1.0, True, cnd,
                                         temperature=1.0, top_k=0, top_p=1.0, sample_logits=False)
                input_pos += 1
            rows = tap.rows
    return torch.stack(rows, dim=1)


def case_bf16_calibration(base="xl_canny_512_cfg1", mk=lambda: C.xl_t2i(1024, "small", "canny"), threads=8):
    torch.set_num_threads(threads)
    gold = np.load(os.path.join(HERE, base + ".npz"))
    B, H, W, seed, _ = [int(x) for x in gold["meta"]]
    cfg = mk()
    gsd, _ = synth.path_state_dicts(cfg, seed=seed)
    model = build_ref_gpt(cfg, gsd, torch.bfloat16)
    img = synth.canny_like_control(B, H, W)
    emb, mask = synth.text_embeddings(B, cfg.gpt.cls_token_num, cfg.gpt.caption_dim)
    t0 = time.time()
    lg = teacher_forced_reference(model, emb, mask, img, torch.from_numpy(gold["tokens"]), torch.bfloat16)
    steps = gold["logits_steps"]
    sub = lg[:, steps][:, :, ::4].numpy().astype(np.float32)
    d = np.abs(sub - gold["logits"])
    agree = (lg.argmax(-1).numpy() == gold["tokens"])
    np.savez_compressed(os.path.join(OUT, base + "_refbf16.npz"), ref_bf16_max=np.float32(d.max()), ref_bf16_mean=np.float32(d.mean()),
                        ref_bf16_agree=np.float32(agree.mean()), threads=np.int64(threads))
    print(f"{base}: reference bf16 vs fp32 teacher-forced: max|d|={d.max():.4f} mean|d|={d.mean():.4f} argmax agree={agree.mean():.4f} ({time.time()-t0:.0f}s)")


CASES["xl_bf16_calibration"] = case_bf16_calibration
CASES["b_bf16_calibration"] = lambda: case_bf16_calibration("b_canny_256_cfg4_c1", lambda: C.b_t2i(256, "small", "canny"))




# --------------------------------------------------------------------------------------------------
# c2i (BASELINE config 1): reference autoregressive/models/gpt.py with the HF ViT-S/16 adapter.  gpt.py:427 hard-casts
# the control tokens to bf16, so the reference only runs this path in bf16 (fp32 raises a dtype error): the golden is
# the reference's bf16 output at a pinned thread count and pins the oracle within bf16 tolerance (teacher-forced).
def build_ref_c2i(cfg, sd, dtype):
    from transformers import ViTConfig, ViTModel
    from autoregressive.models import gpt as ref_c2i
    v = cfg.vit
    transformers.AutoModel.from_pretrained = staticmethod(lambda name, *a, **k: ViTModel(ViTConfig(
        hidden_size=v.hidden, num_hidden_layers=v.layers, num_attention_heads=v.heads, intermediate_size=v.mlp,
        image_size=v.image_size, patch_size=v.patch)))
    g = cfg.gpt
    m = ref_c2i.Transformer(ref_c2i.ModelArgs(dim=g.dim, n_layer=g.n_layer, n_head=g.n_head, vocab_size=g.vocab_size,
                                              block_size=g.block_size, num_classes=g.num_classes, cls_token_num=1, model_type="c2i",
                                              condition_token_num=0, image_size=int(g.block_size ** 0.5) * 16, multiple_of=g.multiple_of))
    if v.hidden != 384:
        m.adapter_mlp = ref_c2i.MLP(v.hidden, g.dim, g.dim)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    allowed = ("condition_embeddings.weight", "condition_mlp.uncond_embedding", "condition_norm.weight")
    assert all(k in allowed for k in missing), missing
    assert not unexpected, unexpected
    transformers.AutoModel.from_pretrained = staticmethod(_fake_from_pretrained)
    return m.to(dtype).eval()


def run_c2i(name, cfg, imgs_u8, labels, cfg_scale, threads=8, seed=0):
    torch.set_num_threads(threads)
    gsd, _ = synth.path_state_dicts(cfg, seed=seed)
    model = build_ref_c2i(cfg, gsd, torch.bfloat16)
    x = torch.from_numpy(imgs_u8).float() / 255          # sample_c2i.py:96-99,106
    x = (2 * (x - 0.5))[:, None].repeat(1, 3, 1, 1)
    n_new = (x.shape[2] // 16) * (x.shape[3] // 16)
    lab = torch.from_numpy(labels).long()
    with _Tap() as tap, torch.no_grad():
        toks = ref_gen.generate(model, lab, n_new, condition=x.to(torch.bfloat16), condition_null=None, condition_token_nums=0,
                                cfg_scale=cfg_scale, cfg_interval=-1, temperature=1.0, top_k=0, top_p=1.0, sample_logits=False)
    logits = torch.stack(tap.rows, dim=1)
    top2 = logits.topk(2, dim=-1).values
    st = 1 if logits.shape[1] * logits.shape[2] <= 64 * 1024 else 8      # big cases keep every 8th step / 4th vocab entry
    np.savez_compressed(os.path.join(OUT, name + ".npz"), tokens=toks.numpy().astype(np.int32),
                        logits=logits.numpy()[:, ::st, ::(2 if st == 1 else 4)].astype(np.float16), logits_step_stride=np.int64(st),
                        margin=(top2[..., 0] - top2[..., 1]).numpy().astype(np.float32), images_u8=imgs_u8, labels=labels,
                        cfg_scale=np.float32(cfg_scale), meta=np.array([x.shape[0], x.shape[2], x.shape[3], seed, threads], dtype=np.int64))
    print(f"{name}: distinct={len(np.unique(toks.numpy()))} ref bf16 -> {os.path.getsize(os.path.join(OUT, name + '.npz'))/1e3:.0f} KB")


def case_c2i_tiny():
    g = torch.Generator().manual_seed(11)
    imgs = ((torch.rand(2, 128, 128, generator=g) > 0.92).numpy() * 255).astype(np.uint8)
    run_c2i("tiny_c2i_cfg1", C.tiny_c2i(64), imgs, np.array([3, 7], dtype=np.int64), 1.0)


def case_c2i_b():
    """The reference's own fixtures: condition/example/c2i/canny/{650,2312,15000,48850}.png + .npy labels (sample_c2i.py:86-99)."""
    from PIL import Image
    ids = [650, 2312, 15000, 48850]
    imgs = np.stack([np.array(Image.open(os.path.join(REF, f"condition/example/c2i/canny/{i}.png"))) for i in ids]).astype(np.uint8)
    labels = np.array([int(np.load(os.path.join(REF, f"condition/example/c2i/canny/{i}.npy"))[0]) for i in ids], dtype=np.int64)
    run_c2i("b_c2i_canny_fixtures_cfg1", C.b_c2i(256), imgs, labels, 1.0)


def case_c2i_l_depth():
    """GPT-L c2i on the reference's depth fixtures: condition/example/c2i/depth/{101,4351,10601,48901}.png + .npy labels (sample_c2i.py:92-99)."""
    from PIL import Image
    ids = [101, 4351, 10601, 48901]
    imgs = np.stack([np.array(Image.open(os.path.join(REF, f"condition/example/c2i/depth/{i}.png"))) for i in ids]).astype(np.uint8)
    labels = np.array([int(np.load(os.path.join(REF, f"condition/example/c2i/depth/{i}.npy"))[0]) for i in ids], dtype=np.int64)
    run_c2i("l_c2i_depth_fixtures_cfg1", C.l_c2i(256), imgs, labels, 1.0)


CASES["tiny_c2i_cfg1"] = case_c2i_tiny
CASES["l_c2i_depth_fixtures_cfg1"] = case_c2i_l_depth
CASES["b_c2i_canny_fixtures_cfg1"] = case_c2i_b




def case_vq_encode(name, cfg, H, W, seed=2):
    """VQModel.encode (Encoder + quant_conv + quantizer arg-min) of the reference on a smooth synthetic image."""
    sd = synth.vq_state_dict(cfg, seed=seed)
    m = ref_vq.VQModel(ref_vq.ModelArgs(codebook_size=cfg.codebook_size, codebook_embed_dim=cfg.codebook_embed_dim,
                                        encoder_ch_mult=list(cfg.ch_mult), decoder_ch_mult=list(cfg.ch_mult), z_channels=cfg.z_channels))
    if cfg.ch != 128:
        m.decoder = ref_vq.Decoder(ch_mult=list(cfg.ch_mult), z_channels=cfg.z_channels, ch=cfg.ch, num_res_blocks=cfg.num_res_blocks)
        m.encoder = ref_vq.Encoder(ch_mult=list(cfg.ch_mult), z_channels=cfg.z_channels, ch=cfg.ch, num_res_blocks=cfg.num_res_blocks)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert all(k.startswith("quantize.codebook_used") for k in missing), missing
    assert not unexpected, unexpected
    m.eval()
    img = synth.smooth_control(2, H, W, seed=77) + 0.1 * synth.canny_like_control(2, H, W, seed=78)
    with torch.no_grad():
        quant, _, info = m.encode(img)
    idx = info[2].view(2, -1)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), tokens=idx.numpy().astype(np.int32), meta=np.array([2, H, W, seed], dtype=np.int64))
    print(name, idx.shape, "distinct", len(np.unique(idx.numpy())))


def case_sampler(name="sampler_v16384"):
    """The reference's own sample() internals (generate.py:17-74) at its real defaults: V = 16384, top_k = 2000 (sample_t2i.py:209),
    with and without nucleus filtering and a temperature.  Pins the oracle's top_k_top_p_filtering and gives the GPU sampler
    (ops.hip sample_stochastic_kernel) its target distribution."""
    g = torch.Generator().manual_seed(5)
    row = (torch.randn(1, 16384, generator=g) * 3.0).float()
    out = {"logits": row.numpy()[0].copy()}
    settings = [(2000, 1.0, 1.0), (2000, 0.9, 1.0), (0, 0.9, 0.7), (50, 1.0, 1.3)]
    out["settings"] = np.array(settings, dtype=np.float64)
    for i, (k, p, t) in enumerate(settings):
        x = row.clone()[:, None, :]                                   # sample() takes [B, T, V] and uses the last position
        idx, probs = ref_gen.sample(x, temperature=t, top_k=int(k), top_p=p, sample_logits=False)
        out[f"probs_{i}"] = probs.numpy()[0].astype(np.float32)
        out[f"greedy_{i}"] = np.int64(idx.item())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, [int((out[f"probs_{i}"] > 0).sum()) for i in range(len(settings))])


CASES["sampler_v16384"] = case_sampler
CASES["vq_encode_tiny"] = lambda: case_vq_encode("vq_encode_tiny", C.tiny_t2i().vq, 128, 128)
CASES["vq_encode_vq16_64x64"] = lambda: case_vq_encode("vq_encode_vq16_64x64", C.VQConfig(), 64, 64)


# ----------------------------------------------------------------------------------------- caption encoder (language/t5.py)
def build_hf_t5(cfg, sd, dtype):
    """HF T5EncoderModel exactly as T5Embedder.__init__ gets it from from_pretrained (language/t5.py:78), random-init of the
    configured size, weights overwritten with the synthetic state dict, eager attention (the explicit matmul/softmax path)."""
    from transformers import T5Config, T5EncoderModel
    hc = T5Config(vocab_size=cfg.vocab_size, d_model=cfg.d_model, d_kv=cfg.d_kv, d_ff=cfg.d_ff, num_layers=cfg.num_layers,
                  num_heads=cfg.num_heads, relative_attention_num_buckets=cfg.relative_attention_num_buckets,
                  relative_attention_max_distance=cfg.relative_attention_max_distance, layer_norm_epsilon=cfg.layer_norm_epsilon,
                  feed_forward_proj="gated-gelu", dropout_rate=0.1, attn_implementation="eager")
    m = T5EncoderModel(hc)
    full = dict(sd)
    full["encoder.embed_tokens.weight"] = sd["shared.weight"]
    missing, unexpected = m.load_state_dict(full, strict=False)
    assert not unexpected and all("embed_tokens" in k or "shared" in k for k in missing), (missing, unexpected)
    return m.to(dtype).eval()


def case_t5(name, cfg, B, lengths=None, threads=8, with_bf16=True):
    torch.set_num_threads(threads)
    sd = synth.t5_state_dict(cfg)
    ids, mask = synth.t5_tokens(B, cfg, lengths=lengths)
    t0 = time.time()
    with torch.no_grad():
        out = build_hf_t5(cfg, sd, torch.float32)(input_ids=ids, attention_mask=mask)["last_hidden_state"]
        arrs = dict(input_ids=ids.numpy(), attention_mask=mask.numpy(), out=out.numpy())
        if with_bf16:      # the reference runs the encoder in bf16 (sample_t2i.py:104): its own round-off vs fp32 calibrates the bf16 tolerance
            ob = build_hf_t5(cfg, sd, torch.bfloat16)(input_ids=ids, attention_mask=mask)["last_hidden_state"]
            arrs["out_bf16"] = ob.float().numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    print(f"{name}: {time.time() - t0:.1f}s  |out| max {out.abs().max():.3f} mean {out.abs().mean():.3f}")


CASES["t5_tiny"] = lambda: case_t5("t5_tiny", C.tiny_t5(), 3, lengths=[1, 17, 120])
CASES["t5_small"] = lambda: case_t5("t5_small", C.small_t5(), 2)
CASES["t5_flan_xl"] = lambda: case_t5("t5_flan_xl", C.flan_t5_xl(), 1, lengths=[23], with_bf16=False)   # full size; fp32 only (1.2 B params)


if __name__ == "__main__":
    names = sys.argv[1:] or DEFAULT
    for n in names:
        CASES[n]()
