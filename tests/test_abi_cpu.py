"""CPU: the C-ABI library loads and exports every symbol include/controlar_hip.h declares;
argument validation that needs no GPU works; the product never imports the oracle."""
import ctypes as C
import os
import re

import pytest

from controlar_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "controlar_hip.h")).read()
    declared = set(re.findall(r"\b(car_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.car_abi_version() == L.CAR_ABI_VERSION


def test_struct_layout_matches_header():
    assert C.sizeof(L.CarConfig) == 4 * (2 + 5 + 3 + 2 + 6 + 2 + 6 + 8 + 1 + 8)
    assert C.sizeof(L.CarSampling) == 56
    assert C.sizeof(L.CarStats) == 8 * 4 + 4 * 8


def test_create_rejects_bad_config_without_gpu():
    lib = L.load()
    h = C.c_void_p()
    cc = L.CarConfig()
    cc.abi_version = 99
    assert lib.car_create(C.byref(h), C.byref(cc)) != 0
    assert b"abi_version" in lib.car_last_error(None)
    cc.abi_version = L.CAR_ABI_VERSION
    assert lib.car_create(C.byref(h), C.byref(cc)) != 0      # zero dims
    assert lib.car_create(None, None) != 0


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "controlar_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                src = open(os.path.join(dp, f)).read()
                assert "controlar_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from controlar_amd import config as Cfg
    from controlar_amd.engine import Engine
    with pytest.raises(RuntimeError):
        Engine(Cfg.tiny_t2i(), "bf16")


def test_e4m3_conversion_matches_torch():
    """The host-side fp8 quantiser of the fp8-weight decode path (BASELINE config 5) == torch.float8_e4m3fn rounding."""
    import numpy as np
    import torch
    lib = L.load()
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(20000, generator=g) * 100, torch.randn(20000, generator=g) * 0.01, torch.randn(5000, generator=g) * 600,
                   torch.tensor([0.0, -0.0, 448.0, 463.9, 464.0, 1e9, -1e9, 2 ** -6, 2 ** -9, 2 ** -10, 1.5 * 2 ** -10, 2 ** -7 + 2 ** -10])]).float().contiguous()
    out = np.empty(x.numel(), dtype=np.uint8)
    assert lib.car_debug_f32_to_e4m3(C.c_void_p(x.data_ptr()), C.c_void_p(out.ctypes.data), x.numel()) == 0
    want = x.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).numpy()     # torch saturates only via clamp
    got_f = torch.from_numpy(out).view(torch.float8_e4m3fn).float()
    want_f = torch.from_numpy(want).view(torch.float8_e4m3fn).float()
    assert torch.equal(got_f, want_f)


def test_dropin_modules_import_and_factories_build_without_gpu():
    """controlar_amd.models / generate import cleanly and the reference-shaped factories build (weights/contexts are lazy)."""
    from controlar_amd import generate as G, models as M
    gpt = M.GPT_models["GPT-XL"](block_size=1024, cls_token_num=120, model_type="t2i", condition_type="canny", adapter_size="small")
    assert gpt.model_type == "t2i" and gpt.cfg.gpt.dim == 1280 and gpt.cfg.vit.hidden == 384
    c2i = M.GPT_models["GPT-B"](vocab_size=16384, block_size=256, num_classes=1000, cls_token_num=1, model_type="c2i",
                                condition_token_num=0, image_size=256)
    assert c2i.cfg.vit.variant == "vit" and c2i.cfg.vit.patch == 16
    vq = M.VQ_models["VQ-16"](codebook_size=16384, codebook_embed_dim=8)
    assert hasattr(vq, "decode_code") and hasattr(vq, "encode_indices") and callable(G.generate)


def test_decode_code_argument_forms_fail_as_in_the_reference():
    """VQModel.decode_code(shape=None) and (channel_first=False) are not usable in the reference either (vq_model.py:53-56,262-277: a 2-D gather, resp.
    a [B,h,w,C] tensor, reaches the NCHW post_quant_conv): the drop-in raises the same exception type with the same reason; when the reference
    tree is present (build container) its own behaviour is checked beside it."""
    import sys
    import torch
    from controlar_amd import models as M
    vq = M.VQ_models["VQ-16"](codebook_size=16384, codebook_embed_dim=8)
    code = torch.zeros(2, 16, dtype=torch.int64)
    with pytest.raises(RuntimeError, match="conv2d"):
        vq.decode_code(code.reshape(-1))                               # shape=None, flat codes: a 2-D gather
    with pytest.raises(RuntimeError, match="channels"):
        vq.decode_code(code)                                           # shape=None, [B, N] codes: [B, N, C] taken for one unbatched image
    with pytest.raises(RuntimeError, match="channels"):
        vq.decode_code(code, (2, 4, 4, 8), channel_first=False)        # [B, h, w, C] with h != C
    ref_root = "/root/reference"
    if os.path.isdir(ref_root):
        sys.path.insert(0, ref_root)
        try:
            from tokenizer.tokenizer_image.vq_model import VQ_models as RefVQ
        except Exception:
            return
        finally:
            sys.path.remove(ref_root)
        ref = RefVQ["VQ-16"](codebook_size=64, codebook_embed_dim=8).eval()
        with torch.no_grad():
            with pytest.raises(RuntimeError, match="conv2d"):
                ref.decode_code(code.reshape(-1))
            with pytest.raises(RuntimeError, match="channels"):
                ref.decode_code(code)
            with pytest.raises(RuntimeError, match="channels"):
                ref.decode_code(code, (2, 4, 4, 8), channel_first=False)
            assert tuple(ref.decode_code(code, (2, 8, 4, 4)).shape) == (2, 3, 64, 64)     # the one form the path uses


def test_header_is_valid_c99_and_example_compiles():
    """include/controlar_hip.h must stay a plain C header (extern "C" boundary): gcc -std=c99 syntax-checks the C example."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "c_abi_example.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    if os.path.isdir("/opt/rocm/include"):       # the runnable client (tests/test_c_example_gpu.py links and runs it on the GPU box)
        r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "examples", "c_abi_run.c")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_decode_weight_fragment_packing_layout():
    """The packed image consumed by dec_linear_kernel: lane l of chunk (rb, kb) holds W[16rb + (l & 15)][32kb + 8(l >> 4) : +8] —
    exactly the A-operand fragment of v_mfma_f32_16x16x32_bf16, so one 16-byte load per lane feeds one MFMA."""
    import numpy as np
    import torch
    lib = L.load()
    N, K = 48, 96
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) / 8.0          # exactly representable in bf16 up to 576
    w = (w % 251.0).contiguous()
    out = np.empty(N * K, dtype=np.uint16)
    assert lib.car_debug_pack_decode_weight(C.c_void_p(w.data_ptr()), N, K, C.c_void_p(out.ctypes.data)) == 0
    got = torch.from_numpy(out.astype(np.int16)).view(torch.bfloat16).float().reshape(N // 16, K // 32, 64, 8)
    wb = w.to(torch.bfloat16).float()
    for rb in range(N // 16):
        for kb in range(K // 32):
            for l in (0, 5, 15, 16, 33, 63):
                r, k0 = rb * 16 + (l & 15), kb * 32 + (l >> 4) * 8
                assert torch.equal(got[rb, kb, l], wb[r, k0:k0 + 8]), (rb, kb, l)
    assert lib.car_debug_pack_decode_weight(C.c_void_p(w.data_ptr()), 40, K, C.c_void_p(out.ctypes.data)) != 0   # N % 16 != 0


def test_decode_gemm_tile_choice_is_valid_for_every_model_size():
    """car_pick_gemm_cfg (decode2.hip, host code) must hand car_launch_dec_gemm_cfg a configuration it accepts for every LlamaGen size with
    64-wide heads (gpt_t2i.py:541-563: B, L, XL, XXL, 1B), every decode linear and every chain length — a rejected tile only shows up
    as a run-time error of generate() on the GPU otherwise.  Mirrors the acceptance rules of car_launch_dec_gemm_cfg."""
    import ctypes as C
    from controlar_amd import _lib
    from controlar_amd.config import ffn_hidden_dim
    lib = _lib.load()
    pick = lib.car_pick_gemm_cfg
    pick.restype = C.c_int; pick.argtypes = [C.c_int] * 4
    EPI_LOGITS, EPI_RESID, EPI_SWIGLU, EPI_QKV = 0, 1, 2, 3
    for dim in (768, 1024, 1280, 1536, 2048):
        fh, V = ffn_hidden_dim(dim), 16384
        linears = [(3 * dim, dim, EPI_QKV), (dim, dim, EPI_RESID), (2 * fh, dim, EPI_SWIGLU), (dim, fh, EPI_RESID), (V, dim, EPI_LOGITS)]
        for M in list(range(1, 18)) + [24, 32, 48, 64, 96, 100, 128, 144, 192, 256, 384, 512, 768, 1024]:
            for N, K, epi in linears:
                cfg = pick(M, N, K, epi)
                I, J, w8 = cfg // 100, (cfg // 10) % 10, cfg % 10
                assert I in (1, 2, 4) and J in (1, 2, 4) and w8 in (0, 1, 2), (dim, M, N, K, epi, cfg)
                # round 6: the one 16-wave tile (cfg 112) exists for the narrow RESID linears of a one-m-block chain only
                assert w8 != 2 or (cfg == 112 and epi == EPI_RESID and M <= 16 and N // 16 <= 128), (dim, M, N, K, epi, cfg)
                assert N % (16 * I) == 0 and K % 32 == 0, (dim, M, N, K, epi, cfg)
                assert epi != EPI_SWIGLU or I >= 2, (dim, M, N, K, epi, cfg)          # the (a, c) pair needs two adjacent row-blocks in one tile


def test_exact_mode_tile_choice_and_norm_partials_are_valid_for_every_model_size():
    """Host-side rules of the exact-mode decode kernels, for every LlamaGen width: (a) car_pick_gemm_f32_cfg (decode_f32.hip) hands car_launch_dec_gemm_f32_cfg a tile
    it accepts (register kernel: N % 16I == 0, K % 16 == 0, two row-blocks per tile for the SwiGLU pair; tiled kernel: K % 128 == 0, N % 32WN == 0) and never lets
    the batch change an output's arithmetic (every configuration computes the canonical 8-slice tree); (b) the on-the-fly RMSNorm (dec_gemm NORM == 2) folds N/32 or N/16 per-tile partials per row with 16-byte loads: that count must be
    a multiple of 4 and at most 128 whatever tile the producer (wo / w2, car_pick_gemm_cfg) takes for chains of up to 48 rows."""
    import ctypes as C
    from controlar_amd import _lib
    from controlar_amd.config import ffn_hidden_dim
    lib = _lib.load()
    pick32 = lib.car_pick_gemm_f32_cfg2
    pick32.restype = C.c_int; pick32.argtypes = [C.c_int] * 5
    lib.car_pick_gemm_f32_cfg.restype = C.c_int; lib.car_pick_gemm_f32_cfg.argtypes = [C.c_int] * 4
    pick = lib.car_pick_gemm_cfg
    pick.restype = C.c_int; pick.argtypes = [C.c_int] * 4
    FEPI_PLAIN, FEPI_RESID, FEPI_SWIGLU, FEPI_QKV = 0, 1, 2, 3
    for dim in (256, 768, 1024, 1280, 1536, 2048):
        fh, V = ffn_hidden_dim(dim), 16384
        for M in list(range(1, 18)) + [32, 36, 64, 96, 192, 384, 768]:
            for N, K, epi in [(3 * dim, dim, FEPI_QKV), (dim, dim, FEPI_RESID), (2 * fh, dim, FEPI_SWIGLU), (dim, fh, FEPI_RESID), (V, dim, FEPI_PLAIN)]:
              for chains in (1, 2, 3):          # the tile choice may depend on how many chains run side by side, never the arithmetic
                cfg = pick32(M, N, K, epi, chains)
                if cfg >= 1000:      # round 5: the LDS-tiled kernel, cfg = 1000 + 100 WN + 10 WM + KG; it needs 8 equal K slices and whole 32·WN-row tiles
                    WN, WM, KG = (cfg - 1000) // 100, (cfg // 10) % 10, cfg % 10
                    assert cfg in (1221, 1212, 1214), (dim, M, N, K, cfg)       # the configurations the product library instantiates
                    assert K % 128 == 0 and N % (32 * WN) == 0 and KG in (1, 2, 4), (dim, M, N, K, cfg)
                    continue
                I, J = cfg // 10, cfg % 10
                assert I in (1, 2, 4) and J in (1, 2, 4), (dim, M, N, K, cfg)
                assert N % (16 * I) == 0 and K % 16 == 0 and (epi != FEPI_SWIGLU or I >= 2), (dim, M, N, K, cfg)
                if chains == 1:
                    assert cfg == lib.car_pick_gemm_f32_cfg(M, N, K, epi)
        for M in range(1, 49):
            for N, K in [(dim, dim), (dim, fh)]:                      # the two RESID producers of the residual stream
                I = pick(M, N, K, 1) // 100
                partials = N // (16 * (2 if I >= 2 else 1))
                assert partials % 4 == 0 and 0 < partials <= 128, (dim, M, N, K, I, partials)


def test_shipped_library_contains_no_debug_switches():
    """The A/B and profiling switches (CAR_* environment variables) are compiled out of libcontrolar_hip.so (csrc/car_common.h CAR_KNOB, csrc/build.sh): no getenv
    import, no switch name among its strings; the development build next to it has them."""
    import subprocess
    def names(path):
        data = open(path, "rb").read()
        return set(m.decode() for m in re.findall(rb"CAR_[A-Z0-9_]{3,}", data))
    rel, dev = names(L.LIB_PATH), names(L.DEV_LIB_PATH)
    allowed = {"CAR_F32", "CAR_BF16"}                       # mode names inside error messages
    assert rel <= allowed, sorted(rel - allowed)
    assert {"CAR_NO_GRAPH", "CAR_DEBUG_SKIP_STEPS", "CAR_CHAINS"} <= dev
    syms = subprocess.run(["nm", "-D", L.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in syms
