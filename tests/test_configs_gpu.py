"""GPU legs (pytest -m gpu) for the BASELINE configurations whose goldens were, until round 3, checked against the oracle only:

  * config 4 — `b_mr_768x512_cfg4`: the multi-resolution geometry of sample_t2i_MR.py:73-78,182-185 at full size (768x512 = 48 x 32 tokens
    on a rope grid of 48, S_max 1656, DINOv2 at 672x448; GPT-B sized): exact tokens over all 1536 positions + bf16 teacher-forced.
  * config 3's encoder — `b_depth_base_256_cfg1p5`: the REAL DINOv2-base, bicubic resize, cfg 1.5, control_strength 0.6.
  * config 1 — `l_c2i_depth_fixtures_cfg1`: GPT-L c2i on the reference's depth fixtures (gpt.py:400-465; bf16 is the only precision the
    reference runs c2i in).
  * the VQ-8 decoder variant (`vq8_real_8x8`, vq_model.py:415-417).
  * config 5 — fp8 decode weights at the XL dimensions (K = 1280 / 3584, N = 3840 / 7168 / 1280 / 16384, b = 8), weight-only and W8A8 on the
    fp8 MFMA, teacher-forced against the oracle running the same arithmetic model on the dequantised weights.
  * the sharp bf16 pin: HIP bf16 against the ORACLE IN BF16 MODE at XL (same rounding points, only the accumulation order differs),
    every one of the 16384 logit columns on 33 steps of two batch rows.

Every test appends what it measured to gpurun_out/parity_measured.jsonl (copied to profiles/ by hand) so the tolerances below can be
read against a number."""
import json
import os

import numpy as np
import pytest
import torch

from tests.cases import GOLDEN

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(name, **vals):
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_measured.jsonl"), "a") as f:
            f.write(json.dumps({"test": name, **{k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in vals.items()}}) + "\n")
    except OSError:
        pass
    print(name, vals)


def _threads():
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))


def _t2i_golden(name, mk, control):
    from controlar_amd import synth
    gold = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    B, H, W, seed, _ = [int(x) for x in gold["meta"]]
    cfg = mk()
    gsd, _ = synth.path_state_dicts(cfg, seed=seed)
    img = synth.canny_like_control(B, H, W) if control == "canny" else synth.smooth_control(B, H, W)
    emb, mask = synth.text_embeddings(B, cfg.gpt.cls_token_num, cfg.gpt.caption_dim)
    return cfg, gold, gsd, img, emb, mask, B, H, W


@pytest.mark.parametrize("name,control", [("b_mr_768x512_cfg4", "canny"), ("b_depth_base_256_cfg1p5", "smooth"),
                                          ("xl_canny_512_cfg4", "canny"), ("xl_mr_768x512_cfg4", "canny")])
def test_t2i_goldens_at_model_size_exact_and_fast(name, control):
    """Reference-minted goldens at model size.  GPT-B: BASELINE config 4's geometry and the real DINOv2-base encoder.  GPT-XL (round 4): BASELINE config 2 as the
    reference runs it (sample_t2i.py:207: cfg_scale 4, one image = 2 rows, 1024 tokens) and config 4 at full width (sample_t2i_MR.py:73-78,182-185: 768x512 = 1536
    tokens on a rope grid of 48, S_max 1656) — exact mode: every token of the reference's greedy sequence, late positions under CFG included; fast mode:
    teacher-forced within the tolerance calibrated on the reference's own bf16 path at XL (tests/golden/xl_canny_512_cfg1_refbf16.npz) scaled by the CFG factor k."""
    from controlar_amd import config as C
    from controlar_amd.engine import Engine
    mk = {"b_mr_768x512_cfg4": lambda: C.b_t2i(2304, "small", "canny"), "b_depth_base_256_cfg1p5": lambda: C.b_t2i(256, "base", "depth"),
          "xl_canny_512_cfg4": lambda: C.xl_t2i(1024, "small", "canny"), "xl_mr_768x512_cfg4": lambda: C.xl_t2i(2304, "small", "canny")}[name]
    xl = name.startswith("xl_")
    cfg, gold, gsd, img, emb, mask, B, H, W = _t2i_golden(name, mk, control)
    n_new = (H // 16) * (W // 16)
    s_, cstr = float(gold["cfg_scale"]), float(gold["control_strength"])
    steps = gold["logits_steps"]
    # ---- exact mode: the reference's greedy tokens bit for bit over the whole image, logits / stages to fp32 round-off
    eng = Engine(cfg, "fp32"); eng.load_state_dict(gsd); eng.finalize()
    a = eng.encode_control(img.cuda(), want_output=True).cpu().numpy()
    np.testing.assert_allclose(a[:, ::7, ::5], gold["adapter_mlp_out"], atol=2e-4, rtol=1e-4)
    toks, logits = eng.generate(emb.cuda(), n_new, mask.cuda(), cfg_scale=s_, control_strength=cstr, return_logits=True)
    eq = toks.cpu().numpy() == gold["tokens"]
    assert eq.all(), f"{int((~eq).sum())} of {eq.size} tokens differ, first at {np.argwhere(~eq)[:1].tolist()}"
    lg = logits.cpu().numpy()[:, steps][:, :, ::4]
    np.testing.assert_allclose(lg, gold["logits"], atol=5e-3, rtol=1e-4)
    b = 2 * B
    c2 = eng.control_tokens(2, b, n_new)
    np.testing.assert_allclose(c2[:B].numpy()[:, ::7, ::5], gold["ctrl2"], atol=2e-4, rtol=1e-4)
    assert float(c2[B:].abs().max()) == 0.0
    assert eng.stats()["graph_used"]
    _record(f"exact[{name}]", tokens_equal=int(eq.sum()), tokens=int(eq.size), logits_max_abs_diff=float(np.abs(lg - gold["logits"]).max()))
    eng.close()
    # ---- fast mode: teacher-forced on the reference's tokens (k = sqrt(s^2 + (s-1)^2) under CFG)
    eng = Engine(cfg, "bf16"); eng.load_state_dict(gsd); eng.finalize()
    eng.encode_control(img.cuda())
    toks, logits = eng.generate(emb.cuda(), n_new, mask.cuda(), cfg_scale=s_, control_strength=cstr,
                                forced_tokens=torch.from_numpy(gold["tokens"]), return_logits=True)
    k = float(np.sqrt(s_ ** 2 + (s_ - 1) ** 2))
    d = np.abs(logits.cpu().numpy()[:, steps][:, :, ::4] - gold["logits"])
    agree = toks.cpu().numpy() == gold["tokens"]
    _record(f"fast[{name}]", max_abs_diff=d.max(), mean_abs_diff=d.mean(), k=k, argmax_agreement=float(agree.mean()))
    if xl:      # 36 layers: the reference's own bf16 path sits 1.45 / 0.24 from its fp32 logits at cfg 1 (91.6 % arg-max agreement); 1.5 x that, times k
        cal = np.load(os.path.join(GOLDEN, "xl_canny_512_cfg1_refbf16.npz"))
        lim_max, lim_mean = 1.5 * float(cal["ref_bf16_max"]) * k, 1.5 * float(cal["ref_bf16_mean"]) * k
        assert d.max() <= lim_max and d.mean() <= lim_mean, (d.max(), d.mean(), lim_max, lim_mean)
        assert agree[gold["margin"] > 2 * lim_max].all() and agree.mean() > 0.8, agree.mean()
    else:
        assert d.max() <= 0.6 * k and d.mean() <= 0.08 * k, (d.max(), d.mean(), k)
        # an arg-max can only flip where the reference's top-2 margin is below the sum of two logit errors: with the per-logit bound 0.6 k that is 1.2 k
        assert agree[gold["margin"] > 1.2 * k].all() and agree.mean() > 0.85, agree.mean()
    eng.close()


def test_c2i_gpt_l_depth_fixtures():
    """GPT-L c2i (24 layers) on condition/example/c2i/depth: exact tokens vs the fp32 oracle; fast mode teacher-forced on the reference's
    bf16 golden with the depth-scaled tolerance of tests/test_oracle_golden.py (the reference's own bf16 drift grows with depth)."""
    from controlar_amd import config as C, synth
    from controlar_amd.engine import Engine
    from oracle import controlar_oracle as O
    _threads()
    gold = np.load(os.path.join(GOLDEN, "l_c2i_depth_fixtures_cfg1.npz"))
    cfg = C.l_c2i(256)
    B, H, W, seed, _ = [int(x) for x in gold["meta"]]
    gsd, _ = synth.path_state_dicts(cfg, seed=seed)
    x = torch.from_numpy(gold["images_u8"]).float() / 255
    x = (2 * (x - 0.5))[:, None].repeat(1, 3, 1, 1)
    labels = torch.from_numpy(gold["labels"])
    n_new, n_chk = gold["tokens"].shape[1], 24
    toks_o, logits_o = O.generate(gsd, cfg, labels, n_chk, None, cfg_scale=1.0, condition=x, return_logits=True)
    eng = Engine(cfg, "fp32"); eng.load_state_dict(gsd); eng.finalize()
    eng.encode_control(x.cuda())
    toks, logits = eng.generate(labels.cuda(), n_chk, None, cfg_scale=1.0, return_logits=True)
    assert np.array_equal(toks.cpu().numpy(), toks_o.numpy())
    np.testing.assert_allclose(logits.cpu().numpy(), logits_o.numpy(), atol=3e-3, rtol=1e-4)
    eng.close()
    eng = Engine(cfg, "bf16"); eng.load_state_dict(gsd); eng.finalize()
    eng.encode_control(x.cuda())
    toks, logits = eng.generate(labels.cuda(), n_new, None, cfg_scale=1.0, forced_tokens=torch.from_numpy(gold["tokens"]), return_logits=True)
    st = int(gold["logits_step_stride"]); vs = 2 if st == 1 else 4
    d = np.abs(logits.cpu().numpy()[:, ::st, ::vs] - gold["logits"].astype(np.float32))
    agree = toks.cpu().numpy() == gold["tokens"]
    _record("fast[l_c2i_depth_fixtures_cfg1]", max_abs_diff=d.max(), mean_abs_diff=d.mean(), argmax_agreement=float(agree.mean()))
    # Calibration at this depth: the reference's bf16 golden sits max 0.94 / mean 0.155 away from fp32 arithmetic (the fp32 oracle against this golden,
    # tests/test_oracle_golden.py).  Two bf16 runs with different summation orders are about as far from each other as each is from fp32 (measured on
    # MI355X: 1.09 / 0.161, profiles/r03_parity_measured.jsonl; at XL the same holds, test_xl_bf16_against_the_oracle_in_bf16_mode): limit = 1.5 x.
    assert d.max() <= 1.5 * 0.94 and d.mean() <= 1.5 * 0.155, (d.max(), d.mean())
    assert agree[gold["margin"] > 2.0 * 1.5 * 0.94].all() and agree.mean() > 0.9, agree.mean()
    eng.close()


@pytest.mark.parametrize("prec,atol,mtol", [("fp32", 2e-3, 1e-4), ("bf16", 0.11, 0.0165)])      # bf16: 2 x measured (0.054 / 0.0082, profiles/r03..r05_parity_measured.jsonl)
def test_vq8_real_architecture(prec, atol, mtol):
    """The VQ-8 decoder variant (ch_mult (1, 2, 2, 4): three upsampling levels) on an 8x8 token grid vs the reference's pixels."""
    from controlar_amd import config as C, synth
    from controlar_amd.engine import Engine
    gold = np.load(os.path.join(GOLDEN, "vq8_real_8x8.npz"))
    cfg = C.tiny_t2i(64, "canny"); cfg.vq = C.VQConfig(ch_mult=(1, 2, 2, 4))
    eng = Engine(cfg, prec)
    eng.load_state_dict(synth.vq_state_dict(cfg.vq, seed=3), finalize=True)
    px = eng.vq_decode(torch.from_numpy(gold["tokens"]), 8, 8).cpu().numpy()
    assert px.shape == (2, 3, 64, 64)
    d = np.abs(px - gold["pixels"])
    _record(f"vq8[{prec}]", max_abs_diff=d.max(), mean_abs_diff=d.mean())
    assert d.max() <= atol and d.mean() <= mtol, (d.max(), d.mean())
    eng.close()


def _quantize_like_library(sd, cfg):
    from tests.test_parity_gpu import _quantize_like_library as q
    return q(sd, cfg)


@pytest.mark.parametrize("mode", [True, "mfma"])
def test_fp8_decode_at_xl_dims_config5(mode):
    """BASELINE config 5 at its real dimensions: LlamaGen-XL, DINOv2-base edge control (bicubic path), 8 images, cfg 1; the tile
    configurations dec_gemm<..., F8> picks for K = 1280 / 3584 and N = 3840 / 7168 / 1280 / 16384 at M = 8.  9 tokens = prefill + 8
    decode steps, teacher-forced on the dequantised-weight oracle's tokens."""
    from controlar_amd import config as C, synth
    from controlar_amd.engine import Engine
    from oracle import controlar_oracle as O
    _threads()
    cfg = C.xl_t2i(1024, "base", "hed")
    gsd, _ = synth.path_state_dicts(cfg, seed=0)
    B, H, W, n_new = 8, 512, 512, 9
    img = synth.smooth_control(B, H, W)
    emb, mask = synth.text_embeddings(B, cfg.gpt.cls_token_num, cfg.gpt.caption_dim)
    qsd = _quantize_like_library(gsd, cfg)
    toks_q, logits_w = O.generate(qsd, cfg, emb, n_new, mask, cfg_scale=1.0, condition=img, return_logits=True)
    ref = logits_w
    if mode == "mfma":        # W8A8: the oracle rounds the inputs of the five decode linears to e4m3 on single-token steps
        _, ref = O.generate(qsd, cfg, emb, n_new, mask, cfg_scale=1.0, condition=img, forced_tokens=toks_q, return_logits=True, act_fp8_decode=True)
    eng = Engine(cfg, "bf16", weights_fp8=mode); eng.load_state_dict(gsd); eng.finalize()
    eng.encode_control(img.cuda())
    toks, logits = eng.generate(emb.cuda(), n_new, mask.cuda(), cfg_scale=1.0, forced_tokens=toks_q, return_logits=True)
    assert eng.stats()["graph_used"]
    lg = logits.cpu()
    assert torch.isfinite(lg).all()
    d = (lg - ref).abs()
    dw = (lg - logits_w).abs()
    model = (ref - logits_w).abs()            # what the activation rounding of the W8A8 MODEL itself moves (0 for the weight-only mode)
    cal = np.load(os.path.join(GOLDEN, "xl_canny_512_cfg1_refbf16.npz"))
    top2 = ref.topk(2, dim=-1).values
    margin = top2[..., 0] - top2[..., 1]
    agree = toks.cpu() == ref.argmax(-1)
    _record(f"fp8_xl[{mode}]", max_abs_diff=float(d.max()), mean_abs_diff=float(d.mean()), vs_weight_only_max=float(dw.max()),
            vs_weight_only_mean=float(dw.mean()), model_act_rounding_max=float(model.max()), model_act_rounding_mean=float(model.mean()),
            argmax_agreement=float(agree.float().mean()))
    # weight-only: the same arithmetic model on both sides, what is left is bf16 rounding over 36 layers — the calibrated XL tolerance
    # (the reference's own bf16 against fp32) x 1.5
    lim_max, lim_mean = 1.5 * float(cal["ref_bf16_max"]), 1.5 * float(cal["ref_bf16_mean"])
    if mode == "mfma":
        # W8A8: rounding every decode-linear input to e4m3 (3 mantissa bits) is a noisy model at 36 layers — the oracle's W8A8 logits sit
        # `model` away from its own weight-only logits — and the e4m3 codes of two implementations decorrelate as soon as their bf16 inputs
        # differ by an ulp.  The kernel check that remains meaningful: the HIP logits must deviate from the weight-only model by no more than the
        # oracle's W8A8 model does (a kernel bug adds deviation), and from the oracle's W8A8 logits by less than two independent draws of that noise.
        assert float(dw.mean()) <= 1.3 * float(model.mean()) and float(dw.max()) <= 1.5 * float(model.max()), (float(dw.mean()), float(model.mean()), float(dw.max()), float(model.max()))
        assert float(d.mean()) <= 1.5 * float(model.mean()), (float(d.mean()), float(model.mean()))
    else:
        assert float(d.max()) <= lim_max and float(d.mean()) <= lim_mean, (float(d.max()), float(d.mean()))
        assert bool(agree[margin > 2.0 * lim_max].all())
    eng.close()


def test_xl_bf16_against_the_oracle_in_bf16_mode():
    """HIP bf16 vs the oracle RUN IN BF16 — the same rounding points, only the accumulation order inside a dot product differs —
    at LlamaGen-XL, two batch rows, prefill + 32 decode steps, all 16384 logit columns (Weak #2).
    Measured on MI355X (profiles/r03_parity_measured.jsonl): HIP-bf16 vs oracle-bf16 max 1.78 / mean 0.255; HIP-bf16 vs oracle-fp32 1.78 / 0.251;
    oracle-bf16 vs oracle-fp32 1.79 / 0.247.  The hoped-for order-of-magnitude tighter pin does not exist: one fp32 ulp of summation-order
    difference flips a bf16 rounding, and 36 layers of those flips decorrelate two bf16 runs almost completely (0.255 against 0.35 = sqrt(2) x 0.247
    for fully independent noise).  What IS sharp is the error BUDGET: a missing, extra or misplaced rounding point, a wrong eps or a biased kernel
    moves HIP-vs-fp32 away from the oracle's bf16-vs-fp32 — so the test pins the ratio (measured 1.017 on the mean, 0.993 on the max), the bias
    (mean signed error ~ 0) and the bf16-vs-bf16 distance below the independent-noise level."""
    from controlar_amd import config as C, synth
    from controlar_amd.engine import Engine
    from oracle import controlar_oracle as O
    _threads()
    cfg = C.xl_t2i(1024, "small", "canny")
    gsd, _ = synth.path_state_dicts(cfg, seed=0)
    B, H, W, n_new = 2, 512, 512, 33
    img = synth.canny_like_control(B, H, W)
    emb, mask = synth.text_embeddings(B, cfg.gpt.cls_token_num, cfg.gpt.caption_dim)
    gsd16 = {k: (v.to(torch.bfloat16) if v.is_floating_point() else v) for k, v in gsd.items()}
    toks_o, logits_o = O.generate(gsd16, cfg, emb, n_new, mask, cfg_scale=1.0, condition=img, dtype=torch.bfloat16, return_logits=True)
    _, logits_32 = O.generate(gsd, cfg, emb, n_new, mask, cfg_scale=1.0, condition=img, forced_tokens=toks_o, return_logits=True)
    eng = Engine(cfg, "bf16"); eng.load_state_dict(gsd); eng.finalize()
    eng.encode_control(img.cuda())
    toks, logits = eng.generate(emb.cuda(), n_new, mask.cuda(), cfg_scale=1.0, forced_tokens=toks_o, return_logits=True)
    lg = logits.cpu().float()
    d = (lg - logits_o.float()).abs()                       # HIP bf16 vs oracle bf16
    d32 = (lg - logits_32).abs()                            # HIP bf16 vs oracle fp32
    o32 = (logits_o.float() - logits_32).abs()              # oracle bf16 vs oracle fp32: the size of bf16 itself on this input
    agree = (toks.cpu() == toks_o)
    top2 = logits_o.float().topk(2, dim=-1).values
    margin = top2[..., 0] - top2[..., 1]
    per_step = d.amax(dim=(0, 2))
    _record("xl_bf16_vs_oracle_bf16", steps=n_new, rows=B, columns=int(lg.shape[-1]),
            hip_vs_oracle_bf16_max=float(d.max()), hip_vs_oracle_bf16_mean=float(d.mean()),
            hip_vs_oracle_fp32_max=float(d32.max()), hip_vs_oracle_fp32_mean=float(d32.mean()),
            oracle_bf16_vs_fp32_max=float(o32.max()), oracle_bf16_vs_fp32_mean=float(o32.mean()),
            argmax_agreement=float(agree.float().mean()), per_step_max_first=float(per_step[0]), per_step_max_last=float(per_step[-1]))
    assert torch.isfinite(lg).all()
    ratio_mean, ratio_max = float(d32.mean()) / float(o32.mean()), float(d32.max()) / float(o32.max())
    bias = float((lg - logits_32).mean()) / float(d32.mean())
    _record("xl_bf16_error_budget", ratio_mean=ratio_mean, ratio_max=ratio_max, signed_bias_over_mean_abs=bias,
            bf16_vs_bf16_over_independent=float(d.mean()) / (float(np.sqrt(2.0)) * float(o32.mean())))
    assert 0.85 <= ratio_mean <= 1.15 and ratio_max <= 1.3, (ratio_mean, ratio_max)       # same error budget as the reference's dataflow
    assert abs(bias) <= 0.05, bias                                                       # no systematic offset
    assert float(d.mean()) <= 0.9 * float(np.sqrt(2.0)) * float(o32.mean()), (float(d.mean()), float(o32.mean()))   # correlated with the oracle's bf16 run
    assert bool(agree[margin > 2.0 * float(d.max())].all())
    eng.close()


def test_w8a8_shallow_model_decorrelates_within_three_layers():
    """W8A8 (e4m3 weights x e4m3 activations on v_mfma_f32_16x16x32_fp8_fp8) at XL width but THREE layers.  Round 4 expected a shallow model to pin the kernel
    tightly; it does not: the HIP logits sit 0.46 (mean) from the oracle's W8A8 logits where the weight-only kernel sits 0.083 from the oracle's weight-only
    logits and the model's own activation rounding moves the logits by 0.37 — a bf16-ulp difference in a linear's input already flips e4m3 codes (3 mantissa
    bits) in the first layer.  The kernel itself is pinned on identical inputs by tests/test_kernels_gpu.py (experiments/f8_check: one bf16 ulp); what is
    asserted here is that the deviation stays at the size of the model's own rounding noise — a misplaced scale or operand half is far outside it."""
    from controlar_amd import config as C, synth
    from controlar_amd.engine import Engine
    from oracle import controlar_oracle as O
    _threads()
    cfg = C.xl_t2i(1024, "small", "canny")
    cfg.gpt.n_layer = 3
    gsd, _ = synth.path_state_dicts(cfg, seed=0)
    B, H, W, n_new = 8, 512, 512, 9
    img = synth.canny_like_control(B, H, W)
    emb, mask = synth.text_embeddings(B, cfg.gpt.cls_token_num, cfg.gpt.caption_dim)
    qsd = _quantize_like_library(gsd, cfg)
    toks_q, logits_w = O.generate(qsd, cfg, emb, n_new, mask, cfg_scale=1.0, condition=img, return_logits=True)
    _, logits_a = O.generate(qsd, cfg, emb, n_new, mask, cfg_scale=1.0, condition=img, forced_tokens=toks_q, return_logits=True, act_fp8_decode=True)
    dev = {}
    for mode, ref in ((True, logits_w), ("mfma", logits_a)):
        eng = Engine(cfg, "bf16", weights_fp8=mode); eng.load_state_dict(gsd); eng.finalize()
        eng.encode_control(img.cuda())
        _, logits = eng.generate(emb.cuda(), n_new, mask.cuda(), cfg_scale=1.0, forced_tokens=toks_q, return_logits=True)
        d = (logits.cpu()[:, 1:] - ref[:, 1:]).abs()            # decode steps only (the prefill runs on the dequantised weights in both modes)
        dev[mode] = (float(d.max()), float(d.mean()))
        eng.close()
    model = (logits_a[:, 1:] - logits_w[:, 1:]).abs()
    _record("w8a8_three_layers", weight_only_max=dev[True][0], weight_only_mean=dev[True][1], w8a8_max=dev["mfma"][0], w8a8_mean=dev["mfma"][1],
            model_act_rounding_max=float(model.max()), model_act_rounding_mean=float(model.mean()))
    assert dev["mfma"][1] <= dev[True][1] + 1.5 * float(model.mean()), (dev, float(model.mean()))
    assert dev["mfma"][0] <= dev[True][0] + 1.5 * float(model.max()), (dev, float(model.max()))


@pytest.mark.parametrize("size", ["tiny", "tiny+w8", "xl"])
def test_e4m3_kv_cache_opt_in(size):
    """car_config.kv_cache_fp8 (opt-in; bf16 KV stays the default and the parity path): rotated K and V are stored as OCP e4m3 bytes, unit scale, and widened
    to bf16 in registers in front of the same MFMAs.  The reference has no such mode: graded teacher-forced against the oracle running the SAME model
    (`kv_fp8=True`: rounding at store time, the prefill attends to its own unrounded rows).  tiny: B = 3 (16-wave one-launch attention, fused norms), ragged
    positions across 32-blocks, cfg 2; xl: 2 rows x 33 steps x every logit column, with the model's own effect and the error budget reported."""
    from controlar_amd import config as C, synth
    from controlar_amd.engine import Engine
    from oracle import controlar_oracle as O
    _threads()
    w8 = size.endswith("+w8")           # together with e4m3 decode weights (weight-only): the two opt-in modes are independent and compose
    size = size.split("+")[0]
    if size == "tiny":
        cfg = C.tiny_t2i(64, "canny"); B, H, W, n_new, s_ = 3, 128, 128, 64, 2.0
    else:
        cfg = C.xl_t2i(1024, "small", "canny"); B, H, W, n_new, s_ = 2, 512, 512, 33, 1.0
    gsd, _ = synth.path_state_dicts(cfg, seed=0)
    img = synth.canny_like_control(B, H, W)
    emb, mask = synth.text_embeddings(B, cfg.gpt.cls_token_num, cfg.gpt.caption_dim)
    osd = _quantize_like_library(gsd, cfg) if w8 else gsd            # the oracle runs on the dequantised weights in the weight-fp8 combination
    toks_o, logits_m = O.generate(osd, cfg, emb, n_new, mask, cfg_scale=s_, condition=img, return_logits=True, kv_fp8=True)
    _, logits_p = O.generate(osd, cfg, emb, n_new, mask, cfg_scale=s_, condition=img, forced_tokens=toks_o, return_logits=True)      # plain model, same tokens
    eng = Engine(cfg, "bf16", kv_fp8=True, weights_fp8=w8); eng.load_state_dict(gsd); eng.finalize()
    eng.encode_control(img.cuda())
    toks, logits = eng.generate(emb.cuda(), n_new, mask.cuda(), cfg_scale=s_, forced_tokens=toks_o, return_logits=True)
    assert eng.stats()["graph_used"]
    lg = logits.cpu()
    assert torch.isfinite(lg).all()
    d = (lg - logits_m).abs()                       # HIP e4m3-KV vs the oracle's e4m3-KV model (fp32 arithmetic)
    model = (logits_m - logits_p).abs()             # what the model itself moves
    k = 1.0 if s_ <= 1 else float(np.sqrt(s_ ** 2 + (s_ - 1) ** 2))
    agree = (toks.cpu() == toks_o)
    _record(f"kv_e4m3[{size}{'+w8' if w8 else ''}]", hip_vs_model_max=float(d.max()), hip_vs_model_mean=float(d.mean()), model_vs_plain_max=float(model.max()),
            model_vs_plain_mean=float(model.mean()), argmax_agreement=float(agree.float().mean()))
    if size == "tiny":
        # the model rounds the same bf16-exact... no: the oracle rounds fp32 K / V, HIP rounds their bf16 values — an e4m3 code can differ where the two disagree by a bf16
        # ulp — so the tolerance is the fast mode's own (0.6 k / 0.08 k) plus the model's effect
        assert float(d.max()) <= 0.6 * k + float(model.max()) and float(d.mean()) <= 0.08 * k + float(model.mean()), (float(d.max()), float(d.mean()))
        top2 = logits_m.topk(2, dim=-1).values
        assert bool(agree[(top2[..., 0] - top2[..., 1]) > 2 * (0.6 * k + float(model.max()))].all()) and float(agree.float().mean()) > 0.85
    else:
        cal = np.load(os.path.join(GOLDEN, "xl_canny_512_cfg1_refbf16.npz"))
        # bf16 arithmetic (calibrated 1.45 / 0.24 at XL) and the e4m3 code flips it induces (bounded by the model's own effect): 1.5 x their sum
        lim_max, lim_mean = 1.5 * (float(cal["ref_bf16_max"]) + float(model.max())), 1.5 * (float(cal["ref_bf16_mean"]) + float(model.mean()))
        assert float(d.max()) <= lim_max and float(d.mean()) <= lim_mean, (float(d.max()), float(d.mean()), lim_max, lim_mean)
        assert float(agree.float().mean()) >= 0.8
    eng.close()
    # the mode exists only in the fast mode
    with pytest.raises(RuntimeError):
        Engine(cfg, "fp32", kv_fp8=True)


def test_e4m3_kv_cache_saturates_outliers():
    """OCP e4m3fn has no infinity: a K / V element beyond +-448 must be stored as +-448 (include/controlar_hip.h and the oracle's kv_fp8 model say
    clamp), never as NaN — one NaN row would poison every later step of the sequence (P * NaN = NaN even at P = 0).  One V channel and one K channel of
    layer 0 are scaled so that most of their cached values exceed 448; the logits must stay finite and follow the oracle running the same clamping model."""
    from controlar_amd import config as C, synth
    from controlar_amd.engine import Engine
    from oracle import controlar_oracle as O
    _threads()
    cfg = C.tiny_t2i(64, "canny"); B, H, W, n_new = 2, 128, 128, 24
    gsd, _ = synth.path_state_dicts(cfg, seed=0)
    gsd = {k: v.clone() for k, v in gsd.items()}
    D = cfg.gpt.dim
    w = gsd["layers.0.attention.wqkv.weight"]
    w[2 * D + 5] *= 4000.0            # V, head 0, dim 5
    w[D + 64 + 7] *= 4000.0           # K, head 1, dim 7 (rotated with its partner dim 6: both can exceed 448)
    img = synth.canny_like_control(B, H, W)
    emb, mask = synth.text_embeddings(B, cfg.gpt.cls_token_num, cfg.gpt.caption_dim)
    toks_o, logits_m = O.generate(gsd, cfg, emb, n_new, mask, cfg_scale=1.0, condition=img, return_logits=True, kv_fp8=True)
    eng = Engine(cfg, "bf16", kv_fp8=True); eng.load_state_dict(gsd); eng.finalize()
    eng.encode_control(img.cuda())
    _, logits = eng.generate(emb.cuda(), n_new, mask.cuda(), cfg_scale=1.0, forced_tokens=toks_o, return_logits=True)
    lg = logits.cpu()
    assert torch.isfinite(lg).all(), "an e4m3 KV outlier was stored as NaN"
    d = (lg - logits_m).abs()
    _record("kv_e4m3_outliers[tiny]", hip_vs_model_max=float(d.max()), hip_vs_model_mean=float(d.mean()))
    assert float(d.max()) <= 0.6 and float(d.mean()) <= 0.08, (float(d.max()), float(d.mean()))      # the fast mode's own tolerance (measured 0.24 / 0.027)
    eng.close()
