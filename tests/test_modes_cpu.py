"""CPU checks of the library-side opt-in MODELS as the oracle states them (what the GPU tests grade the HIP kernels against) and of bench.py's
configuration presets.  No GPU, no /root/reference."""
import sys

import pytest
import torch

from controlar_amd import config as C, synth
from oracle import controlar_oracle as O


@pytest.fixture(scope="module")
def tiny():
    cfg = C.tiny_t2i(64, "canny")
    gsd, _ = synth.path_state_dicts(cfg, seed=0)
    B = 2
    img = synth.canny_like_control(B, 128, 128)
    emb, mask = synth.text_embeddings(B, cfg.gpt.cls_token_num, cfg.gpt.caption_dim)
    return cfg, gsd, img, emb, mask


def test_e4m3_kv_model_rounds_at_store_time_only(tiny):
    """kv_fp8=True (car_config.kv_cache_fp8): K / V are rounded to e4m3 when STORED; the prefill attends to the rows it has just computed, so the first
    sampled token's logits are untouched; every later step reads rounded rows, so its logits move — a little (3 mantissa bits on O(1) values)."""
    cfg, gsd, img, emb, mask = tiny
    t0, l0 = O.generate(gsd, cfg, emb, 12, mask, condition=img, return_logits=True)
    t1, l1 = O.generate(gsd, cfg, emb, 12, mask, condition=img, return_logits=True, forced_tokens=t0, kv_fp8=True)
    assert torch.equal(l0[:, 0], l1[:, 0])                               # prefill step: identical
    d = (l0[:, 1:] - l1[:, 1:]).abs()
    assert float(d.max()) > 0 and float(d.max()) < 0.25 and float(d.mean()) < 0.03, (float(d.max()), float(d.mean()))
    t2, l2 = O.generate(gsd, cfg, emb, 12, mask, condition=img, return_logits=True, forced_tokens=t0, kv_fp8=True)
    assert torch.equal(l1, l2)                                           # deterministic
    # the stored rows really are e4m3 values: rounding them again changes nothing
    k = torch.randn(4, 64) * 3
    r = k.clamp(-448, 448).to(torch.float8_e4m3fn).float()
    assert torch.equal(r, r.to(torch.float8_e4m3fn).float())


def test_w8a8_model_rounds_decode_linear_inputs_only(tiny):
    """act_fp8_decode=True (decode_weight_fp8 = 2): inputs of the decode linears are rounded to e4m3 on single-token steps; the prefill is untouched."""
    cfg, gsd, img, emb, mask = tiny
    t0, l0 = O.generate(gsd, cfg, emb, 8, mask, condition=img, return_logits=True)
    _, l1 = O.generate(gsd, cfg, emb, 8, mask, condition=img, return_logits=True, forced_tokens=t0, act_fp8_decode=True)
    assert torch.equal(l0[:, 0], l1[:, 0])
    assert float((l0[:, 1:] - l1[:, 1:]).abs().max()) > 0


def test_bench_config_presets(monkeypatch):
    """bench.py --config N maps to the BASELINE.json configurations of SURVEY §8d' (1 = the c2i model, new in round 3)."""
    import importlib
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")

    def args(*a):
        monkeypatch.setattr(sys, "argv", ["bench.py", *a])
        return bench.parse()
    a = args()
    assert (a.model, a.batch, a.cfg_scale, a.precision, a.gpus, a.input_dist) == ("xl", 768, 1.0, "bf16", 1, "local") and not a.kv_fp8 and not a.sample_logits
    a = args("--config", "1")
    assert (a.model, a.batch, a.image_size, a.cfg_scale) == ("b_c2i", 4, 256, 1.0)
    assert args("--config", "1", "--batch", "1024").batch == 1024
    a = args("--config", "2"); assert (a.cfg_scale, a.batch) == (4.0, 1)
    a = args("--config", "3"); assert (a.cfg_scale, a.batch, a.condition_type, a.adapter_size) == (4.0, 32, "depth", "base")
    a = args("--config", "4"); assert (a.cfg_scale, a.batch, a.image_h, a.image_w) == (4.0, 1, 768, 512)
    a = args("--config", "5"); assert (a.batch, a.fp8_mfma, a.weights_fp8, a.adapter_size) == (8, True, True, "base")        # BASELINE configs[4] as named: W8A8 on the fp8 MFMA
    a = args("--config", "5", "--fp8-weight-only"); assert (a.batch, a.fp8_mfma, a.weights_fp8) == (8, False, True)          # the weight-only variant (reported beside it)
    assert args().exact_leg_steps == 3 and not args().no_variants                                                            # the default run times the bit-identical mode too
    assert bench.config_legs(args(), 1) == [2, 3, 5, 4, 1] and bench.config_legs(args(), 8) == []                          # ... on one GPU only: at N > 1 no rank spawns child legs
    assert bench.config_legs(args("--config", "2"), 1) == [] and bench.config_legs(args("--no-variants"), 1) == [] and bench.config_legs(args("--precision", "fp32"), 1) == []
    assert args().config_legs == "2,3,5,4,1" and not args().pack_cache                                                       # ... and every other BASELINE config (round 6: the `configs` block)
    a = args("--gpus", "8", "--steps", "20", "--warmup", "5"); assert (a.gpus, a.steps, a.warmup) == (8, 20, 5)
    a = args("--precision", "fp32"); assert (a.batch, a.vq_precision) == (384, "bf16")          # the tokens-exact configuration: fp32 KV of 384 sequences = 162 GB, bf16 pixels
    a = args("--precision", "fp32", "--vq-precision", "fp32", "--batch", "192"); assert (a.batch, a.vq_precision) == (192, "fp32")
    assert args().vq_precision == "bf16"


def test_bench_refuses_library_switches(monkeypatch):
    """A number measured under a CAR_* library switch (CAR_DEBUG_SKIP_STEPS starts the loop late, CAR_NO_GRAPH launches eagerly, ...) is not the benchmark."""
    import importlib
    import os
    import pytest
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    bench.refuse_debug_environment()
    monkeypatch.setenv("CAR_NO_GRAPH", "1")
    with pytest.raises(SystemExit, match="CAR_NO_GRAPH"):
        bench.refuse_debug_environment()


def test_first_valid_position_of_left_padded_masks():
    """The host-side value behind car_sampling.first_valid_hint: min over the batch of the first non-zero mask column; an all-pad row counts as T."""
    from controlar_amd.engine import first_valid_position
    _, mask = synth.text_embeddings_with_lengths([9, 60, 120, 1], 120, 8)
    assert first_valid_position(mask) == 0                       # the unpadded prompt
    assert first_valid_position(mask[:2]) == 60                  # T - 60
    assert first_valid_position(mask[[0, 3]]) == 111             # T - 9
    assert first_valid_position(mask[3:]) == 119
    assert first_valid_position(torch.zeros(2, 120, dtype=torch.int64)) == 120
    m = torch.zeros(1, 120, dtype=torch.int64); m[0, 40] = 1; m[0, 100:] = 1      # a hole in the mask: the first valid column counts
    assert first_valid_position(m) == 40
    _, masks = synth.text_embeddings(64, 120, 8)
    assert first_valid_position(masks) == 120 - int(masks.sum(dim=1).max())
