"""Caption-encoder oracle vs the HF T5EncoderModel goldens (tests/golden/t5_*.npz, minted by make_golden.py: case_t5), plus the
host-side pieces of the T5 front-end that need no GPU."""
import os

import numpy as np
import pytest
import torch

from controlar_amd import config as C
from controlar_amd import synth
from oracle import t5_oracle as TO

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.mark.parametrize("name,cfg", [("t5_tiny", C.tiny_t5()), ("t5_small", C.small_t5())])
def test_oracle_reproduces_hf_fp32(name, cfg):
    g = _load(name)
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    # the token recipe is part of the fixture contract
    ids2, mask2 = synth.t5_tokens(ids.shape[0], cfg, lengths=[1, 17, 120] if name == "t5_tiny" else None)
    assert torch.equal(ids, ids2) and torch.equal(mask, mask2)
    out = TO.encoder_forward(synth.t5_state_dict(cfg), cfg, ids, mask).numpy()
    ref = g["out"]
    err = np.abs(out - ref).max() / np.abs(ref).max()
    assert err <= 2e-5, err


def test_oracle_bf16_within_hf_bf16_roundoff():
    """The reference runs T5 in bf16; the oracle in bf16 must sit where HF's own bf16 run sits relative to fp32."""
    cfg = C.small_t5()
    g = _load("t5_small")
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    out = TO.encoder_forward(synth.t5_state_dict(cfg), cfg, ids, mask, dtype=torch.bfloat16).float().numpy()
    valid = g["attention_mask"].astype(bool)
    hf_err = np.abs(g["out_bf16"] - g["out"])[valid]
    my_err = np.abs(out - g["out"])[valid]
    assert my_err.mean() <= 1.5 * hf_err.mean() + 1e-3, (my_err.mean(), hf_err.mean())
    assert my_err.max() <= 2.0 * hf_err.max() + 1e-2, (my_err.max(), hf_err.max())


def test_relative_position_bucket_known_answers():
    """Known answers of T5's bidirectional bucketing (32 buckets, max distance 128): exact below 8, log-spaced to 127, clamped."""
    rel = torch.tensor([0, 1, 7, 8, 9, 15, 16, 127, 128, 1000, -1, -7, -8, -16, -127, -1000])
    got = TO.relative_position_bucket(rel, 32, 128).tolist()
    assert got == [0, 17, 23, 24, 24, 25, 26, 31, 31, 31, 1, 7, 8, 10, 15, 15]


def test_padded_keys_do_not_influence_valid_rows():
    """Size-independent property of the masked encoder: changing ids at masked positions leaves valid rows unchanged."""
    cfg = C.tiny_t5()
    sd = synth.t5_state_dict(cfg)
    ids, mask = synth.t5_tokens(2, cfg, lengths=[9, 30])
    a = TO.encoder_forward(sd, cfg, ids, mask)
    ids2 = ids.clone()
    ids2[mask == 0] = 5
    b = TO.encoder_forward(sd, cfg, ids2, mask)
    v = mask.bool()
    assert torch.equal(a[v], b[v])


def test_t5_config_from_hf_rejects_relu():
    from controlar_amd.t5 import t5_config_from_hf
    d = dict(vocab_size=32128, d_model=2048, d_kv=64, num_heads=32, d_ff=5120, num_layers=24, feed_forward_proj="gated-gelu")
    assert t5_config_from_hf(d) == C.flan_t5_xl()
    d["feed_forward_proj"] = "relu"
    with pytest.raises(ValueError):
        t5_config_from_hf(d)


def test_t5embedder_keeps_the_reference_keywords_and_never_cleans_silently(tmp_path):
    """language/t5.py:19-35,81-88: the constructor keywords of the reference, the local_cache path rule, and the text cleaning
    default.  ftfy / bs4 are absent from this image, so the reference's clean_caption cannot be borrowed: asking for it must raise."""
    import inspect
    from controlar_amd.t5 import T5Embedder
    ref_kw = ["device", "dir_or_name", "local_cache", "cache_dir", "hf_token", "use_text_preprocessing", "t5_model_kwargs", "torch_dtype",
              "use_offload_folder", "model_max_length"]
    sig = inspect.signature(T5Embedder.__init__)
    assert all(k in sig.parameters for k in ref_kw)
    assert sig.parameters["use_text_preprocessing"].default is True and sig.parameters["local_cache"].default is False
    assert not any(p.kind is inspect.Parameter.VAR_KEYWORD for p in sig.parameters.values())      # nothing is swallowed
    # lower().strip() branch and a user-supplied cleaner
    assert T5Embedder._resolve_preprocessing(False, None)("  A Cat. ") == "a cat."
    assert T5Embedder._resolve_preprocessing(True, lambda t: t.upper())("x") == "X"
    try:
        import ftfy, bs4  # noqa: F401
        have = True
    except ImportError:
        have = False
    if not have:
        with pytest.raises(RuntimeError, match="clean_caption"):
            T5Embedder._resolve_preprocessing(True, None)
    # local_cache: the directory is cache_dir/dir_or_name (it does not exist -> the loader names exactly that path)
    with pytest.raises(FileNotFoundError, match=str(tmp_path / "flan-t5-xl")):
        T5Embedder("cpu", local_cache=True, cache_dir=str(tmp_path), dir_or_name="flan-t5-xl", use_text_preprocessing=False)
