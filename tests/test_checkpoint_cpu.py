"""Checkpoint I/O on CPU: the file-format rules of sample_t2i.py:48-49,64-83 and the key bookkeeping of
load_state_dict(strict=...).  The GPU half (tokens from a checkpoint file / from the packed-image cache equal the in-memory
path) is tests/test_checkpoint_gpu.py."""
import os

import pytest
import torch

from controlar_amd import config as C, synth
from controlar_amd import checkpoint as CK


class _Foreign:                   # something the weights-only unpickler has no business building
    pass


@pytest.fixture(scope="module")
def tiny():
    cfg = C.tiny_t2i(64, "canny")
    gsd, vsd = synth.path_state_dicts(cfg, seed=0)
    return cfg, gsd, vsd


def test_file_formats_follow_the_reference_rules(tiny, tmp_path):
    cfg, gsd, vsd = tiny
    from safetensors.torch import save_file
    p = str(tmp_path / "gpt.safetensors")
    save_file({k: v.contiguous() for k, v in gsd.items()}, p)
    got = CK.load_checkpoint(p)
    assert set(got) == set(gsd) and all(torch.equal(got[k], gsd[k]) for k in gsd)
    for key in ("model", "module", "state_dict"):                      # ddp / deepspeed / plain (sample_t2i.py:72-77)
        q = str(tmp_path / f"gpt_{key}.pt")
        torch.save({key: gsd, "steps": 1}, q)
        got = CK.load_checkpoint(q)
        assert all(torch.equal(got[k], gsd[k]) for k in gsd)
    bad = str(tmp_path / "bad.pt")
    torch.save({"weights": gsd}, bad)
    with pytest.raises(Exception, match="please check model weight"):  # sample_t2i.py:79
        CK.load_checkpoint(bad)
    v = str(tmp_path / "vq.pt")
    torch.save({"model": vsd}, v)                                     # sample_t2i.py:48-49
    assert all(torch.equal(CK.load_checkpoint(v, vq=True)[k], vsd[k]) for k in vsd)


def test_training_script_checkpoints_load(tiny, tmp_path):
    """The .pt files the reference's own training loops write hold an argparse.Namespace and the optimizer state next to the
    weights (train_t2i_canny.py:208, train_c2i_depth.py:252): the weights-only loader must accept them."""
    import argparse
    import pickle
    cfg, gsd, vsd = tiny
    opt = {"state": {0: {"step": torch.tensor(3.0), "exp_avg": torch.zeros(4)}}, "param_groups": [{"lr": 1e-4, "betas": (0.9, 0.95), "params": [0]}]}
    p = str(tmp_path / "0001000.pt")
    torch.save({"model": gsd, "optimizer": opt, "steps": 1000, "args": argparse.Namespace(gpt_model="GPT-XL", image_size=512, lr=1e-4)}, p)
    got = CK.load_checkpoint(p)
    assert set(got) == set(gsd) and all(torch.equal(got[k], gsd[k]) for k in gsd)

    q = str(tmp_path / "foreign.pt")
    torch.save({"model": gsd, "extra": _Foreign()}, q)
    with pytest.raises(pickle.UnpicklingError):
        CK.load_checkpoint(q)
    got = CK.load_checkpoint(q, trust_pickle=True)         # explicit opt-in -> full pickle
    assert all(torch.equal(got[k], gsd[k]) for k in gsd)


def test_expected_keys_match_the_synthetic_reference_state_dicts(tiny):
    cfg, gsd, vsd = tiny
    miss, unexp = CK.key_report(CK.expected_gpt_keys(cfg), gsd.keys(), CK.IGNORED_GPT)
    assert not miss and not unexp, (miss[:4], unexp[:4])
    exp = CK.expected_vq_keys(cfg.vq, "decoder") + CK.expected_vq_keys(cfg.vq, "encoder")
    miss, unexp = CK.key_report(exp, vsd.keys(), ("quantize.codebook_used",))
    assert not miss and not unexp, (miss[:4], unexp[:4])
    c2 = C.tiny_c2i(64)
    g2, _ = synth.path_state_dicts(c2, seed=0)
    miss, unexp = CK.key_report(CK.expected_gpt_keys(c2), g2.keys(), CK.IGNORED_GPT)
    assert not miss and not unexp, (miss[:4], unexp[:4])


def test_load_state_dict_reports_real_missing_and_unexpected_keys(tiny):
    from controlar_amd.models import Transformer, VQModel
    cfg, gsd, vsd = tiny
    m = Transformer(cfg.gpt, cfg.vit)
    sd = dict(gsd)
    del sd["layers.1.attention.wo.weight"]
    sd["something.else"] = torch.zeros(1)
    sd["condition_embeddings.weight"] = torch.zeros(2, 2)              # a training-only tensor: accepted silently
    res = m.load_state_dict(sd, strict=False)
    assert res.missing_keys == ["layers.1.attention.wo.weight"] and res.unexpected_keys == ["something.else"]
    with pytest.raises(RuntimeError, match="Missing key"):
        m.load_state_dict(sd, strict=True)
    assert not m.load_state_dict(gsd, strict=True).missing_keys
    vq = VQModel(cfg.vq)
    dec_only = {k: v for k, v in vsd.items() if not k.startswith(("encoder.", "quant_conv."))}
    assert not vq.load_state_dict(dec_only).missing_keys                # the reference's strict default (sample_t2i.py:49)
    broken = dict(dec_only); del broken["decoder.conv_out.bias"]
    with pytest.raises(RuntimeError):
        vq.load_state_dict(broken)


def test_content_key_depends_on_bytes_and_config(tiny, tmp_path):
    a, b = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    open(a, "wb").write(b"x" * 1000); open(b, "wb").write(b"x" * 999 + b"y")
    k1, k2 = CK.content_key([a], b"cfg"), CK.content_key([b], b"cfg")
    assert k1 != k2 and k1 == CK.content_key([a], b"cfg") and k1 != CK.content_key([a], b"cfg2")
    os.rename(b, str(tmp_path / "a2.bin"))
