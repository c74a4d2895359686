"""Checkpoint I/O on the GPU (pytest -m gpu): a context filled from checkpoint FILES (safetensors / .pt, sample_t2i.py:48-49,64-83)
and a context restored from the packed-image cache must generate exactly what the in-memory load generates."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec,fp8", [("bf16", False), ("bf16", True), ("fp32", False)])
def test_checkpoint_files_and_packed_cache_reproduce_the_in_memory_load(prec, fp8, tmp_path):
    from safetensors.torch import save_file
    from controlar_amd.checkpoint import load_engine_from_checkpoints
    from controlar_amd.engine import Engine
    from tests.cases import load_case
    cs = load_case("tiny_depth_cfg4")
    gp, vp = str(tmp_path / "gpt.safetensors"), str(tmp_path / "vq.pt")
    save_file({k: v.contiguous() for k, v in cs["gsd"].items()}, gp)
    torch.save({"model": cs["vsd"]}, vp)

    def run(eng):
        eng.encode_control(cs["img"].cuda())
        t = eng.generate(cs["emb"].cuda(), cs["n_new"], cs["mask"].cuda(), cfg_scale=cs["cfg_scale"], control_strength=cs["control_strength"])
        px = eng.vq_decode(t, cs["H"] // 16, cs["W"] // 16)
        return t.cpu(), px.cpu()

    ref = Engine(cs["cfg"], prec, weights_fp8=fp8); ref.load_state_dict(cs["gsd"]); ref.load_state_dict(cs["vsd"]); ref.finalize()
    want_t, want_px = run(ref); ref.close()
    infos = []
    for _ in range(2):                                   # first: from the files (cache miss, image written); second: cache hit
        eng = Engine(cs["cfg"], prec, weights_fp8=fp8)
        infos.append(load_engine_from_checkpoints(eng, gp, vp, cache_dir=str(tmp_path / "cache")))
        t, px = run(eng); eng.close()
        assert torch.equal(t, want_t) and torch.equal(px, want_px), infos
    assert [i["cache"] for i in infos] == ["miss", "hit"] and os.path.getsize(infos[0]["file"]) > 0
    # a cache written for another arithmetic mode / config must be refused, not loaded
    other = Engine(cs["cfg"], "fp32" if prec == "bf16" else "bf16")
    assert other.lib.car_import_packed(other._h, infos[0]["file"].encode()) != 0
    other.close()


def test_dropin_transformer_load_checkpoint(tmp_path):
    """GPT_models[...]().load_checkpoint(path) + generate(), as sample_t2i.py:56-83,163 with the file handling folded in."""
    from safetensors.torch import save_file
    from controlar_amd.generate import generate
    from controlar_amd.models import Transformer
    from tests.cases import load_case
    import numpy as np
    cs = load_case("tiny_canny_cfg1")
    p = str(tmp_path / "gpt.safetensors")
    save_file({k: v.contiguous() for k, v in cs["gsd"].items()}, p)
    os.environ["CONTROLAR_PACK_CACHE"] = str(tmp_path / "cache")
    for expect in ("miss", "hit"):
        gpt = Transformer(cs["cfg"].gpt, cs["cfg"].vit).to("cuda", dtype=torch.float32)
        res = gpt.load_checkpoint(p)
        assert not res.missing_keys and not res.unexpected_keys
        toks = generate(gpt, cs["emb"].cuda(), cs["n_new"], cs["mask"].cuda(), condition=cs["img"].cuda(), cfg_scale=1.0, sample_logits=False)
        assert np.array_equal(toks.cpu().numpy(), cs["gold"]["tokens"]) and gpt.cache_info["cache"] == expect
