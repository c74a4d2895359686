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


def test_demo_process_edge_and_depth_end_to_end():
    """controlar_amd.demo.Model.process_edge / process_depth (demo/model.py:92-188, :192-284) on a small graph at the demo's fixed 512x512:
    control map in, [control map, generated image] out as PIL images; the seed argument makes a call reproducible."""
    import numpy as np
    from controlar_amd import config as C, synth
    from controlar_amd.demo import Model
    from controlar_amd.models import Transformer, VQModel
    cfg = C.tiny_t2i(1024, "canny")
    gsd, vsd = synth.path_state_dicts(cfg, seed=0)
    gpt = Transformer(cfg.gpt, cfg.vit).to("cuda", dtype=torch.bfloat16); gpt.load_state_dict(gsd, strict=False)
    vq = VQModel(cfg.vq); vq.to("cuda"); vq.load_state_dict(vsd)
    m = Model(gpt_edge=gpt, gpt_depth=gpt, vq_model=vq)
    ctrl = ((synth.canny_like_control(1, 512, 512)[0] * 0.5 + 0.5) * 255).permute(1, 2, 0).numpy().astype(np.uint8)
    emb, mask = synth.text_embeddings(1, cfg.gpt.cls_token_num, cfg.gpt.caption_dim)
    feats = (torch.flip(emb * mask[:, :, None], dims=[1]), torch.flip(mask, dims=[-1]))      # T5 layout: valid tokens first
    a = m.process_edge(ctrl, feats, 4.0, 1.0, 2000, 1.0, 3, 100, 200, 0.6, "No preprocess")
    b = m.process_edge(ctrl, feats, 4.0, 1.0, 2000, 1.0, 3, 100, 200, 0.6, "No preprocess")
    c = m.process_depth(ctrl, feats, 4.0, 1.0, 2000, 1.0, 4, 0.6, "No preprocess")
    assert len(a) == 2 and a[0].size == (512, 512) and a[1].size == (512, 512)
    assert np.array_equal(np.array(a[0]), ctrl)                                            # the control map comes back first (demo/model.py:178)
    assert np.array_equal(np.array(a[1]), np.array(b[1])) and not np.array_equal(np.array(a[1]), np.array(c[1]))
    with pytest.raises(RuntimeError):
        m.process_edge(ctrl, "a text prompt needs the injected T5 encoder", 4.0, 1.0, 2000, 1.0, 3, 100, 200, 0.6, "No preprocess")
    d = m.process_edge(ctrl, feats, 4.0, 1.0, 2000, 1.0, 3, 100, 200, 0.6, "Canny")     # built-in GPU Canny on the photo
    assert len(d) == 2 and set(np.unique(np.array(d[0]))) <= {0, 255}
    with pytest.raises(RuntimeError):
        m.process_edge(ctrl, feats, 4.0, 1.0, 2000, 1.0, 3, 100, 200, 0.6, "Hed")       # extractor not injected


def test_corrupt_or_foreign_cache_files_are_refused_and_the_loader_falls_back(tmp_path):
    """car_import_packed validates everything it reads (sizes against the file and against numel, the build id in the header)
    before touching the context; a refused file makes load_engine_from_checkpoints fall back to a normal load and rewrite it."""
    import struct
    from safetensors.torch import save_file
    from controlar_amd.checkpoint import load_engine_from_checkpoints
    from controlar_amd.engine import Engine
    from tests.cases import load_case
    cs = load_case("tiny_canny_cfg1")
    gp = str(tmp_path / "gpt.safetensors")
    save_file({k: v.contiguous() for k, v in cs["gsd"].items()}, gp)
    cdir = str(tmp_path / "cache")
    eng = Engine(cs["cfg"], "bf16")
    info = load_engine_from_checkpoints(eng, gp, None, cache_dir=cdir)
    assert info["cache"] == "miss"
    eng.close()
    good = open(info["file"], "rb").read()
    hdr = 8 + 48 + len(bytes(eng._cc)) + 8                    # magic, build id, car_config, entry count
    # first entry: kind u32, name_len u32, name, ndim u32, shape i64 x ndim, numel i64, bytes u64, payload
    nl = struct.unpack_from("<I", good, hdr + 4)[0]
    nd = struct.unpack_from("<I", good, hdr + 8 + nl)[0]
    off_bytes = hdr + 8 + nl + 4 + 8 * nd + 8
    cases = {
        "truncated": good[: len(good) // 2],
        "huge_size": good[:off_bytes] + struct.pack("<Q", 1 << 60) + good[off_bytes + 8:],
        "size_not_numel": good[:off_bytes] + struct.pack("<Q", struct.unpack_from("<Q", good, off_bytes)[0] - 2) + good[off_bytes + 8:],
        "other_build": good[:8] + b"0" * 40 + good[48:],
        "entry_count": good[:hdr - 8] + struct.pack("<Q", 1 << 40) + good[hdr:],
    }
    for name, blob in cases.items():
        bad = str(tmp_path / f"{name}.carpk")
        open(bad, "wb").write(blob)
        e2 = Engine(cs["cfg"], "bf16")
        assert e2.lib.car_import_packed(e2._h, bad.encode()) != 0, name
        assert b"car_import_packed" in e2.lib.car_last_error(e2._h), name
        e2.close()
    open(info["file"], "wb").write(cases["huge_size"])            # the cache entry itself goes bad: fall back, rewrite, then hit again
    for expect in ("miss", "hit"):
        e3 = Engine(cs["cfg"], "bf16")
        assert load_engine_from_checkpoints(e3, gp, None, cache_dir=cdir)["cache"] == expect
        e3.encode_control(cs["img"].cuda())
        t = e3.generate(cs["emb"].cuda(), cs["n_new"], cs["mask"].cuda(), cfg_scale=1.0)
        assert t.shape == (cs["img"].shape[0], cs["n_new"])
        e3.close()
