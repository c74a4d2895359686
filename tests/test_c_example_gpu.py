"""The C ABI from plain C (pytest -m gpu): examples/c_abi_run.c is compiled with gcc (-std=c99), linked against
libcontrolar_hip.so + the HIP runtime, run on a model/input dump, and must produce the tokens the Python shim produces."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_client_links_runs_and_matches(tmp_path):
    from controlar_amd.engine import Engine
    from tests.cases import load_case
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("no gcc / ROCm headers")
    cs = load_case("tiny_canny_cfg1")
    eng = Engine(cs["cfg"], "fp32")
    dump, out, exe = tmp_path / "dump.bin", tmp_path / "tokens.bin", tmp_path / "c_abi_run"
    with open(dump, "wb") as f:
        f.write(struct.pack("<i", 0x43415231)); f.write(bytes(eng._cc))
        sd = {**cs["gsd"], **cs["vsd"]}
        sd = {k: v for k, v in sd.items() if torch.is_floating_point(v)}
        f.write(struct.pack("<i", len(sd)))
        for k, v in sd.items():
            kb = k.encode(); a = v.detach().float().contiguous().numpy()
            f.write(struct.pack("<i", len(kb))); f.write(kb); f.write(struct.pack("<i", a.ndim)); f.write(np.asarray(a.shape, dtype=np.int64).tobytes()); f.write(a.tobytes())
        B, H, W = cs["B"], cs["H"], cs["W"]; T, cap = cs["cfg"].gpt.cls_token_num, cs["cfg"].gpt.caption_dim
        f.write(struct.pack("<6i", B, H, W, T, cap, cs["n_new"]))
        f.write(cs["img"].float().contiguous().numpy().tobytes()); f.write(cs["emb"].float().contiguous().numpy().tobytes())
        f.write(cs["mask"].to(torch.int64).contiguous().numpy().tobytes()); f.write(struct.pack("<f", cs["cfg_scale"]))
    eng.close()
    libdir = os.path.join(ROOT, "controlar_amd", "csrc")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "c_abi_run.c"), "-L", libdir, "-lcontrolar_hip", "-L/opt/rocm/lib", "-lamdhip64",
                        f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe), str(dump), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    raw = open(out, "rb").read()
    toks = np.frombuffer(raw[:-8], dtype=np.int32).reshape(cs["B"], cs["n_new"])
    assert np.array_equal(toks, cs["gold"]["tokens"])                          # exact mode: the reference's greedy tokens
    assert np.isfinite(struct.unpack("<d", raw[-8:])[0])
