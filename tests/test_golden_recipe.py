"""The committed fixtures must be reproducible from the committed recipe: when the reference tree is present (build
container only — the GPU box has no /root/reference) re-mint small goldens into a temp dir with tests/golden/make_golden.py
and require array-for-array equality with tests/golden/*.npz."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF = os.environ.get("CONTROLAR_REFERENCE", "/root/reference")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "autoregressive")), reason="reference tree not mounted")


@pytest.mark.parametrize("name", ["tiny_canny_cfg1", "tiny_mr_192x128"])      # both decode pixels through the reference's VQModel
def test_recipe_reproduces_committed_golden(name, tmp_path):
    env = dict(os.environ, CONTROLAR_GOLDEN_OUT=str(tmp_path))
    subprocess.check_call([sys.executable, os.path.join(GOLDEN, "make_golden.py"), name], env=env, cwd=ROOT)
    new, old = np.load(tmp_path / (name + ".npz")), np.load(os.path.join(GOLDEN, name + ".npz"))
    assert set(new.files) == set(old.files)
    for k in old.files:
        assert np.array_equal(new[k], old[k]), f"{name}: array '{k}' differs from the committed fixture"
