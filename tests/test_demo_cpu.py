"""controlar_amd/demo.py keeps the signatures of the reference's demo entry points (demo/model.py:92-105, :192-203).  The
reference module cannot be imported here (gradio / spaces / its external Preprocessor are absent), so its function signatures
are read from the source with `ast`; the left-padding helper is checked against the reference's inline code."""
import ast
import inspect
import os

import pytest
import torch

REF = os.environ.get("CONTROLAR_REFERENCE", "/root/reference")


def _ref_params(func):
    src = open(os.path.join(REF, "demo", "model.py")).read()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.FunctionDef) and node.name == func:
            return [a.arg for a in node.args.args]
    raise AssertionError(func)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "demo")), reason="reference tree not mounted")
@pytest.mark.parametrize("func", ["process_edge", "process_depth"])
def test_demo_entry_points_keep_the_reference_signature(func):
    from controlar_amd.demo import Model
    mine = list(inspect.signature(getattr(Model, func)).parameters)
    assert mine == _ref_params(func)


def test_left_pad_caption_matches_the_reference_transform():
    from controlar_amd.demo import left_pad_caption
    g = torch.Generator().manual_seed(0)
    embs = torch.randn(3, 120, 16, generator=g)
    masks = torch.zeros(3, 120, dtype=torch.int64)
    for i, n in enumerate((1, 40, 120)):
        masks[i, :n] = 1                       # T5 pads on the right: valid tokens first
    out, new_masks = left_pad_caption(embs, masks)
    for i, n in enumerate((1, 40, 120)):
        assert torch.equal(new_masks[i], torch.flip(masks[i], dims=[-1]))
        assert torch.equal(out[i, 120 - n:], embs[i, :n])            # the caption now ends at slot 119
        assert float(out[i, :120 - n].abs().max()) == 0.0 if n < 120 else True
    with pytest.raises(RuntimeError):
        from controlar_amd.demo import Model
        Model().process_edge(torch.zeros(8, 8, 3).numpy().astype("uint8"), "a prompt", 4.0, 1.0, 2000, 1.0, 0, 100, 200, 1.0, "No preprocess")
