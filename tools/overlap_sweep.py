"""GPU probe: schedule knobs of the multi-chain decode loop at the bench shape (weights loaded once).  Not a test.

The chains of a decode step are parallel branches of one captured graph.  Forked at the same node they run in LOCKSTEP (both do
their linears together, then both their attention): nothing hides under the HBM-bound attention.  CAR_PHASE_OFFSET=1 lets chain
g+1 enter the step right after chain g's first wqkv, so one chain's linears run under the other's attention.  This probe measures
that and the neighbouring knobs, and checks that every variant with the same chain count produces the SAME tokens (the arithmetic
is untouched: only the launch schedule changes).

CAR_ATTN_PERSIST=R turns the attention into a resident grid of R workgroups per CU that walk the (sequence, head) items: a grid of
thousands of attention workgroups keeps the dispatcher busy and the other chain's linears only get slots in its tail.

usage: overlap_sweep.py [B=768] [n_new=1024] [all|quick|persist|diag|hfuse]
"""
import json
import os
os.environ["CONTROLAR_DEV_LIB"] = "1"      # the CAR_* switches exist only in the development build of the library (csrc/build.sh)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from controlar_amd import config as C, synth  # noqa: E402
from controlar_amd.engine import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
n_new = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
which = sys.argv[3] if len(sys.argv) > 3 else "all"
quick = which == "quick"
KNOBS = ["CAR_HFUSE", "CAR_NO_GRAPH", "CAR_ATTN_PERSIST", "CAR_CHAINS", "CAR_SINGLE_CHAIN", "CAR_PHASE_OFFSET", "CAR_GRAPH_STEPS", "CAR_LINEAR_PRIO", "CAR_ATTN_VARIANT", "CAR_ATTN_NSPLIT", "CAR_ATTN_LDS_PAD"]

cfg = C.xl_t2i(1024)
t0 = time.time()
gsd, _ = synth.path_state_dicts(cfg, 0)
eng = Engine(cfg, "bf16"); eng.load_state_dict(gsd); eng.finalize()
print("load %.1fs" % (time.time() - t0), flush=True)


_inputs = {}


def inputs(Bn):       # synthesised once per batch size (22 ms of host time per image)
    if Bn not in _inputs:
        if which == "diag":       # any {-1,+1} map will do for a timing diagnosis
            img = (torch.rand(Bn, 1, 512, 512, generator=torch.Generator().manual_seed(5)) > 0.92).to(torch.bfloat16).mul(2).sub(1).expand(Bn, 3, 512, 512).contiguous().cuda()
        else:
            img = synth.canny_like_control(Bn, 512, 512).to(torch.bfloat16).cuda()
        emb, mask = synth.text_embeddings(Bn, 120, 2048)
        _inputs[Bn] = (img, emb.to(torch.bfloat16).cuda(), mask.cuda())
    return _inputs[Bn]


def run(Bn, env, reps=0):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    img, emb, mask = inputs(Bn)
    eng.encode_control(img)
    best, toks = None, None
    for _ in range(reps + 1):                     # one pass is representative: the graph is captured on the host while the GPU is still in the prefill
        toks = eng.generate(emb, n_new, mask, cfg_scale=1.0); torch.cuda.synchronize()
        st = eng.stats()
        ms = st["decode_ms"] / st["decode_steps"]
        best = ms if best is None else min(best, ms)
    gbs = st["decode_algo_bytes"] / st["decode_steps"] / (best * 1e-3) / 1e9
    return best, gbs, st, toks.cpu()


def sweep(Bn, variants):
    base = {}
    for name, env in variants:
        try:
            ms, gbs, st, toks = run(Bn, env)
        except Exception as e:                    # a knob the library rejects must not end the sweep
            print(json.dumps(dict(B=Bn, variant=name, error=str(e)[:200])), flush=True)
            continue
        ng = "any" if which == "diag" else env.get("CAR_CHAINS", "1" if "CAR_SINGLE_CHAIN" in env else "default")
        ref = base.setdefault(ng, toks)           # first variant of each chain count is the reference
        print(json.dumps(dict(B=Bn, variant=name, env=env, ms_per_step=round(ms, 4), algo_GBps=round(gbs, 1), frac=round(gbs / 8000, 4),
                              kernels=st["decode_kernels_per_step"], graph=st["graph_used"],
                              tokens_equal_to_first_of_same_chain_count=bool(torch.equal(toks, ref)),
                              token_agreement=round(float((toks == ref).float().mean()), 4))), flush=True)


P = {"CAR_PHASE_OFFSET": "1"}
big = [("lockstep (round-2 default)", {}),
       ("phase", dict(P)),
       ("phase+graph8", dict(P, CAR_GRAPH_STEPS="8")),
       ("phase+prio", dict(P, CAR_LINEAR_PRIO="1")),
       ("phase+attn20", dict(P, CAR_ATTN_VARIANT="20")),
       ("phase+nsplit2", dict(P, CAR_ATTN_NSPLIT="2")),
       ("3 chains lockstep", {"CAR_CHAINS": "3"}),
       ("3 chains phase", dict(P, CAR_CHAINS="3")),
       ("4 chains phase", dict(P, CAR_CHAINS="4"))]
if quick:
    big = big[:4]
if which == "persist":
    big = [("lockstep (round-2 default)", {}),
           ("persist4+phase", dict(P, CAR_ATTN_PERSIST="4")),
           ("persist4+phase+graph8", dict(P, CAR_ATTN_PERSIST="4", CAR_GRAPH_STEPS="8")),
           ("persist5+phase", dict(P, CAR_ATTN_PERSIST="5")),
           ("persist3+phase", dict(P, CAR_ATTN_PERSIST="3")),
           ("persist6+phase", dict(P, CAR_ATTN_PERSIST="6")),
           ("persist4 lockstep", {"CAR_ATTN_PERSIST": "4"})]
if which == "hfuse":
    # horizontal fusion (experiments/hfuse_prep.patch / branch hfuse-prep applied): every attention launch carries a linear of the other chain.
    # 4-wave tiles for every linear: tokens are compared with the lockstep schedule by agreement, not bit for bit.
    big = [("lockstep (round-2 default)", {}),
           ("hfuse", {"CAR_HFUSE": "1"}),
           ("hfuse+graph8", {"CAR_HFUSE": "1", "CAR_GRAPH_STEPS": "8"})]
if which == "diag":
    # do the graph's parallel branches overlap at all?  one chain of all rows, and the two chains launched eagerly on ONE stream (strictly serial)
    big = [("2 chains, graph branches (round-2 default)", {}),
           ("1 chain", {"CAR_SINGLE_CHAIN": "1"}),
           ("2 chains, eager launches on one stream", {"CAR_NO_GRAPH": "1"})]
sweep(B, big)
if which == "all":
    # smaller batches: does the phase offset move the batch size from which two chains pay?
    for Bs in (256, 64):
        sweep(Bs, [("1 chain", {"CAR_SINGLE_CHAIN": "1"}), ("2 chains lockstep", {"CAR_CHAINS": "2"}), ("2 chains phase", dict(P, CAR_CHAINS="2")),
                   ("2 chains phase+prio", dict(P, CAR_CHAINS="2", CAR_LINEAR_PRIO="1"))])
eng.close()
