"""CPU, world_size 2 over gloo: the data-parallel edges (broadcast -> shard -> gather) used by bench.py
at N>1 reproduce the single-process result exactly."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, G, q):
    import torch.distributed as dist
    from controlar_amd import synth
    from controlar_amd.dist import broadcast_inputs, shard_slice, gather_tokens
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H = W = 32; T, cap = 12, 64
    import controlar_amd.dist as cdist
    cdist.BCAST_CHUNK = 4096                    # the 56 KB payload goes out in 14 slices, as a 12.7 GB one does in 1 GiB slices
    packed = None
    if rank == 0:                               # bench.py's form: the global batch is written in place into one packed host buffer
        packed, h_img, h_emb, h_mask = cdist.alloc_packed_host(G, H, W, T, cap)
        h_img[:] = synth.canny_like_control(G, H, W, dtype=torch.bfloat16)
        e_, m_ = synth.text_embeddings(G, T, cap); h_emb[:] = e_.to(torch.bfloat16); h_mask[:] = m_
    img, emb, mask = broadcast_inputs(dist, torch.device("cpu"), rank, G, H, W, T, cap, packed=packed)
    # the three-tensor form must deliver the same bytes
    if rank == 0:
        i3, (e3, m3) = synth.canny_like_control(G, H, W).to(torch.bfloat16), synth.text_embeddings(G, T, cap)
        e3 = e3.to(torch.bfloat16)
    else:
        i3 = e3 = m3 = None
    i3, e3, m3 = broadcast_inputs(dist, torch.device("cpu"), rank, G, H, W, T, cap, i3, e3, m3)
    assert torch.equal(i3, img) and torch.equal(e3, emb) and torch.equal(m3, mask)
    sl = shard_slice(G, world, rank)
    # stand-in for the per-rank generate(): a deterministic function of this rank's shard only
    local = (img[sl].float().sum(dim=(1, 2, 3)).round().to(torch.int32)[:, None] +
             mask[sl].sum(dim=1).to(torch.int32)[:, None] + torch.arange(5, dtype=torch.int32)[None])
    allt = gather_tokens(dist, local)
    q.put((rank, allt.clone(), emb.float().abs().sum().item()))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_shard_gather_world2():
    from controlar_amd import synth
    G, world = 6, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, G, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in ps]
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    img = synth.canny_like_control(G, 32, 32).to(torch.bfloat16)
    emb, mask = synth.text_embeddings(G, 12, 64)
    want = (img.float().sum(dim=(1, 2, 3)).round().to(torch.int32)[:, None] + mask.sum(dim=1).to(torch.int32)[:, None] +
            torch.arange(5, dtype=torch.int32)[None])
    for rank, allt, esum in res:
        assert torch.equal(allt, want), rank                      # global order restored on every rank
        assert abs(esum - emb.to(torch.bfloat16).float().abs().sum().item()) < 1e-3


def test_shard_helpers():
    from controlar_amd.dist import shard_slice, pad_to_world
    idx = list(range(10))
    got = sorted(sum([idx[shard_slice(10, 4, r)] for r in range(4)], []))
    assert got == idx
    assert pad_to_world(10, 4) == 12 and pad_to_world(8, 4) == 8


def _timed_worker(rank, world, port, q):
    import time
    import torch.distributed as dist
    from controlar_amd.dist import timed_steps
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def step():
        calls.append(time.perf_counter())
        time.sleep(0.05 * (rank + 1))          # rank 1 is the slow one
        return rank, len(calls)

    elapsed, out = timed_steps(dist, torch.device("cpu"), step, steps=3, warmup=2, sync_fn=None)
    q.put((rank, elapsed, out, len(calls)))
    dist.barrier()
    dist.destroy_process_group()


def test_timed_steps_contract_world2():
    """bench.py's harness: W untimed + exactly K timed steps, barrier-bracketed, elapsed = MAX over ranks."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_timed_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    (r0, e0, out0, n0), (r1, e1, out1, n1) = res
    assert n0 == 5 and n1 == 5 and out0 == (0, 5) and out1 == (1, 5)      # 2 warm-up + 3 timed calls each
    assert abs(e0 - e1) < 1e-9                                              # both ranks report the same (max) time
    assert 0.29 <= e0 <= 0.6                                                # = 3 x 0.10 s of the slow rank, not 3 x 0.05


def _scatter_worker(rank, world, port, n_local, q):
    import torch.distributed as dist
    from controlar_amd import synth
    import controlar_amd.dist as cdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H = W = 32; T, cap = 12, 64
    cdist.BCAST_CHUNK = 4096                    # a shard goes out in slices, as a 1.6 GB one does in 1 GiB slices
    built = []

    def make_shard(r):                          # only the owner is ever asked; one shard at a time; the triple form is accepted too
        built.append(r)
        idx = [r + world * j for j in range(n_local)]
        img = torch.stack([synth.canny_like_control(1, H, W, seed=1234 + g, dtype=torch.bfloat16)[0] for g in idx])
        em = [synth.text_embeddings(1, T, cap, seed=1234 + g) for g in idx]
        return img, torch.stack([e[0][0] for e in em]).to(torch.bfloat16), torch.stack([e[1][0] for e in em])
    img, emb, mask = cdist.scatter_inputs(dist, torch.device("cpu"), rank, world, n_local, H, W, T, cap, make_shard)
    local = (img.float().sum(dim=(1, 2, 3)).round().to(torch.int32)[:, None] + mask.sum(dim=1).to(torch.int32)[:, None] + torch.arange(5, dtype=torch.int32)[None])
    allt = cdist.gather_tokens(dist, local)
    # the wire-only probe bench.py runs beside `--input-dist local`: one shard-sized buffer from rank 0 to every other rank, MAX-over-ranks time
    t_probe, wire = cdist.scatter_probe(dist, torch.device("cpu"), rank, world, 3 * 4096 + 8)
    q.put((rank, allt.clone(), emb.float().abs().sum().item(), list(built), t_probe, wire))
    dist.barrier()
    dist.destroy_process_group()


def test_scatter_shards_gather_world2():
    """bench.py's default input edge: rank 0 owns the inputs, builds ONE shard at a time and sends it point-to-point; every rank ends up with
    exactly its strided shard (images r, r+W, ...), and the gathered tokens come back in global image order."""
    from controlar_amd import synth
    n_local, world = 3, 2
    G = n_local * world
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_scatter_worker, args=(r, world, port, n_local, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in ps]
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    img = synth.canny_like_control(G, 32, 32).to(torch.bfloat16)
    emb, mask = synth.text_embeddings(G, 12, 64)
    want = (img.float().sum(dim=(1, 2, 3)).round().to(torch.int32)[:, None] + mask.sum(dim=1).to(torch.int32)[:, None] +
            torch.arange(5, dtype=torch.int32)[None])
    assert len({r[4] for r in res}) == 1 and res[0][4] > 0 and all(r[5] == (3 * 4096 + 8) * (world - 1) for r in res)      # same (max) time on every rank; (W-1) shards on the wire
    for rank, allt, esum, built, _t, _w in res:
        assert torch.equal(allt, want), rank
        assert built == ([0, 1] if rank == 0 else []), (rank, built)               # only the owner builds shards, in rank order
        mine = emb[rank::world].to(torch.bfloat16).float().abs().sum().item()
        assert abs(esum - mine) < 1e-3, rank                                        # each rank holds ITS shard, not the global batch


def test_parallel_fill_draws_the_same_shard():
    """bench.py draws its per-image inputs on a thread pool (controlar_amd.dist.parallel_fill): independent seeded generators, disjoint slices —
    the shard must be byte-identical to the serial draw, and a single process reports a zero-byte scatter probe."""
    from controlar_amd import synth
    import controlar_amd.dist as cdist
    n, H, W, T, cap = 12, 32, 32, 12, 64
    outs = []
    for workers in (1, 4):
        packed, h_img, h_emb, h_mask = cdist.alloc_packed_host(n, H, W, T, cap)

        def one(j):
            h_img[j] = synth.canny_like_control(1, H, W, seed=1234 + j, dtype=torch.bfloat16)[0]
            e_, m_ = synth.text_embeddings(1, T, cap, seed=1234 + j)
            h_emb[j] = e_[0].to(torch.bfloat16); h_mask[j] = m_[0]
        before = torch.get_num_threads()
        assert cdist.parallel_fill(n, one, workers=workers) == workers
        assert torch.get_num_threads() == before
        outs.append(packed.clone())
    assert torch.equal(outs[0], outs[1])
    assert cdist.scatter_probe(None, torch.device("cpu"), 0, 1, 1024) == (0.0, 0)


def test_gpus_flag_respawns_under_torchrun():
    """`script --gpus 2` started as a plain process (no WORLD_SIZE) must become two ranks by itself (bench.py's start-up): the probe
    re-executes under `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1`, scatters, gathers, and rank 0
    reports n_gpus = 2 with the tokens of all 6 images in global order."""
    import json
    import subprocess
    import sys
    from controlar_amd import synth
    from controlar_amd.dist import respawn_command
    cmd = respawn_command("bench.py", ["--gpus", "4", "--steps", "3"], 4, port=29999)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_spawn_probe.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, probe, "--gpus", "2", "--batch", "3"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["master"] == "127.0.0.1"
    G = 6
    img = synth.canny_like_control(G, 32, 32).to(torch.bfloat16)
    _, mask = synth.text_embeddings(G, 12, 64)
    want = (img.float().sum(dim=(1, 2, 3)).round().to(torch.int32)[:, None] + mask.sum(dim=1).to(torch.int32)[:, None] + torch.arange(4, dtype=torch.int32)[None])
    assert rec["tokens"] == want.tolist()
    # --gpus 1, or an environment that already carries WORLD_SIZE (the driver's torchrun launch): no respawn
    out1 = subprocess.run([sys.executable, probe, "--gpus", "1", "--batch", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert out1.returncode == 0 and json.loads([ln for ln in out1.stdout.splitlines() if ln.startswith("{")][-1])["n_gpus"] == 1


def _weights_once_worker(rank, world, port, tmpdir, q):
    import torch.distributed as dist
    import controlar_amd.dist as cdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    files = [os.path.join(tmpdir, "gpt.carpk"), os.path.join(tmpdir, "vq.carpk")]
    calls = []

    def build_and_export():
        calls.append("build+export")
        for f in files:
            with open(f + ".tmp", "w") as fh:
                fh.write("weights of build X")
            os.replace(f + ".tmp", f)

    def import_packed():
        calls.append("import")
        for f in files:
            if open(f).read() != "weights of build X":
                raise RuntimeError("foreign file")
    how1 = cdist.load_weights_once(dist, rank, files, build_and_export, import_packed, build_only=lambda: calls.append("build"))
    how2 = cdist.load_weights_once(dist, rank, files, build_and_export, import_packed, build_only=lambda: calls.append("build"))     # second start: the files exist
    dist.barrier()
    if rank == 0:
        open(files[1], "w").write("garbage")                                                                                     # a stale / foreign image
    dist.barrier()
    how3 = cdist.load_weights_once(dist, rank, files, build_and_export, import_packed, build_only=lambda: calls.append("build"))
    q.put((rank, how1, how2, how3, list(calls)))
    dist.barrier()
    dist.destroy_process_group()


def test_weights_are_built_once_per_node_world2(tmp_path):
    """bench.py at N > 1: rank 0 synthesises / packs the weights once and exports the packed images, the other ranks import them after a barrier (never before the
    files are complete); existing files are re-used; a rank that cannot import falls back to building."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_weights_once_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    (_, a1, a2, a3, c0), (_, b1, b2, b3, c1) = res
    assert (a1, b1) == ("built", "imported") and (a2, b2) == ("imported", "imported")
    assert a3 == "built" and b3 == "imported"            # rank 0 found a foreign file, rebuilt and re-exported before the barrier
    assert c0 == ["build+export", "import", "import", "build+export"] and c1 == ["import", "import", "import"]
