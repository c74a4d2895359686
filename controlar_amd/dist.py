"""Data-parallel edges of the path.  The reference shards images across ranks with
no collective inside generation (autoregressive/sample/sample_t2i_ddp.py:127-170, index
i*world+rank at :131); north_star adds the distribution of text/control inputs from the rank that owns
them before generation and a gather of tokens after it.  Two forms of the input edge:
  * scatter_inputs  — rank `src` builds ONE shard at a time and sends it point-to-point to its rank: (W-1)/W of the global batch
                      crosses xGMI once, and `src` never holds more than one shard (bench.py's default);
  * broadcast_inputs — everyone receives the whole global batch in one packed broadcast and slices its shard: the right form for a
                      1-GPU-sized batch (tens of MB, latency-bound), W-1 times the bytes for a large one.
On ROCm backend "nccl" is RCCL (xGMI); the same code runs on gloo for the CPU tests.  `dist` may be None (single process):
every function degrades to a no-op.
"""
from __future__ import annotations

import os
import socket
import sys

import torch


def respawn_command(script: str, argv, gpus: int, port: int | None = None):
    """The command line that runs `script argv...` as `gpus` ranks of ONE node, one process per GPU, rendezvous on 127.0.0.1 (the container
    hostname may not resolve) — what the driver itself runs for N > 1."""
    if port is None:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(script)] + list(argv)


def respawn_under_torchrun(script: str, argv, gpus: int) -> None:
    """`script --gpus N` started as a plain process (no WORLD_SIZE in the environment) with N > 1: replace this process by
    `python -m torch.distributed.run ... script argv` so that N ranks run and rank 0 prints the result.  Returns only when no respawn is needed."""
    if gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL / device-tensor sharing across processes needs it on this stack
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(gpus, 1))))
    cmd = respawn_command(script, argv, gpus)
    sys.stdout.flush(); sys.stderr.flush()
    os.execvpe(cmd[0], cmd, env)


def shard_slice(G: int, world: int, rank: int) -> slice:
    """Strided shard r, r+W, r+2W, ... (mirrors sample_t2i_ddp.py:131)."""
    return slice(rank, G, world)


def pad_to_world(G: int, world: int) -> int:
    """'sample a bit more than we need' (sample_t2i_ddp.py:116-123): round the global batch up."""
    return (G + world - 1) // world * world


def packed_layout(G: int, H: int, W: int, T: int, cap: int):
    """Byte sizes (img bf16 [G,3,H,W], emb bf16 [G,T,cap], mask int64 [G,T]) of the packed input buffer; every section starts 8-byte aligned."""
    n_img, n_emb, n_mask = G * 3 * H * W * 2, G * T * cap * 2, G * T * 8
    if n_img % 8 or n_emb % 8:
        raise ValueError(f"packed input sections must be 8-byte aligned (G={G}, H={H}, W={W}, T={T}, cap={cap}): pad the batch or the caption width")
    return n_img, n_emb, n_mask


def packed_views(buf: torch.Tensor, G: int, H: int, W: int, T: int, cap: int):
    """img / emb / mask views into one packed uint8 buffer (host or device)."""
    n_img, n_emb, _ = packed_layout(G, H, W, T, cap)
    img = buf[:n_img].view(torch.bfloat16).view(G, 3, H, W)
    emb = buf[n_img:n_img + n_emb].view(torch.bfloat16).view(G, T, cap)
    mask = buf[n_img + n_emb:].view(torch.int64).view(G, T)
    return img, emb, mask


def alloc_packed_host(G: int, H: int, W: int, T: int, cap: int):
    """Rank `src` fills the returned views in place (chunk by chunk): the global batch of an 8-GPU run is 6144 images = 12.7 GB, built once,
    with no list-of-chunks + torch.cat copies beside it."""
    buf = torch.empty(sum(packed_layout(G, H, W, T, cap)), dtype=torch.uint8)
    return (buf,) + packed_views(buf, G, H, W, T, cap)


BCAST_CHUNK = 1 << 30       # bytes per collective call: one logical broadcast, issued in <= 1 GiB slices (element counts stay far below 2^31)


def broadcast_inputs(dist, device, rank: int, G: int, H: int, W: int, T: int, cap: int, img=None, emb=None, mask=None, src: int = 0, packed=None):
    """Rank `src` holds img [G,3,H,W] bf16, emb [G,T,cap] bf16, mask [G,T] int64 (either as three tensors or already packed by
    alloc_packed_host: `packed`); everyone gets all three through ONE packed byte buffer broadcast over RCCL/xGMI (payloads of a
    1-GPU-sized batch are tens of MB: latency-bound, so one message beats three; large global batches go out in 1 GiB slices)."""
    total = sum(packed_layout(G, H, W, T, cap))
    if rank == src:
        if packed is None:
            packed = torch.cat([img.contiguous().view(torch.uint8).reshape(-1), emb.contiguous().view(torch.uint8).reshape(-1),
                                mask.contiguous().view(torch.uint8).reshape(-1)])
        assert packed.numel() == total and packed.dtype == torch.uint8
        buf = torch.empty(total, dtype=torch.uint8, device=device)
        for o in range(0, total, BCAST_CHUNK):                      # host -> device in slices too: no second full-size staging copy
            buf[o:o + BCAST_CHUNK].copy_(packed[o:o + BCAST_CHUNK])
    else:
        buf = torch.empty(total, dtype=torch.uint8, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        for o in range(0, total, BCAST_CHUNK):
            dist.broadcast(buf[o:o + BCAST_CHUNK], src=src)
    return packed_views(buf, G, H, W, T, cap)


def scatter_inputs(dist, device, rank: int, world: int, n_local: int, H: int, W: int, T: int, cap: int, make_shard, src: int = 0):
    """Rank `src` owns the inputs (the serving front-end: T5 features + control maps).  For every destination rank r it calls
    `make_shard(r)` -> (packed uint8 host buffer, or the (img, emb, mask) triple) of THAT rank's n_local images, copies it to the
    device and sends it point-to-point (<= 1 GiB slices); its own shard is kept.  Returns the local (img, emb, mask) views.
    Wire bytes: (W-1) shards, once — against (W-1) x W shards for broadcast-then-slice."""
    total = sum(packed_layout(n_local, H, W, T, cap))
    multi = dist is not None and dist.is_initialized() and world > 1

    def to_packed(x):
        if isinstance(x, torch.Tensor):
            assert x.dtype == torch.uint8 and x.numel() == total
            return x
        img, emb, mask = x
        return torch.cat([img.contiguous().view(torch.uint8).reshape(-1), emb.contiguous().view(torch.uint8).reshape(-1),
                          mask.contiguous().view(torch.uint8).reshape(-1)])
    mine = torch.empty(total, dtype=torch.uint8, device=device)
    if rank == src:
        stage = torch.empty(total, dtype=torch.uint8, device=device) if multi else None
        for r in range(world):
            host = to_packed(make_shard(r))
            dst = mine if r == src else stage
            for o in range(0, total, BCAST_CHUNK):
                dst[o:o + BCAST_CHUNK].copy_(host[o:o + BCAST_CHUNK])
            if r != src:
                for o in range(0, total, BCAST_CHUNK):
                    dist.send(stage[o:o + BCAST_CHUNK], dst=r)
    elif multi:
        for o in range(0, total, BCAST_CHUNK):
            dist.recv(mine[o:o + BCAST_CHUNK], src=src)
    return packed_views(mine, n_local, H, W, T, cap)


def scatter_probe(dist, device, rank: int, world: int, nbytes: int, src: int = 0, sync_fn=None):
    """The wire half of scatter_inputs alone: rank `src` sends ONE pre-built shard-sized device buffer to every other rank point-to-point (<= 1 GiB slices),
    bracketed by barriers; returns (seconds as the MAX over ranks, bytes that crossed the wire) or (0.0, 0) for a single process.  bench.py runs it beside the
    default `--input-dist local` so that north_star's input edge (T5 / control embeddings from the rank that owns them, over RCCL / xGMI) gets a hardware
    number without the host-side synthesis cost of building W real shards on one rank."""
    import time
    if dist is None or not dist.is_initialized() or world <= 1:
        return 0.0, 0
    sync = sync_fn if sync_fn is not None else (lambda: None)
    buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if rank == src:
        buf.fill_(rank + 1)
    sync(); dist.barrier(); sync()
    t0 = time.perf_counter()
    if rank == src:
        for r in range(world):
            if r != src:
                for o in range(0, nbytes, BCAST_CHUNK):
                    dist.send(buf[o:o + BCAST_CHUNK], dst=r)
    else:
        for o in range(0, nbytes, BCAST_CHUNK):
            dist.recv(buf[o:o + BCAST_CHUNK], src=src)
    sync(); dist.barrier(); sync()
    t = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert int(buf[0].item()) == src + 1 and int(buf[-1].item()) == src + 1        # the payload arrived
    return float(t.item()), nbytes * (world - 1)


def parallel_fill(n: int, fill_one, workers: int = 0):
    """fill_one(j) for j in range(n) on a small thread pool with torch's intra-op pool pinned to one thread (the per-image draws of bench.py are
    independent seeded generators writing disjoint slices; torch's CPU ops release the GIL).  Cuts the start-up of a 768-image shard from ~32 s to a few
    seconds on the GPU box's host — eight ranks of one node do this at the same time."""
    import concurrent.futures as cf
    if workers <= 0:
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        workers = max(1, min(16, cores // max(world, 1)))
    if workers == 1 or n < 4:
        for j in range(n):
            fill_one(j)
        return workers
    prev = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        with cf.ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(fill_one, range(n)))
    finally:
        torch.set_num_threads(prev)
    return workers


def gather_tokens(dist, local_tokens: torch.Tensor) -> torch.Tensor:
    """all_gather of the per-rank token blocks, re-interleaved to global image order
    (inverse of shard_slice)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_tokens
    world = dist.get_world_size()
    parts = [torch.empty_like(local_tokens) for _ in range(world)]
    dist.all_gather(parts, local_tokens.contiguous())
    out = torch.stack(parts, dim=1)                       # [per_rank, world, N]: image r + W*i lives at [i, r]
    return out.reshape(-1, local_tokens.shape[-1])


def timed_steps(dist, device, step_fn, steps: int, warmup: int, sync_fn=None):
    """The bench contract's timing harness: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by a barrier and a
    device synchronise on both sides; returns (elapsed seconds as the MAX over ranks, last step's return value).
    `sync_fn` = torch.cuda.synchronize on GPU ranks, None on CPU (gloo tests)."""
    import time
    multi = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
    sync = sync_fn if sync_fn is not None else (lambda: None)
    out = None
    for _ in range(warmup):
        out = step_fn()
    sync()
    if multi:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step_fn()
    sync()
    if multi:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out


def load_weights_once(dist, rank: int, files, build_and_export, import_packed, build_only=None):
    """N ranks of one node, one set of weights: rank 0 builds them ONCE (checkpoint load or synthesis, device-side packing, car_finalize_weights) and writes the packed
    images (`build_and_export()` -> files under a directory every rank can read, written atomically); after a barrier every other rank restores them with plain
    copies (`import_packed()`, car_import_packed: ~seconds) instead of repeating 10+ s of host work per rank on a host that eight ranks share.  A rank whose import
    fails (foreign build id, truncated file) falls back to `build_only()`, the single-process path.  `dist` None or one rank: `build_only()`.
    Returns "built" | "imported" | "fallback"."""
    import os
    build_only = build_only or build_and_export
    if dist is None or dist.get_world_size() == 1:
        build_only()
        return "built"
    if rank == 0:
        if all(os.path.exists(f) for f in files):
            try:
                import_packed(); how = "imported"
            except Exception:
                build_and_export(); how = "built"
        else:
            build_and_export(); how = "built"
        dist.barrier()
        return how
    dist.barrier()
    try:
        import_packed()
        return "imported"
    except Exception:
        build_only()
        return "fallback"
