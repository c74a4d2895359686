#!/usr/bin/env python3
"""bench.py — images/sec/GPU of the ControlAR conditional-decoding hot path on MI355X.

Workload (BASELINE.json metric): LlamaGen-XL t2i + DINOv2-small canny control, 512x512
(1024 tokens), synthetic inputs and random-init weights of that architecture.
One "step" = one pass of the whole path over one batch per GPU:
    control encoder (resize + DINOv2-S + adapter MLP) -> generate() (text embed, control MLPs,
    prefill, 1023 greedy decode steps) -> VQ decode to 512x512 pixels,
with inputs already resident in HBM when the timed region starts.

Multi-GPU: pure data parallel, one process per GPU.  `--gpus N` started as a plain process re-executes
itself under `python -m torch.distributed.run` (N ranks, 127.0.0.1 rendezvous); under torchrun the
environment decides.  Every rank draws its own strided shard of the global batch (per-image seeds: image g
is the same for any N — the reference's DDP sampler, sample_t2i_ddp.py:127-170, has no input collective
either), generates it with no collective inside the path, and the tokens are all-gathered over RCCL at the
end (timed, reported).  `--input-dist scatter`: rank 0 owns the inputs (a serving front-end holding the T5
features and control maps) and sends every rank ITS shard point-to-point over RCCL/xGMI, one shard built and
sent at a time; timed and reported.  weak scaling.

Prints ONE JSON line (rank 0) with `roofline` (decode step vs the HBM roofline, HIP-event
timed inside the library on its own stream) and `cpu_baseline` (the CPU oracle, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=768, help="images per GPU per step (768 sequences = 163 GB of bf16 KV cache in two chains of 384: sized for the 288 GB of one MI355X; measured plateau, profiles/r02_decode_batch_sweep.txt)")
    ap.add_argument("--cfg-scale", type=float, default=1.0)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"],
                    help="fp32 = the exact mode: greedy tokens bit-identical to the reference's fp32 CPU path (`--precision none` of sample_t2i.py:197); "
                         "decode linears on the exact fp32 MFMA (decode_f32.hip); batch defaults to 384 (fp32 KV cache = 162 GB)")
    ap.add_argument("--vq-precision", default=None, choices=["bf16", "fp32"],
                    help="arithmetic of the VQ decoder context (the reference keeps vq_model a separate module).  Default: bf16 — north_star grades pixels by "
                         "tolerance and tokens bit-exactly, so `--precision fp32` alone is the tokens-exact configuration; pass fp32 for fp32 pixels as well")
    ap.add_argument("--model", default="xl", choices=["xl", "b", "tiny", "b_c2i", "l_c2i"])
    ap.add_argument("--image-size", type=int, default=512)
    ap.add_argument("--image-h", type=int, default=0, help="non-square (MR) height; token grid = H/16 x W/16, rope grid = max side (sample_t2i_MR.py:73-78)")
    ap.add_argument("--image-w", type=int, default=0)
    ap.add_argument("--weights-fp8", action="store_true", help="e4m3 decode weights, weight-only (widened to bf16 in registers, bf16 MFMA)")
    ap.add_argument("--fp8-mfma", action="store_true", help="BASELINE config 5: e4m3 decode weights x e4m3 activations on the fp8 MFMA (W8A8)")
    ap.add_argument("--kv-fp8", action="store_true",
                    help="OPT-IN, not the headline: e4m3 KV cache (car_config.kv_cache_fp8) — K / V stored as OCP e4m3 bytes, widened to bf16 in registers; halves the "
                         "KV stream of the decode step.  The reference has no such mode; tolerance-graded against the oracle's model of it (tests/test_configs_gpu.py)")
    ap.add_argument("--condition-type", default="canny", help="'canny'/'seg' -> nearest resize, anything else -> bicubic (dinov2_adapter.py:19-23)")
    ap.add_argument("--adapter-size", default="small", choices=["small", "base"])
    ap.add_argument("--sample-logits", action="store_true",
                    help="stochastic decoding as every script of the reference runs it (sample_t2i.py:163-170: sample_logits=True, top_k=2000, top_p=1, temperature=1) "
                         "instead of the greedy parity setting; on-device sampler (top-k / top-p by in-LDS sort, Philox keyed by seed, row, step)")
    ap.add_argument("--top-k", type=int, default=2000)
    ap.add_argument("--top-p", type=float, default=1.0)
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--input-dist", default="local", choices=["scatter", "local"],
                    help="N > 1: 'local' = every rank draws its own shard (per-image seeds: the same global batch for any N; no input traffic — the "
                         "reference's DDP sampler); 'scatter' = rank 0 owns the inputs (serving front-end) and sends each rank its shard over RCCL, "
                         "timed and reported.  Synthesis costs ~40 ms per image on the GPU box's host, so a rank-0-owned batch of 8 x 768 images "
                         "takes minutes to draw before anything is sent: not the default")
    ap.add_argument("--overlap-vq", action="store_true",
                    help="run the VQ decode of batch i on a side stream under the token loop of batch i+1 (measured: no gain on MI355X — "
                         "the token loop is HBM-bound and the decoder's GEMM grids take every CU; kept for experiments)")
    ap.add_argument("--cpu-tokens", type=int, default=48, help="decode tokens timed by the CPU baseline sample")
    ap.add_argument("--exact-leg-steps", type=int, default=3,
                    help="headline configuration on one GPU only: after the timed region, time this many steps (+ 1 warm-up) of the BIT-IDENTICAL mode (`--precision fp32`, 384 "
                         "images) in a child process and report it as the `exact` block of the same JSON line — north_star asks for >= 20 images/s WITH the reference's greedy "
                         "tokens; the bf16 headline is tolerance-graded, this leg is the one whose tokens are compared with the reference's.  0 = skip")
    ap.add_argument("--config-legs", default="2,3,5,4,1",
                    help="headline configuration on one GPU only: after the exact leg, time these BASELINE configs (`--config N`, 1 warm-up + --config-leg-steps steps each) in child "
                         "processes and report them as the `configs` block of the same JSON line, so that the driver's record carries every BASELINE config, not only the headline.  '' = skip")
    ap.add_argument("--config-leg-steps", type=int, default=3)
    ap.add_argument("--legs-budget-s", type=float, default=420.0, help="stop starting further config legs once this much wall clock has gone into them")
    ap.add_argument("--pack-cache", action="store_true",
                    help="N = 1: restore the packed weight images from the bench cache directory if this exact configuration was exported there (by the headline run of the same "
                         "build), else build and export — the config legs use it so that a leg on the headline's model does not re-synthesise 750 M parameters")
    ap.add_argument("--fp8-weight-only", action="store_true", help="--config 5: weight-only e4m3 (bf16 MFMA) instead of the W8A8 form BASELINE names; reported as a variant by the default --config 5 run")
    ap.add_argument("--no-variants", action="store_true", help="skip the child-process legs (exact block, config-5 variant)")
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4, 5],
                    help="one of BASELINE.json's other configs (SURVEY §8d'): 1 = LlamaGen-B c2i 256x256 class-conditional + ViT-S/16 control, cfg 1, 4 images (the reference's "
                         "CPU-runnable case); 2 = cfg 4 batch 1; 3 = DINOv2-base depth cfg 4, 32 images/GPU; "
                         "4 = MR 768x512 cfg 4 batch 1; 5 = edge_base fp8 weights batch 8 (weight-only; --fp8-mfma for W8A8).  0 = the headline metric config")
    a = ap.parse_args()
    if a.config == 1:
        a.model, a.image_size, a.cfg_scale = "b_c2i", 256, 1.0
        if a.batch == 768:
            a.batch = 4                                   # the reference's four fixtures (sample_c2i.py:96-106)
    if a.config == 2:
        a.cfg_scale, a.batch = 4.0, 1
    elif a.config == 3:
        a.cfg_scale, a.batch, a.condition_type, a.adapter_size = 4.0, 32, "depth", "base"
    elif a.config == 4:
        a.cfg_scale, a.batch, a.image_h, a.image_w = 4.0, 1, 768, 512
    elif a.config == 5:
        # BASELINE configs[4] names "fp8 weights on CDNA4 fp8 MFMA": e4m3 weights x e4m3 activations on v_mfma_f32_16x16x32_fp8_fp8 (W8A8) is what runs; the
        # weight-only form (widened to bf16 in registers, bf16 MFMA: pinned against the dequantised-weight oracle within the bf16 tolerance) is timed in a child
        # process and reported beside it as config.variants (`--fp8-weight-only` selects it directly)
        a.batch, a.weights_fp8, a.condition_type, a.adapter_size = 8, True, "hed", "base"
        if not a.fp8_weight_only:
            a.fp8_mfma = True
    if a.fp8_mfma:
        a.weights_fp8 = True
    if a.precision == "fp32" and a.batch == 768 and a.config == 0:
        a.batch = 384                                     # fp32 KV rows are twice as wide: 384 sequences = 162 GB
    if a.vq_precision is None:
        a.vq_precision = "bf16"
    return a


def refuse_debug_environment():
    """The library keeps a few A/B and profiling switches behind CAR_* environment variables.  Some of them change WHAT is
    computed (CAR_DEBUG_SKIP_STEPS starts the token loop late) or how it is launched (CAR_NO_GRAPH): a number measured under any of them is
    not the benchmark, so bench.py does not run with them set."""
    bad = sorted(k for k in os.environ if k.startswith("CAR_"))
    if bad:
        raise SystemExit(f"bench.py: refusing to run with library debug / schedule switches set in the environment: {', '.join(bad)}")


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores() -> int:
    """Usable host cores: affinity mask capped by the cgroup CPU quota (os.cpu_count() reports the whole
    host inside a container and oversubscribing torch's CPU pool is pathologically slow)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, min(n, 32))


def cpu_baseline(cfg, gsd, vsd, H, W, n_tok_sample):
    """The CPU oracle ("port" of the reference algorithm) on this box's host cores, bounded sample:
    control encoder + prefill + `n_tok_sample` decode steps + VQ decode for ONE image, extrapolated
    to the full 1024-token image by per-token cost."""
    from oracle import controlar_oracle as O
    from controlar_amd import synth
    cores = host_cores()
    # the decode step on a CPU is a chain of GEMVs over 3 GB of fp32 weights: memory-bound, and on a shared host MORE threads can be slower
    # (BENCH_r02: 161 ms/token on "16 cores" against 51 ms/token on 8 cores of the build container).  Pick the thread count by a 1-second probe.
    probe_w = torch.randn(4096, 8192)
    probe_x = torch.randn(8192, 1)
    best = (None, 1e9)
    for nt in sorted({c for c in (4, 8, 16, 32, cores) if c <= cores}):
        torch.set_num_threads(nt)
        torch.mm(probe_w, probe_x)
        t0 = time.perf_counter()
        for _ in range(6):
            torch.mm(probe_w, probe_x)
        dt = (time.perf_counter() - t0) / 6
        if dt < best[1]:
            best = (nt, dt)
    cores = best[0] or cores
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} threads (GEMV probe {134.2e6 / best[1] / 1e9:.1f} GB/s)")
    img = synth.canny_like_control(1, H, W)
    if cfg.gpt.model_type == "c2i":
        emb, mask = synth.class_labels(1, cfg.gpt.num_classes), None      # class label instead of caption features (gpt.py), no pad mask
    else:
        emb, mask = synth.text_embeddings(1, cfg.gpt.cls_token_num, cfg.gpt.caption_dim)
    n_full = (H // 16) * (W // 16)
    t0 = time.perf_counter()
    a = O.control_encoder(gsd, cfg, img)
    O.mlp(a, gsd["adapter_mlp.fc1.weight"], gsd["adapter_mlp.fc2.weight"])
    t_enc = time.perf_counter() - t0
    log(f"cpu encoder {t_enc:.2f}s")
    t0 = time.perf_counter()
    O.generate(gsd, cfg, emb, 1, mask, cfg_scale=1.0, condition=img)
    t_pre = max(time.perf_counter() - t0 - t_enc, 1e-3)  # generate() re-runs the encoder
    log(f"cpu prefill {t_pre:.2f}s")
    t0 = time.perf_counter()
    O.generate(gsd, cfg, emb, 5, mask, cfg_scale=1.0, condition=img)
    t_tok = max(time.perf_counter() - t0 - t_enc - t_pre, 1e-3) / 4
    if t_tok * n_tok_sample < 40:                        # bounded: only take the longer sample if it fits ~40 s
        t0 = time.perf_counter()
        O.generate(gsd, cfg, emb, n_tok_sample + 1, mask, cfg_scale=1.0, condition=img)
        t_tok = max(time.perf_counter() - t0 - t_enc - t_pre, 1e-3) / n_tok_sample
    else:
        n_tok_sample = 4
    log(f"cpu decode {t_tok*1e3:.1f} ms/token")
    t0 = time.perf_counter()
    g = torch.Generator().manual_seed(0)
    codes = torch.randint(0, cfg.vq.codebook_size, (1, n_full), generator=g, dtype=torch.int32)
    O.vq_decode_code(vsd, cfg.vq, codes, [1, cfg.vq.codebook_embed_dim, H // 16, W // 16])
    t_vq = time.perf_counter() - t0
    t_img = t_enc + t_pre + (n_full - 1) * t_tok + t_vq
    out = {"value": 1.0 / t_img, "unit": "images/sec", "cores": cores, "kind": "port",
           "sample": f"oracle fp32, 1 image: encoder {t_enc:.2f}s + prefill {t_pre:.2f}s + {n_tok_sample} decode tokens "
                     f"({t_tok*1e3:.1f} ms/token, extrapolated to {n_full-1}) + VQ decode {t_vq:.2f}s"}
    # The UNMODIFIED reference cannot travel to the GPU box (no /root/reference there); its timing in the build container is committed and
    # reported beside the port's figure — same workload, same weights, 8 cores (tools/time_reference_cpu.py).  There the port runs at the
    # reference's own speed (fp32: 51 vs 56 ms per decode token), so a gap between `value` and these numbers is the host, not the port.
    try:
        ref = json.load(open(os.path.join(ROOT, "profiles", "r02_reference_cpu.json")))
        out["reference_in_build_container"] = {
            "kind": "reference", "cores": ref["threads"], "bf16_images_per_sec": ref["bf16"]["images_per_sec"], "fp32_images_per_sec": ref["fp32"]["images_per_sec"],
            "bf16_decode_ms_per_token": ref["bf16"]["decode_ms_per_token"], "fp32_decode_ms_per_token": ref["fp32"]["decode_ms_per_token"],
            "source": "profiles/r02_reference_cpu.json"}
    except Exception:
        pass
    return out


def pack_cache_dir():
    """Directory of the packed-weight images shared by the ranks of one node / the legs of one run: private to this user (mode 0700, ownership checked —
    ADVICE r5: a predictable world-writable path would let another local user plant images); anything else falls back to a fresh private directory."""
    import stat
    import tempfile
    d = os.environ.get("CONTROLAR_PACK_CACHE", os.path.join(os.environ.get("TMPDIR", "/tmp"), f"controlar_amd_bench_{os.getuid()}"))
    try:
        os.makedirs(d, mode=0o700, exist_ok=True)
        st_ = os.stat(d)
        if st_.st_uid != os.getuid() or (st_.st_mode & 0o022) or not stat.S_ISDIR(st_.st_mode):
            raise PermissionError(f"{d}: not a private directory of uid {os.getuid()}")
        return d
    except Exception as ex:
        d2 = tempfile.mkdtemp(prefix="controlar_amd_bench_")
        log(f"pack cache: {ex!r}; using {d2}")
        return d2


def config_legs(args, world):
    """The BASELINE configs timed as child legs behind the headline: only for the default headline run on ONE GPU (a multi-GPU run times the metric, nothing else:
    every rank would otherwise spawn its own children on its own device while the others wait at the barrier)."""
    if world != 1 or args.no_variants or args.config != 0 or args.precision != "bf16" or args.model != "xl" or args.batch != 768:
        return []
    if args.weights_fp8 or args.kv_fp8 or args.sample_logits:
        return []
    return [int(x) for x in args.config_legs.split(",") if x.strip()]


def child_leg(extra, timeout=900):
    """One more bench configuration in a child process (its own HIP contexts: the parent has released its device memory), returns its parsed JSON line or an error."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + extra + ["--no-cpu-baseline", "--no-variants", "--exact-leg-steps", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": f"rc {r.returncode}: {(r.stderr or '')[-400:]}"}
        return json.loads(line[-1])
    except Exception as e:      # a failed leg must not take the headline line with it
        return {"error": repr(e)[:400]}


def main():
    args = parse()
    refuse_debug_environment()
    from controlar_amd.dist import respawn_under_torchrun
    respawn_under_torchrun(__file__, sys.argv[1:], args.gpus)     # --gpus N without a torchrun environment: become N ranks
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=dev)     # "nccl" is RCCL on ROCm

    from controlar_amd import config as C, synth
    from controlar_amd.engine import Engine
    from controlar_amd.dist import alloc_packed_host, gather_tokens

    S = args.image_size
    Hh, Ww = (args.image_h or S), (args.image_w or S)
    gh, gw = Hh // 16, Ww // 16
    grid = max(gh, gw)
    c2i = args.model.endswith("_c2i")
    if c2i:                                                   # gpt.py: class-label prefix of length 1, ViT-S/16 control encoder, no text, no pad mask
        cfg = {"b_c2i": C.b_c2i, "l_c2i": C.l_c2i}[args.model](grid * grid)
    elif args.model == "tiny":
        cfg = C.tiny_t2i(grid * grid, args.condition_type)
    else:
        cfg = {"xl": C.xl_t2i, "b": C.b_t2i}[args.model](grid * grid, adapter_size=args.adapter_size, condition_type=args.condition_type)
    n_new = gh * gw
    # Two contexts, as the reference keeps two modules (gpt_model, vq_model).
    eng = Engine(cfg, args.precision, device=dev, weights_fp8=("mfma" if args.fp8_mfma else args.weights_fp8), kv_fp8=args.kv_fp8)
    vq_eng = Engine(cfg, args.vq_precision, device=dev)
    sds = {}

    def build_only():
        log("synthesising weights")
        sds["g"], sds["v"] = synth.path_state_dicts(cfg, seed=0)          # deterministic (seeded CPU generator)
        log("loading weights into the HIP contexts")
        eng.load_state_dict(sds["g"], finalize=True)
        vq_eng.load_state_dict(sds["v"], finalize=True)
    # N > 1: the weights are synthesised and packed ONCE per node (rank 0) and handed to the other ranks as packed images (car_export_packed / car_import_packed:
    # plain copies) — eight ranks each spending 11 s of host synthesis on one shared host is start-up time, not work (dist.load_weights_once)
    import hashlib
    from controlar_amd.dist import load_weights_once
    # the tag covers everything the packed images depend on: the two car_config blocks, the arithmetic, the library build AND the generator of the synthetic weights
    # (a synth.py change with no csrc change must not find stale images)
    synth_src = open(synth.__file__, "rb").read()
    tag = hashlib.blake2b((repr(bytes(eng._cc)) + repr(bytes(vq_eng._cc)) + args.precision + args.vq_precision + eng.lib.car_build_id().decode()).encode() + synth_src, digest_size=8).hexdigest()
    cdir = pack_cache_dir()
    pk = [os.path.join(cdir, f"synth0_{tag}_{k}.carpk") for k in ("gpt", "vq")]

    def build_and_export():
        build_only()
        try:        # a failed export (no space under the cache directory, ...) must not take rank 0 down: the other ranks then fail to import and build for themselves
            for e_, f_ in ((eng, pk[0]), (vq_eng, pk[1])):
                e_._check(e_.lib.car_export_packed(e_._h, (f_ + f".tmp{os.getpid()}").encode()), "car_export_packed")
                os.replace(f_ + f".tmp{os.getpid()}", f_)
        except Exception as ex:
            log(f"packed-image export failed ({ex!r}): every rank builds its own weights")
            for f_ in pk:
                for g_ in (f_, f_ + f".tmp{os.getpid()}"):
                    try:
                        os.remove(g_)
                    except OSError:
                        pass

    def import_packed():
        for e_, f_ in ((eng, pk[0]), (vq_eng, pk[1])):
            e_._check(e_.lib.car_import_packed(e_._h, f_.encode()), "car_import_packed")
    legs = config_legs(args, world)
    if world == 1 and (args.pack_cache or legs):
        # one GPU: a config leg restores the images the headline run exported (same model, same build); the headline exports when legs will follow
        how = None
        if args.pack_cache and all(os.path.exists(f) for f in pk):
            try:
                import_packed(); how = "imported"
            except Exception as ex:
                log(f"packed-image import failed ({ex!r}): building")
        if how is None:
            build_and_export(); how = "built"
    else:
        how = load_weights_once(dist, rank, pk, build_and_export, import_packed, build_only)
    gsd, vsd = sds.get("g"), sds.get("v")                   # None on a rank that imported the packed images (only rank 0 at N = 1 needs them: the CPU baseline)
    side = torch.cuda.Stream(device=dev)
    log(f"weights ready ({how})")

    # ---- inputs.  Global image g (seeds 1234 + g) belongs to rank g % world (strided shards, sample_t2i_ddp.py:131).  Rank 0 builds one shard
    # at a time in a packed host buffer and sends it to its rank (dist.scatter_inputs); --input-dist local: every rank builds its own.
    G = args.batch * world
    T, cap = cfg.gpt.cls_token_num, cfg.gpt.caption_dim
    from controlar_amd.dist import scatter_inputs, scatter_probe, parallel_fill

    def make_shard(r):
        packed, h_img, h_emb, h_mask = alloc_packed_host(args.batch, Hh, Ww, T, cap)

        def one(j):
            g_ = r + world * j                                # global image index of local image j on rank r
            if args.condition_type in ("canny", "seg"):
                h_img[j] = synth.canny_like_control(1, Hh, Ww, seed=1234 + g_, dtype=torch.bfloat16)[0]             # {-1,+1}: exact in bf16
            else:
                h_img[j] = synth.smooth_control(1, Hh, Ww, seed=1234 + g_)[0].to(torch.bfloat16)
            if c2i:                                           # class-conditional: no caption features (the class labels are drawn below)
                h_emb[j] = 0; h_mask[j] = 1
            else:
                e_, m_ = synth.text_embeddings(1, T, cap, seed=1234 + g_)
                h_emb[j] = e_[0].to(torch.bfloat16); h_mask[j] = m_[0]
        parallel_fill(args.batch, one)                       # independent per-image generators, disjoint output slices
        return packed
    t_bc0 = time.perf_counter()
    if args.input_dist == "local" or world == 1:
        img, emb, mask = scatter_inputs(None, dev, 0, 1, args.batch, Hh, Ww, T, cap, lambda _r: make_shard(rank))
    else:
        img, emb, mask = scatter_inputs(dist, dev, rank, world, args.batch, Hh, Ww, T, cap, make_shard)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t_bcast = time.perf_counter() - t_bc0                     # host synthesis of the shards + H2D + the point-to-point sends
    shard_bytes = int(sum(__import__("controlar_amd.dist", fromlist=["packed_layout"]).packed_layout(args.batch, Hh, Ww, T, cap)))
    # the wire half of the input edge, measured on its own (N > 1 only): rank 0 sends one shard-sized device buffer to every other rank over RCCL / xGMI
    t_probe, probe_bytes = scatter_probe(dist, dev, rank, world, shard_bytes, sync_fn=torch.cuda.synchronize)
    img, emb, mask = img.contiguous(), emb.contiguous(), mask.contiguous()
    if args.precision == "fp32" and not c2i:
        # exact mode takes the caption features in fp32 (the packed transport buffer carries bf16): redraw this rank's rows unrounded,
        # so that row 0 is bit for bit the input of the committed XL golden (the control maps are {-1,+1}: exact in either type)
        emb = torch.stack([synth.text_embeddings(1, T, cap, seed=1234 + rank + world * j)[0][0] for j in range(args.batch)]).to(dev)
    # self-check rows: with >= 4 images the first image of the second half (the second decode chain when the batch is cut in
    # two, engine_generate.hip generate_impl) repeats local image 0 — identical inputs must come out as identical tokens and pixels
    twin = args.batch // 2 if (args.batch >= 4 and not args.sample_logits) else -1      # sampled rows draw from their own Philox stream (seed, row, step): twins differ by design
    if twin > 0:
        img[twin], emb[twin], mask[twin] = img[0], emb[0], mask[0]

    # the masks were built on the host: the first valid prompt position of this rank's batch is known without asking the device (car_sampling.first_valid_hint:
    # car_generate then has no host wait at all)
    first_valid = None
    if not c2i:
        from controlar_amd.engine import first_valid_position
        first_valid = first_valid_position(mask)
    labels = None
    if c2i:
        labels = torch.stack([synth.class_labels(1, cfg.gpt.num_classes, seed=1234 + rank + world * j)[0] for j in range(args.batch)]).to(dev)
        if twin > 0:
            labels[twin] = labels[0]

    stage_ev = []      # (encode start, generate start, vq start, end) events of every step, read after the timed region

    def one_step():
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        evs[0].record()
        toks_, px_ = one_step_inner(evs)
        evs[3].record()
        stage_ev.append(evs)
        return toks_, px_

    def one_step_inner(evs):
        eng.encode_control(img)
        evs[1].record()
        if c2i:
            toks = eng.generate(labels, n_new, None, cfg_scale=args.cfg_scale, sample_logits=args.sample_logits, top_k=(args.top_k if args.sample_logits else 0),
                                top_p=args.top_p, temperature=args.temperature, seed=1234)
            evs[2].record()
            return toks, vq_eng.vq_decode(toks, gh, gw)
        if args.sample_logits:
            toks = eng.generate(emb, n_new, mask, cfg_scale=args.cfg_scale, control_strength=1.0, sample_logits=True, top_k=args.top_k, top_p=args.top_p,
                                temperature=args.temperature, seed=1234, first_valid=first_valid)
        else:
            toks = eng.generate(emb, n_new, mask, cfg_scale=args.cfg_scale, control_strength=1.0, first_valid=first_valid)
        evs[2].record()
        if not args.overlap_vq:
            return toks, vq_eng.vq_decode(toks, gh, gw)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            px = vq_eng.vq_decode(toks, gh, gw)     # enqueued asynchronously; the next step's generate() does not wait for it
        return toks, px

    acc = {"dec_ms": 0.0, "pre_ms": 0.0, "st": None, "warm": True}

    def step_and_stats():
        toks, px = one_step()
        st_ = eng.stats()                                   # waits on the library's own stream events (HIP-event timed decode loop)
        if not acc["warm"]:
            acc["dec_ms"] += st_["decode_ms"]; acc["pre_ms"] += st_["prefill_ms"]
        acc["st"] = st_
        return toks, px

    from controlar_amd.dist import timed_steps
    # warmup runs through the same harness with zero timed steps so that the accumulators only see the timed region
    timed_steps(dist, dev, step_and_stats, 0, args.warmup, torch.cuda.synchronize)
    log("warmup done")
    acc["warm"] = False
    elapsed, (toks, px) = timed_steps(dist, dev, step_and_stats, args.steps, 0, torch.cuda.synchronize)
    dec_ms, pre_ms, st = acc["dec_ms"], acc["pre_ms"], acc["st"]
    # every token of every image went through the decode loop: the first comes from the prefill, the other n_new - 1 are one graph replay each
    assert st["decode_steps"] == n_new - 1, f"decode loop ran {st['decode_steps']} steps, expected {n_new - 1}"
    assert st["dev_knobs_active"] == 0, "a CAR_* library switch is set in the environment: not the shipped schedule"
    log(f"timed region {elapsed:.2f}s")
    t_gather = 0.0
    if dist is not None:
        torch.cuda.synchronize(); t_g0 = time.perf_counter()
        all_toks = gather_tokens(dist, toks)                # [G, n_new] on every rank, global image order
        torch.cuda.synchronize(); t_gather = time.perf_counter() - t_g0
        assert all_toks.shape[0] == G
    assert bool(torch.isfinite(px).all())
    parity = {}
    if twin > 0:
        parity["twin_rows_equal"] = bool(torch.equal(toks[0], toks[twin]))
        parity["twin_pixels_max_abs_diff"] = float((px[0] - px[twin]).abs().max())
        assert parity["twin_rows_equal"], "identical inputs in two rows of the batch produced different tokens"
    gpath = os.path.join(ROOT, "tests", "golden", "xl_canny_512_cfg1.npz")
    if (rank == 0 and args.model == "xl" and (Hh, Ww) == (512, 512) and args.cfg_scale <= 1.0 and args.adapter_size == "small"
            and args.condition_type == "canny" and not args.weights_fp8 and not args.kv_fp8 and not args.sample_logits and os.path.exists(gpath)):
        # local image 0 of rank 0 is the input of the committed golden (synth seed 1234): the reference's fp32 greedy tokens.
        # The bf16 fast mode free-runs, so it follows them until the first near-tie and is graded teacher-forced in
        # tests/test_bench_shapes_gpu.py; here the common prefix and overall agreement are reported and sanity-bounded.
        import numpy as np
        gold = np.load(gpath)["tokens"][0]
        mine = toks[0].cpu().numpy()
        neq = np.nonzero(mine != gold)[0]
        parity["golden_prefix_tokens"] = int(neq[0]) if len(neq) else int(len(gold))
        parity["golden_token_agreement"] = float((mine == gold).mean())
        assert parity["golden_prefix_tokens"] >= 2, parity
        if args.precision == "fp32":       # exact mode: the reference's greedy tokens, all of them, whatever the batch (batch-invariant kernels)
            assert parity["golden_token_agreement"] == 1.0, parity

    # HBM traffic of one decode step (read + write) is taken from the committed PMC summary of the SAME configuration
    # (profiles/pmc_decode_step.json, written by tools/pmc_decode.py from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
    # passes, gfx950 corrections of MI355X_MICROARCH.md §HBM applied there); any other configuration reports null.
    def measured_traffic():
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_decode_step.json" if args.precision == "bf16" else "pmc_decode_step_fp32.json")))
        except Exception:
            return None, "no PMC summary for this configuration"
        # the summary is only quoted for the library build it was measured on (tools/pmc_decode.py records car_build_id): after any kernel change the
        # line says null until tools/profile_round.sh has been re-run
        if rec.get("build_id") != eng.lib.car_build_id().decode():
            return None, f"the committed PMC summary was measured on library build {str(rec.get('build_id'))[:12]}, this is {eng.lib.car_build_id().decode()[:12]}: not quoted"
        same = (not args.kv_fp8 and rec.get("model") == args.model and rec.get("batch") == args.batch and abs(rec.get("cfg_scale", 1.0) - args.cfg_scale) < 1e-6
                and rec.get("precision") == args.precision and bool(rec.get("weights_fp8")) == bool(args.weights_fp8)
                and rec.get("image_hw") == [Hh, Ww] and rec.get("adapter_size") == args.adapter_size)
        if not same:
            return None, "no PMC summary for this configuration"
        return (rec["fetch_bytes_per_step"] + rec["write_bytes_per_step"],
                "NOT measured in this run: read from the committed PMC summary of the same configuration (profiles/pmc_decode_step.json, "
                "tools/profile_round.sh). " + rec.get("note", ""))

    if rank == 0:
        value = G * args.steps / elapsed
        per_step_ms = dec_ms / args.steps / max(st["decode_steps"], 1)
        achieved = st["decode_algo_bytes"] / max(st["decode_steps"], 1) / (per_step_ms * 1e-3) / 1e9
        out = {
            "metric": "images/sec/GPU, LlamaGen-XL t2i canny 512x512 (1024 tok); 1/2/4/8-GPU scaling",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": (f"LlamaGen-{args.model[0].upper()} c2i (class-conditional, gpt.py) + ViT-S/16 canny control, {Hh}x{Ww} ({n_new} tokens), " if c2i else
                                    f"LlamaGen-{args.model.upper()} t2i + DINOv2-{args.adapter_size} {args.condition_type} control, {Hh}x{Ww} ({n_new} tokens), ") +
                                   f"{'fp8 (e4m3) decode weights x e4m3 activations on the fp8 MFMA, ' if args.fp8_mfma else ('fp8 (e4m3) decode weights (weight-only), ' if args.weights_fp8 else '')}"
                                   f"{'OPT-IN e4m3 KV cache (not the reference arithmetic: tolerance-graded mode), ' if args.kv_fp8 else ''}"
                                   f"{'EXACT mode (fp32 weights / activations / KV, greedy tokens bit-identical to the fp32 reference), VQ decoder in ' + args.vq_precision + ', ' if args.precision == 'fp32' else ''}"
                                   f"cfg_scale={args.cfg_scale}, {('sampled top_k=%d top_p=%g T=%g' % (args.top_k, args.top_p, args.temperature)) if args.sample_logits else 'greedy'}, {args.batch} images/GPU/step; stages A-H "
                                   "(control encoder, generate, VQ decode) all inside the timed region",
                       "images_per_gpu": args.batch, "global_batch": G, "cfg_scale": args.cfg_scale,
                       "per_gpu_images_per_sec": value / world, "parallelism": f"dp{world}",
                       "decode_fraction_of_step": dec_ms / (elapsed * 1e3), "prefill_ms": pre_ms / args.steps,
                       "input_distribution": ("local (every rank draws its shard)" if (args.input_dist == "local" or world == 1) else
                                              "scatter from rank 0: one shard built, copied and sent point-to-point at a time"),
                       "input_distribution_s": t_bcast, "input_wire_bytes": 0 if (args.input_dist == "local" or world == 1) else shard_bytes * (world - 1),
                       "input_scatter_probe_s": t_probe, "input_scatter_probe_bytes": probe_bytes,
                       "input_scatter_probe_gb_per_s": (probe_bytes / t_probe / 1e9) if t_probe > 0 else None,
                       "token_gather_s": t_gather, "graph": st["graph_used"], "weights": how + (" on rank 0, packed images imported by the other ranks" if world > 1 else ""),
                       "decode_kernels_per_step": st["decode_kernels_per_step"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": measured_traffic()[0], "traffic_note": measured_traffic()[1],
                         "kernel": "decode step (one hipGraph replay = one token for all sequences)",
                         "bytes_per_launch": st["decode_algo_bytes"] / max(st["decode_steps"], 1),
                         "avg_launch_ms": per_step_ms},
        }
        if parity:
            out["config"]["self_check"] = parity
        try:      # where a step goes: events on the caller's stream around the three boundary calls (timed steps only), ms per step
            tl = stage_ev[-args.steps:]
            out["config"]["stage_ms"] = {"encode_control": sum(e[0].elapsed_time(e[1]) for e in tl) / len(tl), "generate": sum(e[1].elapsed_time(e[2]) for e in tl) / len(tl),
                                         "vq_decode": sum(e[2].elapsed_time(e[3]) for e in tl) / len(tl)}
        except Exception:
            pass
        # VQ decoder cost in both arithmetics (the reference keeps vq_model in fp32, sample_t2i.py:43-47; both bench modes decode pixels in bf16 by default):
        # HIP-event time of car_vq_decode on a slice of this step's tokens, outside the timed region
        try:
            nb = min(args.batch, 48)
            vq_ms = {}
            for prec_ in ("bf16", "fp32"):
                if prec_ != args.vq_precision and vsd is None:      # this rank imported packed images: no state dict at hand for a second decoder context
                    continue
                ve = vq_eng if prec_ == args.vq_precision else Engine(cfg, prec_, device=dev)
                if ve is not vq_eng:
                    ve.load_state_dict(vsd, finalize=True)
                ve.vq_decode(toks[:nb], gh, gw); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ve.vq_decode(toks[:nb], gh, gw); e1.record(); torch.cuda.synchronize()
                vq_ms[prec_] = e0.elapsed_time(e1) / nb
                if ve is not vq_eng:
                    ve.close()
            out["config"]["vq_ms_per_image"] = {**vq_ms, "images": nb, "in_timed_region": args.vq_precision}
        except Exception as e:
            out["config"]["vq_ms_per_image"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, gsd, vsd, Hh, Ww, args.cpu_tokens)
        headline = (args.config == 0 and args.precision == "bf16" and args.model == "xl" and (Hh, Ww) == (512, 512) and not args.weights_fp8 and not args.kv_fp8
                    and not args.sample_logits and args.cfg_scale <= 1.0)
        if world == 1 and not args.no_variants and ((headline and (args.exact_leg_steps > 0 or legs)) or (args.config == 5 and args.fp8_mfma)):
            eng.close(); vq_eng.close(); del img, emb, mask, toks, px
            torch.cuda.empty_cache()
            if headline and args.exact_leg_steps > 0:
                log("exact leg (child process): --precision fp32, bit-identical tokens")
                ex = child_leg(["--precision", "fp32", "--steps", str(args.exact_leg_steps), "--warmup", "1"])
                if "error" in ex:
                    out["exact"] = ex
                else:
                    sc = ex["config"].get("self_check", {})
                    out["exact"] = {"value": ex["value"], "unit": ex["unit"], "ms_per_step": ex["ms_per_step"], "steps": ex["steps"], "warmup": ex["warmup"], "dtype": ex["dtype"],
                                    "images_per_gpu": ex["config"]["images_per_gpu"], "roofline": ex["roofline"], "prefill_ms": ex["config"]["prefill_ms"],
                                    "decode_kernels_per_step": ex["config"]["decode_kernels_per_step"],
                                    "golden_token_agreement": sc.get("golden_token_agreement"), "golden_prefix_tokens": sc.get("golden_prefix_tokens"), "twin_rows_equal": sc.get("twin_rows_equal"),
                                    "workload": ex["config"]["workload"],
                                    "note": "same workload as the headline in the mode whose greedy tokens are bit-identical to the fp32 CPU reference (row 0 = the committed "
                                            "reference golden tests/golden/xl_canny_512_cfg1.npz: agreement must be 1.0); timed in a child process after the headline's timed region"}
            if headline and legs:
                # every other BASELINE config, driver-timed: one child process each (its own contexts), the same harness and JSON contract as `--config N`
                out["configs"] = {}
                t_legs = time.perf_counter()
                for cno in legs:
                    if time.perf_counter() - t_legs > args.legs_budget_s:
                        out["configs"][str(cno)] = {"skipped": f"leg budget of {args.legs_budget_s:.0f} s spent"}
                        continue
                    log(f"config {cno} leg (child process)")
                    t_l0 = time.perf_counter()
                    v = child_leg(["--config", str(cno), "--steps", str(args.config_leg_steps), "--warmup", "1", "--pack-cache"], timeout=600)
                    if "error" in v:
                        out["configs"][str(cno)] = v
                        continue
                    out["configs"][str(cno)] = {"value": v["value"], "unit": v["unit"], "per_gpu_images_per_sec": v["config"]["per_gpu_images_per_sec"], "ms_per_step": v["ms_per_step"],
                                                "steps": v["steps"], "warmup": v["warmup"], "dtype": v["dtype"], "images_per_gpu": v["config"]["images_per_gpu"], "cfg_scale": v["config"]["cfg_scale"],
                                                "decode_ms_per_token": v["roofline"]["avg_launch_ms"], "roofline": {k_: v["roofline"][k_] for k_ in ("bound", "achieved", "peak", "unit", "frac", "bytes_per_launch", "avg_launch_ms")},
                                                "decode_kernels_per_step": v["config"]["decode_kernels_per_step"], "stage_ms": v["config"].get("stage_ms"), "weights": v["config"]["weights"],
                                                "workload": v["config"]["workload"], "leg_wall_s": time.perf_counter() - t_l0}
            if not headline:
                log("config 5 variant (child process): weight-only e4m3 on the bf16 MFMA")
                v = child_leg(["--config", "5", "--fp8-weight-only", "--steps", str(args.steps), "--warmup", str(args.warmup)])
                out["config"]["variants"] = {"this_line": "W8A8: e4m3 weights x e4m3 activations on v_mfma_f32_16x16x32_fp8_fp8 (BASELINE configs[4] as named)",
                                             "weight_only_e4m3_bf16_mfma": v if "error" in v else {"value": v["value"], "ms_per_step": v["ms_per_step"], "decode_ms_per_token": v["roofline"]["avg_launch_ms"],
                                                                                                    "roofline_frac": v["roofline"]["frac"]}}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
