#!/usr/bin/env python3
"""Times the UNMODIFIED reference modules (imported from /root/reference, as tests/golden/make_golden.py does) on this
box's host cores, stage by stage, on the bench workload: GPT-XL t2i + DINOv2-small canny, 512x512, cfg 1, B = 1, greedy.
SURVEY.md §8(d) "CPU reference timing beside it": bf16 (the reference default --precision) and fp32, thread count stated.
Bounded sample: `--tokens` decode tokens (default 48) extrapolated to 1023 by per-token cost.  Build container only
(the GPU box has no /root/reference): the result is committed as profiles/r02_reference_cpu.json.
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402
import make_golden as MG  # noqa: E402  (patches AutoModel.from_pretrained, imports the reference)
from controlar_amd import config as C, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tokens", type=int, default=48)
ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_reference_cpu.json"))
args = ap.parse_args()
torch.set_num_threads(args.threads)
cfg = C.xl_t2i(1024, "small", "canny")
gsd, vsd = synth.path_state_dicts(cfg, seed=0)
img = synth.canny_like_control(1, 512, 512)
emb, mask = synth.text_embeddings(1, cfg.gpt.cls_token_num, cfg.gpt.caption_dim)
res = {"threads": args.threads, "workload": "GPT-XL t2i + DINOv2-small canny 512x512 cfg 1 B=1 greedy, unmodified reference modules on CPU", "torch": torch.__version__}
for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
    model = MG.build_ref_gpt(cfg, gsd, dt)
    with torch.no_grad():
        t0 = time.perf_counter(); model.adapter_mlp(model.adapter(img.to(dt))); t_enc = time.perf_counter() - t0
        def gen(n):
            t0 = time.perf_counter()
            MG.ref_gen.generate(model, emb.to(dt), n, mask, cfg_scale=1.0, condition=img.to(dt), temperature=1.0, top_k=0, top_p=1.0, sample_logits=False)
            return time.perf_counter() - t0
        t2 = gen(2)                       # adapter + setup_caches + prefill + 1 decode step (the reference cannot generate 1 token: torch.cat of an empty list, generate.py:203)
        tn = gen(args.tokens + 2)
    t_tok = (tn - t2) / args.tokens
    res[name] = {"adapter_s": t_enc, "generate_1_token_s": t2 - t_tok, "decode_ms_per_token": t_tok * 1e3, "tokens_sampled": args.tokens}
    print(name, res[name], flush=True)
    del model
vq = MG.build_ref_vq(cfg.vq, vsd)
codes = torch.randint(0, cfg.vq.codebook_size, (1, 1024), dtype=torch.int32)
with torch.no_grad():
    t0 = time.perf_counter(); vq.decode_code(codes, [1, 8, 32, 32]); res["vq_decode_fp32_s"] = time.perf_counter() - t0
for name in ("bf16", "fp32"):
    r = res[name]
    r["image_s_extrapolated"] = r["generate_1_token_s"] + 1023 * r["decode_ms_per_token"] / 1e3 + res["vq_decode_fp32_s"]
    r["images_per_sec"] = 1.0 / r["image_s_extrapolated"]
json.dump(res, open(args.out, "w"), indent=1)
print(json.dumps(res))
