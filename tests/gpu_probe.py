"""First-contact diagnostic (not a pytest): prints stage-by-stage errors of the HIP path vs the
oracle for one tiny case in both arithmetic modes.  Run on the GPU box via gpurun."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from controlar_amd.engine import Engine
from oracle import controlar_oracle as O
from tests.cases import load_case

name = sys.argv[1] if len(sys.argv) > 1 else "tiny_canny_cfg1"
cs = load_case(name)
cfg, gold = cs["cfg"], cs["gold"]
print("device", torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count)
toks_o, logits_o, st = O.generate(cs["gsd"], cfg, cs["emb"], cs["n_new"], cs["mask"], cfg_scale=cs["cfg_scale"],
                                  cfg_interval=cs["cfg_interval"], condition=cs["img"], control_strength=cs["control_strength"],
                                  return_logits=True, return_stages=True)
print("oracle tokens == golden:", np.array_equal(toks_o.numpy(), gold["tokens"]))
for prec in ("fp32", "bf16"):
    eng = Engine(cfg, prec)
    t0 = time.time()
    eng.load_state_dict(cs["gsd"]); eng.load_state_dict(cs["vsd"]); eng.finalize()
    print(prec, "load %.2fs" % (time.time() - t0))
    a = eng.encode_control(cs["img"].cuda(), want_output=True).float().cpu()
    ref = st["adapter_mlp_out"]
    print(prec, "adapter_mlp_out max|d| %.3g mean|d| %.3g (mean|x| %.3g)" % ((a - ref).abs().max(), (a - ref).abs().mean(), ref.abs().mean()))
    forced = None if prec == "fp32" else torch.from_numpy(gold["tokens"])
    toks, logits = eng.generate(cs["emb"].cuda(), cs["n_new"], cs["mask"].cuda(), cfg_scale=cs["cfg_scale"], cfg_interval=cs["cfg_interval"],
                                control_strength=cs["control_strength"], forced_tokens=forced, return_logits=True)
    torch.cuda.synchronize()
    b = 2 * cs["B"] if cs["cfg_scale"] > 1 else cs["B"]
    c0 = eng.control_tokens(0, b, cs["n_new"])[: cs["B"]]
    print(prec, "ctrl0 max|d| %.3g (mean|x| %.3g)" % ((c0 - st["ctrl"][0][: cs["B"]].float()).abs().max(), st["ctrl"][0].abs().mean()))
    toks = toks.cpu().numpy(); logits = logits.cpu()
    d = (logits - logits_o).abs()
    print(prec, "logits max|d| %.4g mean|d| %.4g ; first-step max|d| %.4g" % (d.max(), d.mean(), d[:, 0].max()))
    eq = toks == gold["tokens"]
    print(prec, "tokens equal %d/%d ; first mismatch %s" % (eq.sum(), eq.size, np.argwhere(~eq)[:1].tolist()))
    print(prec, "stats", eng.stats())
    if "pixels" in gold:
        px = eng.vq_decode(torch.from_numpy(gold["tokens"]), cs["H"] // 16, cs["W"] // 16).cpu().numpy()
        dp = np.abs(px - gold["pixels"])
        print(prec, "pixels max|d| %.4g mean|d| %.4g (mean|x| %.3g)" % (dp.max(), dp.mean(), np.abs(gold["pixels"]).mean()))
    eng.close()
