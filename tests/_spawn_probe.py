"""Helper for tests/test_dist_gloo.py::test_gpus_flag_respawns_under_torchrun — NOT a test.  A miniature of bench.py's start-up: `--gpus N`
as a plain process re-executes itself under torch.distributed.run (controlar_amd.dist.respawn_under_torchrun), the ranks rendezvous on
127.0.0.1 (gloo here, RCCL in bench.py), rank 0 owns the inputs and scatters the shards, tokens are gathered, rank 0 prints ONE JSON line."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--batch", type=int, default=3)
    a = ap.parse_args()
    from controlar_amd.dist import respawn_under_torchrun, scatter_inputs, gather_tokens, alloc_packed_host
    from controlar_amd import synth
    respawn_under_torchrun(__file__, sys.argv[1:], a.gpus)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("gloo")
    H = W = 32; T, cap = 12, 64

    def make_shard(r):
        packed, h_img, h_emb, h_mask = alloc_packed_host(a.batch, H, W, T, cap)
        for j in range(a.batch):
            g = r + world * j
            h_img[j] = synth.canny_like_control(1, H, W, seed=1234 + g, dtype=torch.bfloat16)[0]
            e_, m_ = synth.text_embeddings(1, T, cap, seed=1234 + g)
            h_emb[j] = e_[0].to(torch.bfloat16); h_mask[j] = m_[0]
        return packed
    img, emb, mask = scatter_inputs(dist, torch.device("cpu"), rank, world, a.batch, H, W, T, cap, make_shard)
    local = (img.float().sum(dim=(1, 2, 3)).round().to(torch.int32)[:, None] + mask.sum(dim=1).to(torch.int32)[:, None] + torch.arange(4, dtype=torch.int32)[None])
    allt = gather_tokens(dist, local)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "tokens": allt.tolist(), "master": os.environ.get("MASTER_ADDR")}), flush=True)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
