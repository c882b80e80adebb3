"""One rank of a sharded, device-resident ensemble that runs the library's OWN move kernels (ensemble.NativeMoves) on the GPU.

Started by tests/test_gpu_sharded_native.py as a fresh child process per rank; 2 or 3 such ranks share the one GPU of the
test box, so the process group is gloo and the all-gather of positions is staged through the host (RCCL wants one GPU per
rank) -- everything else is what rank r of an N-GPU run executes: shard bounds with start != 0, cf_ens_active_set /
cf_ens_propose / cf_ens_accept on the shard's local indices, cf_eval_device for the shard's proposals.

    python tests/sharded_rank_worker.py --rank R --world N --port P --walkers W --steps K --moves ref --randomize 1 --out f.npz
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--walkers", type=int, required=True)
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--moves", default="ref", choices=["ref", "stretch", "de", "kde"])
    ap.add_argument("--randomize", type=int, default=1)
    ap.add_argument("--seed", type=int, default=4)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    if a.world > 1:  # the process group first: nothing has touched the GPU yet
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(a.port))
        dist.init_process_group("gloo", rank=a.rank, world_size=a.world)
    amd = importlib.import_module("cosmology-model-fit_amd")
    if amd.lib().cf_device_count() < 1:
        sys.exit("sharded_rank_worker needs an MI355X; there is no fallback path")
    dev = torch.device("cuda:0")
    syn = amd.synthetic.pantheon_like(n_sn=300, seed=3)
    lk = amd.sn_pantheon.PantheonLikelihood(syn["z_cmb"], syn["z_hel"], syn["obs"], chol=syn["chol"])
    start = amd.synthetic.THETA_TRUE + 1e-2 * np.random.default_rng(1).standard_normal((a.walkers, 4))
    moves = {"ref": amd.ensemble.REFERENCE_MOVES, "stretch": amd.ensemble.STRETCH_ONLY, "de": (("de", 1.0),),
             "kde": (("kde", 1.0),)}[a.moves]
    ens = amd.ensemble.ShardedEnsemble(lk.engine.torch_log_prob(), torch.from_numpy(start).to(dev), seed=a.seed, moves=moves,
                                       randomize_split=bool(a.randomize))
    assert isinstance(ens.impl, amd.ensemble.NativeMoves), "the library's kernels must run the moves"
    assert (ens.start, ens.stop) == amd.ensemble.shard_bounds(a.walkers, a.world, a.rank)
    picked = []
    for _ in range(a.steps):
        picked.append(ens._pick_move())
        ens.step()
    pos, lp = ens.full_state()
    acc = ens.acceptance_fraction()
    torch.cuda.synchronize()
    if a.rank == 0:
        np.savez(a.out, pos=pos.cpu().numpy(), lp=lp.cpu().numpy(), acc=acc, picked=np.array(picked),
                 shard=np.array([ens.start, ens.stop]), host_staged=int(ens._host_staged))
    if a.world > 1:
        dist.barrier()
        dist.destroy_process_group()
    lk.engine.close()


if __name__ == "__main__":
    main()
