"""CPU: the multi-rank ensemble logic (sharding, all-gather of positions, counter-based RNG) on gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_pkg
from oracle import moves_torch


def _tensor_moves():
    """The tensor statement of the moves (oracle/): the driver has no built-in CPU path."""
    return moves_torch.TensorMoves(load_pkg().ensemble.stream_key)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


MU = torch.tensor([0.3, -1.0, 2.0], dtype=torch.float64)
SIG = torch.tensor([0.5, 2.0, 0.1], dtype=torch.float64)


def gauss_logp(theta):
    return -0.5 * (((theta - MU) / SIG) ** 2).sum(dim=1)


def _init_positions(W):
    g = torch.Generator().manual_seed(7)
    return MU + SIG * torch.randn(W, 3, generator=g, dtype=torch.float64)


def _worker(rank, world, port, W, steps, path, moves, randomize):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ens = load_pkg().ensemble.ShardedEnsemble(gauss_logp, _init_positions(W), seed=11, moves=moves, randomize_split=randomize,
                                              moves_impl=_tensor_moves())
    ens.run(steps)
    pos, lp = ens.full_state()
    acc = ens.acceptance_fraction()
    if rank == 0:
        torch.save({"pos": pos, "lp": lp, "acc": acc}, path)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, W, steps, tmp_path, moves=(("stretch", 1.0),), randomize=True):
    path = str(tmp_path / f"w{world}.pt")
    if world == 1:
        ens = load_pkg().ensemble.ShardedEnsemble(gauss_logp, _init_positions(W), seed=11, moves=moves, randomize_split=randomize,
                                                  moves_impl=_tensor_moves())
        ens.run(steps)
        pos, lp = ens.full_state()
        return {"pos": pos, "lp": lp, "acc": ens.acceptance_fraction()}
    mp.spawn(_worker, args=(world, _free_port(), W, steps, path, moves, randomize), nprocs=world, join=True)
    return torch.load(path)


def test_shard_bounds_partition():
    ens = load_pkg().ensemble
    for n, world in ((4096, 1), (4096, 8), (65536, 8), (10, 3), (7, 8)):
        cuts = [ens.shard_bounds(n, world, r) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1


def test_counter_rng_is_a_pure_function_and_uniform():
    ens = _tensor_moves()
    ids = torch.arange(0, 200000, dtype=torch.int64)
    u = ens.uniform01(42, 3, 1, ids, 2)
    assert torch.equal(u, ens.uniform01(42, 3, 1, ids, 2))
    assert torch.equal(u[1000:2000], ens.uniform01(42, 3, 1, ids[1000:2000], 2)), "must not depend on the shard"
    assert 0.0 <= float(u.min()) and float(u.max()) < 1.0
    assert abs(float(u.mean()) - 0.5) < 3e-3 and abs(float(u.var()) - 1 / 12) < 2e-3
    hist = torch.histc(u, bins=20, min=0, max=1) / len(u)
    assert float((hist - 0.05).abs().max()) < 3e-3
    v = ens.uniform01(42, 4, 1, ids, 2)
    assert abs(float(((u - 0.5) * (v - 0.5)).mean())) < 1e-3  # consecutive steps uncorrelated


@pytest.mark.parametrize("world,W,randomize", [(2, 64, True), (2, 64, False), (3, 48, True), (3, 50, False)])
def test_chain_is_bit_identical_for_any_number_of_ranks(tmp_path, world, W, randomize):
    """world_size 2 (equal shards) and 3 (whole-pair shards with the per-step split; ragged shards with the fixed parity
    halves) reproduce the single-process chain exactly."""
    ref = _run(1, W, 25, tmp_path, randomize=randomize)
    got = _run(world, W, 25, tmp_path, randomize=randomize)
    assert torch.equal(ref["pos"], got["pos"])
    assert torch.equal(ref["lp"], got["lp"])
    assert ref["acc"] == got["acc"] and 0.2 < ref["acc"] < 0.9


def test_stretch_move_samples_the_target(tmp_path):
    out = _run(1, 512, 300, tmp_path)
    pos = out["pos"]
    assert torch.allclose(pos.mean(0), MU, atol=float(4 * SIG.max() / np.sqrt(512)))
    assert torch.allclose(pos.std(0), SIG, rtol=0.2)
    assert torch.allclose(out["lp"], gauss_logp(pos))


REF_MOVES = (("kde", 0.30), ("de", 0.70))  # sn/pantheon.py:114-117


def test_reference_move_mixture_is_bit_identical_across_ranks(tmp_path):
    ref = _run(1, 64, 20, tmp_path, REF_MOVES)
    got = _run(2, 64, 20, tmp_path, REF_MOVES)
    assert torch.equal(ref["pos"], got["pos"]) and torch.equal(ref["lp"], got["lp"])
    assert ref["acc"] == got["acc"] and 0.05 < ref["acc"] < 0.95


@pytest.mark.parametrize("moves", [(("de", 1.0),), (("kde", 1.0),), REF_MOVES])
def test_de_and_kde_moves_sample_the_target(tmp_path, moves):
    out = _run(1, 512, 200, tmp_path, moves)
    pos = out["pos"]
    assert torch.allclose(pos.mean(0), MU, atol=float(4 * SIG.max() / np.sqrt(512)))
    assert torch.allclose(pos.std(0), SIG, rtol=0.2)
    assert torch.allclose(out["lp"], gauss_logp(pos))


def test_kde_density_matches_scipy():
    from scipy.stats import gaussian_kde
    g = torch.Generator().manual_seed(3)
    comp = MU + SIG * torch.randn(200, 3, generator=g, dtype=torch.float64)
    pts = MU + SIG * torch.randn(50, 3, generator=g, dtype=torch.float64)
    nc, d = comp.shape
    h = (nc * (d + 2) / 4.0) ** (-1.0 / (d + 4))
    cen = comp - comp.mean(0)
    chol = torch.linalg.cholesky((cen.T @ cen) / (nc - 1) * h * h)
    log_norm = -np.log(nc) - 0.5 * d * np.log(2 * np.pi) - float(torch.log(torch.diagonal(chol)).sum())
    got = moves_torch.TensorMoves.kde_logpdf(pts, comp, torch.linalg.inv(chol).T.contiguous(), log_norm)
    ref = gaussian_kde(comp.numpy().T, bw_method="silverman").logpdf(pts.numpy().T)
    np.testing.assert_allclose(got.numpy(), ref, rtol=1e-10)


def test_stream_keys_never_collide_across_steps_halves_and_streams():
    """with an arithmetic key the KDE move's noise streams of step t were the partner / accept streams of
    step t + 1.  Every (step, half, stream) a run at ndim = 16 touches must have its own key, also across seeds."""
    ens = load_pkg().ensemble
    streams = list(range(0, 4 + 2 * 16 + 2)) + [ens.MAX_STREAMS - 1]
    keys = {}
    for seed in (41, 42, 43):
        for step in list(range(0, 6)) + [1000003, 1000004]:
            for half in (0, 1):
                for s in streams:
                    k = ens.stream_key(seed, step, half, s)
                    assert k not in keys, f"key collision: {(seed, step, half, s)} vs {keys[k]}"
                    keys[k] = (seed, step, half, s)
    ids = torch.arange(0, 4096, dtype=torch.int64)
    tm = _tensor_moves()
    # the two collisions the advisor ran
    assert not torch.equal(tm.uniform01(42, 5, 0, ids, 10), tm.uniform01(42, 6, 0, ids, 2))
    assert not torch.equal(tm.uniform01(42, 5, 0, ids, 8), tm.uniform01(42, 6, 0, ids, 0))
    assert not torch.equal(tm.uniform01(42, 5, 1, ids, 8), tm.uniform01(42, 6, 1, ids, 0))
    assert not torch.equal(tm.uniform01(42, 1000003, 0, ids, 0), tm.uniform01(43, 0, 0, ids, 0))
    # per-walker numbers of two (step, stream) pairs are uncorrelated
    a, b = tm.uniform01(42, 5, 0, ids, 10), tm.uniform01(42, 6, 0, ids, 2)
    assert abs(float(((a - 0.5) * (b - 0.5)).mean())) < 5e-3
    with pytest.raises(ValueError):
        ens.stream_key(1, 0, 0, ens.MAX_STREAMS)


def test_the_driver_has_no_cpu_fallback():
    with pytest.raises(RuntimeError, match="no tensor-library fallback"):
        load_pkg().ensemble.ShardedEnsemble(gauss_logp, _init_positions(8))


def test_randomized_halves_change_every_step_and_stay_balanced():
    """emcee's RedBlueMove re-draws the halves every step; here whole pairs flip (counter-based bit per pair and step):
    each half holds exactly one member of every pair, the assignment differs from step to step (the kernels' own hash is
    checked against this one in tests/test_gpu_parity.py), and odd shard boundaries are refused."""
    E = load_pkg().ensemble
    pairs = torch.arange(0, 4096, dtype=torch.int64)
    flips = []
    for step in range(4):
        key = E.stream_key(7, step, 0, E._SPLIT_STREAM)
        f = moves_torch.flips_from_key(key, pairs)
        assert set(f.tolist()) == {0, 1} and abs(float(f.double().mean()) - 0.5) < 0.03
        flips.append(f)
    assert not torch.equal(flips[0], flips[1]) and not torch.equal(flips[1], flips[2])
    assert torch.equal(moves_torch.flips_from_key(0, pairs), torch.zeros_like(pairs))


def _ragged_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        load_pkg().ensemble.ShardedEnsemble(gauss_logp, _init_positions(50), moves_impl=_tensor_moves(), randomize_split=True)
        ok = False
    except ValueError as e:
        ok = "whole walker pairs" in str(e)
    assert ok
    dist.barrier()
    dist.destroy_process_group()


def test_randomize_split_refuses_shards_that_cut_a_pair():
    mp.spawn(_ragged_worker, args=(3, _free_port()), nprocs=3, join=True)
