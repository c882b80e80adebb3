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


@pytest.mark.parametrize("world,W,randomize", [(2, 64, True), (2, 64, False), (3, 48, True), (3, 50, False), (3, 50, True)])
def test_chain_is_bit_identical_for_any_number_of_ranks(tmp_path, world, W, randomize):
    """world_size 2 (equal shards) and 3 (equal and ragged shards, also with shard boundaries that cut a walker pair) reproduce
    the single-process chain exactly, with the fixed classes and with the per-step re-drawn splits."""
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


@pytest.mark.parametrize("world,W", [(2, 64), (3, 50)])
def test_reference_move_mixture_is_bit_identical_across_ranks(tmp_path, world, W):
    """KDE on two halves + DE on three thirds (emcee's DEMove: nsplits = 3), shards that cut pairs and triples."""
    ref = _run(1, W, 20, tmp_path, REF_MOVES)
    got = _run(world, W, 20, tmp_path, REF_MOVES)
    assert torch.equal(ref["pos"], got["pos"]) and torch.equal(ref["lp"], got["lp"])
    assert ref["acc"] == got["acc"] and 0.05 < ref["acc"] < 0.95


def test_de_move_runs_on_three_splits_like_emcee():
    """emcee's DEMove sets nsplits = 3 (the reference hands it 70 % of the steps, sn/pantheon.py:114-117): a DE step is three
    split updates of W / 3 walkers each, every proposal built from two walkers of the OTHER two thirds; the other moves use two
    halves; de_splits=2 is the two-halves variant."""
    E = load_pkg().ensemble
    calls = []

    class Spy(moves_torch.TensorMoves):
        def split_step(self, e, move, n_splits, split, allpos, split_key):
            all_ids = torch.arange(e.n_total, dtype=torch.int64)
            sp = moves_torch.split_of(split_key, n_splits, all_ids)
            calls.append((move, n_splits, split, int((sp == split).sum()), int((sp != split).sum())))
            return super().split_step(e, move, n_splits, split, allpos, split_key)

    ens = E.ShardedEnsemble(gauss_logp, _init_positions(150), seed=5, moves=(("de", 1.0),), moves_impl=Spy(E.stream_key))
    ens.run(2)
    assert calls == [("de", 3, s, 50, 100) for s in (0, 1, 2)] * 2 and ens.n_proposed == 300
    calls.clear()
    ens = E.ShardedEnsemble(gauss_logp, _init_positions(32), seed=5, moves=(("de", 1.0),), moves_impl=Spy(E.stream_key))
    ens.run(1)  # 32 walkers (BASELINE configs[0]): thirds of 10-11, every walker proposed exactly once per step
    assert [c[:3] for c in calls] == [("de", 3, s) for s in (0, 1, 2)] and sum(c[3] for c in calls) == 32
    assert all(10 <= c[3] <= 11 and c[3] + c[4] == 32 for c in calls)
    calls.clear()
    E.ShardedEnsemble(gauss_logp, _init_positions(32), seed=5, moves=(("kde", 1.0),), moves_impl=Spy(E.stream_key)).run(1)
    E.ShardedEnsemble(gauss_logp, _init_positions(32), seed=5, moves=(("de", 1.0),), de_splits=2, moves_impl=Spy(E.stream_key)).run(1)
    assert [c[:5] for c in calls] == [("kde", 2, 0, 16, 16), ("kde", 2, 1, 16, 16), ("de", 2, 0, 16, 16), ("de", 2, 1, 16, 16)]


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
            for half in (0, 1, 2):
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


def test_randomized_splits_change_every_step_and_stay_balanced():
    """emcee's RedBlueMove re-draws its sets every step; here every consecutive group of S walkers is permuted (counter-based
    permutation per group and step): each split holds exactly one member of every group, the assignment differs from step to step
    (the kernels' own hash is checked against this one in tests/test_gpu_parity.py), all 2 / 6 permutations occur equally often."""
    E = load_pkg().ensemble
    ids = torch.arange(0, 6 * 4096, dtype=torch.int64)
    for S in (2, 3):
        seen = []
        for step in range(4):
            key = E.stream_key(7, step, 0, E._SPLIT_STREAM)
            sp = moves_torch.split_of(key, S, ids).reshape(-1, S)
            assert torch.equal(sp.sort(dim=1).values, torch.arange(S).expand_as(sp))
            code = (sp * torch.tensor([S ** k for k in range(S)])).sum(dim=1)
            freq = torch.bincount(code).double()
            freq = freq[freq > 0] / code.numel()
            assert freq.numel() == (2 if S == 2 else 6) and float((freq - 1.0 / freq.numel()).abs().max()) < 0.015
            seen.append(sp)
        assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
        assert torch.equal(moves_torch.split_of(0, S, ids), ids % S)
    pairs = torch.arange(0, 4096, dtype=torch.int64)  # S = 2 is the pair flip of rounds 1-2 (chains of those rounds unchanged)
    key = E.stream_key(7, 1, 0, E._SPLIT_STREAM)
    assert torch.equal(moves_torch.split_of(key, 2, 2 * pairs), moves_torch.flips_from_key(key, pairs))


def test_odd_and_tiny_ensembles_are_refused():
    E = load_pkg().ensemble
    with pytest.raises(ValueError, match="even number"):
        E.ShardedEnsemble(gauss_logp, _init_positions(9), moves_impl=_tensor_moves())
    with pytest.raises(ValueError, match="even number"):
        E.ShardedEnsemble(gauss_logp, _init_positions(2), moves_impl=_tensor_moves())
    with pytest.raises(ValueError, match="de_splits"):
        E.ShardedEnsemble(gauss_logp, _init_positions(8), moves_impl=_tensor_moves(), de_splits=4)
    # only the three-split DE move needs 6 walkers: the two-set moves run on 4, as the library's own check (ens_check) allows
    for kw in (dict(moves=(("stretch", 1.0),)), dict(moves=(("de", 1.0),), de_splits=2)):
        ens = E.ShardedEnsemble(gauss_logp, _init_positions(4), seed=3, moves_impl=_tensor_moves(), **kw)
        ens.run(2)
        assert ens.n_proposed == 8
    with pytest.raises(ValueError, match=">= 6"):
        E.ShardedEnsemble(gauss_logp, _init_positions(4), moves=(("de", 1.0),), moves_impl=_tensor_moves())
    with pytest.raises(ValueError, match="KDE move needs"):
        E.ShardedEnsemble(gauss_logp, _init_positions(4), moves=(("kde", 1.0),), moves_impl=_tensor_moves())
