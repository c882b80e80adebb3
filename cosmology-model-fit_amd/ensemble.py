"""
Walker ensemble sharded over the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests and for rank processes
that share one GPU, where the all-gather is staged through the host).  Rank r owns a contiguous slice of the W walkers.
The log-probability of a walker depends only on its own theta, so the evaluation itself needs no collective; the
ensemble MOVE does: the update of one split draws the partners of every active walker from the complementary set, which
lives on all ranks -> one all-gather of the walker positions per split update (W_local x ndim doubles per rank; 128 KiB
at 4096 x 4: latency-bound, so a single flat all-gather, no bucketing).

Moves (the ones the reference's scripts configure on emcee): the Goodman & Weare stretch move (emcee's
default, used by the reference's quasars/ scripts), the differential-evolution move and the KDE move with
Silverman's bandwidth (``moves = [(KDEMove(bw_method="silverman"), 0.30), (DEMove(), 0.70)]``,
sn/pantheon.py:114-117).  One move is drawn per step with the configured weights, as emcee does.
Random numbers come from a counter-based generator keyed on (seed, step, split, GLOBAL walker index,
stream), and every quantity a proposal uses is computed from the gathered (global) ensemble with
per-element arithmetic, so a chain is bit-identical for any number of ranks — that is what the gloo
tests check (tensor statement on CPU; the library's kernels with 2 and 3 rank processes on one GPU).

`log_prob_fn(theta[W, ndim] tensor) -> tensor[W]` is pluggable: on a GPU it is
``LikelihoodEngine.torch_log_prob`` (HIP kernels through cf_eval_device on the current stream).

The proposal and accept arithmetic runs in the library's own kernels (cf_ens_active_set / cf_ens_kde_prepare /
cf_ens_propose / cf_ens_accept, csrc/cosmofit_ensemble.hip: three or four launches per split update, no host round trip).
This module is the DRIVER only: sharding, the all-gather, the step loop, the random-stream keys.  There is no
tensor-library fallback: an ensemble on CPU tensors needs an explicit ``moves_impl`` -- the test-suite passes
``oracle.moves_torch.TensorMoves``, the tensor statement of the same moves, to rehearse the multi-rank logic under gloo and
to check the kernels.

Splits.  emcee's RedBlueMove updates ``nsplits`` sets of walkers in turn, each from the union of the others, and re-draws
the sets every step; StretchMove and KDEMove use two sets, **DEMove three** (emcee's ``DEMove.__init__`` sets
``nsplits = 3``), so a DE step here is three split updates of a third of the ensemble each, proposing from the other two
thirds (``de_splits=2`` gives the two-halves variant).  Walkers are taken in consecutive groups of S = nsplits: walker
S c + b belongs to split perm_c[b]; ``randomize_split=True`` (default) draws perm_c as a counter-based random permutation
per group and step, so the sets change every step while every split keeps exactly one member of every group (any shard
owns its fair share of each split, no host round trip, chains independent of the number of ranks and of where the shard
boundaries fall); ``randomize_split=False`` keeps the classes (index mod S) for the whole run.
"""
from __future__ import annotations

import math
from typing import Callable, Tuple

import torch

try:
    import torch.distributed as dist
except Exception:  # pragma: no cover
    dist = None


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, near-equal slices: the first (n_total % world) ranks own one walker more."""
    base, extra = divmod(n_total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


# ---- random-stream keys (counter-based generator: splitmix64 finaliser, identical in the HIP kernels) ---------------
_U64 = 0xFFFFFFFFFFFFFFFF
MAX_STREAMS = 256  # random streams per (seed, step, half): the KDE move uses 4 + 2 ndim + 1 <= 37
_SPLIT_STREAM = MAX_STREAMS - 2  # key of the per-step pair flips
_MOVE_STREAM = MAX_STREAMS - 1   # key of the per-step move draw


def _mix_int(x: int) -> int:
    """splitmix64 finaliser on a Python int (a bijection of 64-bit words)."""
    x &= _U64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _U64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _U64
    return x ^ (x >> 31)


def stream_key(seed: int, step: int, half: int, stream: int = 0) -> int:
    """Unsigned 64-bit key of random stream `stream` for (seed, step, half = the split being updated: 0, 1 [, 2]); stream s has
    key(stream 0) + s.

    seed and step each pass through their own hash round, so the 768 consecutive keys of one (seed, step) -- 256
    streams for each split -- sit at a pseudo-random 64-bit offset: the streams of different steps, halves or seeds
    never share a key (an arithmetic key such as (seed * K + step) * 8 + half * 4 + stream makes the KDE move's noise
    streams of step t coincide with the partner / accept streams of step t + 1)."""
    if not (0 <= stream < MAX_STREAMS) or half not in (0, 1, 2):
        raise ValueError(f"stream must be in 0..{MAX_STREAMS - 1} and the split in (0, 1, 2)")
    base = _mix_int(_mix_int(seed * 0x9E3779B97F4A7C15 + 0x5851F42D4C957F2D) ^ (step & _U64))
    return (base + (half << 8) + stream) & _U64


def uniform01_scalar(key: int, counter: int) -> float:
    """One uniform in [0, 1) for (key, counter): the kernels' ens_uniform on Python ints (the per-step move draw)."""
    x = _mix_int(((counter & _U64) * 0x9E3779B97F4A7C15 + key) & _U64)
    x = _mix_int((x + 0x9E3779B97F4A7C15) & _U64)
    return (x >> 11) * (1.0 / 9007199254740992.0)


_PERM3 = ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0))


def split_perm(split_key: int, n_splits: int, group: int):
    """perm_c of walker group `group`: walker S c + b belongs to split perm_c[b] (cosmofit_ensemble.hip: ens_split_of)."""
    if split_key == 0:
        return tuple(range(n_splits))
    x = _mix_int(((group & _U64) * 0x9E3779B97F4A7C15 + split_key) & _U64)
    x = _mix_int((x + 0x9E3779B97F4A7C15) & _U64)
    if n_splits == 2:
        f = x >> 63
        return (f, 1 - f)
    return _PERM3[((x >> 40) * 6) >> 24]


def active_count(split_key: int, n_splits: int, split: int, start: int, stop: int) -> int:
    """Walkers of split `split` in the shard [start, stop): one per group, less the cut first / last group's member
    (the library's cf_ens_active_count, on Python ints: the driver sizes its launches without a device round trip)."""
    if stop <= start:
        return 0
    c0, c1 = start // n_splits, (stop - 1) // n_splits
    n = c1 - c0 + 1
    id0 = n_splits * c0 + split_perm(split_key, n_splits, c0).index(split)
    id1 = n_splits * c1 + split_perm(split_key, n_splits, c1).index(split)
    if not (start <= id0 < stop):
        n -= 1
    if c1 > c0 and not (start <= id1 < stop):
        n -= 1
    return n


REFERENCE_MOVES = (("kde", 0.30), ("de", 0.70))  # sn/pantheon.py:114-117
STRETCH_ONLY = (("stretch", 1.0),)
_KIND = {"stretch": 0, "de": 1, "kde": 2}


class NativeMoves:
    """The library's kernels on device-resident tensors, asynchronous on torch's current stream."""

    def __init__(self, e):
        from . import _lib as L

        self.L, self.lib = L, L.lib()  # raises if the HIP library is missing
        dev, nmax = e.x.device, (e.stop - e.start) // 2 + 2
        self.y = torch.empty((nmax, e.ndim), dtype=torch.float64, device=dev)
        self.logfac = torch.empty(nmax, dtype=torch.float64, device=dev)
        self.kde_params = torch.empty(2 * e.ndim * e.ndim + 1, dtype=torch.float64, device=dev)
        self.kde_wc = torch.empty(((2 * e.n_total + 2) // 3, e.ndim), dtype=torch.float64, device=dev)
        self.n_acc = torch.zeros(1, dtype=torch.int64, device=dev)
        self.ids = torch.empty(nmax, dtype=torch.int64, device=dev)
        self.idx = torch.empty(nmax, dtype=torch.int64, device=dev)

    def split_step(self, e, move, n_splits, split, allpos, split_key):
        L, lib = self.L, self.lib
        kind = _KIND[move]
        stream = torch.cuda.current_stream(e.x.device).cuda_stream
        # this step's active set: the shard's walkers of split `split` -- one small launch, no host sync (the count is
        # arithmetic on the shard bounds and two group permutations)
        n = active_count(split_key, n_splits, split, e.start, e.stop)
        if n == 0:
            return
        ids, idx = self.ids[:n], self.idx[:n]
        L.check(lib.cf_ens_active_set(split_key, n_splits, split, e.start, e.stop, ids.data_ptr(), idx.data_ptr(), stream))
        key0 = stream_key(e.seed, e.step_count, split)
        if kind == 2:
            L.check(lib.cf_ens_kde_prepare(allpos.data_ptr(), e.n_total, e.ndim, n_splits, split, split_key,
                                           self.kde_params.data_ptr(), self.kde_wc.data_ptr(), stream))
        y, logfac = self.y[:n], self.logfac[:n]
        L.check(lib.cf_ens_propose(kind, allpos.data_ptr(), e.n_total, e.ndim, n_splits, split, split_key, ids.data_ptr(), n, key0,
                                   float(e.a), float(e.de_sigma), self.kde_params.data_ptr(), self.kde_wc.data_ptr(),
                                   y.data_ptr(), logfac.data_ptr(), stream))
        lp_new = e.log_prob_fn(y)
        L.check(lib.cf_ens_accept(ids.data_ptr(), idx.data_ptr(), n, e.ndim, key0, y.data_ptr(), lp_new.data_ptr(),
                                  logfac.data_ptr(), e.x.data_ptr(), e.logp.data_ptr(), self.n_acc.data_ptr(), stream))
        e._n_proposed += n


class ShardedEnsemble:
    def __init__(self, log_prob_fn: Callable[[torch.Tensor], torch.Tensor], positions: torch.Tensor, *,
                 seed: int = 42, a: float = 2.0, moves=STRETCH_ONLY, de_sigma: float = 1e-5, group=None,
                 randomize_split: bool = True, de_splits: int = 3, moves_impl=None):
        """positions: [W_total, ndim] float64 initial ensemble, identical on every rank (it is sliced here).
        moves: sequence of (name, weight), name in {"stretch", "de", "kde"}; one is drawn per step.
        randomize_split: re-draw the splits every step (a random permutation per group of walkers); any shard boundaries.
        de_splits: sets of a DE step: 3 = emcee's DEMove (default), 2 = two halves like the other moves.
        moves_impl: None = the library's kernels (positions must live on a GPU); tests pass the tensor statement."""
        names = set(_KIND)
        if not moves or any(m not in names or w <= 0 for m, w in moves):
            raise ValueError(f"moves must be a non-empty sequence of (name in {sorted(names)}, weight > 0)")
        if de_splits not in (2, 3):
            raise ValueError("de_splits must be 3 (emcee's DEMove) or 2")
        tot = float(sum(w for _, w in moves))
        self.moves = [(m, w / tot) for m, w in moves]
        self.de_sigma, self.de_splits = de_sigma, de_splits
        self.group = group
        self.distributed = dist is not None and dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.n_total, self.ndim = positions.shape
        # three splits (the DE move as emcee runs it) need a walker and an ordered pair of partners in the other two thirds: 6;
        # the two-set moves need 4 (cosmofit_ensemble.hip: ens_check keeps the same two bounds)
        need = 6 if (de_splits == 3 and any(m == "de" for m, _ in self.moves)) else 4
        if self.n_total % 2 or self.n_total < need:
            raise ValueError(f"the ensemble needs an even number (>= {need}) of walkers")
        if any(m == "kde" for m, _ in self.moves) and self.n_total // 2 <= self.ndim:
            raise ValueError("the KDE move needs more than ndim walkers in each half (the complementary set's covariance must be regular)")
        self.start, self.stop = shard_bounds(self.n_total, self.world, self.rank)
        counts = [shard_bounds(self.n_total, self.world, r) for r in range(self.world)]
        self.randomize_split = randomize_split
        self.log_prob_fn = log_prob_fn
        self.x = positions[self.start:self.stop].clone().contiguous()
        self.ids = torch.arange(self.start, self.stop, dtype=torch.int64, device=self.x.device)
        self.seed, self.a, self.step_count = seed, a, 0
        self.logp = self.log_prob_fn(self.x)
        self._n_accepted = 0
        self._n_proposed = 0
        self._equal = len({b - a_ for a_, b in counts}) == 1
        self._max_local = max(b - a_ for a_, b in counts)
        self._counts = counts
        self._allpos = None
        # a backend without device collectives (gloo: rank processes sharing one GPU, CPU rehearsals): the all-gather of
        # device-resident positions is staged through the host
        self._host_staged = self.distributed and self.x.is_cuda and "nccl" not in str(dist.get_backend(group)).lower()
        if moves_impl is None:
            if not self.x.is_cuda:
                raise RuntimeError("ShardedEnsemble runs its moves in the library's HIP kernels: the positions must be on an "
                                   "MI355X (there is no tensor-library fallback; tests pass oracle.moves_torch.TensorMoves)")
            moves_impl = NativeMoves(self)
        self.impl = moves_impl

    # ---- the exchange step ---------------------------------------------------------------------------
    def _all_gather(self, out: torch.Tensor, inp: torch.Tensor) -> torch.Tensor:
        """dist.all_gather_into_tensor(out, inp); through host copies when the backend has no device collectives."""
        if self._host_staged:
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, inp.cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, inp, group=self.group)
        return out

    def _gather_rows(self, local: torch.Tensor) -> torch.Tensor:
        """Every rank's rows of `local` ([W_local, ...]) in rank order ([W_total, ...]) on every rank."""
        if self._equal:
            return self._all_gather(torch.empty((self.n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device), local)
        # ragged shards: pad to the largest shard, gather, strip
        pad = torch.zeros((self._max_local,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        buf = self._all_gather(torch.empty((self.world * self._max_local,) + tuple(local.shape[1:]), dtype=local.dtype,
                                           device=local.device), pad)
        return torch.cat([buf[r * self._max_local: r * self._max_local + (b - a_)] for r, (a_, b) in enumerate(self._counts)])

    def gather_positions(self) -> torch.Tensor:
        """All walkers' positions [W_total, ndim] on every rank (one collective)."""
        if self.world == 1:
            return self.x
        if self._equal:
            if self._allpos is None:  # one buffer for the life of the ensemble: the collective runs every split update
                self._allpos = torch.empty((self.n_total, self.ndim), dtype=self.x.dtype, device=self.x.device)
            return self._all_gather(self._allpos, self.x)
        return self._gather_rows(self.x)

    def _pick_move(self) -> str:
        u = uniform01_scalar(stream_key(self.seed, self.step_count, 0, _MOVE_STREAM), 0)
        acc = 0.0
        for name, w in self.moves:
            acc += w
            if u < acc:
                return name
        return self.moves[-1][0]

    # ---- one ensemble step = the split updates of the step's move (two halves; three thirds for DE) -------------------
    def step(self):
        """Per split: all-gather -> [KDE fit] -> propose -> log P -> accept; with the library's kernels everything is
        asynchronous on the current stream."""
        move = self._pick_move()
        n_splits = self.de_splits if move == "de" else 2
        split_key = stream_key(self.seed, self.step_count, 0, _SPLIT_STREAM) if self.randomize_split else 0
        for split in range(n_splits):
            self.impl.split_step(self, move, n_splits, split, self.gather_positions(), split_key)
        self.step_count += 1

    @property
    def n_accepted(self) -> int:
        """Accepted moves of this rank's walkers so far (the kernels count on the device: reading synchronises)."""
        return int(self.impl.n_acc.item()) if isinstance(self.impl, NativeMoves) else self._n_accepted

    @property
    def n_proposed(self) -> int:
        return self._n_proposed

    def run(self, n_steps: int):
        for _ in range(n_steps):
            self.step()
        return self

    def acceptance_fraction(self) -> float:
        acc = torch.tensor([self.n_accepted, self.n_proposed], dtype=torch.float64,
                           device="cpu" if self._host_staged else self.x.device)
        if self.world > 1:
            dist.all_reduce(acc, group=self.group)
        return float(acc[0] / torch.clamp(acc[1], min=1.0))

    def full_state(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(positions [W_total, ndim], log-prob [W_total]) gathered on every rank (for tests / check-pointing)."""
        pos = self.gather_positions()
        if self.world == 1:
            return pos, self.logp
        return pos, self._gather_rows(self.logp)
