"""
Walker ensemble sharded over the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).
Rank r owns a contiguous slice of the W walkers.  The log-probability of a walker depends only on its own
theta, so the evaluation itself needs no collective; the ensemble MOVE does: a red/blue half-step draws the
partner of every active walker from the complementary half, which lives on all ranks -> one all-gather of the
walker positions per half-step (W_local x ndim doubles per rank; 128 KiB at 4096 x 4: latency-bound, so a
single flat all-gather, no bucketing).

Moves (the ones the reference's scripts configure on emcee): the Goodman & Weare stretch move (emcee's
default, used by the reference's quasars/ scripts), the differential-evolution move and the KDE move with
Silverman's bandwidth (``moves = [(KDEMove(bw_method="silverman"), 0.30), (DEMove(), 0.70)]``,
sn/pantheon.py:114-117).  One move is drawn per step with the configured weights, as emcee does.
Random numbers come from a counter-based generator keyed on (seed, step, half, GLOBAL walker index,
stream), and every quantity a proposal uses is computed from the gathered (global) ensemble with
per-element arithmetic, so a chain is bit-identical for any number of ranks — that is what the gloo
tests check.

`log_prob_fn(theta[W, ndim] tensor) -> tensor[W]` is pluggable: on a GPU it is
``LikelihoodEngine.torch_log_prob`` (HIP kernels through cf_eval_device on the current stream).

On a GPU the proposal and accept arithmetic runs in the library's own kernels (cf_ens_kde_prepare / cf_ens_propose /
cf_ens_accept, csrc/cosmofit_ensemble.hip: the same counter-based random numbers and formulae as the tensor code
below, two or three launches per half-step and no host round trip); the tensor code is the CPU path of the gloo
tests and the statement of what those kernels compute.
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


# ---- counter-based uniform random numbers (splitmix64 finaliser), identical on CPU and GPU -----------------
_M1 = -7046029254386353131  # 0x9E3779B97F4A7C15 as int64
_M2 = -4658895280553007687  # 0xBF58476D1CE4E5B9
_M3 = -7723592293110705685  # 0x94D049BB133111EB


def _lsr(x: torch.Tensor, s: int) -> torch.Tensor:
    """Logical shift right of int64 (torch's >> is arithmetic)."""
    return (x >> s) & ((1 << (64 - s)) - 1)


def _mix(x: torch.Tensor) -> torch.Tensor:
    x = (x ^ _lsr(x, 30)) * _M2
    x = (x ^ _lsr(x, 27)) * _M3
    return x ^ _lsr(x, 31)


_U64 = 0xFFFFFFFFFFFFFFFF
MAX_STREAMS = 256  # random streams per (seed, step, half): the KDE move uses 4 + 2 ndim + 1 <= 37


def _mix_int(x: int) -> int:
    """splitmix64 finaliser on a Python int (the same bijection of 64-bit words as `_mix`)."""
    x &= _U64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _U64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _U64
    return x ^ (x >> 31)


def stream_key(seed: int, step: int, half: int, stream: int = 0) -> int:
    """Unsigned 64-bit key of random stream `stream` for (seed, step, half); stream s has key(stream 0) + s.

    seed and step each pass through their own hash round, so the 512 consecutive keys of one (seed, step) -- 256
    streams for each half -- sit at a pseudo-random 64-bit offset: the streams of different steps, halves or seeds
    never share a key (an arithmetic key such as (seed * K + step) * 8 + half * 4 + stream makes the KDE move's noise
    streams of step t coincide with the partner / accept streams of step t + 1)."""
    if not (0 <= stream < MAX_STREAMS) or half not in (0, 1):
        raise ValueError(f"stream must be in 0..{MAX_STREAMS - 1} and half in (0, 1)")
    base = _mix_int(_mix_int(seed * 0x9E3779B97F4A7C15 + 0x5851F42D4C957F2D) ^ (step & _U64))
    return (base + (half << 8) + stream) & _U64


def uniform01(seed: int, step: int, half: int, walker_ids: torch.Tensor, stream: int) -> torch.Tensor:
    """float64 uniforms in [0, 1), a pure function of its arguments (walker_ids: int64 tensor)."""
    key = stream_key(seed, step, half, stream)
    key = key - (1 << 64) if key >= (1 << 63) else key  # as signed int64
    x = _mix(walker_ids * _M1 + key)
    x = _mix(x + _M1)
    return _lsr(x, 11).to(torch.float64) * (1.0 / 9007199254740992.0)


def normal01(seed: int, step: int, half: int, walker_ids: torch.Tensor, stream: int) -> torch.Tensor:
    """Standard normals by Box-Muller from two counter-based uniforms (streams `stream`, `stream + 1`)."""
    u1 = 1.0 - uniform01(seed, step, half, walker_ids, stream)  # (0, 1]
    u2 = uniform01(seed, step, half, walker_ids, stream + 1)
    return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos((2.0 * math.pi) * u2)


REFERENCE_MOVES = (("kde", 0.30), ("de", 0.70))  # sn/pantheon.py:114-117
STRETCH_ONLY = (("stretch", 1.0),)


class ShardedEnsemble:
    def __init__(self, log_prob_fn: Callable[[torch.Tensor], torch.Tensor], positions: torch.Tensor, *,
                 seed: int = 42, a: float = 2.0, moves=STRETCH_ONLY, de_sigma: float = 1e-5, group=None):
        """positions: [W_total, ndim] float64 initial ensemble, identical on every rank (it is sliced here).
        moves: sequence of (name, weight), name in {"stretch", "de", "kde"}; one is drawn per step."""
        names = {"stretch", "de", "kde"}
        if not moves or any(m not in names or w <= 0 for m, w in moves):
            raise ValueError(f"moves must be a non-empty sequence of (name in {sorted(names)}, weight > 0)")
        tot = float(sum(w for _, w in moves))
        self.moves = [(m, w / tot) for m, w in moves]
        self.de_sigma = de_sigma
        self.group = group
        self.distributed = dist is not None and dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.n_total, self.ndim = positions.shape
        if self.n_total % 2:
            raise ValueError("the ensemble needs an even number of walkers (two halves)")
        self.start, self.stop = shard_bounds(self.n_total, self.world, self.rank)
        self.log_prob_fn = log_prob_fn
        self.x = positions[self.start:self.stop].clone().contiguous()
        self.ids = torch.arange(self.start, self.stop, dtype=torch.int64, device=self.x.device)
        self.seed, self.a, self.step_count = seed, a, 0
        self.logp = self.log_prob_fn(self.x)
        self._n_accepted = 0
        self._n_proposed = 0
        counts = [shard_bounds(self.n_total, self.world, r) for r in range(self.world)]
        self._equal = len({b - a_ for a_, b in counts}) == 1
        self._max_local = max(b - a_ for a_, b in counts)
        self._counts = counts
        self._allpos = None
        # the two halves of this rank's shard: local indices and global ids, fixed for the life of the ensemble
        self._act_idx = [torch.nonzero((self.ids % 2) == h, as_tuple=False)[:, 0].contiguous() for h in (0, 1)]
        self._act_ids = [self.ids[i].contiguous() for i in self._act_idx]
        self._native = self.x.is_cuda
        if self._native:
            from . import _lib as L

            self._L = L
            self._lib = L.lib()  # raises if the HIP library is missing: no silent tensor-code fallback on a GPU
            dev, nmax = self.x.device, max(int(i.numel()) for i in self._act_idx)
            self._y = torch.empty((max(nmax, 1), self.ndim), dtype=torch.float64, device=dev)
            self._logfac = torch.empty(max(nmax, 1), dtype=torch.float64, device=dev)
            self._kde_params = torch.empty(2 * self.ndim * self.ndim + 1, dtype=torch.float64, device=dev)
            self._kde_wc = torch.empty((self.n_total // 2, self.ndim), dtype=torch.float64, device=dev)
            self._n_acc_dev = torch.zeros(1, dtype=torch.int64, device=dev)

    # ---- the exchange step ---------------------------------------------------------------------------
    def gather_positions(self) -> torch.Tensor:
        """All walkers' positions [W_total, ndim] on every rank (one collective)."""
        if self.world == 1:
            return self.x
        if self._equal:
            if self._allpos is None:  # one buffer for the life of the ensemble: the collective runs every half-step
                self._allpos = torch.empty((self.n_total, self.ndim), dtype=self.x.dtype, device=self.x.device)
            dist.all_gather_into_tensor(self._allpos, self.x, group=self.group)
            return self._allpos
        # ragged shards: pad to the largest shard, gather, strip
        pad = torch.zeros((self._max_local, self.ndim), dtype=self.x.dtype, device=self.x.device)
        pad[: self.x.shape[0]] = self.x
        buf = torch.empty((self.world * self._max_local, self.ndim), dtype=self.x.dtype, device=self.x.device)
        dist.all_gather_into_tensor(buf, pad, group=self.group)
        return torch.cat([buf[r * self._max_local: r * self._max_local + (b - a_)] for r, (a_, b) in enumerate(self._counts)])

    # ---- proposals: (y [n, ndim], log of the Hastings factor [n]) for the active walkers ----------------
    def _propose_stretch(self, xa, ids, comp, half):
        nc = comp.shape[0]
        j = torch.clamp((uniform01(self.seed, self.step_count, half, ids, 0) * nc).to(torch.int64), max=nc - 1)
        z = ((self.a - 1.0) * uniform01(self.seed, self.step_count, half, ids, 1) + 1.0) ** 2 / self.a
        partner = comp[j]
        return partner + z[:, None] * (xa - partner), (self.ndim - 1) * torch.log(z)

    def _propose_de(self, xa, ids, comp, half):
        """emcee DEMove: q = s + g0 (1 + sigma N(0,1)) (c_j - c_k), j != k, g0 = 2.38 / sqrt(2 ndim); symmetric."""
        nc = comp.shape[0]
        j = torch.clamp((uniform01(self.seed, self.step_count, half, ids, 0) * nc).to(torch.int64), max=nc - 1)
        k = torch.clamp((uniform01(self.seed, self.step_count, half, ids, 1) * (nc - 1)).to(torch.int64), max=nc - 2)
        k = k + (k >= j).to(torch.int64)
        gamma = (2.38 / math.sqrt(2 * self.ndim)) * (1.0 + self.de_sigma * normal01(self.seed, self.step_count, half, ids, 3))
        return xa + gamma[:, None] * (comp[j] - comp[k]), torch.zeros_like(gamma)

    def _kde_logpdf(self, pts, comp, chol_inv_t, log_norm):
        """log of the Gaussian-KDE density of `comp` at `pts`; per-element arithmetic only (rank-count invariant)."""
        out = torch.empty(pts.shape[0], dtype=pts.dtype, device=pts.device)
        wc = comp @ chol_inv_t  # whitened data, same on every rank
        chunk = max(1, (1 << 24) // max(1, comp.shape[0] * self.ndim))
        for a0 in range(0, pts.shape[0], chunk):
            wp = pts[a0:a0 + chunk] @ chol_inv_t
            d2 = ((wp[:, None, :] - wc[None, :, :]) ** 2).sum(dim=2)
            out[a0:a0 + chunk] = torch.logsumexp(-0.5 * d2, dim=1) + log_norm
        return out

    def _propose_kde(self, xa, ids, comp, half):
        """emcee KDEMove(bw_method="silverman"): independence proposal from the Gaussian KDE of the complementary
        set, factor = log kde(s) - log kde(q)."""
        nc, d = comp.shape
        h = (nc * (d + 2) / 4.0) ** (-1.0 / (d + 4))  # scipy.stats.gaussian_kde.silverman_factor
        mean = comp.mean(dim=0)
        cen = comp - mean
        cov = (cen.T @ cen) / (nc - 1) * (h * h)
        chol = torch.linalg.cholesky(cov)
        chol_inv_t = torch.linalg.inv(chol).T.contiguous()
        log_norm = -math.log(nc) - 0.5 * d * math.log(2.0 * math.pi) - float(torch.log(torch.diagonal(chol)).sum())
        j = torch.clamp((uniform01(self.seed, self.step_count, half, ids, 0) * nc).to(torch.int64), max=nc - 1)
        noise = torch.stack([normal01(self.seed, self.step_count, half, ids, 4 + 2 * k) for k in range(d)], dim=1)
        q = comp[j] + noise @ chol.T
        return q, self._kde_logpdf(xa, comp, chol_inv_t, log_norm) - self._kde_logpdf(q, comp, chol_inv_t, log_norm)

    def _pick_move(self) -> str:
        u = float(uniform01(self.seed, self.step_count, 0, torch.tensor([-1], dtype=torch.int64), MAX_STREAMS - 1)[0])
        acc = 0.0
        for name, w in self.moves:
            acc += w
            if u < acc:
                return name
        return self.moves[-1][0]

    # ---- one ensemble step = two red/blue half-steps --------------------------------------------------------
    def _step_native(self, move: str):
        """One step with the library's kernels: per half-step all-gather -> [KDE fit] -> propose -> log P -> accept,
        everything asynchronous on the current stream."""
        L, lib = self._L, self._lib
        kind = {"stretch": 0, "de": 1, "kde": 2}[move]
        stream = torch.cuda.current_stream(self.x.device).cuda_stream
        for half in (0, 1):
            allpos = self.gather_positions()
            ids, idx = self._act_ids[half], self._act_idx[half]
            n = int(ids.numel())
            if n == 0:
                continue
            key0 = stream_key(self.seed, self.step_count, half)
            if kind == 2:
                L.check(lib.cf_ens_kde_prepare(allpos.data_ptr(), self.n_total, self.ndim, half, self._kde_params.data_ptr(),
                                               self._kde_wc.data_ptr(), stream))
            y, logfac = self._y[:n], self._logfac[:n]
            L.check(lib.cf_ens_propose(kind, allpos.data_ptr(), self.n_total, self.ndim, half, ids.data_ptr(), n, key0,
                                       float(self.a), float(self.de_sigma), self._kde_params.data_ptr(), self._kde_wc.data_ptr(),
                                       y.data_ptr(), logfac.data_ptr(), stream))
            lp_new = self.log_prob_fn(y)
            L.check(lib.cf_ens_accept(ids.data_ptr(), idx.data_ptr(), n, self.ndim, key0, y.data_ptr(), lp_new.data_ptr(),
                                      logfac.data_ptr(), self.x.data_ptr(), self.logp.data_ptr(), self._n_acc_dev.data_ptr(), stream))
            self._n_proposed += n
        self.step_count += 1

    def step(self):
        move = self._pick_move()
        if self._native:
            return self._step_native(move)
        propose = {"stretch": self._propose_stretch, "de": self._propose_de, "kde": self._propose_kde}[move]
        for half in (0, 1):
            allpos = self.gather_positions()
            # active set: walkers with global index parity == half; the complementary set is the other parity
            active = (self.ids % 2) == half
            if bool(active.any()):
                ids = self.ids[active]
                comp = allpos[(1 - half)::2]
                u_acc = uniform01(self.seed, self.step_count, half, ids, 2)
                y, log_factor = propose(self.x[active], ids, comp, half)
                lp_new = self.log_prob_fn(y.contiguous())
                log_q = log_factor + lp_new - self.logp[active]
                accept = torch.log(u_acc) < log_q
                idx = torch.nonzero(active, as_tuple=False)[:, 0][accept]
                self.x[idx] = y[accept]
                self.logp[idx] = lp_new[accept]
                self._n_accepted += int(accept.sum())
                self._n_proposed += int(active.sum())
        self.step_count += 1

    @property
    def n_accepted(self) -> int:
        """Accepted moves of this rank's walkers so far (reads the device counter: synchronises)."""
        return int(self._n_acc_dev.item()) if self._native else self._n_accepted

    @property
    def n_proposed(self) -> int:
        return self._n_proposed

    def run(self, n_steps: int):
        for _ in range(n_steps):
            self.step()
        return self

    def acceptance_fraction(self) -> float:
        acc = torch.tensor([self.n_accepted, self.n_proposed], dtype=torch.float64, device=self.x.device)
        if self.world > 1:
            dist.all_reduce(acc, group=self.group)
        return float(acc[0] / torch.clamp(acc[1], min=1.0))

    def full_state(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(positions [W_total, ndim], log-prob [W_total]) gathered on every rank (for tests / check-pointing)."""
        pos = self.gather_positions()
        if self.world == 1:
            return pos, self.logp
        pad = torch.zeros(self._max_local, dtype=self.logp.dtype, device=self.logp.device)
        pad[: self.logp.shape[0]] = self.logp
        buf = torch.empty(self.world * self._max_local, dtype=self.logp.dtype, device=self.logp.device)
        dist.all_gather_into_tensor(buf, pad, group=self.group)
        lp = torch.cat([buf[r * self._max_local: r * self._max_local + (b - a_)] for r, (a_, b) in enumerate(self._counts)])
        return pos, lp
