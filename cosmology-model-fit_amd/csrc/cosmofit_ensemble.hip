// cosmofit_ensemble.hip — proposal and accept kernels of a device-resident ensemble sampler.
//
// The reference hands its likelihood to emcee with KDEMove (30 %) + DEMove (70 %) (sn/pantheon.py:114-117) and emcee's
// default StretchMove elsewhere; with the ensemble resident in HBM the update of one split (a half, or a third for DE) is
//     all-gather of positions -> propose -> log P (cf_eval_device) -> accept,
// and the proposal / accept arithmetic is a handful of flops per walker: done with tensor-library calls it costs
// 8-20x the likelihood itself in launches and host round trips.  These kernels do it in two launches (three for KDE).
//
// Random numbers are counter-based (splitmix64 finaliser keyed by walker id, step, split and stream), bit-identical
// to cosmology-model-fit_amd/ensemble.py's uniform01 / normal01: a chain does not depend on how walkers are sharded.
// The splits of a step (two halves; three thirds for the DE move, as emcee's DEMove sets nsplits = 3): see ens_split_of.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <string>

#include "../../include/cosmofit.h"

#define CF_ENS_MAX_NDIM 16

extern int cf_set_error(int code, const std::string& msg);  // cosmofit_api.hip

__device__ __forceinline__ uint64_t ens_mix(uint64_t x) {
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// uniform in [0, 1): ensemble.py uniform01 with key = key0 + stream
__device__ __forceinline__ double ens_uniform(uint64_t key0, int stream, int64_t id) {
  uint64_t x = ens_mix((uint64_t)id * 0x9E3779B97F4A7C15ull + key0 + (uint64_t)stream);
  x = ens_mix(x + 0x9E3779B97F4A7C15ull);
  return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}
// standard normal by Box-Muller from streams `stream`, `stream + 1`: ensemble.py normal01
__device__ __forceinline__ double ens_normal(uint64_t key0, int stream, int64_t id) {
  const double u1 = 1.0 - ens_uniform(key0, stream, id);
  const double u2 = ens_uniform(key0, stream + 1, id);
  return sqrt(-2.0 * log(u1)) * cos((2.0 * 3.14159265358979323846) * u2);
}

// ---- the splits of a step (emcee's RedBlueMove: `nsplits` sets updated in turn, each proposing from the others) ----------
// Walkers are taken in consecutive groups of S = n_splits (S = 2: pairs -- StretchMove, KDEMove; S = 3: triples -- emcee's
// DEMove sets nsplits = 3): walker S c + b belongs to split perm_c[b], where perm_c is the identity for split_key == 0 (fixed
// classes id mod S) and otherwise a counter-based random permutation of (split_key, c), re-drawn every step.  Every split
// then holds exactly one member of every group, so any contiguous shard owns its fair share of each split, and the partition
// does not depend on the walkers' positions (detailed balance as for emcee's shuffled index array).
// S = 2: perm = (flip, 1 - flip) with flip = the top bit of the same two-round hash as ens_uniform (ensemble.py: split_perm).
__device__ __host__ __forceinline__ uint64_t ens_mix_h(uint64_t x) {
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __host__ __forceinline__ uint64_t ens_group_hash(uint64_t split_key, int64_t c) {
  return ens_mix_h(ens_mix_h((uint64_t)c * 0x9E3779B97F4A7C15ull + split_key) + 0x9E3779B97F4A7C15ull);
}
// split of member b of group c, packed permutations of three: entry p holds perm[b] in bits 2b .. 2b + 1
// p: 0 = (0,1,2), 1 = (0,2,1), 2 = (1,0,2), 3 = (1,2,0), 4 = (2,0,1), 5 = (2,1,0)
__device__ __host__ __forceinline__ int ens_split_of(uint64_t split_key, int S, int64_t c, int b) {
  if (split_key == 0) return b;
  const uint64_t x = ens_group_hash(split_key, c);
  if (S == 2) return b ^ (int)(x >> 63);
  const unsigned p = (unsigned)(((x >> 40) * 6ull) >> 24);  // floor(6 u), u = the top 24 bits as a fraction
  const unsigned packed = 0x24u | (0x18u << 6) | (0x21u << 12) | (0x09u << 18) | (0x12u << 24);  // p = 0..4; p = 5 below
  const unsigned e = p == 5 ? 0x06u : (packed >> (6 * p)) & 0x3Fu;
  return (int)((e >> (2 * b)) & 3u);
}
// member b of group c that belongs to split s (exactly one)
__device__ __host__ __forceinline__ int ens_member_in(uint64_t split_key, int S, int64_t c, int s) {
  for (int b = 0; b < S - 1; ++b)
    if (ens_split_of(split_key, S, c, b) == s) return b;
  return S - 1;
}

// row m of the complementary set of split s: the walkers of every other split in ascending index order, i.e. the r-th
// (r = m mod (S - 1)) member of group c = m / (S - 1) that is not in split s
__device__ __forceinline__ const double* comp_row(const double* all_pos, int ndim, int S, int s, uint64_t split_key, int64_t m) {
  if (S == 2) return all_pos + (2 * m + ens_member_in(split_key, 2, m, 1 - s)) * ndim;
  const int64_t c = m / (S - 1);
  int r = (int)(m - c * (S - 1)), b = 0;
  for (; b < S - 1; ++b)
    if (ens_split_of(split_key, S, c, b) != s && r-- == 0) break;
  return all_pos + (S * c + b) * ndim;
}

// host-side counts (no device needed): active walkers of split s in the shard [start, stop), size of the complementary set
static int64_t ens_count_in(uint64_t split_key, int S, int s, int64_t start, int64_t stop) {
  if (stop <= start) return 0;
  const int64_t c0 = start / S, c1 = (stop - 1) / S;
  int64_t n = c1 - c0 + 1;
  const int64_t id0 = S * c0 + ens_member_in(split_key, S, c0, s), id1 = S * c1 + ens_member_in(split_key, S, c1, s);
  if (id0 < start || id0 >= stop) --n;
  if (c1 > c0 && (id1 < start || id1 >= stop)) --n;
  return n;
}

// KDE preparation for a compile-time dimension D <= 8 (one 256-thread workgroup): ONE pass over the complementary set for the first
// and second moments of x - x_ref (x_ref = its first row, inside the cloud, so the subtraction in the covariance cancels nothing it
// needs), every row loaded once with unconditional loads; covariance, Cholesky factor and inverse by thread 0 in registers; the
// whitened set in a second pass.  The run-time-dimension form in the kernel below reads the rows D + D (D + 1) / 2 times with a block
// reduction each, branches around every element load and does its linear algebra through LDS: 57 us at nc = 2048, D = 4
// (profiles/r02_kde_kernels_ab.txt).
template <int D>
__device__ __forceinline__ void kde_prepare_small(const double* __restrict__ all_pos, int64_t nc, int S, int half, uint64_t split_key, double h,
                                                  double* __restrict__ params, double* __restrict__ wc) {
  constexpr int NM = D + D * (D + 1) / 2;
  __shared__ double wsum[4][NM], tot[NM], inv_s[D * D];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double x0[D], mom[NM];
  {
    const double* r0 = comp_row(all_pos, D, S, half, split_key, 0);
#pragma unroll
    for (int k = 0; k < D; ++k) x0[k] = r0[k];
  }
#pragma unroll
  for (int k = 0; k < NM; ++k) mom[k] = 0.0;
  for (int64_t c = tid; c < nc; c += 256) {
    const double* r = comp_row(all_pos, D, S, half, split_key, c);
    double dd[D];
#pragma unroll
    for (int k = 0; k < D; ++k) dd[k] = r[k] - x0[k];
#pragma unroll
    for (int a = 0; a < D; ++a) {
      mom[a] += dd[a];
#pragma unroll
      for (int b = 0; b <= a; ++b) mom[D + a * (a + 1) / 2 + b] += dd[a] * dd[b];
    }
  }
#pragma unroll
  for (int k = 0; k < NM; ++k) {
    double v = mom[k];
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) wsum[wave][k] = v;
  }
  __syncthreads();
  if (tid < NM) tot[tid] = ((wsum[0][tid] + wsum[1][tid]) + wsum[2][tid]) + wsum[3][tid];
  __syncthreads();
  if (tid == 0) {
    double Lm[D][D], It[D][D];
#pragma unroll
    for (int j = 0; j < D; ++j) {  // Cholesky (lower) of the covariance h^2 (S2 - S1 S1^T / n) / (n - 1)
      auto cov = [&](int a, int b) {
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        return (tot[D + hi * (hi + 1) / 2 + lo] - tot[hi] * tot[lo] / (double)nc) / (double)(nc - 1) * (h * h);
      };
      double sj = cov(j, j);
#pragma unroll
      for (int k = 0; k < j; ++k) sj -= Lm[j][k] * Lm[j][k];
      Lm[j][j] = sqrt(sj);
#pragma unroll
      for (int i = 0; i < D; ++i) {
        if (i < j) Lm[i][j] = 0.0;
        if (i > j) {
          double t = cov(i, j);
#pragma unroll
          for (int k = 0; k < j; ++k) t -= Lm[i][k] * Lm[j][k];
          Lm[i][j] = t / Lm[j][j];
        }
      }
    }
    double log_det = 0.0;
#pragma unroll
    for (int col = 0; col < D; ++col) {  // inv(chol) by forward substitution, stored transposed: It[k][m] = inv[m][k]
#pragma unroll
      for (int i = 0; i < D; ++i) {
        double t = i == col ? 1.0 : 0.0;
#pragma unroll
        for (int k = col; k < i; ++k) t -= Lm[i][k] * It[col][k];
        It[col][i] = i < col ? 0.0 : t / Lm[i][i];
      }
      log_det += log(Lm[col][col]);
    }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) {
        params[a * D + b] = Lm[a][b];
        params[D * D + a * D + b] = It[a][b];
        inv_s[a * D + b] = It[a][b];
      }
    params[2 * D * D] = -log((double)nc) - 0.5 * D * log(2.0 * 3.14159265358979323846) - log_det;
  }
  __syncthreads();
  double inv[D * D];
#pragma unroll
  for (int k = 0; k < D * D; ++k) inv[k] = inv_s[k];
  for (int64_t c = tid; c < nc; c += 256) {  // whitened complementary set: wc = comp @ inv_t
    const double* r = comp_row(all_pos, D, S, half, split_key, c);
    double x[D];
#pragma unroll
    for (int k = 0; k < D; ++k) x[k] = r[k];
#pragma unroll
    for (int m = 0; m < D; ++m) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) t += x[k] * inv[k * D + m];
      wc[c * D + m] = t;
    }
  }
}

// ---- KDE (scipy.stats.gaussian_kde, bw_method="silverman", as emcee's KDEMove uses it) -----------------
// params = { chol[d*d] (lower), chol_inv_t[d*d], log_norm }, wc = comp @ chol_inv_t  [nc * d]
extern "C" __global__ void __launch_bounds__(256)
ens_kde_prepare_kernel(const double* __restrict__ all_pos, int64_t nc, int ndim, int S, int half, uint64_t split_key, double* __restrict__ params,
                       double* __restrict__ wc) {
  __shared__ double red[256];
  __shared__ double mean[CF_ENS_MAX_NDIM], cov[CF_ENS_MAX_NDIM * CF_ENS_MAX_NDIM], chol[CF_ENS_MAX_NDIM * CF_ENS_MAX_NDIM],
      inv_t[CF_ENS_MAX_NDIM * CF_ENS_MAX_NDIM];
  const int tid = threadIdx.x, d = ndim;
  auto block_sum = [&](double v) {
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) red[tid] += red[tid + s];
      __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
  };
  const double h = pow((double)nc * (d + 2) / 4.0, -1.0 / (d + 4));  // silverman_factor
  if (d <= 8) {  // compile-time dimension: everything in registers, every load unconditional (see kde_prepare_small)
    switch (d) {
      case 1: kde_prepare_small<1>(all_pos, nc, S, half, split_key, h, params, wc); break;
      case 2: kde_prepare_small<2>(all_pos, nc, S, half, split_key, h, params, wc); break;
      case 3: kde_prepare_small<3>(all_pos, nc, S, half, split_key, h, params, wc); break;
      case 4: kde_prepare_small<4>(all_pos, nc, S, half, split_key, h, params, wc); break;
      case 5: kde_prepare_small<5>(all_pos, nc, S, half, split_key, h, params, wc); break;
      case 6: kde_prepare_small<6>(all_pos, nc, S, half, split_key, h, params, wc); break;
      case 7: kde_prepare_small<7>(all_pos, nc, S, half, split_key, h, params, wc); break;
      default: kde_prepare_small<8>(all_pos, nc, S, half, split_key, h, params, wc); break;
    }
    return;
  } else {
    for (int k = 0; k < d; ++k) {
      double s = 0.0;
      for (int64_t c = tid; c < nc; c += 256) s += comp_row(all_pos, d, S, half, split_key, c)[k];
      const double tot = block_sum(s);
      if (tid == 0) mean[k] = tot / (double)nc;
    }
    __syncthreads();
    for (int a = 0; a < d; ++a)
      for (int b = 0; b <= a; ++b) {
        double s = 0.0;
        for (int64_t c = tid; c < nc; c += 256) {
          const double* r = comp_row(all_pos, d, S, half, split_key, c);
          s += (r[a] - mean[a]) * (r[b] - mean[b]);
        }
        const double tot = block_sum(s);
        if (tid == 0) cov[a * d + b] = cov[b * d + a] = tot / (double)(nc - 1) * (h * h);
      }
  }
  __syncthreads();
  if (tid == 0) {
    for (int i = 0; i < d * d; ++i) chol[i] = 0.0;
    for (int j = 0; j < d; ++j) {  // Cholesky, lower
      double s = cov[j * d + j];
      for (int k = 0; k < j; ++k) s -= chol[j * d + k] * chol[j * d + k];
      chol[j * d + j] = sqrt(s);
      for (int i = j + 1; i < d; ++i) {
        double t = cov[i * d + j];
        for (int k = 0; k < j; ++k) t -= chol[i * d + k] * chol[j * d + k];
        chol[i * d + j] = t / chol[j * d + j];
      }
    }
    // inv(chol) by forward substitution, stored transposed: inv_t[k][m] = inv[m][k]
    double log_det = 0.0;
    for (int col = 0; col < d; ++col) {
      for (int i = 0; i < d; ++i) {
        double t = i == col ? 1.0 : 0.0;
        for (int k = col; k < i; ++k) t -= chol[i * d + k] * inv_t[col * d + k];
        inv_t[col * d + i] = i < col ? 0.0 : t / chol[i * d + i];
      }
      log_det += log(chol[col * d + col]);
    }
    for (int i = 0; i < d * d; ++i) {
      params[i] = chol[i];
      params[d * d + i] = inv_t[i];
    }
    params[2 * d * d] = -log((double)nc) - 0.5 * d * log(2.0 * 3.14159265358979323846) - log_det;
  }
  __syncthreads();
  for (int64_t c = tid; c < nc; c += 256) {
    const double* r = comp_row(all_pos, d, S, half, split_key, c);
    for (int m = 0; m < d; ++m) {
      double s = 0.0;
      for (int k = 0; k < d; ++k) s += r[k] * inv_t[k * d + m];
      wc[c * d + m] = s;
    }
  }
}

// log Hastings factor of the KDE move, log kde(x) - log kde(q): one 256-thread WORKGROUP per active walker, thread t takes
// the complementary walkers t, t + 256, ...; log-sum-exp in two passes over registers (maximum, then the sum), both reduced
// over the workgroup (wave shuffles + four LDS slots).  One WAVE per walker left the chip with 2 waves per SIMD on chains of
// dependent exp evaluations: 44 us per half-step of 2048 walkers against 28 us (profiles/r02_kde_kernels_ab.txt).
#define CF_ENS_KDE_CHUNK 8  // complementary walkers per thread held in registers at a time
extern "C" __global__ void __launch_bounds__(256)
ens_kde_logfactor_kernel(const double* __restrict__ all_pos, int64_t nc, int ndim, const int64_t* __restrict__ ids,
                         int64_t n_active, const double* __restrict__ kde_params, const double* __restrict__ wc,
                         const double* __restrict__ y, double* __restrict__ log_factor) {
  __shared__ double xch[4][4];  // per wave {max a, max q, sum a, sum q}
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t i = blockIdx.x;
  const int d = ndim;
  const double* inv_t = kde_params + d * d;
  const double log_norm = kde_params[2 * d * d];
  const double* xa = all_pos + ids[i] * d;
  const double* q = y + i * d;
  double wa[CF_ENS_MAX_NDIM], wq[CF_ENS_MAX_NDIM];
  for (int m = 0; m < d; ++m) {
    double sa = 0.0, sq = 0.0;
    for (int k = 0; k < d; ++k) {
      sa += xa[k] * inv_t[k * d + m];
      sq += q[k] * inv_t[k * d + m];
    }
    wa[m] = sa;
    wq[m] = sq;
  }
  auto wave_max = [](double v) {
    for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
  };
  auto wave_add = [](double v) {
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  // running (maximum, scaled sum) per point over chunks of 256 * CF_ENS_KDE_CHUNK complementary walkers
  double mxa = -INFINITY, mxq = -INFINITY, sa = 0.0, sq = 0.0;
  for (int64_t c0 = 0; c0 < nc; c0 += 256 * CF_ENS_KDE_CHUNK) {
    double ea[CF_ENS_KDE_CHUNK], eq[CF_ENS_KDE_CHUNK];
    double la = -INFINITY, lq = -INFINITY;
#pragma unroll
    for (int t = 0; t < CF_ENS_KDE_CHUNK; ++t) {
      const int64_t c = c0 + tid + 256 * t;
      ea[t] = eq[t] = -INFINITY;
      if (c < nc) {
        double da = 0.0, dq = 0.0;
        for (int m = 0; m < d; ++m) {
          const double w = wc[c * d + m];
          da += (wa[m] - w) * (wa[m] - w);
          dq += (wq[m] - w) * (wq[m] - w);
        }
        ea[t] = -0.5 * da;
        eq[t] = -0.5 * dq;
      }
      la = fmax(la, ea[t]);
      lq = fmax(lq, eq[t]);
    }
    la = wave_max(la);
    lq = wave_max(lq);
    if (lane == 0) {
      xch[wave][0] = la;
      xch[wave][1] = lq;
    }
    __syncthreads();
    const double na = fmax(mxa, fmax(fmax(xch[0][0], xch[1][0]), fmax(xch[2][0], xch[3][0])));
    const double nq = fmax(mxq, fmax(fmax(xch[0][1], xch[1][1]), fmax(xch[2][1], xch[3][1])));
    double pa = 0.0, pq = 0.0;
#pragma unroll
    for (int t = 0; t < CF_ENS_KDE_CHUNK; ++t) {
      pa += exp(ea[t] - na);  // exp(-inf) = 0 for the slots past nc
      pq += exp(eq[t] - nq);
    }
    pa = wave_add(pa);
    pq = wave_add(pq);
    if (lane == 0) {
      xch[wave][2] = pa;
      xch[wave][3] = pq;
    }
    __syncthreads();
    sa = sa * exp(mxa - na) + (((xch[0][2] + xch[1][2]) + xch[2][2]) + xch[3][2]);
    sq = sq * exp(mxq - nq) + (((xch[0][3] + xch[1][3]) + xch[2][3]) + xch[3][3]);
    mxa = na;
    mxq = nq;
    __syncthreads();  // xch is rewritten by the next chunk
  }
  if (tid == 0) log_factor[i] = ((mxa + log(sa)) + log_norm) - ((mxq + log(sq)) + log_norm);
}

// kind 0 stretch (emcee StretchMove, a), 1 DE (emcee DEMove, gamma0 = 2.38 / sqrt(2 ndim), sigma; an ordered pair j != k of the
// complementary set, which for emcee's DEMove is the other TWO thirds of the ensemble: n_splits = 3), 2 KDE (independence
// proposal from the Gaussian KDE of the complementary set).  y[i] = proposal of active walker i, log_factor[i] = log of
// the Hastings factor.
extern "C" __global__ void __launch_bounds__(256)
ens_propose_kernel(int kind, const double* __restrict__ all_pos, int64_t nc, int ndim, int S, int half, uint64_t split_key,
                   const int64_t* __restrict__ ids,
                   int64_t n_active, uint64_t key0, double a, double de_sigma, const double* __restrict__ kde_params,
                   const double* __restrict__ kde_wc, double* __restrict__ y, double* __restrict__ log_factor) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n_active) return;
  const int d = ndim;
  const int64_t id = ids[i];
  const double* xa = all_pos + id * d;
  int64_t j = (int64_t)(ens_uniform(key0, 0, id) * (double)nc);
  j = j > nc - 1 ? nc - 1 : j;
  const double* cj = comp_row(all_pos, d, S, half, split_key, j);
  if (kind == 0) {
    const double t = (a - 1.0) * ens_uniform(key0, 1, id) + 1.0;
    const double z = t * t / a;
    for (int k = 0; k < d; ++k) y[i * d + k] = cj[k] + z * (xa[k] - cj[k]);
    log_factor[i] = (d - 1) * log(z);
  } else if (kind == 1) {
    int64_t k2 = (int64_t)(ens_uniform(key0, 1, id) * (double)(nc - 1));
    k2 = k2 > nc - 2 ? nc - 2 : k2;
    k2 += k2 >= j ? 1 : 0;
    const double* ck = comp_row(all_pos, d, S, half, split_key, k2);
    const double gamma = (2.38 / sqrt(2.0 * d)) * (1.0 + de_sigma * ens_normal(key0, 3, id));
    for (int k = 0; k < d; ++k) y[i * d + k] = xa[k] + gamma * (cj[k] - ck[k]);
    log_factor[i] = 0.0;
  } else {
    const double* chol = kde_params;
    double noise[CF_ENS_MAX_NDIM], q[CF_ENS_MAX_NDIM];
    for (int k = 0; k < d; ++k) noise[k] = ens_normal(key0, 4 + 2 * k, id);
    for (int r = 0; r < d; ++r) {
      double s = 0.0;
      for (int k = 0; k < d; ++k) s += noise[k] * chol[r * d + k];  // noise @ chol.T
      q[r] = cj[r] + s;
      y[i * d + r] = q[r];
    }
    // log_factor: ens_kde_logfactor_kernel (one wave per walker)
  }
}

// accept with probability min(1, exp(log_factor + lp_new - lp_old)); NaN never accepts
extern "C" __global__ void __launch_bounds__(256)
ens_accept_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ local_idx, int64_t n_active, int ndim, uint64_t key0,
                  const double* __restrict__ y, const double* __restrict__ lp_new, const double* __restrict__ log_factor,
                  double* __restrict__ x_local, double* __restrict__ logp_local, unsigned long long* __restrict__ n_accepted) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool acc = false;
  if (i < n_active) {
    const int64_t li = local_idx[i];
    const double log_q = log_factor[i] + lp_new[i] - logp_local[li];
    acc = log(ens_uniform(key0, 2, ids[i])) < log_q;
    if (acc) {
      for (int k = 0; k < ndim; ++k) x_local[li * ndim + k] = y[i * ndim + k];
      logp_local[li] = lp_new[i];
    }
  }
  const unsigned long long m = __ballot(acc);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_accepted, (unsigned long long)__popcll(m));
}

// The active walkers of split s in the shard [shard_start, shard_stop): one thread per group c0 + i that touches the shard; its
// member in split s is written if the shard owns it, at position i less the first group's miss (only the first and the last
// group of a shard can be cut).  ens_count_in is the host's count of the entries written.
extern "C" __global__ void __launch_bounds__(256)
ens_active_set_kernel(uint64_t split_key, int S, int s, int64_t shard_start, int64_t shard_stop, int64_t* __restrict__ ids,
                      int64_t* __restrict__ local_idx) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t c0 = shard_start / S, c1 = (shard_stop - 1) / S;
  if (c0 + i > c1) return;
  const int64_t c = c0 + i, id = S * c + ens_member_in(split_key, S, c, s);
  if (id < shard_start || id >= shard_stop) return;
  const int64_t first = S * c0 + ens_member_in(split_key, S, c0, s);
  const int64_t pos = i - (first < shard_start ? 1 : 0);
  ids[pos] = id;
  local_idx[pos] = id - shard_start;
}

// ------------------------------------------------------------------------------------------------
static int ens_check(int64_t w_total, int32_t ndim, int32_t n_splits, int32_t split, const char* fn) {
  if (w_total < 4 || (w_total & 1)) return cf_set_error(CF_ERR_INVALID, std::string(fn) + ": the ensemble needs an even number (>= 4) of walkers");
  if (ndim < 1 || ndim > CF_ENS_MAX_NDIM) return cf_set_error(CF_ERR_INVALID, std::string(fn) + ": ndim must be in 1..16");
  if (n_splits != 2 && n_splits != 3) return cf_set_error(CF_ERR_INVALID, std::string(fn) + ": n_splits must be 2 or 3");
  if (split < 0 || split >= n_splits) return cf_set_error(CF_ERR_INVALID, std::string(fn) + ": split must be in 0..n_splits - 1");
  if (n_splits == 3 && w_total < 6) return cf_set_error(CF_ERR_INVALID, std::string(fn) + ": three splits need at least 6 walkers");
  return 0;
}

extern "C" int64_t cf_ens_active_count(uint64_t split_key, int32_t n_splits, int32_t split, int64_t shard_start, int64_t shard_stop) {
  if ((n_splits != 2 && n_splits != 3) || split < 0 || split >= n_splits || shard_start < 0 || shard_stop < shard_start) return -1;
  return ens_count_in(split_key, n_splits, split, shard_start, shard_stop);
}

extern "C" int64_t cf_ens_comp_count(uint64_t split_key, int32_t n_splits, int32_t split, int64_t w_total) {
  if ((n_splits != 2 && n_splits != 3) || split < 0 || split >= n_splits || w_total < 0) return -1;
  return w_total - ens_count_in(split_key, n_splits, split, 0, w_total);
}

extern "C" int cf_ens_kde_prepare(const double* d_all_pos, int64_t w_total, int32_t ndim, int32_t n_splits, int32_t split,
                                  uint64_t split_key, double* d_params, double* d_wc, void* hip_stream) {
  int rc = ens_check(w_total, ndim, n_splits, split, "cf_ens_kde_prepare");
  if (rc) return rc;
  if (!d_all_pos || !d_params || !d_wc) return cf_set_error(CF_ERR_INVALID, "cf_ens_kde_prepare: null argument");
  const int64_t nc = cf_ens_comp_count(split_key, n_splits, split, w_total);
  hipLaunchKernelGGL(ens_kde_prepare_kernel, dim3(1), dim3(256), 0, (hipStream_t)hip_stream, d_all_pos, nc, (int)ndim, (int)n_splits,
                     (int)split, split_key, d_params, d_wc);
  return hipGetLastError() == hipSuccess ? CF_OK : cf_set_error(CF_ERR_HIP, "cf_ens_kde_prepare: launch failed");
}

extern "C" int cf_ens_propose(int32_t kind, const double* d_all_pos, int64_t w_total, int32_t ndim, int32_t n_splits, int32_t split,
                              uint64_t split_key, const int64_t* d_ids, int64_t n_active, uint64_t key0, double a, double de_sigma,
                              const double* d_kde_params, const double* d_kde_wc, double* d_y, double* d_log_factor,
                              void* hip_stream) {
  int rc = ens_check(w_total, ndim, n_splits, split, "cf_ens_propose");
  if (rc) return rc;
  if (kind < 0 || kind > 2) return cf_set_error(CF_ERR_INVALID, "cf_ens_propose: kind must be 0 (stretch), 1 (DE) or 2 (KDE)");
  if (!d_all_pos || !d_ids || !d_y || !d_log_factor || (kind == 2 && (!d_kde_params || !d_kde_wc)))
    return cf_set_error(CF_ERR_INVALID, "cf_ens_propose: null argument");
  if (n_active <= 0) return CF_OK;
  const int64_t nc = cf_ens_comp_count(split_key, n_splits, split, w_total);
  hipLaunchKernelGGL(ens_propose_kernel, dim3((unsigned)((n_active + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, (int)kind,
                     d_all_pos, nc, (int)ndim, (int)n_splits, (int)split, split_key, d_ids, n_active, key0, a, de_sigma, d_kde_params,
                     d_kde_wc, d_y, d_log_factor);
  if (kind == 2)
    hipLaunchKernelGGL(ens_kde_logfactor_kernel, dim3((unsigned)n_active), dim3(256), 0, (hipStream_t)hip_stream,
                       d_all_pos, nc, (int)ndim, d_ids, n_active, d_kde_params, d_kde_wc, (const double*)d_y, d_log_factor);
  return hipGetLastError() == hipSuccess ? CF_OK : cf_set_error(CF_ERR_HIP, "cf_ens_propose: launch failed");
}

extern "C" int cf_ens_accept(const int64_t* d_ids, const int64_t* d_local_idx, int64_t n_active, int32_t ndim, uint64_t key0,
                             const double* d_y, const double* d_lp_new, const double* d_log_factor, double* d_x_local,
                             double* d_logp_local, uint64_t* d_n_accepted, void* hip_stream) {
  if (ndim < 1 || ndim > CF_ENS_MAX_NDIM) return cf_set_error(CF_ERR_INVALID, "cf_ens_accept: ndim must be in 1..16");
  if (!d_ids || !d_local_idx || !d_y || !d_lp_new || !d_log_factor || !d_x_local || !d_logp_local || !d_n_accepted)
    return cf_set_error(CF_ERR_INVALID, "cf_ens_accept: null argument");
  if (n_active <= 0) return CF_OK;
  hipLaunchKernelGGL(ens_accept_kernel, dim3((unsigned)((n_active + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, d_ids,
                     d_local_idx, n_active, (int)ndim, key0, d_y, d_lp_new, d_log_factor, d_x_local, d_logp_local,
                     (unsigned long long*)d_n_accepted);
  return hipGetLastError() == hipSuccess ? CF_OK : cf_set_error(CF_ERR_HIP, "cf_ens_accept: launch failed");
}

extern "C" int cf_ens_active_set(uint64_t split_key, int32_t n_splits, int32_t split, int64_t shard_start, int64_t shard_stop,
                                 int64_t* d_ids, int64_t* d_local_idx, void* hip_stream) {
  if ((n_splits != 2 && n_splits != 3) || split < 0 || split >= n_splits)
    return cf_set_error(CF_ERR_INVALID, "cf_ens_active_set: n_splits must be 2 or 3 and split in 0..n_splits - 1");
  if (shard_start < 0 || shard_stop < shard_start) return cf_set_error(CF_ERR_INVALID, "cf_ens_active_set: bad shard range");
  if (!d_ids || !d_local_idx) return cf_set_error(CF_ERR_INVALID, "cf_ens_active_set: null argument");
  if (shard_stop == shard_start) return CF_OK;
  const int64_t n_groups = (shard_stop - 1) / n_splits - shard_start / n_splits + 1;
  hipLaunchKernelGGL(ens_active_set_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, split_key,
                     (int)n_splits, (int)split, shard_start, shard_stop, d_ids, d_local_idx);
  return hipGetLastError() == hipSuccess ? CF_OK : cf_set_error(CF_ERR_HIP, "cf_ens_active_set: launch failed");
}
