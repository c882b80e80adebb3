// cf_pack.h — host-side packing of the lower Cholesky factor into the fragment streams that
// trsm_chi2_kernel consumes (layout documented above that kernel and in DESIGN.md).
//
// Reference semantics kept: only L[i][j<=i] is read (cho_factor(..., lower=True)[0] leaves
// garbage above the diagonal: sn/pantheon.py:14, solve_triangular.py:13).
#ifndef CF_PACK_H
#define CF_PACK_H

#include <cmath>
#include <cstdint>
#include <thread>
#include <vector>

#include "cosmofit_device.h"

struct cf_host_pack {
  int64_t n = 0, n_pad = 0;
  int32_t n_blocks = 0, ksplit = 1, tclasses = 4;
  std::vector<cf_d2> frags;        // 64 lane elements (1 KiB) per fragment
  std::vector<int64_t> upd_off;    // [n_blocks*4*ksplit] in fragments
  std::vector<int64_t> diag_off;   // [n_blocks*4*ksplit]
};

static inline int cf_tiles_in_block(int64_t T, int b) {
  int64_t left = T - (int64_t)b * CF_BLOCK_TILES;
  return (int)(left < CF_BLOCK_TILES ? left : CF_BLOCK_TILES);
}
static inline int cf_slots_of_wave(int tiles_b, int tq, int TC) { return tiles_b > tq ? (tiles_b - tq + TC - 1) / TC : 0; }

// Highest valid tile of wave v in the diagonal phase of a block with tiles_b tiles (-1: none).
static inline int cf_diag_ml_max(int NW, int v, int tiles_b) {
  int m = -1;
  for (int j = 0; j < cf_diag_slots(NW); ++j) {
    int t = cf_diag_tile(NW, v, j);
    if (t >= 0 && t < tiles_b && t > m) m = t;
  }
  return m;
}

// Returns 0, or -1 when a pivot is not finite-positive.  (ksplit, tclasses) in {(1,4), (2,4), (4,4), (2,8)}.
static int cf_pack_cholesky(const double* L, int64_t n, int64_t ld, cf_host_pack& out, int ksplit = 2, int tclasses = 4) {
  const int64_t n_pad = (n + 15) / 16 * 16;
  const int64_t T = n_pad / 16;
  const int n_blocks = (int)((T + CF_BLOCK_TILES - 1) / CF_BLOCK_TILES);
  const int KS = ksplit, TC = tclasses, NW = TC * KS, NTD = cf_diag_slots(NW);
  out.n = n;
  out.tclasses = TC;
  out.n_pad = n_pad;
  out.n_blocks = n_blocks;
  out.ksplit = KS;
  out.upd_off.assign((size_t)n_blocks * NW, 0);
  out.diag_off.assign((size_t)n_blocks * NW, 0);
  for (int64_t i = 0; i < n; ++i) {
    double p = L[i * ld + i];
    if (!(p > 0.0) || !std::isfinite(p)) return -1;
  }
  // padded element accessor: identity on the padding rows, zero above the diagonal
  auto Lp = [&](int64_t i, int64_t j) -> double {
    if (j > i) return 0.0;
    if (i >= n) return i == j ? 1.0 : 0.0;
    return L[i * ld + j];
  };
  // size pass
  int64_t total = 0;
  for (int b = 0; b < n_blocks; ++b) {
    const int tiles_b = cf_tiles_in_block(T, b);
    for (int wq = 0; wq < TC; ++wq)
      for (int g = 0; g < KS; ++g) {
        out.upd_off[(b * TC + wq) * KS + g] = total;
        total += (int64_t)(32 * b / KS) * cf_slots_of_wave(tiles_b, wq, TC);
      }
    for (int v = 0; v < NW; ++v) {
      out.diag_off[b * NW + v] = total;
      const int mlm = cf_diag_ml_max(NW, v, tiles_b);
      if (mlm >= 0) total += (int64_t)(2 * mlm + 2) * NTD;
    }
  }
  total += 32;  // slack: the kernel's software pipelines read up to 4 steps past a stream's end
  out.frags.assign((size_t)total * 64, cf_d2{0.0, 0.0});

  std::vector<long double> inv;  // inverse of the current diagonal block, row-major nb x nb
  for (int b = 0; b < n_blocks; ++b) {
    const int tiles_b = cf_tiles_in_block(T, b);
    const int nb = tiles_b * 16;
    const int64_t r0 = (int64_t)b * CF_BLOCK_ROWS;
    // inverse of the lower-triangular diagonal block by column-wise forward substitution, in
    // extended precision so that the stored double is the correctly rounded inverse entry
    inv.assign((size_t)nb * nb, 0.0L);
    for (int j = 0; j < nb; ++j) {
      inv[(size_t)j * nb + j] = 1.0L / (long double)Lp(r0 + j, r0 + j);
      for (int i = j + 1; i < nb; ++i) {
        long double s = 0.0L;
        for (int k = j; k < i; ++k) s += (long double)Lp(r0 + i, r0 + k) * inv[(size_t)k * nb + j];
        inv[(size_t)i * nb + j] = -s / (long double)Lp(r0 + i, r0 + i);
      }
    }
    const int64_t n_s2g = 32 * b / KS;
    for (int wq = 0; wq < TC; ++wq) {
      const int nt = cf_slots_of_wave(tiles_b, wq, TC);
      for (int g = 0; g < KS && nt > 0; ++g) {
        cf_d2* up = out.frags.data() + out.upd_off[(b * TC + wq) * KS + g] * 64;
        for (int64_t q = 0; q < n_s2g; ++q) {
          const int64_t s2 = g * n_s2g + q;
          for (int j = 0; j < nt; ++j)
            for (int lane = 0; lane < 64; ++lane) {
              const int64_t row = r0 + 16 * (wq + TC * j) + (lane & 15);
              const int64_t c0 = 8 * s2 + (lane >> 4);
              cf_d2& f = up[(q * nt + j) * 64 + lane];
              f.x = -Lp(row, c0);
              f.y = -Lp(row, c0 + 4);
            }
        }
      }
    }
    for (int v = 0; v < NW; ++v) {
      const int mlm = cf_diag_ml_max(NW, v, tiles_b);
      if (mlm < 0) continue;
      cf_d2* dg = out.frags.data() + out.diag_off[b * NW + v] * 64;
      for (int sl2 = 0; sl2 <= 2 * mlm + 1; ++sl2)
        for (int j = 0; j < NTD; ++j) {
          const int ml = cf_diag_tile(NW, v, j);
          if (ml < 0 || ml >= tiles_b) continue;  // stays zero
          for (int lane = 0; lane < 64; ++lane) {
            const int rl = 16 * ml + (lane & 15);
            const int c0 = 8 * sl2 + (lane >> 4);
            cf_d2& f = dg[((int64_t)sl2 * NTD + j) * 64 + lane];
            f.x = (c0 <= rl) ? (double)inv[(size_t)rl * nb + c0] : 0.0;
            f.y = (c0 + 4 <= rl) ? (double)inv[(size_t)rl * nb + c0 + 4] : 0.0;
          }
        }
    }
  }
  return 0;
}

// Replays the fragment streams for ONE right-hand side on the host, with the kernel's block
// structure (update through -L fragments per K-split group, diagonal through the inverse
// fragments).  Used only by the CPU test-suite to validate the packing logic without a GPU;
// cf_eval never calls it.
static double cf_pack_replay_host(const cf_host_pack& pk, const double* b_in) {
  const int64_t n_pad = pk.n_pad, T = n_pad / 16;
  const int KS = pk.ksplit, TC = pk.tclasses, NW = TC * KS, NTD = cf_diag_slots(NW);
  std::vector<double> y((size_t)n_pad, 0.0), rhs(CF_BLOCK_ROWS);
  double chi = 0.0;
  for (int b = 0; b < pk.n_blocks; ++b) {
    const int tiles_b = cf_tiles_in_block(T, b);
    const int64_t r0 = (int64_t)b * CF_BLOCK_ROWS;
    const int64_t n_s2g = 32 * b / KS;
    for (int i = 0; i < tiles_b * 16; ++i) rhs[i] = (r0 + i < pk.n) ? b_in[r0 + i] : 0.0;
    for (int wq = 0; wq < TC; ++wq) {
      const int nt = cf_slots_of_wave(tiles_b, wq, TC);
      for (int g = 0; g < KS; ++g) {
        const cf_d2* up = pk.frags.data() + pk.upd_off[(b * TC + wq) * KS + g] * 64;
        for (int64_t q = 0; q < n_s2g; ++q)
          for (int j = 0; j < nt; ++j)
            for (int lane = 0; lane < 64; ++lane) {
              const int rl = 16 * (wq + TC * j) + (lane & 15);
              const int64_t c0 = 8 * (g * n_s2g + q) + (lane >> 4);
              const cf_d2& f = up[(q * nt + j) * 64 + lane];
              rhs[rl] += f.x * y[c0] + f.y * y[c0 + 4];
            }
      }
    }
    for (int v = 0; v < NW; ++v) {
      const int mlm = cf_diag_ml_max(NW, v, tiles_b);
      if (mlm < 0) continue;
      const cf_d2* dg = pk.frags.data() + pk.diag_off[b * NW + v] * 64;
      for (int j = 0; j < NTD; ++j) {
        const int ml = cf_diag_tile(NW, v, j);
        if (ml < 0 || ml >= tiles_b) continue;
        for (int li = 0; li < 16; ++li) {
          double s = 0.0;
          for (int sl2 = 0; sl2 <= 2 * ml + 1; ++sl2)
            for (int kq = 0; kq < 4; ++kq) {
              const cf_d2& f = dg[((int64_t)sl2 * NTD + j) * 64 + kq * 16 + li];
              s += f.x * rhs[8 * sl2 + kq] + f.y * rhs[8 * sl2 + 4 + kq];
            }
          y[r0 + 16 * ml + li] = s;
        }
      }
    }
    for (int i = 0; i < tiles_b * 16; ++i) chi += y[r0 + i] * y[r0 + i];
  }
  return chi;
}

// Probe right-hand sides shared by both packings.  mode 0: pseudo-random in [-1, 1) (splitmix64); 1: all ones (an
// offset-like residual); 2: residual-shaped -- what Delta = m - M - mu(z; theta) looks like for a theta away from
// the truth on z-sorted supernovae: a smooth trend of a few tenths of a magnitude across the sample plus 0.15 mag of
// scatter.  A smooth right-hand side is the hard case for an explicit inverse: neighbouring (strongly correlated)
// supernovae make rows of L^-1 large and alternating, and the products cancel.
#define CF_PROBE_MODES 3
static void cf_probe_rhs(int mode, int64_t n, std::vector<double>& b) {
  b.resize((size_t)n);
  uint64_t s = 0x1234567ull;
  for (int64_t i = 0; i < n; ++i) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const double u = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    const double x = n > 1 ? (double)i / (double)(n - 1) : 0.0;
    b[i] = mode == 0 ? u : mode == 1 ? 1.0 : 0.08 + 0.25 * x - 0.4 * x * x + 0.15 * u;
  }
}

// || L^-1 b ||^2 by plain row-by-row forward substitution (solve_triangular.py:12-14): what a probe compares with.
static double cf_probe_reference(const double* L, int64_t n, int64_t ld, const std::vector<double>& b) {
  std::vector<double> y((size_t)n);
  double ref = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const double* row = L + i * ld;
    double acc = 0.0;
    for (int64_t j = 0; j < i; ++j) acc += row[j] * y[j];
    y[i] = (b[i] - acc) / row[i];
    ref += y[i] * y[i];
  }
  return ref;
}

// Accuracy probe of a packed factor: || L^-1 b ||^2 through the fragment streams (blocked, diagonal blocks through
// their inverses) against row-by-row forward substitution, worst of the CF_PROBE_MODES right-hand sides.  An
// ill-conditioned diagonal block shows up here (cond(L_bb) * eps) before any walker is evaluated.
static double cf_pack_probe(const cf_host_pack& pk, const double* L, int64_t ld) {
  double worst = 0.0;
  std::vector<double> b;
  for (int mode = 0; mode < CF_PROBE_MODES; ++mode) {
    cf_probe_rhs(mode, pk.n, b);
    const double ref = cf_probe_reference(L, pk.n, ld, b);
    const double got = cf_pack_replay_host(pk, b.data());
    const double rel = std::fabs(got - ref) / (std::fabs(ref) > 0 ? std::fabs(ref) : 1.0);
    if (!(rel <= worst)) worst = rel;
  }
  return worst;
}

// ------------------------------------------------------------------------------------------------
// Latency mode (cf_desc.solve_mode = CF_SOLVE_INVERSE_GEMM): Y = X Delta with X = L^-1 computed
// once on the host -- a triangular GEMM with no dependency chain, so rows split over as many
// workgroups as the batch needs ("Hard parts": the alternative to the blocked solve).
// Stream of (row block rb of 4 tiles = 64 rows, wave g of 4): the wave's quarter of the K range
// [0, 64 (rb+1)), as [q < 2 (rb+1)][tile j < 4][lane] -> {X[row][8 s2 + k], X[row][8 s2 + 4 + k]},
// row = 64 rb + 16 j + (l&15), k = l>>4, s2 = g * 2 (rb+1) + q.
// ------------------------------------------------------------------------------------------------
struct cf_host_invpack {
  int64_t n = 0, n_pad = 0;
  int32_t n_rowblocks = 0;
  std::vector<cf_d2> frags;
  std::vector<int64_t> off;  // [n_rowblocks*4] in fragments
};

// X = L^-1 (lower triangular, row-major n_pad x n_pad, identity on the padding), column by column: column j solves
// L x = e_j by forward substitution with every product and sum in `long double` (x87 extended: 64-bit significand),
// four accumulation chains per dot product, and the stored double is the rounding of that extended result -- so an
// entry of X carries one rounding, not the ~sqrt(n) eps of a double recurrence.  Columns are independent: a few host
// threads share them (n^3 / 6 = 0.8 G multiply-adds at n = 1701).
static void cf_invert_lower(const double* L, int64_t n, int64_t ld, int64_t n_pad, std::vector<double>& X) {
  X.assign((size_t)n_pad * n_pad, 0.0);
  for (int64_t j = n; j < n_pad; ++j) X[(size_t)j * n_pad + j] = 1.0;
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt == 0 ? 1 : (nt > 16 ? 16 : nt);
  if (n < 256) nt = 1;
  auto work = [&](unsigned t) {
    std::vector<long double> col((size_t)n);
    // interleaved columns: column j costs (n - j)^2 / 2, so neighbours cost the same
    for (int64_t j = t; j < n; j += nt) {
      col[j] = 1.0L / (long double)L[j * ld + j];
      for (int64_t i = j + 1; i < n; ++i) {
        const double* row = L + i * ld;
        long double a0 = 0.0L, a1 = 0.0L, a2 = 0.0L, a3 = 0.0L;
        int64_t k = j;
        for (; k + 3 < i; k += 4) {
          a0 += (long double)row[k] * col[k];
          a1 += (long double)row[k + 1] * col[k + 1];
          a2 += (long double)row[k + 2] * col[k + 2];
          a3 += (long double)row[k + 3] * col[k + 3];
        }
        for (; k < i; ++k) a0 += (long double)row[k] * col[k];
        col[i] = -((a0 + a1) + (a2 + a3)) / (long double)row[i];
      }
      for (int64_t i = j; i < n; ++i) X[(size_t)i * n_pad + j] = (double)col[i];
    }
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
}

// Column of X (= row of Delta) that lane group kq feeds in K-step h (0 / 1) of K-step pair s2: the two K-steps
// of a pair take ADJACENT columns per lane group, so that the matching B fragment {Delta[w][c], Delta[w][c+1]}
// is one aligned 16-byte load from the row-major residual vector.
static inline int64_t cf_inv_col(int64_t s2, int kq, int h) { return 8 * s2 + 2 * kq + h; }

static void cf_pack_inverse(const double* L, int64_t n, int64_t ld, cf_host_invpack& out) {
  const int64_t n_pad = (n + 15) / 16 * 16, T = n_pad / 16;
  const int RB = (int)((T + 3) / 4);
  out.n = n;
  out.n_pad = n_pad;
  out.n_rowblocks = RB;
  out.off.assign((size_t)RB * 4, 0);
  std::vector<double> X;
  cf_invert_lower(L, n, ld, n_pad, X);
  int64_t total = 0;
  for (int rb = 0; rb < RB; ++rb)
    for (int g = 0; g < 4; ++g) {
      out.off[rb * 4 + g] = total;  // == cf_inv_stream_off(rb, g), which the kernels evaluate (cf_invpack_probe checks it)
      total += (int64_t)2 * (rb + 1) * 4;
    }
  total += 32;  // slack for the kernel's software pipeline
  out.frags.assign((size_t)total * 64, cf_d2{0.0, 0.0});
  for (int rb = 0; rb < RB; ++rb)
    for (int g = 0; g < 4; ++g) {
      cf_d2* st = out.frags.data() + out.off[rb * 4 + g] * 64;
      const int64_t nq = 2 * (rb + 1);
      for (int64_t q = 0; q < nq; ++q)
        for (int j = 0; j < 4; ++j)
          for (int lane = 0; lane < 64; ++lane) {
            const int64_t row = 64 * rb + 16 * j + (lane & 15);
            if (row >= n_pad) continue;  // missing tile of the last row block stays zero
            const int64_t c0 = cf_inv_col(g * nq + q, lane >> 4, 0), c1 = cf_inv_col(g * nq + q, lane >> 4, 1);
            cf_d2& f = st[(q * 4 + j) * 64 + lane];
            f.x = c0 <= row ? X[(size_t)row * n_pad + c0] : 0.0;
            f.y = c1 <= row ? X[(size_t)row * n_pad + c1] : 0.0;
          }
    }
}

static double cf_invpack_replay_host(const cf_host_invpack& pk, const double* b_in) {
  const int64_t n_ld = (int64_t)pk.n_rowblocks * 64;
  std::vector<double> b((size_t)n_ld + 64, 0.0);
  for (int64_t i = 0; i < pk.n; ++i) b[i] = b_in[i];
  double chi = 0.0;
  for (int rb = 0; rb < pk.n_rowblocks; ++rb) {
    double y[64] = {0.0};
    const int64_t nq = 2 * (rb + 1);
    for (int g = 0; g < 4; ++g) {
      const cf_d2* st = pk.frags.data() + pk.off[rb * 4 + g] * 64;
      for (int64_t q = 0; q < nq; ++q)
        for (int j = 0; j < 4; ++j)
          for (int lane = 0; lane < 64; ++lane) {
            const cf_d2& f = st[(q * 4 + j) * 64 + lane];
            y[16 * j + (lane & 15)] += f.x * b[cf_inv_col(g * nq + q, lane >> 4, 0)] + f.y * b[cf_inv_col(g * nq + q, lane >> 4, 1)];
          }
    }
    for (int i = 0; i < 64; ++i) chi += y[i] * y[i];
  }
  return chi;
}

// Probe of the inverse pack: worst relative chi^2 discrepancy against row-by-row substitution over the
// CF_PROBE_MODES right-hand sides (cf_probe_rhs).
static double cf_invpack_probe(const cf_host_invpack& pk, const double* L, int64_t ld) {
  double worst = 0.0;
  // the solve kernels compute a stream's offset in closed form: a pack laid out differently must not reach them
  for (int rb = 0; rb < pk.n_rowblocks; ++rb)
    for (int g = 0; g < 4; ++g)
      if (pk.off[rb * 4 + g] != cf_inv_stream_off(rb, g)) return INFINITY;
  std::vector<double> b;
  for (int mode = 0; mode < CF_PROBE_MODES; ++mode) {
    cf_probe_rhs(mode, pk.n, b);
    const double ref = cf_probe_reference(L, pk.n, ld, b);
    const double got = cf_invpack_replay_host(pk, b.data());
    const double rel = std::fabs(got - ref) / (std::fabs(ref) > 0 ? std::fabs(ref) : 1.0);
    if (!(rel <= worst)) worst = rel;
  }
  return worst;
}

#endif
