/*
 * cosmofit_oracle.c — plain-C CPU restatement of the reference's walker log-likelihood hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Built into oracle/_build/libcosmofit_oracle.so by oracle/Makefile and
 * used (a) by tests/ as the checker for the HIP path at sizes numpy is too slow for and (b) by
 * bench.py's `cpu_baseline` leg (kind "port": the compiled analogue of the reference's @njit +
 * multiprocessing.Pool).  The product library never links, loads or calls it.
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against the golden vectors generated
 * by running the reference itself (tests/golden/generate_golden.py) to <= 1e-12 relative.
 *
 * Same discretisation and the same operation order as the reference: linspace grid, trapezoid
 * with a sequential left-to-right cumsum, searchsorted-left-minus-one interval rule, forward
 * substitution row by row.  Compiled with -O2 -ffp-contract=off (no FMA contraction, no
 * fast-math) so that the arithmetic is the IEEE arithmetic numpy/numba perform.
 *
 * Reference lines restated (paths relative to the reference repo):
 *   grid, dz                 sn/pantheon.py:16-17
 *   H_z / f_DE               sn/pantheon.py:28-31, bao/desi.py:26-35, sn/pantheon_and_sh0es.py:26-28,
 *                            bao/desi_fs_lya_cmb.py:19-22
 *   cumulative trapezoid     sn/pantheon.py:35-39
 *   Hermite / PCHIP          interpolator.py:5-119
 *   mu_corr, mu_theory, res. sn/pantheon.py:43-61
 *   forward substitution     solve_triangular.py:5-14
 *   prior / log_probability  sn/pantheon.py:80-97
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { CO_FDE_LCDM = 0, CO_FDE_WCDM = 1, CO_FDE_THAWING = 2, CO_FDE_CPL = 3 };
enum { CO_P_OFFSET = 0, CO_P_H0, CO_P_OM, CO_P_OBH2, CO_P_OCH2, CO_P_W0, CO_P_WA, CO_P_V, CO_P_RD, CO_P_NSLOTS };

typedef struct co_slot {
  int32_t idx;
  int32_t pad;
  double scale;
  double fixed;
} co_slot;

typedef struct co_desc {
  int32_t ndim, n_grid, ez_model, fde;
  double z_max, c;
  co_slot slot[CO_P_NSLOTS];
  int64_t n_sn;
  const double *z_cmb, *z_hel, *obs, *step; /* step: per-SN sign/weight (never NULL here) */
  const double* chol;                       /* [n_sn*ld], only j<=i read */
  int64_t ld;
  const double* bounds; /* [ndim*2] or NULL */
  int32_t n_gauss, pad;
  const double* gauss; /* [n_gauss*3]: idx, mean, sigma */
} co_desc;

static inline double slot_get(const co_slot* s, const double* th) {
  return s->idx >= 0 ? s->scale * th[s->idx] : s->fixed;
}

/* ---- interpolator.py ---------------------------------------------------------------------- */
/* np.searchsorted(x, v, 'left'): first index i with x[i] >= v. */
static int64_t searchsorted_left(const double* x, int64_t n, double v) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = lo + (hi - lo) / 2;
    if (x[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* interpolator.py:71-108 */
static double cubic_eval(double xi, const double* x, const double* y, const double* d, int64_t n, int exact) {
  if (xi <= x[0]) return exact ? y[0] + d[0] * (xi - x[0]) : y[0];
  if (xi >= x[n - 1]) return exact ? y[n - 1] + d[n - 1] * (xi - x[n - 1]) : y[n - 1];
  int64_t i = searchsorted_left(x, n, xi) - 1;
  double h_i = x[i + 1] - x[i];
  double t = (xi - x[i]) / h_i;
  double t2 = t * t, t3 = t2 * t;
  double h00 = 2 * t3 - 3 * t2 + 1;
  double h10 = t3 - 2 * t2 + t;
  double h01 = -2 * t3 + 3 * t2;
  double h11 = t3 - t2;
  return h00 * y[i] + h10 * h_i * d[i] + h01 * y[i + 1] + h11 * h_i * d[i + 1];
}

static double sgn(double v) { return (v > 0) - (v < 0); }

/* interpolator.py:5-68 */
void co_pchip_slopes(const double* x, const double* y, int64_t n, double* d) {
  if (n < 2) { for (int64_t i = 0; i < n; i++) d[i] = 0.0; return; }
  double* h = (double*)malloc(sizeof(double) * (size_t)(n - 1));
  double* delta = (double*)malloc(sizeof(double) * (size_t)(n - 1));
  for (int64_t i = 0; i < n - 1; i++) { h[i] = x[i + 1] - x[i]; delta[i] = (y[i + 1] - y[i]) / h[i]; }
  if (n == 2) { d[0] = d[1] = delta[0]; free(h); free(delta); return; }
  for (int64_t i = 1; i < n - 1; i++) {
    double dl = delta[i - 1], dr = delta[i];
    if (dl != 0.0 && dr != 0.0 && dl * dr > 0.0) {
      double w1 = 2.0 * h[i] + h[i - 1], w2 = h[i] + 2.0 * h[i - 1];
      d[i] = (w1 + w2) / (w1 / dl + w2 / dr);
    } else d[i] = 0.0;
  }
  double d0 = ((2 * h[0] + h[1]) * delta[0] - h[0] * delta[1]) / (h[0] + h[1]);
  if (delta[0] == 0.0 || sgn(d0) != sgn(delta[0])) d[0] = 0.0;
  else if (sgn(delta[0]) != sgn(delta[1]) && fabs(d0) > fabs(3 * delta[0])) d[0] = 3 * delta[0];
  else d[0] = d0;
  double dn = ((2 * h[n - 2] + h[n - 3]) * delta[n - 2] - h[n - 2] * delta[n - 3]) / (h[n - 2] + h[n - 3]);
  if (delta[n - 2] == 0.0 || sgn(dn) != sgn(delta[n - 2])) d[n - 1] = 0.0;
  else if (sgn(delta[n - 2]) != sgn(delta[n - 3]) && fabs(dn) > fabs(3 * delta[n - 2])) d[n - 1] = 3 * delta[n - 2];
  else d[n - 1] = dn;
  free(h); free(delta);
}

void co_interp_hermite(const double* xq, int64_t nq, const double* x, const double* y, const double* yp,
                       int64_t n, double* out) {
  for (int64_t k = 0; k < nq; k++) out[k] = cubic_eval(xq[k], x, y, yp, n, 1);
}

void co_interp_pchip(const double* xq, int64_t nq, const double* x, const double* y, int64_t n, double* out) {
  double* d = (double*)malloc(sizeof(double) * (size_t)n);
  co_pchip_slopes(x, y, n, d);
  for (int64_t k = 0; k < nq; k++) out[k] = cubic_eval(xq[k], x, y, d, n, 0);
  free(d);
}

/* ---- solve_triangular.py:5-14 ------------------------------------------------------------ */
double co_solve_triangular(const double* L, int64_t n, int64_t ld, const double* b, double* ywork) {
  for (int64_t i = 0; i < n; i++) {
    const double* row = L + i * ld;
    double s = 0.0;
    for (int64_t j = 0; j < i; j++) s += row[j] * ywork[j];
    ywork[i] = (b[i] - s) / row[i];
  }
  double acc = 0.0;
  for (int64_t i = 0; i < n; i++) acc += ywork[i] * ywork[i];
  return acc;
}

/* ---- expansion rate ---------------------------------------------------------------------- */
static double f_de(const co_desc* d, double z, const double* th) {
  double zp1 = 1.0 + z;
  switch (d->fde) {
    case CO_FDE_LCDM: return 1.0;
    case CO_FDE_WCDM: return pow(zp1, 3 * (1 + slot_get(&d->slot[CO_P_W0], th)));
    case CO_FDE_THAWING: {
      double w0 = slot_get(&d->slot[CO_P_W0], th), cubed = zp1 * zp1 * zp1;
      double r = 2 * cubed / ((1.0 + w0) + (1.0 - w0) * cubed);
      return r * r;
    }
    default: {
      double w0 = slot_get(&d->slot[CO_P_W0], th), wa = slot_get(&d->slot[CO_P_WA], th);
      return pow(zp1, 3 * (1 + w0 + wa)) * exp(-3 * wa * z / zp1);
    }
  }
}

static double H_of_z(const co_desc* d, double z, const double* th) {
  double H0 = slot_get(&d->slot[CO_P_H0], th);
  double Om = slot_get(&d->slot[CO_P_OM], th);
  double zp1 = 1.0 + z, cubed = zp1 * zp1 * zp1;
  if (d->fde == CO_FDE_LCDM) return H0 * sqrt(Om * cubed + (1.0 - Om)); /* sn/pantheon.py:31 */
  return H0 * sqrt(Om * cubed + (1.0 - Om) * f_de(d, z, th));             /* bao/desi.py:35 */
}

/* Per-thread scratch */
typedef struct co_work {
  double *zg, *dz, *dh, *cum, *delta, *y;
} co_work;

static co_work work_alloc(const co_desc* d) {
  co_work w;
  size_t G = (size_t)d->n_grid, N = (size_t)(d->n_sn > 0 ? d->n_sn : 1);
  w.zg = (double*)malloc(8 * G); w.dz = (double*)malloc(8 * G); w.dh = (double*)malloc(8 * G);
  w.cum = (double*)malloc(8 * G); w.delta = (double*)malloc(8 * N); w.y = (double*)malloc(8 * N);
  /* np.linspace(0, z_max, G): arange*step, last node forced to stop; dz = np.diff */
  double step = d->z_max / (double)(d->n_grid - 1);
  for (int64_t i = 0; i < d->n_grid; i++) w.zg[i] = (double)i * step;
  w.zg[d->n_grid - 1] = d->z_max;
  for (int64_t i = 0; i + 1 < d->n_grid; i++) w.dz[i] = w.zg[i + 1] - w.zg[i];
  return w;
}
static void work_free(co_work* w) { free(w->zg); free(w->dz); free(w->dh); free(w->cum); free(w->delta); free(w->y); }

/* sn/pantheon.py:35-39 */
static void dm_grid(const co_desc* d, const double* th, co_work* w) {
  int64_t G = d->n_grid;
  for (int64_t i = 0; i < G; i++) w->dh[i] = d->c / H_of_z(d, w->zg[i], th);
  w->cum[0] = 0.0;
  double acc = 0.0;
  for (int64_t i = 0; i + 1 < G; i++) {
    double mid = (w->dh[i] + w->dh[i + 1]) / 2;
    acc += mid * w->dz[i];
    w->cum[i + 1] = acc;
  }
}

/* sn/pantheon.py:43-61; dm_obs / mu_corr may be NULL */
static void sn_delta(const co_desc* d, const double* th, co_work* w, double* dm_obs, double* mu_corr_out) {
  dm_grid(d, th, w);
  double off = slot_get(&d->slot[CO_P_OFFSET], th);
  double v = slot_get(&d->slot[CO_P_V], th);
  for (int64_t i = 0; i < d->n_sn; i++) {
    double zc = d->z_cmb[i];
    double DM = cubic_eval(zc, w->zg, w->cum, w->dh, d->n_grid, 1);
    double v_km_s = 100 * v * d->step[i];
    double z_pec = v_km_s / d->c;
    double z_cosmo = -1.0 + (1.0 + zc) / (1.0 + z_pec);
    double mu_corr = 5.0 * log10(cubic_eval(z_cosmo, w->zg, w->cum, w->dh, d->n_grid, 1) / DM);
    double mu_th = 25.0 + 5 * log10((1.0 + d->z_hel[i]) * DM);
    w->delta[i] = d->obs[i] - off - mu_corr - mu_th;
    if (dm_obs) dm_obs[i] = DM;
    if (mu_corr_out) mu_corr_out[i] = mu_corr;
  }
}

static double chi2_one(const co_desc* d, const double* th, co_work* w) {
  sn_delta(d, th, w, NULL, NULL);
  return co_solve_triangular(d->chol, d->n_sn, d->ld, w->delta, w->y);
}

/* sn/pantheon.py:80-85 */
static double log_prior(const co_desc* d, const double* th) {
  double lp = 0.0;
  if (d->bounds) {
    for (int k = 0; k < d->ndim; k++)
      if (!(d->bounds[2 * k] < th[k] && th[k] < d->bounds[2 * k + 1])) return -INFINITY;
    double s = 0.0;
    for (int k = 0; k < d->ndim; k++) s += log(d->bounds[2 * k + 1] - d->bounds[2 * k]);
    lp = -s;
  }
  for (int g = 0; g < d->n_gauss; g++) {
    int idx = (int)d->gauss[3 * g];
    double diff = th[idx] - d->gauss[3 * g + 1], sg = d->gauss[3 * g + 2];
    lp = lp - 0.5 * (diff * diff) / (sg * sg);
  }
  return lp;
}

/* out_kind: 0 chi2, 1 logL, 2 logP.  Returns the number of threads used. */
int co_eval_batch(const co_desc* d, const double* theta, int64_t W, double* out, int out_kind, int nthreads) {
  int used = 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
  {
#pragma omp single
    used = omp_get_num_threads();
#else
  {
#endif
    co_work w = work_alloc(d);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int64_t k = 0; k < W; k++) {
      const double* th = theta + k * d->ndim;
      if (out_kind == 2) {
        double lp = log_prior(d, th); /* likelihood not evaluated outside the box: sn/pantheon.py:90-92 */
        out[k] = isinf(lp) ? -INFINITY : lp - 0.5 * chi2_one(d, th, &w);
      } else {
        double c2 = chi2_one(d, th, &w);
        out[k] = out_kind == 1 ? -0.5 * c2 : c2;
      }
    }
    work_free(&w);
  }
  return used;
}

/* Intermediates of one walker (any pointer may be NULL). */
void co_sn_parts(const co_desc* d, const double* th, double* dm_obs, double* mu_corr, double* delta,
                 double* cum_dm, double* dh_grid) {
  co_work w = work_alloc(d);
  sn_delta(d, th, &w, dm_obs, mu_corr);
  if (delta) memcpy(delta, w.delta, 8 * (size_t)d->n_sn);
  if (cum_dm) memcpy(cum_dm, w.cum, 8 * (size_t)d->n_grid);
  if (dh_grid) memcpy(dh_grid, w.dh, 8 * (size_t)d->n_grid);
  work_free(&w);
}

int co_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
