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
#define _USE_MATH_DEFINES
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { CO_FDE_LCDM = 0, CO_FDE_WCDM = 1, CO_FDE_THAWING = 2, CO_FDE_CPL = 3 };
enum { CO_P_OFFSET = 0, CO_P_H0, CO_P_OM, CO_P_OBH2, CO_P_OCH2, CO_P_W0, CO_P_WA, CO_P_V, CO_P_RD, CO_P_FCC, CO_P_NSLOTS };

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
  /* --- physical-density E(z): cmb/data_planck_act_compression.py:29-66 --- */
  int32_t has_vstep, cpl_wall;
  double or_h2, omnu_h2, o_gamma_h2, nu_m0, nu_rho0;
  double nu_qs_sq[5], nu_ws[5];
  /* --- BAO block: bao/desi_cmb_des5y.py:69-100,132-135 --- */
  int32_t n_bao, bao_dh_exact, rd_from_fit, pad2;
  const double *bao_z, *bao_val, *bao_inv_cov;
  const int32_t* bao_qty;
  double rd_fit[11];
  /* --- compressed CMB: cmb/data_planck_act_compression.py:86-212 --- */
  int32_t cmb_mode, n_gl;
  const double *gl_x, *gl_w;
  double cmb_prior[3], cmb_inv_cov[9], zstar_fit[4];
  int32_t n_chi2_gauss, pad3;
  const double* chi2_gauss; /* [n*3] */
  /* --- SH0ES calibrators (sn/pantheon_and_sh0es.py:63-69) and cosmic chronometers
         (bao/desi_union3_cc_theta_star.py:129-139) --- */
  const double* fixed_mu; /* [n_sn] NaN -> theory, or NULL */
  int32_t n_cc, pad4;
  const double *cc_z, *cc_h, *cc_inv_cov;
  double cc_logdet;
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

/* cmb/data_planck_act_compression.py:53-66 */
static double omnu_z(const co_desc* d, double z) {
  double zp1 = 1.0 + z;
  double r = d->nu_m0 / zp1, mz_sq = r * r;
  double f0 = sqrt(d->nu_qs_sq[0] + mz_sq), f1 = sqrt(d->nu_qs_sq[1] + mz_sq), f2 = sqrt(d->nu_qs_sq[2] + mz_sq);
  double f3 = sqrt(d->nu_qs_sq[3] + mz_sq), f4 = sqrt(d->nu_qs_sq[4] + mz_sq);
  double ws = f0 * d->nu_ws[0] + f1 * d->nu_ws[1] + f2 * d->nu_ws[2] + f3 * d->nu_ws[3] + f4 * d->nu_ws[4];
  double zp1_2 = zp1 * zp1;
  return zp1_2 * zp1_2 * ws / d->nu_rho0;
}

static double H_of_z(const co_desc* d, double z, const double* th) {
  double H0 = slot_get(&d->slot[CO_P_H0], th);
  double zp1 = 1.0 + z, cubed = zp1 * zp1 * zp1;
  if (d->ez_model == 0) {
    double Om = slot_get(&d->slot[CO_P_OM], th);
    if (d->fde == CO_FDE_LCDM) return H0 * sqrt(Om * cubed + (1.0 - Om)); /* sn/pantheon.py:31 */
    return H0 * sqrt(Om * cubed + (1.0 - Om) * f_de(d, z, th));             /* bao/desi.py:35 */
  }
  /* bao/desi_cmb_des5y.py:34-54 */
  double h = H0 / 100;
  double Onu = d->omnu_h2 / (h * h), Or = d->or_h2 / (h * h);
  double Obc = (slot_get(&d->slot[CO_P_OBH2], th) + slot_get(&d->slot[CO_P_OCH2], th)) / (h * h);
  double Ode = 1.0 - Obc - Or - Onu;
  double rad = Or * (cubed * zp1), mat = Obc * cubed, nu = Onu * omnu_z(d, z);
  double de = d->fde == CO_FDE_LCDM ? Ode : Ode * f_de(d, z, th);
  return H0 * sqrt(rad + mat + de + nu);
}

/* cmb/data_planck_act_compression.py:86-99 */
static double z_star_fit(const double* f, double wb, double wm) {
  wb = pow(wb, f[2]);
  wm = pow(wm, f[3]);
  return pow(wm, -0.7316314841257655) + f[0] * 391.6723594873167 * pow(wb, 0.9368102670600895) * pow(wm, -0.35300106475765136) +
         f[1] * 937.4224935298015 * pow(wm, 0.0192950634264157) * pow(wb, -0.04285000485853785);
}

/* cmb/data_planck_act_compression.py:102-124; f = b, m, a1..a9 */
static double r_drag_fit(const double* f, double wb, double wm) {
  wb = pow(wb, f[0]);
  wm = pow(wm, f[1]);
  double den = (f[2] * pow(wb, f[3])) + (f[4] * pow(wb, f[5]) * pow(wm, f[6])) + (f[7] * pow(wm, f[8]));
  return 1.0 / den - f[9] / pow(wm, f[10]);
}

/* cmb/data_planck_act_compression.py:160-212 */
static void cmb_distances(const co_desc* d, const double* th, double out[3]) {
  double Ob = slot_get(&d->slot[CO_P_OBH2], th), Oc = slot_get(&d->slot[CO_P_OCH2], th);
  double Om_h2 = Oc + Ob + d->omnu_h2;
  double zstar = z_star_fit(d->zstar_fit, Ob, Om_h2);
  double a_lim = 1.0 / (1.0 + zstar), half = a_lim / 2.0, integral = 0.0;
  for (int i = 0; i < d->n_gl; i++) {
    double a = half * d->gl_x[i] + half;
    double z = (1.0 / a) - 1.0;
    double Rb = (3.0 / 4.0) * (Ob / d->o_gamma_h2) * a;
    integral += d->gl_w[i] * (d->c / (a * a * H_of_z(d, z, th) * sqrt(3.0 * (1.0 + Rb))));
  }
  double rs_star = half * integral;
  half = zstar / 2.0;
  integral = 0.0;
  for (int i = 0; i < d->n_gl; i++) integral += d->gl_w[i] * (d->c / H_of_z(d, half * d->gl_x[i] + half, th));
  double DM_star = half * integral;
  if (d->cmb_mode == 3) { out[0] = rs_star / DM_star; out[1] = Ob; out[2] = Om_h2; return; }
  out[0] = 100 * sqrt(Om_h2) * DM_star / d->c;
  out[1] = M_PI * DM_star / rs_star;
  out[2] = Ob;
}

static double chi2_cmb(const co_desc* d, const double* th) {
  double v[3], dl[3];
  cmb_distances(d, th, v);
  for (int i = 0; i < 3; i++) dl[i] = d->cmb_prior[i] - v[i];
  if (d->cmb_mode == 2) return dl[1] * dl[1] * d->cmb_inv_cov[4];
  double acc = 0.0;
  for (int j = 0; j < 3; j++) {
    double t = 0.0;
    for (int i = 0; i < 3; i++) t += dl[i] * d->cmb_inv_cov[3 * i + j];
    acc += t * dl[j];
  }
  return acc;
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

/* bao/desi_cmb_des5y.py:82-100 (tables already built in w) */
static void bao_theory(const co_desc* d, const double* th, co_work* w, double* out) {
  double rd;
  if (d->rd_from_fit) {
    double Ob = slot_get(&d->slot[CO_P_OBH2], th), Oc = slot_get(&d->slot[CO_P_OCH2], th);
    rd = r_drag_fit(d->rd_fit, Ob, Ob + Oc + d->omnu_h2);
  } else rd = slot_get(&d->slot[CO_P_RD], th);
  double* slopes = NULL;
  if (!d->bao_dh_exact) {
    slopes = (double*)malloc(8 * (size_t)d->n_grid);
    co_pchip_slopes(w->zg, w->dh, d->n_grid, slopes); /* interpolator.py:113: all 4000 slopes, like the reference */
  }
  for (int k = 0; k < d->n_bao; k++) {
    double z = d->bao_z[k];
    double DM = cubic_eval(z, w->zg, w->cum, w->dh, d->n_grid, 1);
    double DH = d->bao_dh_exact ? d->c / H_of_z(d, z, th) : cubic_eval(z, w->zg, w->dh, slopes, d->n_grid, 0);
    switch (d->bao_qty[k]) {
      case 2: out[k] = DH / rd; break;
      case 1: out[k] = DM / rd; break;
      case 0: out[k] = pow(z * DH * (DM * DM), 1.0 / 3) / rd; break;
      default: out[k] = DM / DH; break;
    }
  }
  free(slopes);
}

static double chi2_bao(const co_desc* d, const double* th, co_work* w) {
  double th_[64], dl[64];
  bao_theory(d, th, w, th_);
  for (int k = 0; k < d->n_bao; k++) dl[k] = d->bao_val[k] - th_[k];
  double acc = 0.0;
  for (int j = 0; j < d->n_bao; j++) {
    double t = 0.0;
    for (int i = 0; i < d->n_bao; i++) t += dl[i] * d->bao_inv_cov[i * d->n_bao + j];
    acc += t * dl[j];
  }
  return acc;
}

/* sn/pantheon.py:43-61; dm_obs / mu_corr may be NULL.  The distance tables must be in w. */
static void sn_delta(const co_desc* d, const double* th, co_work* w, double* dm_obs, double* mu_corr_out) {
  double off = slot_get(&d->slot[CO_P_OFFSET], th);
  double v = slot_get(&d->slot[CO_P_V], th);
  for (int64_t i = 0; i < d->n_sn; i++) {
    double zc = d->z_cmb[i];
    double DM = cubic_eval(zc, w->zg, w->cum, w->dh, d->n_grid, 1);
    double mu_corr = 0.0;
    if (d->has_vstep) {
      double v_km_s = 100 * v * d->step[i];
      double z_pec = v_km_s / d->c;
      double z_cosmo = -1.0 + (1.0 + zc) / (1.0 + z_pec);
      mu_corr = 5.0 * log10(cubic_eval(z_cosmo, w->zg, w->cum, w->dh, d->n_grid, 1) / DM);
    }
    double mu_th = 25.0 + 5 * log10((1.0 + d->z_hel[i]) * DM);
    if (d->fixed_mu && !isnan(d->fixed_mu[i])) mu_th = d->fixed_mu[i];
    w->delta[i] = d->obs[i] - off - mu_corr - mu_th;
    if (dm_obs) dm_obs[i] = DM;
    if (mu_corr_out) mu_corr_out[i] = mu_corr;
  }
}

/* blocks[0..2] = sn, bao, cmb */
static double chi2_one_blocks(const co_desc* d, const double* th, co_work* w, double* blocks) {
  double sn = 0.0, bao = 0.0, cmb = 0.0;
  if (d->n_sn > 0 || d->n_bao > 0) dm_grid(d, th, w);
  if (d->n_sn > 0) {
    sn_delta(d, th, w, NULL, NULL);
    sn = co_solve_triangular(d->chol, d->n_sn, d->ld, w->delta, w->y);
  }
  if (d->n_bao > 0) bao = chi2_bao(d, th, w);
  if (d->cmb_mode) cmb = chi2_cmb(d, th);
  if (blocks) { blocks[0] = sn; blocks[1] = bao; blocks[2] = cmb; }
  double total = cmb + bao + sn; /* bao/desi_cmb_des5y.py:141 */
  if (d->n_cc > 0) { /* bao/desi_union3_cc_theta_star.py:129-130 */
    double dl[64], acc = 0.0, f = slot_get(&d->slot[CO_P_FCC], th);
    for (int k = 0; k < d->n_cc; k++) dl[k] = d->cc_h[k] - H_of_z(d, d->cc_z[k], th);
    for (int j = 0; j < d->n_cc; j++) {
      double t = 0.0;
      for (int i = 0; i < d->n_cc; i++) t += dl[i] * d->cc_inv_cov[i * d->n_cc + j];
      acc += t * dl[j];
    }
    total += acc * (f * f);
  }
  for (int g = 0; g < d->n_chi2_gauss; g++) {
    double diff = th[(int)d->chi2_gauss[3 * g]] - d->chi2_gauss[3 * g + 1], sg = d->chi2_gauss[3 * g + 2];
    total += diff * diff / (sg * sg);
  }
  return total;
}
static double chi2_one(const co_desc* d, const double* th, co_work* w) { return chi2_one_blocks(d, th, w, NULL); }

static double logl_one(const co_desc* d, const double* th, co_work* w) {
  if (d->cpl_wall && slot_get(&d->slot[CO_P_W0], th) + slot_get(&d->slot[CO_P_WA], th) >= 0.0) return -1e8;
  double ll = -0.5 * chi2_one(d, th, w);
  if (d->n_cc > 0) /* bao/desi_union3_cc_theta_star.py:135-139 */
    ll -= 0.5 * (d->n_cc * log(2 * M_PI) + d->cc_logdet - 2 * d->n_cc * log(slot_get(&d->slot[CO_P_FCC], th)));
  return ll;
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
        out[k] = isinf(lp) ? -INFINITY : lp + logl_one(d, th, &w);
      } else if (out_kind == 1) {
        out[k] = logl_one(d, th, &w);
      } else {
        out[k] = chi2_one(d, th, &w);
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
  dm_grid(d, th, &w);
  sn_delta(d, th, &w, dm_obs, mu_corr);
  if (delta) memcpy(delta, w.delta, 8 * (size_t)d->n_sn);
  if (cum_dm) memcpy(cum_dm, w.cum, 8 * (size_t)d->n_grid);
  if (dh_grid) memcpy(dh_grid, w.dh, 8 * (size_t)d->n_grid);
  work_free(&w);
}

/* chi^2 blocks (sn, bao, cmb), BAO theory vector and CMB distance vector of one walker (pointers may be NULL). */
void co_blocks(const co_desc* d, const double* th, double* blocks, double* bao_th, double* cmb_vec) {
  co_work w = work_alloc(d);
  chi2_one_blocks(d, th, &w, blocks);
  if (bao_th && d->n_bao > 0) bao_theory(d, th, &w, bao_th);
  if (cmb_vec && d->cmb_mode) cmb_distances(d, th, cmb_vec);
  work_free(&w);
}

int co_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
