/*
 * cosmofit.h — C-ABI of the MI355X (gfx950) batched cosmological log-likelihood engine.
 *
 * This is the drop-in boundary for ONE hot path of franciscotln/cosmology-model-fit:
 * the per-walker chi^2 / log-likelihood that emcee / nautilus call through
 * `log_probability(theta)` or `log_probs_vectorized(Theta[W,ndim])`.
 * Plain C types only (pointers + sizes); loadable with ctypes.
 *
 * Reference interfaces each entry point replaces (paths relative to the reference repo):
 *   cf_create            module-level data set-up        sn/pantheon.py:10-19, bao/desi_cmb_des5y.py:14-22
 *   cf_eval              chi_squared / log_likelihood /
 *                        log_probability / *_vectorized  sn/pantheon.py:57-97, bao/desi.py:63-106,
 *                                                        bao/desi_cmb.py:137-143, bao/desi_cmb_des5y.py:138-145
 *   cf_eval_device       the same on device-resident Theta (on-device ensemble; replaces the
 *                        emcee pool.map dispatch          sn/pantheon.py:119-125)
 *   cf_eval_parts        DM_z / mu_theory / mu_corr accessors used by the post-fit plots
 *                                                        sn/pantheon.py:34-54,152-155
 *   cf_eval_table        the (cum_dm, dh_grid) pair inside DM_z / DM_grid  sn/pantheon.py:35-39, bao/desi_cmb_des5y.py:60-66
 *   cf_eval_bao_at       bao_theory(z, qty, params) at ARBITRARY redshifts (post-fit plots)  bao/desi.py:38-56, bao/plot_predictions.py:24-45
 *   cf_eval_hz           H_z(z, params) at arbitrary redshifts (post-fit plots)             ohd/cc.py:95-96, ohd/plot_predictions.py:7-32
 *   cf_eval_fs8_at       fs8_theory(a, params) at arbitrary scale factors (post-fit plots) fs8/fs8.py:84-98,221-226
 *   cf_interp_hermite    interp_hermite                   interpolator.py:117-119
 *   cf_interp_pchip      interp_pchip                     interpolator.py:111-114
 *   cf_solve_triangular  solve_triangular (returns y.y)   solve_triangular.py:5-14
 *
 * Error convention: every function returns 0 on success, a negative cf_status otherwise;
 * the message is available through cf_last_error() (thread-local). Nothing throws across
 * the ABI. Numerical convention: theta outside the strict prior box -> -inf for CF_OUT_LOGP
 * (sn/pantheon.py:82-83); a non-finite chi^2 for an in-box theta -> -inf (never NaN: emcee
 * raises ValueError on NaN) and is counted in cf_info.nonfinite_count.
 *
 * Threading: one caller per handle at a time (the reference callers are single-threaded per
 * process). Different handles may be used from different threads.
 *
 * Current device: a handle lives on the device(s) named at cf_create; every function taking a handle
 * switches the calling thread to that device for its own HIP calls and restores the thread's previous
 * current device before it returns. The cf_ens_* functions take device pointers and a stream and launch
 * on the calling thread's current device, which must be the one those belong to.
 */
#ifndef COSMOFIT_H
#define COSMOFIT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CF_ABI_VERSION 8

typedef struct cf_handle cf_handle;

enum cf_status {
  CF_OK = 0,
  CF_ERR_INVALID = -1,   /* bad descriptor / argument */
  CF_ERR_NO_DEVICE = -2, /* no HIP device visible */
  CF_ERR_HIP = -3,       /* a HIP runtime call failed */
  CF_ERR_NOT_POSDEF = -4,/* a zero / negative / non-finite pivot in L */
  CF_ERR_UNSUPPORTED = -5,
  CF_ERR_ILL_CONDITIONED = -6 /* the blocked solve would lose more than 1e-11 relative on this factor */
};

/* Expansion-rate family (a2-a4). */
enum cf_ez_model {
  /* H = H0 * sqrt(Om*(1+z)^3 + (1-Om)*f_DE(z))            sn/pantheon.py:28-31, bao/desi.py:32-35 */
  CF_EZ_LATE_FLAT = 0,
  /* H = H0 * sqrt(Or*zp1^4 + Obc*zp1^3 + Ode*f_DE + Onu*Omnu_z(z)), densities = omega/h^2,
     Ode = 1-Obc-Or-Onu                                   bao/desi_cmb_des5y.py:34-54 */
  CF_EZ_PHYSICAL = 1
};

/* Dark-energy density ratio f_DE(z) (SURVEY section 2 "f_DE(z)"). */
enum cf_fde {
  CF_FDE_LCDM = 0,    /* 1 */
  CF_FDE_WCDM = 1,    /* zp1^(3(1+w0))                        sn/pantheon_and_sh0es.py:26-28 */
  CF_FDE_THAWING = 2, /* (2 zp1^3 / (1+w0+(1-w0) zp1^3))^2    bao/desi.py:26-28 */
  CF_FDE_CPL = 3      /* zp1^(3(1+w0+wa)) exp(-3 wa z/zp1)    bao/desi_fs_lya_cmb.py:19-22 */
};

/* Physical-parameter slots. Each slot is either read from theta[idx]*scale or fixed. */
enum cf_param_slot {
  CF_P_OFFSET = 0, /* M (absolute magnitude) or delta-M offset, subtracted from the SN residual */
  CF_P_H0 = 1,     /* km/s/Mpc; use scale=100 when the sampler parameter is h */
  CF_P_OM = 2,     /* Omega_m (CF_EZ_LATE_FLAT) */
  CF_P_OBH2 = 3,   /* omega_b (CF_EZ_PHYSICAL) */
  CF_P_OCH2 = 4,   /* omega_c (CF_EZ_PHYSICAL) */
  CF_P_W0 = 5,
  CF_P_WA = 6,
  CF_P_V = 7,      /* peculiar-velocity step amplitude in units of 100 km/s (mu_corr) */
  CF_P_RD = 8,     /* sound horizon in Mpc when fixed or free (BAO block) */
  CF_P_FCC = 9,    /* error-rescale factor of the cosmic-chronometer block (default: fixed 1) */
  CF_P_LIN = 10,   /* amplitude of the per-SN linear magnitude term sn_lin_coef[i] (bulk-flow correction of M,
                      bao/desi_cmb_pantheon_H0trgb.py:102-106) */
  CF_P_V2 = 11,    /* second and third velocity components (x 100 km/s) of a direction-dependent peculiar velocity: */
  CF_P_V3 = 12,    /*   v_los,i = n_i . (V, V2, V3)                     sn/pantheon_dipole_xyz.py:50-60 */
  CF_P_S8 = 13,    /* sigma_8(z = 0) of the growth-rate block          fs8/fs8.py:84-98 */
  CF_P_FS8ERR = 14,/* error-rescale factor f_err of the growth-rate block (default: fixed 1)  fs8/fs8.py:116-125 */
  CF_P_NSLOTS = 15
};

typedef struct cf_param {
  int32_t idx;   /* index into theta, or -1 when the slot is fixed / unused */
  int32_t _pad;
  double scale;  /* value = theta[idx]*scale  (scale is applied as `scale*theta`, cf. `100 * h`) */
  double fixed;  /* value when idx < 0 */
} cf_param;

/* BAO quantity codes (bao/desi_cmb_des5y.py:69-79). */
enum cf_bao_qty { CF_BAO_DV = 0, CF_BAO_DM = 1, CF_BAO_DH = 2, CF_BAO_FAP = 3 };

enum cf_bao_dh_mode {
  CF_BAO_DH_PCHIP = 0, /* DH interpolated from the grid with PCHIP  bao/desi_cmb_des5y.py:88 */
  CF_BAO_DH_EXACT = 1  /* DH = c/H(z) evaluated at the datum       bao/desi_cmb.py:54-56 */
};

enum cf_rd_mode {
  CF_RD_PARAM = 0,  /* from slot CF_P_RD (fixed constant or free parameter) */
  CF_RD_FIT = 1     /* r_drag(omega_b, omega_m) fitting formula, coefficients in cf_desc.rd_fit */
};

enum cf_cmb_mode {
  CF_CMB_NONE = 0,
  CF_CMB_R_LA_WB = 1,   /* (R, l_A, omega_b) 3-vector x 3x3 inverse cov   bao/desi_cmb_des5y.py:126-129 */
  CF_CMB_LA_ONLY = 2,   /* only the l_A component                        bao/desi_des5y_bbn_theta_star.py:110-111 */
  CF_CMB_THETA_WB_WM = 3 /* (100 theta*, omega_b, omega_m)               cmb/data_early_lcdm_compression.py:198-207 */
};

/* How chi^2 = || L^-1 Delta ||^2 is evaluated (both on FP64 matrix cores; solve_triangular.py:5-14). */
enum cf_solve_mode {
  CF_SOLVE_BLOCKED_TRSM = 0, /* blocked forward substitution, one workgroup per 16-walker panel: 256-row block rows,
                                diagonal blocks through their host-computed inverses */
  CF_SOLVE_INVERSE_GEMM = 1, /* triangular GEMM against X = L^-1 (inverted once on the host in extended precision,
                                probed against row-by-row substitution at cf_create): no dependency between row
                                blocks, so the matrix cores stay busy for large batches and a batch of 1..2048
                                walkers (log_evidence.py:20-46, small emcee ensembles) spreads over the whole chip */
  CF_SOLVE_AUTO = 2          /* CF_SOLVE_INVERSE_GEMM when its probe passes (<= 1e-11 relative), otherwise
                                CF_SOLVE_BLOCKED_TRSM; cf_info.solve_mode tells which one is in effect */
};

/* Output selector for cf_eval*. */
enum cf_out {
  CF_OUT_CHI2 = 0, /* chi^2 */
  CF_OUT_LOGL = 1, /* -0.5 chi^2                                  (nautilus: likelihood only) */
  CF_OUT_LOGP = 2  /* log prior + log L, -inf outside the box     (emcee log_probability) */
};

typedef struct cf_gauss_prior {
  int32_t idx;   /* theta index */
  int32_t _pad;
  double mean;
  double sigma;  /* term: -0.5*(theta[idx]-mean)^2/sigma^2        sn/pantheon.py:85 */
} cf_gauss_prior;

/*
 * Immutable likelihood descriptor. All arrays are host pointers, copied at cf_create and
 * free to be released as soon as it returns.
 */
typedef struct cf_desc {
  int32_t abi_version;  /* CF_ABI_VERSION */
  int32_t struct_size;  /* sizeof(cf_desc) as seen by the caller */
  int32_t device;       /* HIP device ordinal */
  int32_t ndim;         /* length of one theta row */

  int32_t ez_model;     /* cf_ez_model */
  int32_t fde;          /* cf_fde */
  int32_t n_grid;       /* G, points of the uniform z grid (4000 in every reference script) */
  int32_t _pad0;
  double z_max;         /* grid = linspace(0, z_max, G)            sn/pantheon.py:16 */
  double c_km_s;        /* speed of light in km/s                  sn/pantheon.py:12 */

  cf_param param[CF_P_NSLOTS];

  /* ---- SN block (may be absent: n_sn = 0) ---- */
  int64_t n_sn;
  const double* sn_z_cmb;   /* [n_sn] */
  const double* sn_z_hel;   /* [n_sn] */
  const double* sn_obs;     /* [n_sn] m_b or mu */
  const double* sn_step;    /* [n_sn] per-SN sign/weight s_i multiplying 100*v/c in z_cosmo; NULL -> use z_turn */
  double sn_z_turn;         /* s_i = +1 if z_cmb <= z_turn else -1     sn/pantheon.py:46 */
  const double* sn_chol;    /* [n_sn*ld] row-major lower Cholesky factor; ONLY L[i][j<=i] is read
                               (cho_factor leaves garbage above the diagonal, sn/pantheon.py:14) */
  int64_t sn_chol_ld;       /* row stride of sn_chol in elements (>= n_sn) */

  /* ---- BAO block (n_bao = 0 -> absent) ---- */
  int32_t n_bao;
  int32_t bao_dh_mode;      /* cf_bao_dh_mode */
  int32_t rd_mode;          /* cf_rd_mode */
  int32_t _pad1;
  const double* bao_z;      /* [n_bao] */
  const double* bao_val;    /* [n_bao] */
  const int32_t* bao_qty;   /* [n_bao] cf_bao_qty */
  const double* bao_inv_cov;/* [n_bao*n_bao] row-major explicit inverse */
  double rd_fit[11];        /* r_drag coefficients b, m, a1..a9     cmb/data_planck_act_compression.py:102-124 */

  /* ---- compressed-CMB block ---- */
  int32_t cmb_mode;         /* cf_cmb_mode */
  int32_t n_gl;             /* Gauss-Legendre nodes (100) */
  const double* gl_x;       /* [n_gl] nodes on [-1,1]   (np.polynomial.legendre.leggauss) */
  const double* gl_w;       /* [n_gl] weights */
  double cmb_prior[3];
  double cmb_inv_cov[9];
  double zstar_fit[11];     /* z_star: s1, s2, b, m, then e0, c1, e1, e2, c2, e3, e4 of
                               wm^e0 + s1 c1 wb^e1 wm^e2 + s2 c2 wm^e3 wb^e4 (wb, wm raised to b, m first)
                               cmb/data_planck_act_compression.py:86-99 */
  double o_gamma_h2;        /* photon density for R_b                 cmb/...:29 */

  /* ---- radiation + massive neutrino constants (CF_EZ_PHYSICAL) ---- */
  double or_h2;             /* cmb.Or_h2 */
  double omnu_h2;           /* cmb.Omnu_h2 */
  double nu_m0;             /* m0 */
  double nu_rho0;           /* rho0 */
  double nu_qs_sq[5];       /* qs**2 */
  double nu_ws[5];          /* weights */

  /* ---- priors (CF_OUT_LOGP only) ---- */
  const double* bounds;     /* [ndim*2] (lo,hi) strict box; NULL -> no box */
  int32_t n_gauss;
  int32_t cpl_wall;         /* 1: w0+wa >= 0 -> -1e8             bao/desi_fs_lya_cmb.py:118-121 */
  const cf_gauss_prior* gauss; /* [n_gauss] Gaussian terms added to the log-prior */
  /* Gaussian terms added to chi^2 itself (e.g. BBN omega_b, bao/desi_des5y_bbn_theta_star.py:139) */
  int32_t n_chi2_gauss;
  int32_t _pad2;
  const cf_gauss_prior* chi2_gauss;

  /* ---- SN block extension: fixed distance moduli (SH0ES Cepheid hosts) ---- */
  const double* sn_fixed_mu; /* [n_sn] or NULL; entry NaN -> mu_theory, else this value replaces
                                25 + 5 log10((1+z_hel) DM)          sn/pantheon_and_sh0es.py:63-69 */

  /* ---- cosmic-chronometer block (n_cc = 0 -> absent)  bao/desi_union3_cc_theta_star.py:129-139 ----
   * chi2_cc = (H_obs - H(z))^T inv_cov (H_obs - H(z)) * f_cc^2 and
   * log L -= 0.5 * (n_cc ln(2 pi) + logdet - 2 n_cc ln f_cc) */
  int32_t n_cc;
  int32_t _pad3;
  const double* cc_z;       /* [n_cc] */
  const double* cc_h;       /* [n_cc] km/s/Mpc */
  const double* cc_inv_cov; /* [n_cc*n_cc] */
  double cc_logdet;         /* ln det of the CC covariance */

  int32_t solve_mode;       /* cf_solve_mode */
  int32_t _pad4;
  double probe_limit;       /* acceptance limit of the create-time accuracy probe of the packed factor (relative chi^2
                               discrepancy against row-by-row substitution on three probe vectors); 0 = the default 1e-11.
                               A smaller value makes CF_SOLVE_AUTO fall back to the blocked solve earlier. */

  /* ---- several GPUs behind one handle (the sampler stays in ONE host process, sn/pantheon.py:119-125) ----
   * n_devices = 0: the single ordinal `device`.  n_devices > 0: `devices[n_devices]` HIP ordinals; the data and the
   * packed factor are replicated on each at cf_create, cf_eval splits the rows of theta contiguously over them (one
   * host thread + one stream per device, disjoint slices of `out`), results are bit-identical to one device.
   * n_devices = -1: every visible device.  An ordinal may repeat (two replicas on one GPU).  cf_eval_device needs a
   * single-device handle. */
  int32_t n_devices;
  int32_t _pad5;
  const int32_t* devices;

  /* ---- parameterisation variants of the scripts ("Physics variants") ---- */
  int32_t om_mode;          /* 0: Omega_m = slot CF_P_OM; 1: slot CF_P_OM is omega_m = Omega_m h^2 and
                               Omega_m = omega_m / (H0/100)^2          bao/desi_omh2.py:18-20 */
  int32_t rd_wm_mode;       /* matter density handed to the r_drag fit (CF_RD_FIT): 0 = omega_b + omega_c + omega_nu (physical
                               densities, bao/desi_cmb_des5y.py:84-85); 1 = Omega_m (H0/100)^2 of the late-time flat model with
                               omega_b from slot CF_P_OBH2                bao/desi_bbn.py:46-60 */
  const double* sn_lin_coef;/* [n_sn] or NULL: the SN offset becomes offset + theta_LIN * sn_lin_coef[i]; with
                               sn_lin_coef[i] = 100 (5/ln 10) / (c z_cmb,i) this is the linearised bulk-flow magnitude
                               term                                    bao/desi_cmb_pantheon_H0trgb.py:102-106 */
  const double* sn_dir;     /* [n_sn*3] or NULL: unit vectors n_i; then the peculiar velocity of SN i is
                               100 * (n_i . (V, V2, V3)) * sn_step[i] km/s (sn_step = attenuation x survey mask)
                                                                       sn/pantheon_dipole_xyz.py:50-60 */

  /* ---- growth-rate block f sigma_8(z) (n_fs8 = 0 -> absent)  fs8/fs8.py:64-125, bao/desi_cmb_union3_fs8.py:147-207 ----
   * delta'' = -(3/a + E'/E) delta' + (3/2) Omega_m delta / (a^5 E^2) from a_init (delta = a, delta' = 1) to a = 1, with
   * Omega_m = slot OM (CF_EZ_LATE_FLAT) or (omega_b + omega_c)/h^2 (CF_EZ_PHYSICAL);
   * theory_k = (sigma_8 / delta(1)) a_k delta'(a_k) / q_k,  q_k = H(z_k) D_M(z_k) / fs8_fid[k]  (Alcock-Paczynski);
   * chi2_fs8 = f_err^2 (val - theory)^T inv_cov (val - theory)  and  log L += n_fs8 ln f_err.
   * The reference integrates with scipy's adaptive RK45 at rtol = 1e-6; here a fixed-step RK4 in ln a: fs8_steps steps, rounded
   * up to 256, 512, 1024 (0 = default) or 2048 (256 lanes per walker, 1-8 steps per lane; the ODE is linear, so the steps are
   * 2 x 2 matrices combined by a parallel scan): the two agree to the reference's own integration error (~1e-6 relative on theory);
   * against the reference's equation integrated to convergence the default leaves < 1e-9 on the theory (fs8_n_agrid below). */
  int32_t n_fs8;
  int32_t fs8_steps;
  const double* fs8_z;      /* [n_fs8] */
  const double* fs8_val;    /* [n_fs8] */
  const double* fs8_inv_cov;/* [n_fs8*n_fs8] */
  const double* fs8_fid;    /* [n_fs8] H_fid(z_k) D_M,fid(z_k) in the units H(z) D_M(z) has for this descriptor */
  double logl_const;        /* constant added to log L (CF_OUT_LOGL / CF_OUT_LOGP): Gaussian normalisations a script keeps in its
                               log-likelihood, e.g. -0.5 (N ln 2 pi + logdet) of the growth-rate block, fs8/fs8_cmb.py:20,181-183 */
  double fs8_a_init;        /* 10^-2.15 (fs8/fs8.py:79), 10^-2.7 (bao/desi_cmb_union3_fs8.py:168), 1/201 (ohd/cc_fs8.py:86-87) */

  /* ---- further per-script conventions ---- */
  int32_t sn_vel_mode;      /* 0: z_cosmo = -1 + (1 + z_cmb) / (1 + z_pec)                         sn/pantheon.py:43-48
                               1: z_cosmo = max((1 + z_cmb) (1 + z_pec) - 1, 1e-8)                 bao/desi_pantheon_cc.py:84-90 */
  int32_t cc_f_mode;        /* 0: chi2_cc * f_cc^2, log L += n_cc ln f_cc  (f_cc divides the errors, bao/desi_union3_cc_theta_star.py:129-139)
                               1: chi2_cc * f_cc^-2, log L -= n_cc ln f_cc (f_cc multiplies them)   ohd/cc_pantheon.py:64,92 */
  int32_t prior_norm_mode;  /* 0: log prior inside the box = -sum log(hi - lo) (sn/pantheon.py:77); 1: 0.0 (ohd/cc_cmb.py:70-73) */
  int32_t fs8_n_agrid;      /* growth block: 0 = delta'(a_k) read from the integration directly; N >= 4 = as the scripts do, by interp_pchip
                             * of delta' sampled on a_span = np.logspace(log10 fs8_a_init, 0, N) (fs8/fs8.py:79-98: N = 1000;
                             * bao/desi_cmb_union3_fs8.py:169: 2500; ohd/cc_fs8.py:87: 1000; fs8/fs8_cmb.py:129: 5000) */
} cf_desc;

typedef struct cf_info {
  int64_t n_sn;
  int64_t n_sn_pad;        /* rows after padding to the 16-row MFMA tile */
  int64_t packed_chol_bytes;
  int64_t workspace_bytes; /* current device workspace (grows with W) */
  int64_t max_walkers;     /* walkers the current workspace holds */
  int64_t nonfinite_count; /* in-box evaluations whose chi^2 was not finite since create */
  int32_t device;
  int32_t cu_count;
  char gcn_arch[64];
  double pack_probe_rel;   /* |chi2(packed streams) - chi2(row-by-row substitution)| / chi2 on a probe vector,
                              measured on the host at cf_create (refused above 1e-11) */
  int32_t solve_mode;      /* cf_solve_mode in effect (never CF_SOLVE_AUTO) */
  int32_t n_devices;       /* replicas behind this handle (1 unless cf_desc.n_devices asked for more) */
  int32_t devices[16];     /* their HIP ordinals (the first 16) */
} cf_info;

int cf_device_count(void);
const char* cf_last_error(void);
int cf_abi_version(void);

int cf_create(const cf_desc* desc, cf_handle** out);
void cf_destroy(cf_handle* h);
int cf_get_info(cf_handle* h, cf_info* info);

/* Host buffers. theta: [W*ndim] C-order float64; out: [W] float64. Synchronous.  A handle created over several
 * devices (cf_desc.n_devices) splits the rows as cf_split_rows says, one host thread and one stream per device; this is
 * what lets a sampler that lives in one host process (emcee / nautilus with a vectorised callback, sn/pantheon.py:119-125,
 * bao/desi.py:100-129) use every GPU of the node. */
int cf_eval(cf_handle* h, const double* theta, int64_t W, double* out, int32_t out_kind);

/* Rows [*begin, *end) of a W-walker batch that replica k of n evaluates: contiguous, whole 32-walker panels, sizes
 * differing by at most one panel.  Pure host arithmetic (no device needed). */
void cf_split_rows(int64_t W, int32_t n, int32_t k, int64_t* begin, int64_t* end);

/* Device buffers on the handle's device; asynchronous on `hip_stream` (a hipStream_t; NULL = HIP's
 * default stream, which is what torch.cuda.current_stream().cuda_stream reports as 0), ordered like any
 * other work on that stream. Grows the workspace if needed (then it synchronises once).
 * A handle owns ONE workspace (residual rows, chi^2 shares, arrival counters): evaluations of one handle are
 * ordered with respect to each other by the library -- on one stream by that stream; when an evaluation arrives on a
 * DIFFERENT stream than the handle's previous one (cf_eval and the accessors run on a stream the handle owns), the
 * library first waits ON THE HOST until the previous evaluation has drained (hipDeviceSynchronize if that one ran on
 * a caller's stream: the caller's stream handle is never passed back to HIP, so it may be destroyed at any time after
 * the call that used it).  Hence: calls that stay on one stream are purely asynchronous and capturable into a
 * hipGraph; the FIRST call after a stream switch blocks the host and is illegal while `hip_stream` is capturing.
 * For concurrent evaluations create one handle per stream. */
int cf_eval_device(cf_handle* h, const double* d_theta, int64_t W, double* d_out,
                   int32_t out_kind, void* hip_stream);

/* Intermediates for plots/tests. Any output pointer may be NULL. Host buffers.
 *   dm_obs [W*n_sn]  DM(z_cmb)           sn/pantheon.py:58
 *   mu_corr[W*n_sn]                      sn/pantheon.py:43-49
 *   delta  [W*n_sn]  residual vector     sn/pantheon.py:59-60
 *   chi2_blocks[W*10] (chi2_sn, chi2_bao, chi2_cmb, cmb distance vector[3], chi2_cc, chi2_fs8, z_star, r_drag: the two
 *                     fitting formulae as the blocks evaluated them, 0 where no block needs them;
 *                     cmb/data_planck_act_compression.py:86-124, the blobs of cmb/cmb.py:45-58)
 *                                        bao/desi_cmb_des5y.py:126-141, cmb/data_planck_act_compression.py:200-212
 *   bao_theory[W*n_bao]                  bao/desi_cmb_des5y.py:82-100
 *   fs8_theory[W*n_fs8]                  fs8_theory(a, theta) before the Alcock-Paczynski division, fs8/fs8.py:84-98 */
int cf_eval_parts(cf_handle* h, const double* theta, int64_t W, double* dm_obs, double* mu_corr,
                  double* delta, double* chi2_blocks, double* bao_theory, double* fs8_theory);

/* The distance table of W <= 4096 walkers on the reference's grid z_grid = linspace(0, z_max, G): cum_dm[W*G] (the
 * cumulative trapezoid of c/H) and dh[W*G] (c/H at the nodes), host buffers -- the pair that DM_z(params, z) of the
 * scripts hands to interp_hermite (sn/pantheon.py:34-40, bao/desi_cmb_des5y.py:60-66), for the post-fit plots that
 * evaluate distances at arbitrary redshifts (sn/pantheon.py:152-155). */
int cf_eval_table(cf_handle* h, const double* theta, int64_t W, double* cum_dm, double* dh);

/* bao_theory(z, qty, params) of the scripts for ONE theta at n arbitrary (redshift, quantity code) pairs -- what the post-fit
 * block hands to plot_bao_predictions as a smooth curve (bao/plot_predictions.py:24-45, bao/desi.py:38-56, :204-211): the
 * handle's own E(z) model, D_H convention (PCHIP or c / H) and sound horizon (slot, fixed or fitted), evaluated by the same
 * kernels as the BAO block of the likelihood.  qty: 0 D_V / r_d, 1 D_M / r_d, 2 D_H / r_d, 3 F_AP.  Host buffers. */
int cf_eval_bao_at(cf_handle* h, const double* theta, const double* z, const int32_t* qty, int64_t n, double* out);

/* H_z(z, params) of the scripts in km/s/Mpc for ONE theta at n arbitrary redshifts: the curve and the residuals of
 * plot_cc_predictions (ohd/plot_predictions.py:7-32, ohd/cc.py:95-101, bao/desi_cc.py:193-199).  Host buffers. */
int cf_eval_hz(cf_handle* h, const double* theta, const double* z, int64_t n, double* out);

/* fs8_theory(a, params) of the growth-rate scripts -- f sigma_8 before the Alcock-Paczynski division -- for ONE theta at n
 * arbitrary redshifts z = 1 / a - 1 with a_init <= a <= 1: the smooth curve of fs8/plot_predictions.py:7-32
 * (fs8/fs8.py:221-226).  The handle must have a growth-rate block (its a_init, E(z) model and sigma_8 slot are used). */
int cf_eval_fs8_at(cf_handle* h, const double* theta, const double* z, int64_t n, double* out);

/* Per-kernel timing with HIP events recorded on the stream the kernels are launched on.
 * cf_enable_timing(h, slots): keep events for the last `slots` evaluation calls (0 = off, the
 * default) and reset the call counter.  cf_kernel_ms(h, call, t): t[0] = distance + residual
 * kernel (and the small-blocks kernel of a joint likelihood), t[1] = solve + chi^2 kernel of evaluation number `call` (0-based since
 * cf_enable_timing); waits for that call.  cf_last_kernel_ms = the most recent call. */
int cf_enable_timing(cf_handle* h, int slots);
/* record the events on every `stride`-th evaluation only (default 1): sampled timing of a long loop */
int cf_set_timing_stride(cf_handle* h, int stride);
int64_t cf_timed_calls(cf_handle* h);
int cf_kernel_ms(cf_handle* h, int64_t call, float t[2]);
/* the same with the per-walker stage split: t[0] = walker_kernel, t[1] = small-block (+ growth) kernels, t[2] = solve kernel */
int cf_kernel_ms3(cf_handle* h, int64_t call, float t[3]);
int cf_last_kernel_ms(cf_handle* h, float t[2]);

/* ---- stand-alone operators with the reference's signatures (host buffers) ---- */
/* out[k] = Hermite(xq[k]; x, y, y_prime), linear extrapolation outside   interpolator.py:117-119 */
int cf_interp_hermite(const double* xq, int64_t nq, const double* x, const double* y,
                      const double* y_prime, int64_t n, double* out);
/* out[k] = PCHIP(xq[k]; x, y), clamped outside                             interpolator.py:111-114 */
int cf_interp_pchip(const double* xq, int64_t nq, const double* x, const double* y, int64_t n,
                    double* out);
/* out[w] = || L^-1 b_w ||^2 for nrhs right-hand sides b[nrhs*n] (row w = b_w)  solve_triangular.py:5-14 */
int cf_solve_triangular(const double* L, int64_t n, int64_t ld, const double* b, int64_t nrhs,
                        double* out);

/* Host-only self-test of the factor packing (validates the fragment streams the solve kernel
 * consumes by replaying them for one right-hand side).  For the CPU test-suite; no evaluation
 * entry point ever calls it. */
int cf_selftest_pack_host(const double* L, int64_t n, int64_t ld, const double* b, double* chi2_out,
                          int64_t* packed_bytes);

/* The same for the inverse-GEMM packing (explicit inverse); probe_out (may be NULL) receives the value
 * cf_create compares with 1e-11. */
int cf_selftest_invpack_host(const double* L, int64_t n, int64_t ld, const double* b, double* chi2_out,
                             double* probe_out);

/* Device self-tests of the two in-kernel log10 routines: out[k] = log10(x[k]).  cf_selftest_log10: the <= 1 ulp
 * routine of the accessor / calibrator paths (mu_corr = 5 log10 of a ratio near 1); cf_selftest_log10_tab: the
 * table-driven routine of the production SN loop (absolute error ~2e-16 max(1, |log10 x|)). */
int cf_selftest_log10(const double* x, int64_t n, double* out);
int cf_selftest_log10_tab(const double* x, int64_t n, double* out);
/* The table-driven exp of the wCDM / CPL table build (|x| < ~700, no special cases): out[k] = exp(x[k]), <= 1.5 ulp. */
int cf_selftest_exp_tab(const double* x, int64_t n, double* out);
/* sqrt / quotient of the compressed-CMB Gauss-Legendre nodes (positive, finite, normal operands: the library routines' instruction
 * sequences without their exponent scaling and class selects): out[4 k .. 4 k + 3] = {sqrt_pos(a[k]), the library's sqrt(a[k]),
 * div_pos(a[k], b[k]), the library's a[k] / b[k]}, all on the device -- the pairs must be the same bits. */
int cf_selftest_pos_ops(const double* a, const double* b, int64_t n, double* out);

/* ---- Device-resident ensemble moves (the sampler side of sn/pantheon.py:108-125: emcee with KDEMove 30 % +
 * DEMove 70 %, StretchMove by default elsewhere).  All pointers are device pointers on the current device, all
 * calls are asynchronous on `hip_stream`.
 * Splits (emcee's RedBlueMove: `nsplits` sets updated in turn, each proposing from the union of the others): n_splits = 2 for
 * the stretch and KDE moves, 3 for the DE move (emcee's DEMove sets nsplits = 3).  Walkers are taken in consecutive groups of
 * n_splits: walker n_splits c + b belongs to split perm_c[b], where perm_c is the identity for split_key = 0 (fixed classes
 * index mod n_splits) and otherwise a counter-based random permutation of (split_key, c) -- the splits are re-drawn every step
 * (emcee shuffles its index array) while every split still holds one member of every group, so every process keeps its fair
 * share of each split.  d_all_pos [w_total * ndim] holds every walker's position (after the all-gather of a sharded
 * ensemble), d_ids [n_active] the global indices of the active walkers this process owns, and the complementary set of
 * split `split` is every walker of the other splits in ascending index order.  Random numbers are counter-based: key0 = the
 * 64-bit key of stream 0 for this (seed, step, split) (cosmology-model-fit_amd/ensemble.py: stream_key), so a chain does not
 * depend on how the walkers are sharded over processes.
 *   cf_ens_active_count / cf_ens_comp_count: HOST functions (no device needed): the number of walkers of split `split` in the
 *     shard [shard_start, shard_stop), and the size of its complementary set in an ensemble of w_total walkers; -1 on bad arguments.
 *   cf_ens_active_set: the active walkers of split `split` that the shard [shard_start, shard_stop) owns, ascending:
 *     d_ids [cf_ens_active_count] global indices, d_local_idx = d_ids - shard_start.
 *   cf_ens_kde_prepare: Silverman-bandwidth Gaussian KDE of the complementary set: d_params [2 ndim^2 + 1] =
 *     {chol (lower), inv(chol)^T, log normalisation}, d_wc [cf_ens_comp_count * ndim] = whitened complementary positions.
 *   cf_ens_propose: kind 0 stretch (scale a), 1 differential evolution (gamma0 = 2.38 / sqrt(2 ndim), jitter de_sigma),
 *     2 KDE independence proposal; d_y [n_active * ndim], d_log_factor [n_active] = log Hastings factor.
 *   cf_ens_accept: accept with probability min(1, exp(log_factor + lp_new - lp_old)) (NaN never accepts); updates
 *     d_x_local / d_logp_local at d_local_idx [n_active] and adds the number of accepted moves to *d_n_accepted. */
int64_t cf_ens_active_count(uint64_t split_key, int32_t n_splits, int32_t split, int64_t shard_start, int64_t shard_stop);
int64_t cf_ens_comp_count(uint64_t split_key, int32_t n_splits, int32_t split, int64_t w_total);
int cf_ens_active_set(uint64_t split_key, int32_t n_splits, int32_t split, int64_t shard_start, int64_t shard_stop,
                      int64_t* d_ids, int64_t* d_local_idx, void* hip_stream);
int cf_ens_kde_prepare(const double* d_all_pos, int64_t w_total, int32_t ndim, int32_t n_splits, int32_t split,
                       uint64_t split_key, double* d_params, double* d_wc, void* hip_stream);
int cf_ens_propose(int32_t kind, const double* d_all_pos, int64_t w_total, int32_t ndim, int32_t n_splits, int32_t split,
                   uint64_t split_key, const int64_t* d_ids, int64_t n_active, uint64_t key0, double a, double de_sigma,
                   const double* d_kde_params, const double* d_kde_wc, double* d_y, double* d_log_factor,
                   void* hip_stream);
int cf_ens_accept(const int64_t* d_ids, const int64_t* d_local_idx, int64_t n_active, int32_t ndim, uint64_t key0,
                  const double* d_y, const double* d_lp_new, const double* d_log_factor, double* d_x_local,
                  double* d_logp_local, uint64_t* d_n_accepted, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* COSMOFIT_H */
