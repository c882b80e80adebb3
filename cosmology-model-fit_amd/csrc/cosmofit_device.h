// cosmofit_device.h — structures shared between the host C-ABI layer and the gfx950 kernels.
// Passed BY VALUE as kernel arguments (kernarg segment), so they hold only scalars, small fixed
// arrays and device pointers.
#ifndef COSMOFIT_DEVICE_H
#define COSMOFIT_DEVICE_H

#include <stdint.h>

#define CF_MAX_NDIM 16
#define CF_MAX_GAUSS 8
#define CF_MAX_BAO 64
#define CF_MAX_GL 256
#define CF_MAX_CC 64
#define CF_MAX_FS8 64
#define CF_N_SLOTS 15

// One block row of the blocked solve = 16 MFMA tiles of 16 rows.
#define CF_BLOCK_TILES 16
#define CF_BLOCK_ROWS 256

// Mirrors of the public enums (include/cosmofit.h), kept numeric so device code needs no host header.
#define CF_FDE_LCDM_D 0
#define CF_FDE_WCDM_D 1
#define CF_FDE_THAWING_D 2
#define CF_FDE_CPL_D 3

#define CF_EZ_LATE_FLAT_D 0
#define CF_EZ_PHYSICAL_D 1

#define CF_P_OFFSET_D 0
#define CF_P_H0_D 1
#define CF_P_OM_D 2
#define CF_P_OBH2_D 3
#define CF_P_OCH2_D 4
#define CF_P_W0_D 5
#define CF_P_WA_D 6
#define CF_P_V_D 7
#define CF_P_RD_D 8
#define CF_P_FCC_D 9
#define CF_P_LIN_D 10
#define CF_P_V2_D 11
#define CF_P_V3_D 12
#define CF_P_S8_D 13
#define CF_P_FS8ERR_D 14

#define CF_OUT_CHI2_D 0
#define CF_OUT_LOGL_D 1
#define CF_OUT_LOGP_D 2

struct cf_dev_slot {
  int32_t idx;
  int32_t pad;
  double scale;
  double fixed;
};

// one supernova of the production loop: {z_cmb, step weight, z_hel, observed magnitude}
struct cf_d4 {
  double x, y, z, w;
};

struct cf_dev_desc {
  int32_t ndim, n_grid, ez_model, fde;
  double z_max, step, c;
  double inv_step, inv_last;  // 1/(z[1]-z[0]) and 1/(z[G-1]-z[G-2]) of the linspace grid
  int32_t chunk_shift, pad0;  // residual kernel: each of its 512 threads owns 2^chunk_shift grid nodes
  cf_dev_slot slot[CF_N_SLOTS];
  // SN block (device pointers)
  int32_t n_sn, n_pad;  // n_pad: n_sn rounded up to the 16-row MFMA tile
  int32_t n_ld, pad_ld;  // n_ld: leading dimension of the residual rows Delta[w][n_ld] (multiple of 64, zero filled past n_sn)
  const double* z_cmb;
  const double* z_hel;
  const double* obs;
  const double* sn_step;
  const cf_d4* sn_rec;  // [n_ld + 512] {has_vstep ? 1 + z_cmb : z_cmb, step, 1 + z_hel, obs} per SN, one 32-byte record (the
                        // production SN loop); padding {1, 1, 1, 0}
  const void* log10_tab;  // cf_d2[64] {1 / c_j, log10 c_j}: reduction table of the production loop's log10
  int32_t has_vstep;  // 0: the likelihood has no peculiar-velocity step (z_cosmo = z_cmb)
  int32_t step_pm1;   // 1: every sn_step entry is +1 or -1
  const double* sn_fixed_mu;  // [n_sn] or null; non-NaN entries replace mu_theory (SH0ES calibrators)
  const double* sn_lin;       // [n_sn] or null: offset_i = offset + theta_LIN * sn_lin[i] (bulk-flow magnitude term)
  const double* sn_dir;       // [n_sn * 3] or null: unit vectors of a direction-dependent peculiar velocity
  int32_t om_mode, lin_in_rec;  // om_mode 1: slot OM holds omega_m = Omega_m h^2; lin_in_rec: sn_rec[i].y carries sn_lin[i]
  // cosmic chronometers
  int32_t n_cc, pad3;
  const double* cc_z;
  const double* cc_h;
  const double* cc_inv_cov;
  double cc_logdet;
  // growth-rate block (fs8/fs8.py:64-120): f sigma_8 data, explicit inverse covariance, fiducial H(z) D_M(z) of the
  // Alcock-Paczynski correction; the growth ODE is integrated in ln a by RK4 from a_init to 1 in fs8_steps steps
  int32_t n_fs8, fs8_steps;
  const double* fs8_z;
  const double* fs8_val;
  const double* fs8_inv_cov;
  const double* fs8_fid;
  const int32_t* fs8_step_of;  // [n_fs8] RK4 step that contains ln a_k, data sorted by step (ascending a)
  const int32_t* fs8_order;    // [n_fs8] datum index of the k-th entry in that order
  double fs8_a_init;
  const double* fs8_pts;       // [n_fs8][32] per datum in step order: see fs8_points (cosmofit_api.hip)
  const double* fs8_tab;       // [2 fs8_steps + 1][4] {a, 1 + z, nu(z), nu 3 (1 + w_nu) / (1 + z)} at x = ln a_init (1 - m / (2 steps))
  int32_t n_aux, fs8_n_agrid;   // fs8_n_agrid: points of the scripts' logarithmic a-grid (delta' by PCHIP on it), 0: direct read-out; table-node copies for small_blocks / growth kernels: n_bao BAO data then n_fs8 growth data
  // radiation + massive neutrinos (CF_EZ_PHYSICAL)   cmb/data_planck_act_compression.py:29-66
  double or_h2, omnu_h2, o_gamma_h2, nu_m0, nu_rho0;
  double nu_qs_sq[5], nu_ws[5];
  const double* nu_grid;  // [n_grid] massive-neutrino density ratio at the grid nodes (theta-independent), or null
  const double* ln_grid;  // [n_grid] ln(1 + z) at the grid nodes (wCDM / CPL dark energy: zp1^a = exp(a ln zp1)), or null
  // the same two tables in the order the 512 threads of walker_kernel fetch them (thread t owns nodes 8 t .. 8 t + 7):
  // [k * 512 + t] = table[min(8 t + k, n_grid - 1)], so that a wave's load of its k-th nodes is 512 contiguous bytes instead of
  // 64 cache lines; only for the register path of the table build (n_grid <= 4096), else null
  const double* nu_sw;
  const double* ln_sw;
  const double* exp2_tab;   // [64] 2^(j/64): reduction table of the table build's exp (wCDM / CPL), or null
  // BAO block
  int32_t n_bao, bao_dh_exact, rd_from_fit, rd_wm_late;  // rd_wm_late: the r_drag fit takes wm = Omega_m h^2 (late-time flat model)
  const double* bao_z;
  const double* bao_val;
  const double* bao_inv_cov;
  const int32_t* bao_qty;
  const int32_t* bao_base;  // [n_bao] first of the CF_BAO_NODES table nodes copied out for datum k (small_blocks_kernel)
  double rd_fit[11];
  // compressed-CMB block
  int32_t cmb_mode, n_gl;
  const double* gl_x;
  const double* gl_w;
  double cmb_prior[3], cmb_inv_cov[9], zstar_fit[11];
  // priors
  int32_t has_bounds, n_gauss, n_chi2_gauss, cpl_wall;
  double log_norm;  // -sum(log(hi-lo))
  int32_t sn_vel_mult, cc_f_inverse;  // cf_desc.sn_vel_mode / cc_f_mode == 1
  double logl_const;  // constant added to log L (Gaussian normalisations kept in a script's log-likelihood)
  double lo[CF_MAX_NDIM], hi[CF_MAX_NDIM];
  int32_t gauss_idx[CF_MAX_GAUSS];
  double gauss_mean[CF_MAX_GAUSS], gauss_sigma[CF_MAX_GAUSS];
  int32_t chi2_gauss_idx[CF_MAX_GAUSS];
  double chi2_gauss_mean[CF_MAX_GAUSS], chi2_gauss_sigma[CF_MAX_GAUSS];
};

// What the PRODUCTION per-walker kernel (walker_fast_kernel) needs of the descriptor, and nothing else: ~0.5 KB of kernel
// arguments instead of cf_dev_desc's 1.9 KB.  With the whole descriptor by value the kernel's prologue was a chain of
// scalar loads of fields of code paths it never takes, each waited for and spilled (74 SGPR spills for the physical-density
// CPL model).  Same field names as cf_dev_desc: the device helpers are templates over the descriptor type.
struct cf_walker_args {
  int32_t ndim, n_grid, ez_model, fde;
  int32_t chunk_shift, om_mode;
  int32_t n_sn, n_ld;
  int32_t has_vstep, step_pm1, lin_in_rec, n_aux;
  double z_max, step, c, inv_step, inv_last;
  double or_h2, omnu_h2;
  cf_dev_slot slot[CF_N_SLOTS];
  const cf_d4* sn_rec;
  const void* log10_tab;
  const double* nu_sw;
  const double* ln_sw;
  const double* exp2_tab;
  const int32_t* bao_base;
};

// What the prior / output epilogue (finalize_value: sn/pantheon.py:80-97) reads, and nothing else: 0.7 KB.  The solve kernels take
// a POINTER to the handle's device copy (cf_handle::epi): with the 1.9 KB cf_dev_desc by value they carried 44-46 SGPR spills,
// because the epilogue of the one last-arriving workgroup kept the whole descriptor live in scalar registers in all of them.
// Every workgroup fetches one 8-byte word per lane of it beside its hand-off; the last arriver parks them in LDS.
struct cf_epilogue {
  int32_t ndim, has_bounds, n_gauss, n_chi2_gauss;
  int32_t cpl_wall, n_fs8, n_cc, cc_f_inverse;
  double log_norm, logl_const, cc_logdet;
  cf_dev_slot w0, wa, fs8err, fcc;  // the four parameter slots the epilogue consults (CPL wall, growth and chronometer error scales)
  double lo[CF_MAX_NDIM], hi[CF_MAX_NDIM];
  int32_t gauss_idx[CF_MAX_GAUSS];
  double gauss_mean[CF_MAX_GAUSS], gauss_sigma[CF_MAX_GAUSS];
  int32_t chi2_gauss_idx[CF_MAX_GAUSS];
  double chi2_gauss_mean[CF_MAX_GAUSS], chi2_gauss_sigma[CF_MAX_GAUSS];
};
#define CF_EPI_WORDS ((int)((sizeof(cf_epilogue) + 7) / 8))

#ifdef __HIPCC__
typedef double cf_d2 __attribute__((ext_vector_type(2)));
#else
typedef struct { double x, y; } cf_d2;
#endif

// Packed Cholesky factor: fragment streams per (block row, wave). Offsets in 1 KiB fragments.
// The solve workgroup has NW = tclasses*ksplit waves.  In the update phase wave
// (tq = wave % tclasses, g = wave / tclasses) owns the tiles tq, tq+tclasses, .. of the block row
// and the K-step pairs [g*n, (g+1)*n), n = 32 b / ksplit.
struct cf_dev_pack {
  const cf_d2* frags;
  const int64_t* upd_off;   // [n_blocks*NW]  index (b*tclasses + tq)*ksplit + g
  const int64_t* diag_off;  // [n_blocks*NW]  index b*NW + wave
  int32_t n_blocks;
  int32_t ksplit;
  int32_t tclasses;
  int32_t pad;
};

// Inverse-GEMM solve: fragment streams of X = L^-1 per (64-row block, K quarter); see cf_pack.h.
struct cf_dev_invpack {
  const cf_d2* frags;
  const int64_t* off;  // [n_rowblocks*4]
  int32_t n_rowblocks;
  int32_t pad;
};

#ifdef __HIPCC__
#define CF_HD __host__ __device__
#else
#define CF_HD
#endif

// Inverse-GEMM solve: offset (in 1 KiB fragments) of the stream of (row block rb, K quarter g) -- row block r holds 4 quarters of
// 2 (r + 1) K-step pairs x 4 tiles, so the streams before it add up to 16 rb (rb + 1).  cf_pack_inverse lays the streams out by this
// formula and the solve kernels evaluate it instead of loading cf_dev_invpack::off.
CF_HD static inline int64_t cf_inv_stream_off(int rb, int g) { return (int64_t)8 * (rb + 1) * (2 * rb + g); }

// Tiles of a block row that wave v (of NW = 4, 8 or 16) solves in the diagonal phase, slot j.
// Pairs (v, 2NW-1-v) balance the triangular work.  Returns -1 when the slot does not exist.
CF_HD static inline int cf_diag_slots(int NW) { return CF_BLOCK_TILES / NW; }
CF_HD static inline int cf_diag_tile(int NW, int v, int j) {
  if (NW >= CF_BLOCK_TILES) return j == 0 ? v : -1;
  const int chunk = j >> 1, span = 2 * NW;
  return (j & 1) ? chunk * span + span - 1 - v : chunk * span + v;
}

#endif
