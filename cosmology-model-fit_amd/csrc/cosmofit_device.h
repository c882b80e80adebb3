// cosmofit_device.h — structures shared between the host C-ABI layer and the gfx950 kernels.
// Passed BY VALUE as kernel arguments (kernarg segment), so they hold only scalars, small fixed
// arrays and device pointers.
#ifndef COSMOFIT_DEVICE_H
#define COSMOFIT_DEVICE_H

#include <stdint.h>

#define CF_MAX_NDIM 16
#define CF_MAX_GAUSS 8
#define CF_N_SLOTS 9

// One block row of the blocked solve = 16 MFMA tiles of 16 rows.
#define CF_BLOCK_TILES 16
#define CF_BLOCK_ROWS 256

// Mirrors of the public enums (include/cosmofit.h), kept numeric so device code needs no host header.
#define CF_FDE_LCDM_D 0
#define CF_FDE_WCDM_D 1
#define CF_FDE_THAWING_D 2
#define CF_FDE_CPL_D 3

#define CF_P_OFFSET_D 0
#define CF_P_H0_D 1
#define CF_P_OM_D 2
#define CF_P_OBH2_D 3
#define CF_P_OCH2_D 4
#define CF_P_W0_D 5
#define CF_P_WA_D 6
#define CF_P_V_D 7
#define CF_P_RD_D 8

#define CF_OUT_CHI2_D 0
#define CF_OUT_LOGL_D 1
#define CF_OUT_LOGP_D 2

struct cf_dev_slot {
  int32_t idx;
  int32_t pad;
  double scale;
  double fixed;
};

struct cf_dev_desc {
  int32_t ndim, n_grid, ez_model, fde;
  double z_max, step, c;
  cf_dev_slot slot[CF_N_SLOTS];
  // SN block (device pointers)
  int32_t n_sn, n_pad;
  const double* z_cmb;
  const double* z_hel;
  const double* obs;
  const double* sn_step;
  // priors
  int32_t has_bounds, n_gauss, n_chi2_gauss, cpl_wall;
  double log_norm;  // -sum(log(hi-lo))
  double lo[CF_MAX_NDIM], hi[CF_MAX_NDIM];
  int32_t gauss_idx[CF_MAX_GAUSS];
  double gauss_mean[CF_MAX_GAUSS], gauss_sigma[CF_MAX_GAUSS];
  int32_t chi2_gauss_idx[CF_MAX_GAUSS];
  double chi2_gauss_mean[CF_MAX_GAUSS], chi2_gauss_sigma[CF_MAX_GAUSS];
};

#ifdef __HIPCC__
typedef double cf_d2 __attribute__((ext_vector_type(2)));
#else
typedef struct { double x, y; } cf_d2;
#endif

// Packed Cholesky factor: fragment streams per (block row, wave). Offsets in 1 KiB fragments.
struct cf_dev_pack {
  const cf_d2* frags;
  const int64_t* upd_off;   // [n_blocks*4]
  const int64_t* diag_off;  // [n_blocks*4]
  int32_t n_blocks;
  int32_t pad;
};

#endif
