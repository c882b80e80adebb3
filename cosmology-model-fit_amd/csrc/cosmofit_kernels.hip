// cosmofit_kernels.hip — gfx950 (MI355X, CDNA4) device code of the batched log-likelihood engine.
//
// The hot path is two launches per evaluation:
//
//   walker_fast_kernel / walker_kernel   (a1-a10)   one 512-thread workgroup per walker.  E(z) on the G-point grid,
//        cumulative trapezoid as chunk-sequential sums + a wave64 DPP scan + LDS carry, the table {cum_dm, dh} staged
//        in LDS (G * 16 B = 64 KB), cubic Hermite at z_cosmo(theta) of the N supernovae, residual -> Delta[w][0..n_ld).
//        walker_fast_kernel is the production form (lean kernel arguments, theta row across the lanes); walker_kernel
//        keeps the accessor / calibrator / direction-dependent / long-grid paths.
//
//   tri_gemm_chi2_kernel / tri_gemm_small_kernel   (a11)   chi^2 = || X Delta ||^2 with X = L^-1 inverted once on the
//        host: a triangular GEMM on FP64 matrix cores (v_mfma_f64_16x16x4_f64) with no dependency between 64-row
//        blocks; one workgroup per (row block, panel of 16-32 walkers), or per (panel, row block, one or two 16-row tiles)
//        for batches of <= 160 walkers (bit-identical; the per-walker kernel then writes the residuals in this kernel's
//        fragment order and runs several workgroups per walker); the workgroup that arrives last for a panel adds the
//        shares in a fixed order and applies the prior / output epilogue.  trsm_chi2_kernel (blocked forward
//        substitution) is the fallback when the explicit inverse fails its create-time probe.
//
// Joint likelihoods add small_blocks_kernel (BAO, compressed CMB, cosmic chronometers: sixteen lanes per walker when the
// batch fills the chip, one or two waves per walker below, the same bits either way) and growth_kernel (f sigma_8) between
// the two.  A walker's result never depends on the batch it is evaluated in.  Written for wave64 / gfx950 only.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "cosmofit_device.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define CF_WAVE 64

// ------------------------------------------------------------------------------------------------
// Parameter slots
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double slot_get(const cf_dev_desc& d, int s, const double* __restrict__ th) {
  const cf_dev_slot& p = d.slot[s];
  return p.idx >= 0 ? p.scale * th[p.idx] : p.fixed;
}

// The walker's theta row held ACROSS THE LANES of a wave (lane k = theta[k], one coalesced load at the head of the kernel);
// a slot is then a v_readlane at the slot's (wave-uniform) index.  Read through the pointer, every slot was its own basic
// block -- load, wait for it, next slot: seven serialised round trips to the theta row at the head of walker_kernel for the
// CPL joint likelihood, 7-9 k of its 23 k cycles per workgroup (in-kernel stamps, profiles/r03_walker_stamps_config3.txt).
struct ThetaRow {
  double v;
};
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
template <class D>
__device__ __forceinline__ double slot_get(const D& d, int s, const ThetaRow& th) {
  const cf_dev_slot& p = d.slot[s];
  const double v = readlane_f64(th.v, p.idx >= 0 ? p.idx : 0);
  return p.idx >= 0 ? p.scale * v : p.fixed;
}

// The same for a GROUP of LANES lanes per walker (small_blocks_kernel: several walkers per wave, so the slot's lane differs from
// group to group -- a shuffle inside the group instead of v_readlane): lane sl of the group holds theta[sl].
template <int LANES>
struct ThetaGroup {
  double v;
};
template <class D, int LANES>
__device__ __forceinline__ double slot_get(const D& d, int s, const ThetaGroup<LANES>& th) {
  const cf_dev_slot& p = d.slot[s];
  const double v = __shfl(th.v, p.idx >= 0 ? p.idx : 0, LANES);
  return p.idx >= 0 ? p.scale * v : p.fixed;
}

// ------------------------------------------------------------------------------------------------
// Expansion rate.  dh(z) = c / H(z).   Reference: sn/pantheon.py:28-31, bao/desi.py:26-35,
// sn/pantheon_and_sh0es.py:26-28, bao/desi_fs_lya_cmb.py:19-22.  (1+z)^3 by multiplies (numba).
// ------------------------------------------------------------------------------------------------
struct WalkerCosmo {
  double H0, Om, w0, wa, c;
  double Or, Obc, Ode, Onu;  // CF_EZ_PHYSICAL densities (omega / h^2)
  int fde, model;
};

// D: cf_dev_desc or the lean cf_walker_args; TH: const double* (theta row in memory) or ThetaRow (across the lanes)
template <class D, class TH>
__device__ __forceinline__ WalkerCosmo make_cosmo(const D& d, const TH& th) {
  WalkerCosmo wc;
  wc.H0 = slot_get(d, CF_P_H0_D, th);
  wc.Om = slot_get(d, CF_P_OM_D, th);
  if (d.om_mode) {  // the sampler's parameter is omega_m = Omega_m h^2    bao/desi_omh2.py:18-20
    const double h = wc.H0 / 100;
    wc.Om = wc.Om / (h * h);
  }
  wc.w0 = slot_get(d, CF_P_W0_D, th);
  wc.wa = slot_get(d, CF_P_WA_D, th);
  wc.c = d.c;
  wc.fde = d.fde;
  wc.model = d.ez_model;
  wc.Or = wc.Obc = wc.Ode = wc.Onu = 0.0;
  if (d.ez_model == CF_EZ_PHYSICAL_D) {  // bao/desi_cmb_des5y.py:35-39
    const double h = wc.H0 / 100;
    const double h2 = h * h;
    wc.Onu = d.omnu_h2 / h2;
    wc.Or = d.or_h2 / h2;
    wc.Obc = (slot_get(d, CF_P_OBH2_D, th) + slot_get(d, CF_P_OCH2_D, th)) / h2;
    wc.Ode = 1.0 - wc.Obc - wc.Or - wc.Onu;
  }
  return wc;
}

// MODEL / FDE are compile-time: one kernel instantiation per expansion-rate family keeps the hot
// loops free of the other families' code (the all-in-one kernel was 14 k instructions, well past
// the instruction cache, and ran its table build 3x slower than its VALU work).

// exp(x) for the dark-energy density of the table build: |x| < ~40 (x = 3 (1 + w0 + wa) ln(1 + z) - 3 wa z / (1 + z) inside any
// prior box a sampler would draw from), so none of the library routine's overflow / underflow / NaN handling, and a
// table-driven reduction instead of its long polynomial:  x = (64 k + j) ln2 / 64 + r,  |r| <= ln2 / 128,
//   exp(x) = 2^k 2^(j/64) (1 + r + r^2/2 + ... + r^6/720)         (next term 1.5e-19 relative)
// tab[j] = 2^(j/64) correctly rounded from extended precision (cf_create), 512 B staged in LDS; ln2 / 64 as a hi + lo pair.
// <= 1.5 ulp; 17 instructions where the library's exp takes ~60 -- the table build of the CPL model was 90 instructions per
// grid node against 19 for LambdaCDM (profiles/r03_cpl_exp_ab.txt).
__device__ __forceinline__ double exp_tab(double x, const double* __restrict__ tab) {
  const double n = __builtin_rint(x * 0x1.71547652b82fep+6);  // 64 / ln 2
  double r = fma(-n, 0x1.62e42fefa39efp-7, x);
  r = fma(-n, 0x1.abc9e3b39803fp-62, r);
  const int ni = (int)n;
  const double t = tab[ni & 63];
  double p = fma(r, 1.0 / 720, 1.0 / 120);
  p = fma(p, r, 1.0 / 24);
  p = fma(p, r, 1.0 / 6);
  p = fma(p, r, 0.5);
  p = fma(p * r, r, r);  // expm1(r)
  return ldexp(fma(t, p, t), ni >> 6);
}

// `lnzp1` >= 0: ln(1 + z) of a grid node, tabulated at cf_create (theta-independent): the power of the wCDM / CPL forms
// becomes ONE exp of a product instead of pow (+ exp): zp1^a = exp(a ln zp1), a few ulp from the library pow and a third
// of its instructions -- the table build of the CPL model was 2.3x the LCDM one.  Absent (< 0): the reference's expression.
// TAB: `lnzp1` is known to be a tabulated value at compile time (the table build): the pow fallback is
// not even compiled in -- inlined eight times per thread it put ~1500 cold instructions into the hot loop -- and the exp is the
// table-driven exp_tab on `etab` (LDS).
template <int FDE, bool TAB = false>
__device__ __forceinline__ double f_de(const WalkerCosmo& wc, double z, double zp1, double cubed, double lnzp1 = -1.0,
                                       const double* __restrict__ etab = nullptr) {
  if (FDE == CF_FDE_LCDM_D) return 1.0;
  if (FDE == CF_FDE_WCDM_D) {
    if (TAB) return exp_tab(3 * (1 + wc.w0) * lnzp1, etab);
    return exp(3 * (1 + wc.w0) * (lnzp1 >= 0.0 ? lnzp1 : log(zp1)));  // zp1^a = exp(a ln zp1): log + exp, half of pow's instructions
  }
  if (FDE == CF_FDE_THAWING_D) {
    double r = 2 * cubed / ((1.0 + wc.w0) + (1.0 - wc.w0) * cubed);
    return r * r;
  }
  if (TAB) {  // z / (1 + z) through a refined reciprocal (1 + z in [1, 1 + z_max]: no special cases), <= 1 ulp from the quotient
    double r = __builtin_amdgcn_rcp(zp1);
    r = fma(fma(-zp1, r, 1.0), r, r);
    return exp_tab(fma(3 * (1 + wc.w0 + wc.wa), lnzp1, (-3 * wc.wa) * (z * r)), etab);
  }
  // the reference's pow(zp1, a) * exp(b) (bao/desi_fs_lya_cmb.py:19-22) as ONE exp: a ln zp1 + b carries |a ln zp1| ulps of the
  // logarithm's rounding into the result, <= 2e-14 relative at the Gauss-Legendre nodes of the sound horizon (z ~ 1e6) where the
  // dark-energy term is negligible anyway; 110 instructions instead of pow + exp's 180 (small_blocks_kernel evaluates it 400 times
  // per walker for the compressed-CMB block)
  return exp(fma(3 * (1 + wc.w0 + wc.wa), lnzp1 >= 0.0 ? lnzp1 : log(zp1), -3 * wc.wa * z / zp1));
}

// 5-node massive-neutrino density, cmb/data_planck_act_compression.py:53-66
__device__ __forceinline__ double omnu_z(const cf_dev_desc& d, double zp1) {
  const double r = d.nu_m0 / zp1, mz_sq = r * r;
  const double ws = sqrt(d.nu_qs_sq[0] + mz_sq) * d.nu_ws[0] + sqrt(d.nu_qs_sq[1] + mz_sq) * d.nu_ws[1] +
                    sqrt(d.nu_qs_sq[2] + mz_sq) * d.nu_ws[2] + sqrt(d.nu_qs_sq[3] + mz_sq) * d.nu_ws[3] +
                    sqrt(d.nu_qs_sq[4] + mz_sq) * d.nu_ws[4];
  const double zp1_2 = zp1 * zp1;
  return zp1_2 * zp1_2 * ws / d.nu_rho0;
}

// E^2(z) of both families.  `nu` < 0: evaluate the massive-neutrino density here; otherwise it is the
// value tabulated at cf_create for this grid node (it does not depend on theta: 5 sqrt + 2 divides saved
// per node and per walker, and the tabulated value is the reference's own arithmetic).
// TAB (the table build's register path): `nu` and `lnzp1` ARE tabulated values, the fallbacks are compiled out.
template <int MODEL, int FDE, bool TAB = false, class D = cf_dev_desc>
__device__ __forceinline__ double e2_of_z(const D& d, const WalkerCosmo& wc, double z, double nu = -1.0, double lnzp1 = -1.0,
                                          const double* __restrict__ etab = nullptr) {
  const double zp1 = 1.0 + z;
  const double cubed = zp1 * zp1 * zp1;
  if (MODEL == CF_EZ_LATE_FLAT_D)
    return (FDE == CF_FDE_LCDM_D) ? wc.Om * cubed + (1.0 - wc.Om)
                                  : wc.Om * cubed + (1.0 - wc.Om) * f_de<FDE, TAB>(wc, z, zp1, cubed, lnzp1, etab);
  const double de = (FDE == CF_FDE_LCDM_D) ? wc.Ode : wc.Ode * f_de<FDE, TAB>(wc, z, zp1, cubed, lnzp1, etab);
  if constexpr (!TAB) {
    if (nu < 0.0) nu = omnu_z(d, zp1);
  }
  return wc.Or * (cubed * zp1) + wc.Obc * cubed + de + wc.Onu * nu;  // bao/desi_cmb_des5y.py:43-48
}

// E^2(z) from a dark-energy density ratio f already in hand and the node's tabulated massive-neutrino density (table build).
template <int MODEL>
__device__ __forceinline__ double e2_from_fde(const WalkerCosmo& wc, double z, double f, double nu) {
  const double zp1 = 1.0 + z;
  const double cubed = zp1 * zp1 * zp1;
  if (MODEL == CF_EZ_LATE_FLAT_D) return wc.Om * cubed + (1.0 - wc.Om) * f;
  return wc.Or * (cubed * zp1) + wc.Obc * cubed + wc.Ode * f + wc.Onu * nu;  // bao/desi_cmb_des5y.py:43-48
}

// H(z) in the reference's form H0 * sqrt(E^2) (used where only a few values are needed).
template <int MODEL, int FDE>
__device__ __forceinline__ double H_of_z(const cf_dev_desc& d, const WalkerCosmo& wc, double z) {
  return wc.H0 * sqrt(e2_of_z<MODEL, FDE>(d, wc, z));
}

// dh(z) = c/H(z) on the grid without the sqrt + divide pair: (c/H0) * rsqrt(E^2), <= 2 ulp away.
template <int MODEL, int FDE>
__device__ __forceinline__ double dh_of_z_fast(const cf_dev_desc& d, const WalkerCosmo& wc, double c_over_H0, double z) {
  return c_over_H0 * rsqrt(e2_of_z<MODEL, FDE>(d, wc, z));
}

// rsqrt for positive finite normal arguments (E^2 on the grid): the library's seed + refinement without its
// zero / infinity / NaN selects (3 instructions per grid node).
__device__ __forceinline__ double rsqrt_pos(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y0, y0, 1.0);
  return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

// sqrt and quotient for positive, finite, normal operands whose results are normal too (densities, 1 + z, E^2 at the Gauss-Legendre
// nodes of the CMB distances): the library routines' own sequences -- v_rsq + one Goldschmidt step + two corrections; v_rcp + two
// Newton steps + the residual step of v_div_fmas -- without their exponent scaling and class selects, which is what such operands
// never take: the SAME BITS as sqrt() and a / b there, 10 instead of 19 and 8 instead of 12 instructions.
__device__ __forceinline__ double sqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  g = fma(fma(-g, g, x), h, g);
  return fma(fma(-g, g, x), h, g);
}
__device__ __forceinline__ double div_pos(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
}

// Grid node i of np.linspace(0, z_max, G): i*step, last node forced to z_max (sn/pantheon.py:16).
__device__ __forceinline__ double grid_z(int i, int G, double step, double z_max) {
  return i == G - 1 ? z_max : (double)i * step;
}

// the same at grid node g, with the tabulated neutrino density of that node
template <int MODEL, int FDE>
__device__ __forceinline__ double dh_of_node(const cf_dev_desc& d, const WalkerCosmo& wc, double c_over_H0, int g) {
  const double nu = (MODEL == CF_EZ_PHYSICAL_D && d.nu_grid) ? d.nu_grid[g] : -1.0;
  return c_over_H0 * rsqrt(e2_of_z<MODEL, FDE>(d, wc, grid_z(g, d.n_grid, d.step, d.z_max), nu));
}

// ------------------------------------------------------------------------------------------------
// Distance table of one walker in LDS: element g = {cum_dm[g], dh[g]} (16 B), stored at the
// skewed position g + (g >> CHS).  Thread t owns the 2^CHS contiguous nodes starting at t << CHS,
// so during the chunk-sequential prefix pass lane t touches position t*(2^CHS + 1) + k: a lane
// stride of an odd number of 16-byte slots, which is bank-conflict free for ds_write_b128 /
// ds_read_b128 (a power-of-two stride would put all 64 lanes on the same banks).
// ------------------------------------------------------------------------------------------------
#define CF_TPB_A 512

struct DistTable {
  const d2* tab;
  int G, chs;
  double step, inv_step, inv_last, z_max;
  __device__ __forceinline__ d2 at(int g) const { return tab[g + (g >> chs)]; }
};

// The CF_BAO_NODES table nodes around one BAO redshift, copied out of a walker's LDS table by walker_kernel
// (node `base` first): the same readers as for the full table, used by small_blocks_kernel.
#define CF_BAO_NODES 6
struct NodeView {
  const d2* p;
  int base, G;
  double step, inv_step, inv_last, z_max;
  __device__ __forceinline__ d2 at(int g) const {
    int o = g - base;
    o = o < 0 ? 0 : (o > CF_BAO_NODES - 1 ? CF_BAO_NODES - 1 : o);  // never outside the copy
    return p[o];
  }
};

// Cubic Hermite on the uniform grid (nodes cum_dm, slopes dh).  interpolator.py:71-108 with
// exact=True: interval i = searchsorted_left(x, xi) - 1, i.e. x[i] < xi <= x[i+1]; linear
// extrapolation outside.  t = (xi - x_i)/h_i is formed with the precomputed 1/h_i (<= 1 ulp).
template <class TAB>
__device__ __forceinline__ double hermite_tab(const TAB& T, double xi) {
  const int G = T.G;
  if (xi <= 0.0) {
    const d2 e = T.at(0);
    return e.x + e.y * (xi - 0.0);
  }
  if (xi >= T.z_max) {
    const d2 e = T.at(G - 1);
    return e.x + e.y * (xi - T.z_max);
  }
  int i = (int)(xi * T.inv_step);
  i = i > G - 2 ? G - 2 : i;
  // the estimate is off by at most one node; restore x[i] < xi <= x[i+1] (interpolator.py:94)
  double x0 = (double)i * T.step;
  if (x0 >= xi) {  // xi > 0, so i >= 1 here
    --i;
    x0 = (double)i * T.step;
  }
  double x1 = grid_z(i + 1, G, T.step, T.z_max);
  if (x1 < xi) {  // then i + 1 <= G - 2 because xi < z_max
    ++i;
    x0 = x1;
    x1 = grid_z(i + 1, G, T.step, T.z_max);
  }
  const double h_i = x1 - x0;
  const double t = (xi - x0) * (i == G - 2 ? T.inv_last : T.inv_step);
  const double t2 = t * t, t3 = t2 * t;
  const double h00 = 2 * t3 - 3 * t2 + 1;
  const double h10 = t3 - 2 * t2 + t;
  const double h01 = -2 * t3 + 3 * t2;
  const double h11 = t3 - t2;
  const d2 e0 = T.at(i), e1 = T.at(i + 1);
  return h00 * e0.x + h10 * h_i * e0.y + h01 * e1.x + h11 * h_i * e1.y;
}

// v_fma_f64 with its addend / one factor in a scalar register pair.  Written out because the compiler turns fma(p, r, C) with a loop-
// invariant C into a 64-bit register copy + the two-address v_fmac (one more issue slot per term of every polynomial of the SN loop).
__device__ __forceinline__ double fma_vvs(double a, double b, double c_uniform) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_uniform));
  return r;
}
__device__ __forceinline__ double fma_vsv(double a, double b_uniform, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
  return r;
}

// The same interpolant for the production SN loop.  The interval comes from one truncation, without restoring
// x[i] < xi <= x[i+1] at the ulp level: at a node both neighbouring cubics give the node value (C1 interpolant), so
// an xi within rounding of a node may use either; t = xi/step - i instead of (xi - x_i)/h_i and h = step also in
// the last interval (z_max vs (G-1)*step: a rounding apart) move the result by ~1e-14 relative, far inside the
// 1e-10 bar (tests/test_gpu_parity.py).  With the cubic in Horner form ~55 instructions fewer per supernova than
// hermite_tab.  Node i + 1 is read from the slot behind node i: the table builders put a copy of each chunk's first
// node into the spare slot behind the previous chunk.
__device__ __forceinline__ double hermite_fast(const DistTable& T, double xi) {
  const int G = T.G;
  if (__builtin_expect(xi > 0.0 && xi < T.z_max, 1)) {
    const double u = xi * T.inv_step;
    int i = (int)u;
    i = i > G - 2 ? G - 2 : i;
    const double t = u - (double)i;
    const d2* p = T.tab + (i + (i >> T.chs));
    const d2 e0 = p[0], e1 = p[1];
    // the same cubic in powers of t: y0 + t (h m0 + t (c + t d)), c = 3 dy - h (2 m0 + m1), d = h (m0 + m1) - 2 dy
    const double b = T.step * e0.y, hm1 = T.step * e1.y, dy = e1.x - e0.x;
    const double sm = b + hm1;
    const double dd = fma(-2.0, dy, sm);
    const double cc = fma(3.0, dy, -(sm + b));
    return fma(t, fma(t, fma(t, dd, cc), b), e0.x);
  }
  if (xi <= 0.0) {
    const d2 e = T.at(0);
    return e.x + e.y * xi;
  }
  const d2 e = T.at(G - 1);  // xi >= z_max, or NaN (which the sum keeps)
  return e.x + e.y * (xi - T.z_max);
}

// log10 for positive, finite, normal arguments (distances in Mpc): the fdlibm / msun algorithm
// (log1p kernel: 14-term odd polynomial in s = f/(2+f); hi/lo split of 1/ln10 and log10(2)), < 1 ulp.
// About a third of the instructions of the generic library call, which also handles zero, negative,
// subnormal and non-finite inputs; those fall back to it here.
__device__ __forceinline__ double log10_pos(double x) {
  if (!(x > 2.2250738585072014e-308 && x < 1.7976931348623157e308)) return log10(x);
  const double ivln10hi = 4.34294481878168880939e-01, ivln10lo = 2.50829467116452752298e-11;
  const double log10_2hi = 3.01029995663611771306e-01, log10_2lo = 3.69423907715893078616e-13;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  int k;
  double m = frexp(x, &k);  // m in [0.5, 1)
  if (m < 0.70710678118654752440) {
    m *= 2.0;
    k -= 1;
  }
  const double f = m - 1.0;  // in [sqrt(1/2) - 1, sqrt(2) - 1)
  const double hfsq = 0.5 * f * f;
  // s = f / (2 + f) through a refined reciprocal (one division's worth of accuracy, half its cost)
  const double den = 2.0 + f;
  double r = __builtin_amdgcn_rcp(den);
  r = fma(fma(-den, r, 1.0), r, r);
  double sq = f * r;
  sq = fma(fma(-den, sq, f), r, sq);
  const double z = sq * sq, w = z * z;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  const double R = sq * (hfsq + (t2 + t1));
  double hi = f - hfsq;
  hi = __longlong_as_double(__double_as_longlong(hi) & 0xFFFFFFFF00000000ll);
  const double lo = (f - hi) - hfsq + R;
  double val_hi = hi * ivln10hi;
  const double dk = (double)k;
  const double y2 = dk * log10_2hi;
  double val_lo = dk * log10_2lo + (lo + hi) * ivln10lo + lo * ivln10hi;
  const double ww = y2 + val_hi;
  val_lo += (y2 - ww) + val_hi;
  return val_lo + ww;
}

// log10 for the production SN loop (distances in Mpc): table-driven reduction instead of the long polynomial.
// x = m 2^e, m in [0.5, 1); c_j = centre of the j-th of 64 equal cells of [0.5, 1); r = m / c_j - 1, |r| <= 1/128;
//   log10 x = e log10(2) + log10(c_j) + log1p(r) / ln 10,      log1p by its Taylor series to r^8 (next term 5e-21).
// tab[j] = {1 / c_j, log10 c_j}, correctly rounded from long double on the host (cf_create), 1 KiB, staged in LDS.
// Absolute error <= ~2e-16 max(1, |log10 x|) (cf_selftest_log10, mode 1): what a distance modulus needs.  Near x = 1
// the RELATIVE error is not bounded by ulps (e log10(2) + log10 c_j cancel), which is why mu_corr = 5 log10(ratio ~ 1)
// of the accessor / calibrator paths stays on log10_pos.  ~20 instructions instead of ~48.
__device__ __forceinline__ double log10_tab(double x, const d2* __restrict__ tab) {
  if (!(x > 2.2250738585072014e-308 && x < 1.7976931348623157e308)) return log10(x);
  const double C1 = 4.34294481903251827651e-01, C2 = -2.17147240951625913826e-01, C3 = 1.44764827301083942550e-01,
               C4 = -1.08573620475812956913e-01, C5 = 8.68588963806503655303e-02, C6 = -7.23824136505419712752e-02,
               C7 = 6.20420688433216896645e-02, C8 = -5.42868102379064784564e-02;
  const double log10_2hi = 3.01029995663611771306e-01, log10_2lo = 3.69423907715893078616e-13;
  const int e = __builtin_amdgcn_frexp_exp(x);
  const double m = __builtin_amdgcn_frexp_mant(x);
  const d2 t = tab[(__double2hiint(m) >> 14) & 63];
  const double r = fma(m, t.x, -1.0);
  double p = fma_vsv(r, C8, C7);
  p = fma_vvs(p, r, C6);
  p = fma_vvs(p, r, C5);
  p = fma_vvs(p, r, C4);
  p = fma_vvs(p, r, C3);
  p = fma_vvs(p, r, C2);
  p = fma_vvs(p, r, C1);
  const double ek = (double)e;
  return fma_vsv(ek, log10_2hi, t.y) + fma(r, p, ek * log10_2lo);
}

// Inclusive scan across the 64 lanes of a wave on DPP row operations (no LDS round trips, unlike
// ds_bpermute-based shuffles): Hillis-Steele inside each row of 16 lanes (row_shr 1, 2, 4, 8), then
// lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast:15) and lane 31 into rows 2-3 (row_bcast:31).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);  // lanes without a source (or outside ROW_MASK) get 0
}

__device__ __forceinline__ double wave_inclusive_scan(double v) {
  v += dpp_move<0x111, 0xF>(v);  // row_shr:1
  v += dpp_move<0x112, 0xF>(v);  // row_shr:2
  v += dpp_move<0x114, 0xF>(v);  // row_shr:4
  v += dpp_move<0x118, 0xF>(v);  // row_shr:8
  v += dpp_move<0x142, 0xA>(v);  // row_bcast:15 -> rows 1 and 3
  v += dpp_move<0x143, 0xC>(v);  // row_bcast:31 -> rows 2 and 3
  return v;
}

// Build the table.  All CF_TPB_A threads call it.
//   dh[g]  = c/H(z_g)
//   cum[g] = sum_{k<g} (dh[k]+dh[k+1])/2 * (z[k+1]-z[k])          (sn/pantheon.py:35-39)
// Thread t owns CH = 2^chs contiguous nodes (CH*CF_TPB_A >= G): chunk-sequential trapezoid sums in
// registers, wave64 DPP scan over the chunk totals, wave totals and the wave-boundary intervals carried
// through LDS, then ONE 16-byte LDS store per node.
// One thread's chunk: dh at its CH nodes and the chunk-local trapezoid prefix, branch-free (the CH evaluation
// chains are independent and interleave; a per-node branch would serialise them).  Every wave runs the same arithmetic:
// nodes at g * step and the nominal half step as the one factor of every interval -- the reference's np.diff(z_grid)
// is `step` up to the rounding of i * step (1 ulp of z, i.e. <= 4e-13 of step, and the deviations telescope), also in the
// last interval (z_max vs (G-1) * step: a rounding apart).  LAST = the wave that holds the last grid node and the threads
// past the grid: the last node sits at z_max exactly (np.linspace forces it) and the nodes past the grid contribute
// nothing -- two selects per node, no other difference.  (Until round 2 the first and the last wave ran a separate path
// with true node differences; in-kernel stamps showed the other six waves waiting ~2.6 k of the workgroup's 13 k cycles
// for them at the barrier: profiles/r02_walker_stamps.txt.)
// The interval in front of a chunk belongs to it, with dh of the node before taken from the neighbouring lane
// (DPP wave_shr:1); lane 0 leaves its first interval out -- build_distance_table_regs adds the eight
// wave-boundary intervals with the carries.
template <int MODEL, int FDE, int CH, bool LAST, class D>
__device__ __forceinline__ double chunk_eval(const D& d, const WalkerCosmo& wc, double c_over_H0, int g0, int lane,
                                             const double (&nu_pre)[CH], const double (&ln)[CH], const double* __restrict__ etab,
                                             double (&dh)[CH], double (&loc)[CH]) {
  const int G = d.n_grid;
  // nu_pre / ln: the tabulated neutrino density and ln(1 + z) of the nodes (theta-independent tables, fetched by the kernel
  // before anything else); the tables exist whenever the model needs them (cf_create), so no fallback is compiled in
  constexpr bool POWER_LAW = FDE == CF_FDE_WCDM_D || FDE == CF_FDE_CPL_D;
  double fde[CH];
  if (POWER_LAW) {
    // The dark-energy density of a thread's CH consecutive nodes: f = exp(x), x = 3 (1 + w0 + wa) ln(1 + z) - 3 wa z / (1 + z)
    // (wCDM: wa = 0).  ONE table-driven exp (17 instructions) for the first node; from node to node x moves by
    // |dx| <= 3 |1 + w0 + wa| dz / (1 + z) + 3 |wa| dz < 0.02 on a 4000-node grid inside any prior box of the scripts, so
    // f_k = f_(k-1) (1 + expm1(dx)) with a degree-7 series (|dx|^8 / 8! < 2.3e-17 below the 2^-5 guard): 8 instead of 17
    // instructions per node, the chain of CH - 1 products costs <= (CH - 1) ulp (profiles/r04_fde_chain_ab.txt).
    double x[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int g = g0 + k;
      double z = (double)g * d.step;
      if (LAST) z = g >= G - 1 ? d.z_max : z;
      if (FDE == CF_FDE_CPL_D) {  // z / (1 + z) through a refined reciprocal (1 + z in [1, 1 + z_max]: no special cases), <= 1 ulp from the quotient
        const double zp1 = 1.0 + z;
        double r = __builtin_amdgcn_rcp(zp1);
        r = fma(fma(-zp1, r, 1.0), r, r);
        x[k] = fma(3 * (1 + wc.w0 + wc.wa), ln[k], (-3 * wc.wa) * (z * r));
      } else {
        x[k] = 3 * (1 + wc.w0) * ln[k];
      }
    }
    fde[0] = exp_tab(x[0], etab);
    double dx_max = 0.0;
#pragma unroll
    for (int k = 1; k < CH; ++k) {
      const double dx = x[k] - x[k - 1];
      dx_max = fmax(dx_max, fabs(dx));
      double p = fma(dx, 1.0 / 5040, 1.0 / 720);
      p = fma(p, dx, 1.0 / 120);
      p = fma(p, dx, 1.0 / 24);
      p = fma(p, dx, 1.0 / 6);
      p = fma(p, dx, 0.5);
      p = fma(p * dx, dx, dx);  // expm1(dx)
      fde[k] = fma(fde[k - 1], p, fde[k - 1]);
    }
    if (!(dx_max < 0x1p-5)) {  // a coarse grid or an extreme prior box: the series does not apply, an exp per node (NaN lands here too)
#pragma unroll
      for (int k = 1; k < CH; ++k) fde[k] = exp_tab(x[k], etab);
    }
  }
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int g = g0 + k;
    double z = (double)g * d.step;
    if (LAST) z = g >= G - 1 ? d.z_max : z;
    const double e2 = POWER_LAW ? e2_from_fde<MODEL>(wc, z, fde[k], nu_pre[k]) : e2_of_z<MODEL, FDE, true, D>(d, wc, z, nu_pre[k], ln[k], etab);
    const double v = c_over_H0 * rsqrt_pos(e2);
    dh[k] = (LAST && g >= G) ? 0.0 : v;
  }
  double prev = dpp_move<0x138, 0xF>(dh[CH - 1]);  // wave_shr:1; lane 0 gets 0 and skips its first interval
  double run = 0.0;
  const double half = 0.5 * d.step;
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const bool counts = (k > 0 || lane > 0) && (!LAST || g0 + k < G);
    run = counts ? fma_vsv(prev + dh[k], half, run) : run;  // three-address: loc[k - 1] stays where it is
    loc[k] = run;
    prev = dh[k];
  }
  return run;
}

template <int MODEL, int FDE, int CH, class D>
__device__ __forceinline__ void build_distance_table_regs(const D& d, const WalkerCosmo& wc, d2* tab,
                                                          d4* wave_pub, const double (&nu_pre)[CH], const double (&ln_pre)[CH],
                                                          const double* __restrict__ etab) {
  const int G = d.n_grid;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g0 = tid * CH;
  const int base = g0 + tid;  // skewed position of node g0
  const double c_over_H0 = wc.c / wc.H0;
  double dh[CH], loc[CH];
  const int wave_last = (tid - lane) * CH + 64 * CH - 1;  // last node of this wave
  const double run = wave_last < G - 1 ? chunk_eval<MODEL, FDE, CH, false>(d, wc, c_over_H0, g0, lane, nu_pre, ln_pre, etab, dh, loc)
                                       : chunk_eval<MODEL, FDE, CH, true>(d, wc, c_over_H0, g0, lane, nu_pre, ln_pre, etab, dh, loc);
  const double incl = wave_inclusive_scan(run);
  // per wave: {sum of its intervals, dh of its first node, dh of its last node}
  const double first_dh = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(dh[0])),
                                           __builtin_amdgcn_readfirstlane(__double2loint(dh[0])));
  if (lane == 63) wave_pub[wave] = (d4){incl, first_dh, dh[CH - 1], 0.0};
  __syncthreads();
  // exclusive prefix inside the wave + the earlier waves' sums + the wave-boundary intervals up to this wave:
  // lane v < 8 forms wave v's contribution, a DPP row reduction adds the eight, lane 7 broadcasts the sum
  double term = 0.0;
  if (lane < CF_TPB_A / 64) {
    const d4 pv = wave_pub[lane];
    term = lane < wave ? pv[0] : 0.0;
    if (lane >= 1) {
      const int gb = lane * 64 * CH;  // first node of wave `lane`; the interval (gb - 1, gb) exists if gb < G
      const double zb = gb == G - 1 ? d.z_max : (double)gb * d.step;
      const double bnd = (wave_pub[lane - 1][2] + pv[1]) / 2 * (zb - (double)(gb - 1) * d.step);
      term += (lane <= wave && gb < G) ? bnd : 0.0;
    }
  }
  term += dpp_move<0x111, 0xF>(term);  // row_shr:1
  term += dpp_move<0x112, 0xF>(term);  // row_shr:2
  term += dpp_move<0x114, 0xF>(term);  // row_shr:4 -> lane 7 holds lanes 0..7
  const double carry = (incl - run) + __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(term), 7),
                                                       __builtin_amdgcn_readlane(__double2loint(term), 7));
#pragma unroll
  for (int k = 0; k < CH; ++k)
    if (g0 + k < G) tab[base + k] = (d2){loc[k] + carry, dh[k]};
  if (tid > 0 && g0 < G) tab[base - 1] = (d2){loc[0] + carry, dh[0]};  // hermite_fast reads node i + 1 behind node i
  __syncthreads();
}

// Same result for long grids (CH > 8 would not fit the register budget): the chunk lives in LDS.
template <int MODEL, int FDE>
__device__ __forceinline__ void build_distance_table_lds(const cf_dev_desc& d, const WalkerCosmo& wc, d2* tab,
                                                         d4* wave_pub) {
  double* wave_tot = reinterpret_cast<double*>(wave_pub);
  const int G = d.n_grid, chs = d.chunk_shift, CH = 1 << chs;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g0 = tid << chs;
  const int base = g0 + tid;  // skewed position of node g0
  const int n_own = max(0, min(CH, G - g0));
  const double c_over_H0 = wc.c / wc.H0;
  for (int k = 0; k < n_own; ++k)
    tab[base + k].y = dh_of_node<MODEL, FDE>(d, wc, c_over_H0, g0 + k);
  __syncthreads();
  double run = 0.0;
  double prev = g0 > 0 && n_own > 0 ? tab[base - 2].y : 0.0;  // last node of the previous thread's chunk
  for (int k = 0; k < n_own; ++k) {
    const int g = g0 + k;
    const double cur = tab[base + k].y;
    if (g >= 1) {
      const double dz = grid_z(g, G, d.step, d.z_max) - grid_z(g - 1, G, d.step, d.z_max);
      run += (prev + cur) / 2 * dz;
    }
    tab[base + k].x = run;
    prev = cur;
  }
  const double incl = wave_inclusive_scan(run);
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  double carry = incl - run;
  for (int w = 0; w < wave; ++w) carry += wave_tot[w];
  for (int k = 0; k < n_own; ++k) tab[base + k].x += carry;
  if (tid > 0 && n_own > 0) tab[base - 1] = tab[base];  // hermite_fast reads node i + 1 behind node i
  __syncthreads();
}

template <int MODEL, int FDE>
__device__ __forceinline__ void build_distance_table(const cf_dev_desc& d, const WalkerCosmo& wc, d2* tab,
                                                     d4* wave_pub, const double (&nu_pre)[8], const double (&ln_pre)[8],
                                                     const double* __restrict__ etab) {
  // grids up to 4096 nodes (the reference uses 4000) take the register path, 8 nodes per thread
  if (d.chunk_shift == 3) build_distance_table_regs<MODEL, FDE, 8>(d, wc, tab, wave_pub, nu_pre, ln_pre, etab);
  else build_distance_table_lds<MODEL, FDE>(d, wc, tab, wave_pub);
}

// ------------------------------------------------------------------------------------------------
// PCHIP of the dh grid at xq (BAO D_H, bao/desi_cmb_des5y.py:88 -> interpolator.py:111-114): the
// reference builds all G Fritsch-Carlson slopes per walker; only the two at the bracketing nodes
// are needed, each from a 3-point stencil, with the exact branch logic of interpolator.py:25-66.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double sgn_d(double v) { return (double)((v > 0) - (v < 0)); }

template <class TAB>
__device__ double pchip_slope_tab(const TAB& T, int i) {
  const int n = T.G;
  auto X = [&](int k) { return grid_z(k, n, T.step, T.z_max); };
  auto Y = [&](int k) { return T.at(k).y; };
  if (i > 0 && i < n - 1) {
    const double hl = X(i) - X(i - 1), hr = X(i + 1) - X(i);
    const double dl = (Y(i) - Y(i - 1)) / hl, dr = (Y(i + 1) - Y(i)) / hr;
    if (dl != 0.0 && dr != 0.0 && dl * dr > 0.0) {
      const double w1 = 2.0 * hr + hl, w2 = hr + 2.0 * hl;
      return (w1 + w2) / (w1 / dl + w2 / dr);
    }
    return 0.0;
  }
  double h0, h1, d0, d1;
  if (i == 0) {
    h0 = X(1) - X(0); h1 = X(2) - X(1);
    d0 = (Y(1) - Y(0)) / h0; d1 = (Y(2) - Y(1)) / h1;
  } else {
    h0 = X(n - 1) - X(n - 2); h1 = X(n - 2) - X(n - 3);
    d0 = (Y(n - 1) - Y(n - 2)) / h0; d1 = (Y(n - 2) - Y(n - 3)) / h1;
  }
  const double e = ((2 * h0 + h1) * d0 - h0 * d1) / (h0 + h1);
  if (d0 == 0.0 || sgn_d(e) != sgn_d(d0)) return 0.0;
  if (sgn_d(d0) != sgn_d(d1) && fabs(e) > fabs(3 * d0)) return 3 * d0;
  return e;
}

template <class TAB>
__device__ double pchip_dh_tab(const TAB& T, double xi) {
  const int G = T.G;
  if (xi <= 0.0) return T.at(0).y;            // clamped outside (exact=False), interpolator.py:80-85
  if (xi >= T.z_max) return T.at(G - 1).y;
  int i = (int)(xi * T.inv_step);
  i = i > G - 2 ? G - 2 : i;
  if (i > 0 && grid_z(i, G, T.step, T.z_max) >= xi) --i;
  if (i < G - 2 && grid_z(i + 1, G, T.step, T.z_max) < xi) ++i;
  const double x0 = grid_z(i, G, T.step, T.z_max);
  const double h_i = grid_z(i + 1, G, T.step, T.z_max) - x0;
  const double t = (xi - x0) / h_i;
  const double t2 = t * t, t3 = t2 * t;
  const double h00 = 2 * t3 - 3 * t2 + 1, h10 = t3 - 2 * t2 + t, h01 = -2 * t3 + 3 * t2, h11 = t3 - t2;
  return h00 * T.at(i).y + h10 * h_i * pchip_slope_tab(T, i) + h01 * T.at(i + 1).y + h11 * h_i * pchip_slope_tab(T, i + 1);
}

// Fitting formulae of arXiv:2106.00428 with the reference's coefficients.
// zstar_fit = s1, s2, b, m, e0, c1, e1, e2, c2, e3, e4     cmb/data_planck_act_compression.py:86-99
__device__ double z_star_fit(const double* f, double wb, double wm) {
  wb = pow(wb, f[2]);
  wm = pow(wm, f[3]);
  return pow(wm, f[4]) + f[0] * f[5] * pow(wb, f[6]) * pow(wm, f[7]) + f[1] * f[8] * pow(wm, f[9]) * pow(wb, f[10]);
}
// rd_fit = b, m, a1..a9                                    cmb/data_planck_act_compression.py:102-124
__device__ double r_drag_fit(const double* f, double wb, double wm) {
  wb = pow(wb, f[0]);
  wm = pow(wm, f[1]);
  const double den = (f[2] * pow(wb, f[3])) + (f[4] * pow(wb, f[5]) * pow(wm, f[6])) + (f[7] * pow(wm, f[8]));
  return 1.0 / den - f[9] / pow(wm, f[10]);
}

// ------------------------------------------------------------------------------------------------
// SN residuals, production path (no accessor outputs, no fixed distance moduli): one Hermite, one
// log10 per SN; the per-SN inputs of the NEXT iteration are fetched while the current one computes.
// PM1: every step weight is +1 or -1 (the Heaviside step of sn/pantheon.py:46), so 1 + z_pec takes
// two values per walker and z_cosmo = -1 + (1+z) * (1/(1+z_pec)) needs no per-SN division
// (<= 1 ulp of 1+z_cosmo away from the quotient form).
// ------------------------------------------------------------------------------------------------
// LIN: the record's second field carries the coefficient of the linear magnitude term instead of a step weight (only
// likelihoods without a velocity step): offset_i = offset + lin * coef_i    bao/desi_cmb_pantheon_H0trgb.py:102-106
// FRAG: the residual goes to the small-batch solve kernel's B-fragment order instead of the walker's row (`out` then points at
// the walker's column of its 16-walker panel): row i of walker 16 px + col is double (i / 8) * 128 + (i / 2 % 4) * 32 + 2 col + i % 2
// of the panel's 16 n_ld doubles -- what lane (kq = i / 2 % 4, col) of the MFMA reads for K-step pair i / 8 (tri_gemm_small_kernel).
template <bool FRAG>
__device__ __forceinline__ int delta_index(int i) {
  return FRAG ? ((i >> 3) * 128 + ((i >> 1) & 3) * 32 + (i & 1)) : i;
}
template <bool PM1, bool LIN = false, bool FRAG = false, class D = cf_dev_desc>
__device__ __forceinline__ void sn_fast_loop(const D& d, const DistTable& T, const d2* __restrict__ log_tab,
                                             double* __restrict__ out, double off, double v100, int tid, double lin = 0.0,
                                             int part = 0, int n_parts = 1) {
  // part / n_parts: a small batch spreads a walker's SNe over n_parts workgroups (walker_fast_kernel): this one takes the SNe
  // part * CF_TPB_A + tid, then every n_parts * CF_TPB_A-th
  const int n_sn = d.n_sn, first = part * CF_TPB_A + tid, stride = n_parts * CF_TPB_A;
  // PM1: z_cosmo = fma(za, r, c0) -- {r, c0} = {1 / (1 + z_pec), -1} on 1 + z_cmb with a velocity step, {1, 0} on z_cmb (exact) without
  double r_pos = 1.0, r_neg = 1.0, c0 = 0.0;
  if (PM1 && d.has_vstep) {
    r_pos = 1.0 / (1.0 + v100 / d.c);
    r_neg = 1.0 / (1.0 + (-v100) / d.c);
    c0 = -1.0;
  }
  // {has_vstep ? 1 + z_cmb : z_cmb, step, 1 + z_hel, obs}: the two sums are the reference's own first operations
  const d4* __restrict__ rec = reinterpret_cast<const d4*>(d.sn_rec) + first;
  // rows n_sn .. n_ld - 1 are zero padding for the MFMA tiles / the 64-row blocks of the inverse-GEMM solve
  if (part == 0 && tid < d.n_ld - n_sn) out[delta_index<FRAG>(n_sn + tid)] = 0.0;
  auto one_sn = [&](const d4& r, int i) {
    const double za = r[0], st = r[1], zhp1 = r[2], ob = r[3];
    double z_cosmo = za;
    if (!LIN && PM1) {
      z_cosmo = fma(za, st > 0.0 ? r_pos : r_neg, c0);
    } else if (!LIN && d.has_vstep) {  // general weights (dipole fits): sn/pantheon.py:43-48 as written
      const double z_pec = (v100 * st) / d.c;
      z_cosmo = -1.0 + za / (1.0 + z_pec);
    }
    const double off_i = LIN ? off + lin * st : off;
    out[delta_index<FRAG>(i)] = ob - off_i - fma_vsv(log10_tab(zhp1 * hermite_fast(T, z_cosmo), log_tab), 5.0, 25.0);
  };
  // two records in flight, ping-pong (no register copies); the record array carries CF_SN_REC_SLACK spare entries past n_ld.
  // (A variant with three branch-free evaluations per iteration -- selects instead of the extrapolation / range guards, so
  // that the three dependent chains share one basic block -- measured the same 0.051 ms: the loop is issue-, not
  // latency-bound; profiles/r02_sn_loop_ab.txt.)
  d4 ra = rec[0], rb;
  int i = first;
  while (i < n_sn) {
    rb = rec[stride];
    one_sn(ra, i);
    i += stride;
    if (i >= n_sn) break;
    rec += 2 * stride;
    ra = rec[0];
    one_sn(rb, i);
    i += stride;
  }
}

// ------------------------------------------------------------------------------------------------
// Kernel A: one 512-thread workgroup per walker: distance table and SN residual vector.
//
// SN fast path (dm_out == mucorr_out == NULL): the two magnitude terms of the reference,
//   mu_corr + mu_theory = 5 log10(DM(z_cosmo)/DM(z_cmb)) + 25 + 5 log10((1+z_hel) DM(z_cmb)),
// are evaluated as the algebraically identical 25 + 5 log10((1+z_hel) DM(z_cosmo)): one Hermite
// and one log10 per SN instead of two and two (difference ~1e-15 mag, tests/test_gpu_parity.py).
// The accessor path (cf_eval_parts) keeps the reference's exact sequence  sn/pantheon.py:43-61.
//
// The small blocks of the joint likelihoods (BAO, compressed CMB, cosmic chronometers) are evaluated by
// small_blocks_kernel after this kernel; the BAO block needs a few table nodes per datum, copied out here.
// ------------------------------------------------------------------------------------------------
template <int MODEL, int FDE>
__global__ void __launch_bounds__(CF_TPB_A, 4)
walker_kernel(cf_dev_desc d, const double* __restrict__ theta, int64_t W, double* __restrict__ delta,
              double* __restrict__ dm_out, double* __restrict__ mucorr_out, d2* __restrict__ bao_nodes, d2* __restrict__ table_out) {
  extern __shared__ __align__(16) d2 lds_tab[];
  __shared__ __align__(16) d4 wave_pub[CF_TPB_A / 64];  // per-wave {interval sum, first dh, last dh} of the table build
  __shared__ __align__(16) d2 log_tab[64];  // log10_tab's reduction table; the table build's barriers order the fill
  __shared__ double exp2_tab[64];           // exp_tab's 2^(j/64) (wCDM / CPL table build)

  const int64_t w = blockIdx.x;
  if (w >= W) return;
  const double* th = theta + w * d.ndim;
  const int tid = threadIdx.x;
  if (tid < 64 && d.n_sn > 0) log_tab[tid] = reinterpret_cast<const d2*>(d.log10_tab)[tid];
  // tabulated massive-neutrino density of this thread's 8 grid nodes (register path of the table build): fetched
  // before anything else so that the loads fly while theta is read and the cosmology scalars are formed
  double nu_pre[8], ln_pre[8];
  constexpr bool POWER_LAW = FDE == CF_FDE_WCDM_D || FDE == CF_FDE_CPL_D;  // ln(1 + z) of the nodes: zp1^a = exp(a ln zp1)
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // swizzled copies: thread-contiguous chunks of the natural table would make every one of these loads touch 64 cache lines
    nu_pre[k] = (MODEL == CF_EZ_PHYSICAL_D && d.chunk_shift == 3) ? d.nu_sw[k * CF_TPB_A + tid] : -1.0;
    ln_pre[k] = (POWER_LAW && d.chunk_shift == 3) ? d.ln_sw[k * CF_TPB_A + tid] : -1.0;
  }
  // (only the register path of the table build uses it; longer grids take build_distance_table_lds and the library exp,
  // and their descriptor carries no table)
  if (POWER_LAW && d.chunk_shift == 3 && tid >= 64 && tid < 128) exp2_tab[tid - 64] = d.exp2_tab[tid - 64];
  if (POWER_LAW && d.chunk_shift == 3) __syncthreads();  // the table build reads exp2_tab before its own first barrier
  const WalkerCosmo wc = make_cosmo(d, th);
  DistTable T;
  T.tab = lds_tab;
  T.G = d.n_grid;
  T.chs = d.chunk_shift;
  T.step = d.step;
  T.inv_step = d.inv_step;
  T.inv_last = d.inv_last;
  T.z_max = d.z_max;
  if (d.n_sn > 0 || d.n_aux > 0 || table_out) build_distance_table<MODEL, FDE>(d, wc, lds_tab, wave_pub, nu_pre, ln_pre, exp2_tab);
  if (table_out)  // accessor path (cf_eval_table): the walker's whole {cum_dm, dh} table, node order
    for (int g = tid; g < d.n_grid; g += CF_TPB_A) table_out[w * d.n_grid + g] = T.at(g);
  // the table nodes around each BAO redshift go to small_blocks_kernel (the BAO block is evaluated there, one wave
  // per walker); copied by the last threads of the workgroup, whose waves have the lightest share of the SN loop
  for (int e = CF_TPB_A - 1 - tid; e < CF_BAO_NODES * d.n_aux; e += CF_TPB_A) {
    const int k = e / CF_BAO_NODES, o = e % CF_BAO_NODES;
    bao_nodes[(w * d.n_aux + k) * CF_BAO_NODES + o] = T.at(d.bao_base[k] + o);
  }

  // ---- SN residual vector ----
  if (d.n_sn > 0) {
    double* out = delta + w * d.n_ld;
    const double off = slot_get(d, CF_P_OFFSET_D, th);
    const double v100 = 100 * slot_get(d, CF_P_V_D, th);
    const double lin = d.sn_lin ? slot_get(d, CF_P_LIN_D, th) : 0.0;
    const bool parts = dm_out != nullptr || mucorr_out != nullptr;
    if (!parts && !d.sn_fixed_mu && !d.sn_dir && !d.sn_vel_mult && (!d.sn_lin || d.lin_in_rec)) {
      if (d.lin_in_rec) sn_fast_loop<false, true>(d, T, log_tab, out, off, v100, tid, lin);
      else if (d.step_pm1) sn_fast_loop<true>(d, T, log_tab, out, off, v100, tid);
      else sn_fast_loop<false>(d, T, log_tab, out, off, v100, tid);
    } else {
      // direction-dependent velocity: v_los = n . (V, V2, V3), weight = sn_step    sn/pantheon_dipole_xyz.py:54-57
      const double vx = slot_get(d, CF_P_V_D, th), vy = slot_get(d, CF_P_V2_D, th), vz = slot_get(d, CF_P_V3_D, th);
      for (int i = tid; i < d.n_ld; i += CF_TPB_A) {
        double res = 0.0;
        if (i < d.n_sn) {
          const double zc = d.z_cmb[i];
          const double off_i = d.sn_lin ? off + lin * d.sn_lin[i] : off;
          double z_cosmo = zc;
          if (d.has_vstep) {  // sn/pantheon.py:43-49
            const double v_km_s = d.sn_dir ? 100 * (d.sn_dir[3 * i] * vx + d.sn_dir[3 * i + 1] * vy + d.sn_dir[3 * i + 2] * vz) * d.sn_step[i]
                                           : v100 * d.sn_step[i];
            const double z_pec = v_km_s / d.c;
            z_cosmo = -1.0 + (1.0 + zc) / (1.0 + z_pec);
            if (d.sn_vel_mult) z_cosmo = fmax((1.0 + zc) * (1.0 + z_pec) - 1.0, 1e-8);  // bao/desi_pantheon_cc.py:84-87
          }
          const double DMc = hermite_tab(T, z_cosmo);
          const double fixed = d.sn_fixed_mu ? d.sn_fixed_mu[i] : __longlong_as_double(0x7ff8000000000000ll);
          if (fixed == fixed) {  // calibrator: its distance modulus is data, only mu_corr is theory (sn/pantheon_and_sh0es.py:65-67)
            const double DM = hermite_tab(T, zc);
            const double mu_corr = d.has_vstep ? 5.0 * log10_pos(DMc / DM) : 0.0;
            res = d.obs[i] - off_i - mu_corr - fixed;
            if (dm_out) dm_out[w * d.n_sn + i] = DM;
            if (mucorr_out) mucorr_out[w * d.n_sn + i] = mu_corr;
          } else if (!parts) {
            res = d.obs[i] - off_i - (25.0 + 5 * log10_pos((1.0 + d.z_hel[i]) * DMc));
          } else {
            const double DM = hermite_tab(T, zc);
            const double mu_corr = d.has_vstep ? 5.0 * log10(DMc / DM) : 0.0;
            const double mu_th = 25.0 + 5 * log10((1.0 + d.z_hel[i]) * DM);  // sn/pantheon.py:52-54
            res = d.obs[i] - off_i - mu_corr - mu_th;                        // sn/pantheon.py:59-60
            if (dm_out) dm_out[w * d.n_sn + i] = DM;
            if (mucorr_out) mucorr_out[w * d.n_sn + i] = mu_corr;
          }
        }
        out[i] = res;  // rows >= n_sn are zero padding for the 16-row MFMA tiles
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The PRODUCTION form of Kernel A: the same table build and SN loop for the evaluations that take none of the accessor /
// calibrator / direction-dependent paths (cf_eval, cf_eval_device of every likelihood whose SN block fits sn_fast_loop) and a
// grid of at most 4096 nodes.  Lean kernel arguments (cf_walker_args), the walker's theta row across the lanes (ThetaRow):
// the prologue is one theta load + the scalar loads of the fields this path uses, instead of ~60 serialised scalar loads.
// ------------------------------------------------------------------------------------------------
template <int MODEL, int FDE>
__global__ void __launch_bounds__(CF_TPB_A, 4)
walker_fast_kernel(cf_walker_args d, const double* __restrict__ theta, int64_t W, double* __restrict__ delta,
                   d2* __restrict__ bao_nodes, double* __restrict__ theta_copy, int frag_b, int sn_parts) {
  extern __shared__ __align__(16) d2 lds_tab[];
  __shared__ __align__(16) d4 wave_pub[CF_TPB_A / 64];
  __shared__ __align__(16) d2 log_tab[64];
  __shared__ double exp2_all[CF_TPB_A / 64][64];  // exp_tab's 2^(j/64), one copy PER WAVE: filled with no workgroup barrier

  // sn_parts > 1 (a small batch: most of the chip would idle): sn_parts workgroups per walker, each builds the walker's table and
  // evaluates every sn_parts-th stretch of CF_TPB_A SNe (the residual rows are disjoint); part 0 writes everything else
  const int64_t w = sn_parts > 1 ? blockIdx.x / (unsigned)sn_parts : blockIdx.x;
  const int part = sn_parts > 1 ? (int)(blockIdx.x % (unsigned)sn_parts) : 0;
  if (w >= W) return;
  const int tid = threadIdx.x, lane = tid & 63;
  double* const exp2_tab = exp2_all[tid >> 6];
  // loads in the order their values are needed: the theta row (the cosmology scalars wait for nothing else), the two small
  // reduction tables, then the theta-independent node tables of the table build
  const ThetaRow th{theta[w * d.ndim + (lane < d.ndim ? lane : 0)]};
  if (theta_copy && part == 0 && tid < d.ndim) theta_copy[w * d.ndim + tid] = th.v;  // for the later kernels of a zero-copy evaluation
  constexpr bool POWER_LAW = FDE == CF_FDE_WCDM_D || FDE == CF_FDE_CPL_D;
  d2 lt = (d2){0.0, 0.0};
  double et = 0.0;
  if (tid < 64 && d.n_sn > 0) lt = reinterpret_cast<const d2*>(d.log10_tab)[tid];
  if (POWER_LAW) et = d.exp2_tab[lane];
  double nu_pre[8], ln_pre[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    nu_pre[k] = MODEL == CF_EZ_PHYSICAL_D ? d.nu_sw[k * CF_TPB_A + tid] : -1.0;
    ln_pre[k] = POWER_LAW ? d.ln_sw[k * CF_TPB_A + tid] : -1.0;
  }
  if (tid < 64 && d.n_sn > 0) log_tab[tid] = lt;
  if (POWER_LAW) exp2_tab[lane] = et;  // read back by this wave only: a wave's LDS accesses complete in order
  const WalkerCosmo wc = make_cosmo(d, th);
  const double off = slot_get(d, CF_P_OFFSET_D, th);
  const double v100 = 100 * slot_get(d, CF_P_V_D, th);
  const double lin = d.lin_in_rec ? slot_get(d, CF_P_LIN_D, th) : 0.0;
  DistTable T;
  T.tab = lds_tab;
  T.G = d.n_grid;
  T.chs = 3;
  T.step = d.step;
  T.inv_step = d.inv_step;
  T.inv_last = d.inv_last;
  T.z_max = d.z_max;
  build_distance_table_regs<MODEL, FDE, 8>(d, wc, lds_tab, wave_pub, nu_pre, ln_pre, exp2_tab);
  for (int e = CF_TPB_A - 1 - tid; part == 0 && e < CF_BAO_NODES * d.n_aux; e += CF_TPB_A) {
    const int k = e / CF_BAO_NODES, o = e % CF_BAO_NODES;
    bao_nodes[(w * d.n_aux + k) * CF_BAO_NODES + o] = T.at(d.bao_base[k] + o);
  }
  if (d.n_sn > 0 && !frag_b) {
    double* out = delta + w * d.n_ld;
    if (d.lin_in_rec) sn_fast_loop<false, true>(d, T, log_tab, out, off, v100, tid, lin, part, sn_parts);
    else if (d.step_pm1) sn_fast_loop<true>(d, T, log_tab, out, off, v100, tid, 0.0, part, sn_parts);
    else sn_fast_loop<false>(d, T, log_tab, out, off, v100, tid, 0.0, part, sn_parts);
  } else if (d.n_sn > 0) {  // a small batch: the residuals in the solve kernel's fragment order (sn_fast_loop, FRAG)
    double* out = delta + (w >> 4) * (16 * (int64_t)d.n_ld) + 2 * (w & 15);
    if (d.lin_in_rec) sn_fast_loop<false, true, true>(d, T, log_tab, out, off, v100, tid, lin, part, sn_parts);
    else if (d.step_pm1) sn_fast_loop<true, false, true>(d, T, log_tab, out, off, v100, tid, 0.0, part, sn_parts);
    else sn_fast_loop<false, false, true>(d, T, log_tab, out, off, v100, tid, 0.0, part, sn_parts);
  }
}

// H(z) at a Gauss-Legendre node of the compressed-CMB distances (small_blocks_kernel evaluates it 2 n_gl times per walker: this is
// where that kernel's instructions go, and it is bound by the instructions it issues -- profiles/NOTES_r04.md).  E^2 as e2_of_z
// writes it, with sqrt_pos / div_pos (the library's bits) and, for the power-law dark-energy forms, ONE table-driven exp of
// a ln(1 + z) + b with ln through the table-driven log10 (both tables in LDS): 3 (1 + w0 + wa) <= 12 times the logarithm's
// <= 3e-16 max(1, log10(1 + z)) leaves the density ratio within ~4e-14 of the library's exp(a log(1 + z) + b) where 1 + z ~ 1e7
// and the term is negligible, within 1e-15 at z < 10 where it is not.
template <int MODEL, int FDE>
__device__ __forceinline__ double H_at_gl_node(const cf_dev_desc& d, const WalkerCosmo& wc, double z, const double* __restrict__ etab,
                                               const d2* __restrict__ ltab) {
  const double zp1 = 1.0 + z;
  const double cubed = zp1 * zp1 * zp1;
  double f = 1.0;
  if (FDE == CF_FDE_WCDM_D) {
    f = exp_tab((3 * (1 + wc.w0)) * (log10_tab(zp1, ltab) * 2.30258509299404568402), etab);
  } else if (FDE == CF_FDE_CPL_D) {
    f = exp_tab(fma(3 * (1 + wc.w0 + wc.wa), log10_tab(zp1, ltab) * 2.30258509299404568402, div_pos(-3 * wc.wa * z, zp1)), etab);
  } else if (FDE == CF_FDE_THAWING_D) {
    const double r = div_pos(2 * cubed, (1.0 + wc.w0) + (1.0 - wc.w0) * cubed);
    f = r * r;
  }
  double e2;
  if (MODEL == CF_EZ_LATE_FLAT_D) {
    e2 = (FDE == CF_FDE_LCDM_D) ? wc.Om * cubed + (1.0 - wc.Om) : wc.Om * cubed + (1.0 - wc.Om) * f;
  } else {
    const double de = (FDE == CF_FDE_LCDM_D) ? wc.Ode : wc.Ode * f;
    // omnu_z with the positive-operand sqrt / quotient   cmb/data_planck_act_compression.py:53-66
    const double r = div_pos(d.nu_m0, zp1), mz_sq = r * r;
    const double ws = sqrt_pos(d.nu_qs_sq[0] + mz_sq) * d.nu_ws[0] + sqrt_pos(d.nu_qs_sq[1] + mz_sq) * d.nu_ws[1] +
                      sqrt_pos(d.nu_qs_sq[2] + mz_sq) * d.nu_ws[2] + sqrt_pos(d.nu_qs_sq[3] + mz_sq) * d.nu_ws[3] +
                      sqrt_pos(d.nu_qs_sq[4] + mz_sq) * d.nu_ws[4];
    const double zp1_2 = zp1 * zp1;
    const double nu = div_pos(zp1_2 * zp1_2 * ws, d.nu_rho0);
    e2 = wc.Or * (cubed * zp1) + wc.Obc * cubed + de + wc.Onu * nu;  // bao/desi_cmb_des5y.py:43-48
  }
  return wc.H0 * sqrt_pos(e2);
}

// ------------------------------------------------------------------------------------------------
// The small blocks of the joint likelihoods -- z* / r_drag fitting formulae, compressed CMB (2 x n_gl
// Gauss-Legendre nodes), cosmic chronometers, BAO -- after walker_kernel, SIXTEEN LANES per walker (four walkers
// per wave, sixteen per 256-thread workgroup).  Only the BAO block needs the distance table, and of it only the few
// nodes around each BAO redshift, which walker_kernel copies out; so thousands of walkers fill the chip here
// (inside walker_kernel these serial sections held a 72 KB table workgroup for as long as the table build and the
// SN loop together).  Sixteen lanes, not a wave: the powers and the BAO data occupy 4-14 lanes per walker, and a
// SIMD spends a whole wave-instruction on them however many lanes are active.
// chi2_extra[w] = chi2_bao + chi2_cmb + chi2_cc;  blocks_out[w] = (bao, cmb, cmb vector[3], cc, z*, r_d), bao_out[w][k] optional.
//   z* / r_drag: sums of products of powers (14 calls of pow), one power per lane, two dependent rounds, combined
//   in the reference's order                               cmb/data_planck_act_compression.py:86-124
//   CMB: lane l takes nodes l, l + 16, ... of r_s(z*) (in a) and D_M(z*) (in z), a butterfly adds the lanes
//   (the reference adds in node order: differs at the 1e-16 level)   cmb/data_planck_act_compression.py:160-212
//   CC / BAO: lane k (+16, ...) forms delta_k -- H_obs - H(z_k) (bao/desi_union3_cc_theta_star.py:129-139), or BAO
//   datum k: Hermite D_M; D_H by PCHIP with the two Fritsch-Carlson slopes it needs, or exactly as c/H
//   (bao/desi_cmb_des5y.py:82-100,132-135) -- then column j of delta @ inv_cov, then ... @ delta by a butterfly
// ------------------------------------------------------------------------------------------------
template <int LANES>
__device__ __forceinline__ double group_sum(double v) {  // over the LANES lanes of a walker
#pragma unroll
  for (int o = LANES / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, LANES);
  return v;
}

// A sum over items k = 0 .. n - 1 spread over the lanes of a walker's group that does NOT depend on the group's width (16, 32 or
// 64 lanes: a batch of a few walkers takes wide groups -- fewer items per lane, a shorter serial chain -- a large one narrow
// groups, and a walker's result must be the same bits in both).  64 VIRTUAL lanes: virtual lane v adds its items v, v + 64, ...
// in ascending order, and the 64 partial sums meet in the xor butterfly of a 64-lane group.  A group of LANES < 64 lanes keeps
// NV = 64 / LANES partial sums per lane (lane sl holds the virtual lanes sl, sl + LANES, ...) and folds them in the butterfly's own
// order -- its first log2(NV) levels -- before the remaining levels run across the lanes.
// Use: every lane calls push() the same number of times, iteration `it` with the term of item sl + it LANES (0.0 past the end);
// the partial sums rotate through a[] so that the one being added to is always a[0] (registers, no dynamic index).
template <int LANES>
struct VirtualLaneSum {
  static constexpr int NV = 64 / LANES;
  double a[NV];
  __device__ __forceinline__ VirtualLaneSum() {
#pragma unroll
    for (int j = 0; j < NV; ++j) a[j] = 0.0;
  }
  __device__ __forceinline__ void push(double t) {
    const double tmp = a[0] + t;
#pragma unroll
    for (int j = 0; j + 1 < NV; ++j) a[j] = a[j + 1];
    a[NV - 1] = tmp;
  }
  __device__ __forceinline__ double total(int pushes) const {  // after `pushes` calls virtual lane sl + j LANES sits in a[(j - pushes) mod NV]
    double v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      v[j] = a[0];
#pragma unroll
      for (int i = 1; i < NV; ++i) v[j] = (((j - pushes) & (NV - 1)) == i) ? a[i] : v[j];
    }
    double r = v[0];
    if (NV == 2) r = v[0] + v[1];
    if (NV == 4) r = (v[0] + v[2]) + (v[1] + v[3]);
    return group_sum<LANES>(r);
  }
};

// delta @ inv_cov @ delta for a vector held in LDS (n <= 64): a lane forms column j (j = sl, sl + LANES, ...) in the reference's
// order of operations; the columns' products meet in a VirtualLaneSum
template <int LANES>
__device__ __forceinline__ double group_quadratic_form(const double* __restrict__ dl, const double* __restrict__ inv_cov, int n,
                                                       int sl) {
  VirtualLaneSum<LANES> acc;
  const int iters = (n + LANES - 1) / LANES;
  for (int it = 0; it < iters; ++it) {
    const int j = sl + it * LANES, jc = j < n ? j : n - 1;
    // eight entries of the column in flight at a time (one load + one add per iteration waited a memory round trip each: the kernel
    // spent half its life in s_waitcnt, profiles/NOTES_r04.md); the additions keep their order
    double t = 0.0;
    for (int i0 = 0; i0 < n; i0 += 8) {
      double c[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) c[k] = inv_cov[(i0 + k < n ? i0 + k : n - 1) * n + jc];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (i0 + k < n) t += dl[i0 + k] * c[k];
    }
    acc.push(j < n ? t * dl[jc] : 0.0);
  }
  return acc.total(iters);
}

// ROLES = 2 (only with a wave per walker, LANES = 64: batches that leave the chip mostly idle): TWO waves per walker.  Wave A runs the
// z* / r_drag powers, then the compressed-CMB integrals; wave B the cosmic chronometers and the BAO numerators (table look-ups, the
// cube root), which need nothing from A but the final division by r_d.  Two workgroup barriers: after the first B reads r_d from LDS
// and finishes the BAO block while A integrates, after the second A adds B's two chi^2 and writes the walker's results.  Every
// operation and every order of operations is that of the one-role form: the same bits (tests/test_gpu_joint.py, test_gpu_random_shapes.py).
template <int MODEL, int FDE, int LANES, int ROLES>
__global__ void __launch_bounds__(256)
small_blocks_kernel(cf_dev_desc d, const double* __restrict__ theta, int64_t W, const d2* __restrict__ bao_nodes,
                    double* __restrict__ chi2_extra, double* __restrict__ blocks_out, double* __restrict__ bao_out) {
  static_assert(LANES == 16 || LANES == 32 || LANES == 64, "lanes per walker");
  static_assert(ROLES == 1 || (ROLES == 2 && LANES == 64 && CF_MAX_BAO <= 64), "two roles: a wave each, one BAO datum per lane");
  constexpr int CF_SB_LANES = LANES;
  constexpr bool SPLIT = ROLES == 2;
  constexpr int WALKERS_PER_WG = 256 / (LANES * ROLES);
  __shared__ double delta_s[WALKERS_PER_WG][CF_MAX_BAO > CF_MAX_CC ? CF_MAX_BAO : CF_MAX_CC];
  __shared__ double xch[WALKERS_PER_WG][4];  // SPLIT: {r_d from wave A; chi2_bao, chi2_cc from wave B}
  // reduction tables of exp_tab / log10_tab for the dark-energy factor at the Gauss-Legendre nodes (H_at_gl_node); cf_create uploads
  // them whenever a compressed-CMB block meets a power-law dark energy
  constexpr bool POWER_LAW = FDE == CF_FDE_WCDM_D || FDE == CF_FDE_CPL_D;
  __shared__ double exp2_s[POWER_LAW ? 64 : 1];
  __shared__ __align__(16) d2 log_s[POWER_LAW ? 64 : 1];
  // ... and the Gauss-Legendre nodes and weights: fetched by the loop they were two loads per iteration, each waited for on the spot
  static_assert(CF_MAX_GL <= 256, "one Gauss-Legendre node per thread of the workgroup");
  __shared__ __align__(16) d2 gl_s[CF_MAX_GL];  // {x, w}
  if (d.cmb_mode) {
    if (threadIdx.x < d.n_gl) gl_s[threadIdx.x] = (d2){d.gl_x[threadIdx.x], d.gl_w[threadIdx.x]};  // n_gl <= CF_MAX_GL = the workgroup's 256 threads
    if (POWER_LAW) {
      if (threadIdx.x >= 128 && threadIdx.x < 192) exp2_s[threadIdx.x - 128] = d.exp2_tab[threadIdx.x - 128];
      else if (threadIdx.x >= 192) log_s[threadIdx.x - 192] = reinterpret_cast<const d2*>(d.log10_tab)[threadIdx.x - 192];
    }
    __syncthreads();
  }
  const int grp = threadIdx.x / (LANES * ROLES), sl = threadIdx.x % LANES;
  const bool do_a = !SPLIT || ((threadIdx.x / LANES) & 1) == 0;  // powers, CMB, output
  const bool do_b = !SPLIT || ((threadIdx.x / LANES) & 1) == 1;  // cosmic chronometers, BAO
  const int64_t w_raw = (int64_t)blockIdx.x * WALKERS_PER_WG + grp;
  const bool live = w_raw < W;
  const int64_t w = live ? w_raw : W - 1;  // spare groups of the last workgroup shadow the last walker and write nothing
  // the walker's theta row across the lanes of its group: ONE load that waits for nothing (read slot by slot through the pointer,
  // each slot was a load behind the scalar load of its index)
  const ThetaGroup<LANES> th{theta[w * d.ndim + (sl < d.ndim ? sl : 0)]};
  const WalkerCosmo wc = make_cosmo(d, th);
  const double Ob = slot_get(d, CF_P_OBH2_D, th), Oc = slot_get(d, CF_P_OCH2_D, th);
  double* dl = delta_s[grp];
  if (Ob == 1.2345e300) dl[0] = wc.H0;  // (stamps: keeps the loads in front of the stamp)
  // dl is written and read by the lanes of one group: a workgroup barrier when a wave holds several groups' neighbours, the wave's
  // own LDS order when the group IS the wave (SPLIT: the other role's waves are not at this point of the program)
  auto group_sync = [&]() {
    if (SPLIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else __syncthreads();
  };
  double z_star = 0.0, r_d = 0.0;
  auto pow_block = [&]() {
    const double h_late = wc.H0 / 100;
    const double wm_z = Oc + Ob + d.omnu_h2, wm_r = d.rd_wm_late ? wc.Om * (h_late * h_late) : Ob + Oc + d.omnu_h2;
    const double* fz = d.zstar_fit;  // s1 s2 b m e0 c1 e1 e2 c2 e3 e4
    const double* fr = d.rd_fit;     // b m a1..a9
    // round 1: lane 0 wb^b(z*), 1 wm^m(z*), 2 wb^b(rd), 3 wm^m(rd)
    const double base1 = (sl & 1) ? ((sl & 2) ? wm_r : wm_z) : Ob;
    const double exp1 = sl == 0 ? fz[2] : sl == 1 ? fz[3] : sl == 2 ? fr[0] : fr[1];
    const double p1 = sl < 4 ? pow(base1, exp1) : 0.0;
    const double wbz = __shfl(p1, 0, CF_SB_LANES), wmz = __shfl(p1, 1, CF_SB_LANES), wbr = __shfl(p1, 2, CF_SB_LANES),
                 wmr = __shfl(p1, 3, CF_SB_LANES);
    // round 2: lanes 0-4 the z* powers, 5-9 the r_drag powers
    double base2 = 1.0, exp2 = 1.0;
    switch (sl) {
      case 0: base2 = wmz; exp2 = fz[4]; break;
      case 1: base2 = wbz; exp2 = fz[6]; break;
      case 2: base2 = wmz; exp2 = fz[7]; break;
      case 3: base2 = wmz; exp2 = fz[9]; break;
      case 4: base2 = wbz; exp2 = fz[10]; break;
      case 5: base2 = wbr; exp2 = fr[3]; break;
      case 6: base2 = wbr; exp2 = fr[5]; break;
      case 7: base2 = wmr; exp2 = fr[6]; break;
      case 8: base2 = wmr; exp2 = fr[8]; break;
      case 9: base2 = wmr; exp2 = fr[10]; break;
      default: break;
    }
    const double p2 = sl < 10 ? pow(base2, exp2) : 0.0;
    double q[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) q[k] = __shfl(p2, k, CF_SB_LANES);
    // cmb/data_planck_act_compression.py:94-99
    z_star = q[0] + fz[0] * fz[5] * q[1] * q[2] + fz[1] * fz[8] * q[3] * q[4];
    // cmb/data_planck_act_compression.py:121-124
    const double den = (fr[2] * q[5]) + (fr[4] * q[6] * q[7]) + (fr[7] * q[8]);
    r_d = 1.0 / den - fr[9] / q[9];
  };

  double c_cmb = 0.0, vec[3] = {0.0, 0.0, 0.0};
  auto cmb_block = [&]() {
    VirtualLaneSum<LANES> s_rs, s_dm;
    const double half_a = (1.0 / (1.0 + z_star)) / 2.0, half_z = z_star / 2.0;
    const int gl_iters = (d.n_gl + LANES - 1) / LANES;  // the same for every lane (VirtualLaneSum); a lane past the last node adds 0
    for (int it = 0; it < gl_iters; ++it) {
      const int k_raw = sl + it * LANES, k = k_raw < d.n_gl ? k_raw : d.n_gl - 1;
      const d2 xw = gl_s[k];
      const double gw = k_raw < d.n_gl ? xw.y : 0.0;
      const double a = half_a * xw.x + half_a;
      const double z = div_pos(1.0, a) - 1.0;
      const double Rb = (3.0 / 4.0) * (Ob / d.o_gamma_h2) * a;
      s_rs.push(gw * div_pos(d.c, a * a * H_at_gl_node<MODEL, FDE>(d, wc, z, exp2_s, log_s) * sqrt_pos(3.0 * (1.0 + Rb))));
      s_dm.push(gw * div_pos(d.c, H_at_gl_node<MODEL, FDE>(d, wc, half_z * xw.x + half_z, exp2_s, log_s)));
    }
    const double i_rs = s_rs.total(gl_iters), i_dm = s_dm.total(gl_iters);
    const double rs_star = half_a * i_rs;
    const double DM_star = half_z * i_dm;
    const double Om_h2 = Oc + Ob + d.omnu_h2;
    if (d.cmb_mode == 3) {  // (theta*, wb, wm)  cmb/data_early_lcdm_compression.py:206-207
      vec[0] = rs_star / DM_star; vec[1] = Ob; vec[2] = Om_h2;
    } else {                // (R, lA, wb)       cmb/data_planck_act_compression.py:209-212
      vec[0] = 100 * sqrt(Om_h2) * DM_star / d.c; vec[1] = 3.14159265358979323846 * DM_star / rs_star; vec[2] = Ob;
    }
    double dv[3];
    for (int i = 0; i < 3; ++i) dv[i] = d.cmb_prior[i] - vec[i];
    if (d.cmb_mode == 2) {
      c_cmb = dv[1] * dv[1] * d.cmb_inv_cov[4];  // bao/desi_des5y_bbn_theta_star.py:110-111
    } else {
      for (int j = 0; j < 3; ++j) {
        double t = 0.0;
        for (int i = 0; i < 3; ++i) t += dv[i] * d.cmb_inv_cov[3 * i + j];
        c_cmb += t * dv[j];
      }
    }
  };

  double c_cc = 0.0;
  auto cc_block = [&]() {  // delta @ inv_cov @ delta * f_cc^2
    for (int k = sl; k < d.n_cc; k += CF_SB_LANES) dl[k] = d.cc_h[k] - H_of_z<MODEL, FDE>(d, wc, d.cc_z[k]);
    group_sync();
    c_cc = group_quadratic_form<LANES>(dl, d.cc_inv_cov, d.n_cc, sl);
    const double f = slot_get(d, CF_P_FCC_D, th);
    c_cc = d.cc_f_inverse ? c_cc * pow(f, -2.0) : c_cc * (f * f);  // ohd/cc_pantheon.py:64 / bao/desi_union3_cc_theta_star.py:130
    group_sync();  // dl is reused by the BAO block
  };

  // BAO datum k in two parts: the look-ups that need only the walker's table, then the division by r_d and the residual
  auto bao_num = [&](int k, double& z, double& DM, double& DH) {
    NodeView T;
    T.p = bao_nodes + (w * d.n_aux + k) * CF_BAO_NODES;
    T.base = d.bao_base[k];
    T.G = d.n_grid;
    T.step = d.step;
    T.inv_step = d.inv_step;
    T.inv_last = d.inv_last;
    T.z_max = d.z_max;
    z = d.bao_z[k];
    DM = hermite_tab(T, z);
    DH = d.bao_dh_exact ? d.c / H_of_z<MODEL, FDE>(d, wc, z) : pchip_dh_tab(T, z);
  };
  auto bao_fin = [&](int k, double z, double DM, double DH) {
    double t;
    switch (d.bao_qty[k]) {
      case 2: t = DH / r_d; break;
      case 1: t = DM / r_d; break;
      case 0: t = pow(z * DH * (DM * DM), 1.0 / 3) / r_d; break;
      default: t = DM / DH; break;
    }
    dl[k] = d.bao_val[k] - t;
    if (bao_out && live) bao_out[w * d.n_bao + k] = t;
  };

  double c_bao = 0.0;
  if (!SPLIT) {
    if (d.cmb_mode || d.rd_from_fit) pow_block();
    if (!d.rd_from_fit) r_d = slot_get(d, CF_P_RD_D, th);
    if (d.cmb_mode) cmb_block();
    if (d.n_cc > 0) cc_block();
    if (d.n_bao > 0) {
      for (int k = sl; k < d.n_bao; k += CF_SB_LANES) {
        double z, DM, DH;
        bao_num(k, z, DM, DH);
        bao_fin(k, z, DM, DH);
      }
      group_sync();
      c_bao = group_quadratic_form<LANES>(dl, d.bao_inv_cov, d.n_bao, sl);
    }
  } else {
    double bz = 0.0, bDM = 0.0, bDH = 0.0;
    const bool has_datum = sl < d.n_bao;  // CF_MAX_BAO <= LANES: one datum per lane
    if (do_a && (d.cmb_mode || d.rd_from_fit)) pow_block();
    if (!d.rd_from_fit) r_d = slot_get(d, CF_P_RD_D, th);
    if (do_b) {
      if (d.n_cc > 0) cc_block();
      if (has_datum) bao_num(sl, bz, bDM, bDH);
    }
    if (do_a && sl == 0) xch[grp][0] = r_d;
    __syncthreads();  // r_d is in LDS; B's numerators are in its registers
    if (do_a) {
      if (d.cmb_mode) cmb_block();
    } else {
      r_d = xch[grp][0];
      if (d.n_bao > 0) {
        if (has_datum) bao_fin(sl, bz, bDM, bDH);
        group_sync();
        c_bao = group_quadratic_form<LANES>(dl, d.bao_inv_cov, d.n_bao, sl);
      }
      if (sl == 0) {
        xch[grp][1] = c_bao;
        xch[grp][2] = c_cc;
      }
    }
    __syncthreads();  // B's chi^2 are in LDS
    if (do_a) {
      c_bao = xch[grp][1];
      c_cc = xch[grp][2];
    }
  }
  if (sl == 0 && live && do_a) {
    chi2_extra[w] = c_cmb + c_bao + c_cc;
    if (blocks_out) {
      blocks_out[8 * w + 0] = c_bao; blocks_out[8 * w + 1] = c_cmb; blocks_out[8 * w + 5] = c_cc;
      if (d.cmb_mode) { blocks_out[8 * w + 2] = vec[0]; blocks_out[8 * w + 3] = vec[1]; blocks_out[8 * w + 4] = vec[2]; }
      blocks_out[8 * w + 6] = z_star;  // 0 where no block needs them
      blocks_out[8 * w + 7] = r_d;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Growth-rate block f sigma_8(z): fs8/fs8.py:26-120, bao/desi_cmb_union3_fs8.py:27-207.  ONE LANE per walker.
//   delta'' = -(3/a + E_a/E) delta' + (3/2) Om delta / (a^5 E^2),   E_a/E = -(dE^2/dz) / (2 a^2 E^2),
//   from a_init with delta = a_init, delta' = 1 to a = 1 (fs8/fs8.py:64-90).  The reference hands this to scipy's adaptive
//   RK45 (rtol 1e-6, atol 1e-8) and reads delta' at the data points by PCHIP from a 1000 / 2500-point log grid; here
//   classical RK4 in x = ln a with a fixed step (y1 = delta, y2 = delta'; dy/dx = a dy/da), and delta'(a_k) by cubic Hermite
//   inside the step that contains ln a_k (the slope dy2/dx is the first stage of the next step): error ~1e-9, far below
//   the reference's own ~1e-6 (tests/test_fs8.py states the resulting parity bar).
//   theory_k = (sigma8 / delta(1)) a_k delta'(a_k)                                        fs8/fs8.py:84-98
//   q_k = H(z_k) D_M(z_k) / (H D_M)_fid,k ;  Delta = val - theory / q ;  chi2 = f_err^2 Delta^T inv_cov Delta    :111-120
// ------------------------------------------------------------------------------------------------
template <int FDE>
__device__ __forceinline__ double dfde_dz(const WalkerCosmo& wc, double z, double zp1, double cubed, double f) {
  if (FDE == CF_FDE_LCDM_D) return 0.0;
  if (FDE == CF_FDE_WCDM_D) return f * 3 * (1.0 + wc.w0) / zp1;
  if (FDE == CF_FDE_THAWING_D) {  // fs8/fs8.py:26-41: Ode_z * 3 (1 + w_de(z)) / (1 + z)
    const double w = -1.0 + 2 * (1.0 + wc.w0) / ((1.0 + wc.w0) + (1.0 - wc.w0) * cubed);
    return f * 3 * (1.0 + w) / zp1;
  }
  return f * 3 * (1.0 + wc.w0 + wc.wa * z / zp1) / zp1;  // CPL: w(z) = w0 + wa z / (1 + z)
}

// massive-neutrino equation of state, cmb/data_planck_act_compression.py:70-83
__device__ __forceinline__ double w_nu_z(const cf_dev_desc& d, double zp1) {
  const double r = d.nu_m0 / zp1, mz_sq = r * r;
  double num = 0.0, den = 0.0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const double f = sqrt(d.nu_qs_sq[i] + mz_sq);
    num += d.nu_ws[i] / f;
    den += d.nu_ws[i] * f;
  }
  return (1.0 / 3) - (1.0 / 3) * mz_sq * num / den;
}

// {E^2, dE^2/dz} at z                       fs8/fs8.py:44-56, bao/desi_cmb_union3_fs8.py:46-66,127-140
template <int MODEL, int FDE>
__device__ __forceinline__ void e2_and_slope(const cf_dev_desc& d, const WalkerCosmo& wc, double z, double& e2, double& de2) {
  const double zp1 = 1.0 + z, sq = zp1 * zp1, cubed = sq * zp1;
  const double f = (FDE == CF_FDE_LCDM_D) ? 1.0 : f_de<FDE>(wc, z, zp1, cubed);
  const double df = dfde_dz<FDE>(wc, z, zp1, cubed, f);
  if (MODEL == CF_EZ_LATE_FLAT_D) {
    e2 = wc.Om * cubed + (1.0 - wc.Om) * f;
    de2 = 3 * wc.Om * sq + (1.0 - wc.Om) * df;
  } else {
    const double nu = omnu_z(d, zp1);
    e2 = wc.Or * (cubed * zp1) + wc.Obc * cubed + wc.Ode * f + wc.Onu * nu;
    de2 = 3 * wc.Obc * sq + 4 * wc.Or * cubed + wc.Onu * (nu * 3 * (1.0 + w_nu_z(d, zp1)) / zp1) + wc.Ode * df;
  }
}

// The ODE is LINEAR in y = (delta, delta'):  dy/dx = A(x) y,  A = [[0, a], [s, -p]],  s = (3/2) Om / (a^4 E^2),
// p = 3 - (dE^2/dz) / (2 a E^2)  (x = ln a).  One classical RK4 step is therefore a 2 x 2 matrix,
//   M = I + (h/6) (K1 + 2 K2 + 2 K3 + K4),  K1 = A0,  K2 = Ah (I + h/2 K1),  K3 = Ah (I + h/2 K2),  K4 = A1 (I + h K3),
// the same arithmetic as the vector form up to the association of the products, and the 2048 dependent right-hand sides of
// the step-by-step integration become: the coefficients at all 2 S + 1 points in parallel (C steps per lane, 256 lanes per
// walker), the C step matrices of a lane multiplied together, ONE scan of 2 x 2 products over the 256 lanes, and a C-step
// re-walk from the lane's own start value that leaves (delta', d delta'/dx) at every step boundary in LDS for the Hermite
// read-out at the data points.  Everything that does not depend on theta -- a = exp(x), 1 + z, the massive-neutrino density and
// its slope -- is tabulated at cf_create (fs8_tab).
struct M22 {
  double a, b, c, d;  // [[a, b], [c, d]]
};
__device__ __forceinline__ M22 mm(const M22& L, const M22& R) {
  return {L.a * R.a + L.b * R.c, L.a * R.b + L.b * R.d, L.c * R.a + L.d * R.c, L.c * R.b + L.d * R.d};
}
__device__ __forceinline__ M22 shfl_up_m(const M22& m, int delta) {
  return {__shfl_up(m.a, delta), __shfl_up(m.b, delta), __shfl_up(m.c, delta), __shfl_up(m.d, delta)};
}

// A(x) at the table point stored at index m of fs8_tab (see growth_kernel for the order)
template <int MODEL, int FDE>
__device__ __forceinline__ void growth_coef(const cf_dev_desc& d, const WalkerCosmo& wc, double om, int m, double lnzp1, double& a,
                                            double& s, double& p) {
  const d4 t = reinterpret_cast<const d4*>(d.fs8_tab)[m];  // {a, 1 + z, nu(z), nu 3 (1 + w_nu) / (1 + z)}
  a = t[0];
  const double zp1 = t[1], z = zp1 - 1.0, sq = zp1 * zp1, cubed = sq * zp1;
  // dark-energy density ratio f and df/dz = f 3 (1 + w(z)) / (1 + z), with 1 / (1 + z) = a and one reciprocal shared
  // (fs8/fs8.py:26-41; the forms of f_de / dfde_dz above)
  double f = 1.0, df = 0.0;
  if (FDE == CF_FDE_WCDM_D) {
    f = exp(3 * (1 + wc.w0) * lnzp1);
    df = f * 3 * (1.0 + wc.w0) * a;
  } else if (FDE == CF_FDE_THAWING_D) {
    const double inv = 1.0 / ((1.0 + wc.w0) + (1.0 - wc.w0) * cubed), r = 2 * cubed * inv;
    f = r * r;
    df = f * 3 * (2 * (1.0 + wc.w0) * inv) * a;  // 1 + w(z) = 2 (1 + w0) / ((1 + w0) + (1 - w0) (1 + z)^3)
  } else if (FDE == CF_FDE_CPL_D) {
    const double za = z * a;
    f = exp(fma(3 * (1 + wc.w0 + wc.wa), lnzp1, -3 * wc.wa * za));
    df = f * 3 * (1.0 + wc.w0 + wc.wa * za) * a;
  }
  double e2, de2;
  if (MODEL == CF_EZ_LATE_FLAT_D) {  // fs8/fs8.py:44-56
    e2 = wc.Om * cubed + (1.0 - wc.Om) * f;
    de2 = 3 * wc.Om * sq + (1.0 - wc.Om) * df;
  } else {  // bao/desi_cmb_union3_fs8.py:46-66,127-140
    e2 = wc.Or * (cubed * zp1) + wc.Obc * cubed + wc.Ode * f + wc.Onu * t[2];
    de2 = 3 * wc.Obc * sq + 4 * wc.Or * cubed + wc.Onu * t[3] + wc.Ode * df;
  }
  const double r = zp1 / e2;  // 1 / (a E^2)
  s = 1.5 * om * r * cubed;   // (3/2) Om / (a^4 E^2)
  p = 3.0 - 0.5 * de2 * r;
}

// interp_pchip (interpolator.py:5-108, exact=False) on a window of four consecutive nodes x0 < x1 < x2 < x3 of a longer grid:
// the query lies in the interval [x_li, x_{li+1}] (li = 0 only when the window starts at the grid's first node -- flags & 1 --
// and li = 2 only when it ends at the last -- flags & 2), so the two Fritsch-Carlson slopes it needs are inside the window:
// interior nodes by the weighted harmonic mean of the neighbouring secants (:25-40), the grid's end nodes by the three-point
// formula with its sign / overshoot guards (:41-66).
__device__ __forceinline__ double pchip_window4(double x0, double x1, double x2, double x3, double y0, double y1, double y2, double y3,
                                                int li, int flags, double xq) {
  const double x[4] = {x0, x1, x2, x3}, y[4] = {y0, y1, y2, y3};
  auto interior = [&](int i) {
    const double hl = x[i] - x[i - 1], hr = x[i + 1] - x[i];
    const double dl = (y[i] - y[i - 1]) / hl, dr = (y[i + 1] - y[i]) / hr;
    if (dl != 0.0 && dr != 0.0 && dl * dr > 0.0) {
      const double w1 = 2.0 * hr + hl, w2 = hr + 2.0 * hl;
      return (w1 + w2) / (w1 / dl + w2 / dr);
    }
    return 0.0;
  };
  auto end_point = [&](int i0, int i1, int i2) {  // slope at node i0 from (i0, i1, i2), i0 an end of the GRID
    const double h0 = fabs(x[i1] - x[i0]), h1 = fabs(x[i2] - x[i1]);
    const double d0 = (y[i1] - y[i0]) / (x[i1] - x[i0]), d1 = (y[i2] - y[i1]) / (x[i2] - x[i1]);
    const double e = ((2 * h0 + h1) * d0 - h0 * d1) / (h0 + h1);
    if (d0 == 0.0 || sgn_d(e) != sgn_d(d0)) return 0.0;
    if (sgn_d(d0) != sgn_d(d1) && fabs(e) > fabs(3 * d0)) return 3 * d0;
    return e;
  };
  li = li < 0 ? 0 : (li > 2 ? 2 : li);
  const double s_lo = li == 0 ? ((flags & 1) ? end_point(0, 1, 2) : interior(1)) : interior(li);
  const double s_hi = li == 2 ? ((flags & 2) ? end_point(3, 2, 1) : interior(2)) : interior(li + 1);
  const double xa = x[li], h = x[li + 1] - xa;
  if (xq <= x[0] && (flags & 1)) return y[0];  // clamped outside the grid (:80-85)
  if (xq >= x[3] && (flags & 2)) return y[3];
  const double t = (xq - xa) / h, t2 = t * t, t3 = t2 * t;
  const double h00 = 2 * t3 - 3 * t2 + 1, h10 = t3 - 2 * t2 + t, h01 = -2 * t3 + 3 * t2, h11 = t3 - t2;
  return h00 * y[li] + h10 * h * s_lo + h01 * y[li + 1] + h11 * h * s_hi;
}

#define CF_GROWTH_TPB 256
template <int MODEL, int FDE, int C>
__global__ void __launch_bounds__(CF_GROWTH_TPB)
growth_kernel(cf_dev_desc d, const double* __restrict__ theta, int64_t W, const d2* __restrict__ aux_nodes,
              double* __restrict__ chi2_extra, int accumulate, double* __restrict__ blocks_out, double* __restrict__ theory_out) {
  extern __shared__ double growth_lds[];
  constexpr int S = C * CF_GROWTH_TPB;
  d2* bnd = reinterpret_cast<d2*>(growth_lds);                   // [S + 1] {delta', d delta'/dx} at the step boundaries
  M22* wave_tot = reinterpret_cast<M22*>(growth_lds + 2 * (S + 1));  // [4] product of each wave's lanes
  double* resid = growth_lds + 2 * (S + 1) + 16;                  // [CF_MAX_FS8] residuals; [CF_MAX_FS8] = delta(a = 1)
  double* part = resid + CF_MAX_FS8 + 2;                          // [4][64] partial column sums of the quadratic form
  const int64_t w = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double* th = theta + w * d.ndim;
  const WalkerCosmo wc = make_cosmo(d, th);
  const double om = MODEL == CF_EZ_LATE_FLAT_D ? wc.Om : wc.Obc;
  const int n = d.n_fs8;
  const double x0 = log(d.fs8_a_init), h = -x0 / S, hh = 0.5 * h;

  // ---- the C step matrices of this lane and their product ----
  M22 M[C];
  double s_at[C], p_at[C];  // A at the start boundary of each step: d delta'/dx = s delta - p delta'
  double a0, s0, p0, s_end, p_end;
  // table point m (boundary / midpoint number) lives at [(m % 2C) * 257 + m / 2C]: lane t reads its j-th point from row j at
  // column t -- a wave's load is contiguous (from [m] every lane would touch its own cache line); cf_create stores it so
  auto sw = [](int m) { return (m % (2 * C)) * 257 + m / (2 * C); };
  growth_coef<MODEL, FDE>(d, wc, om, sw(2 * C * tid), -(x0 + (2 * C * tid) * hh), a0, s0, p0);
  M22 Q{1.0, 0.0, 0.0, 1.0};
#pragma unroll
  for (int j = 0; j < C; ++j) {
    const int m = 2 * (C * tid + j);
    double am, sm, pm, a1, s1, p1;
    growth_coef<MODEL, FDE>(d, wc, om, sw(m + 1), -(x0 + (m + 1) * hh), am, sm, pm);
    growth_coef<MODEL, FDE>(d, wc, om, sw(m + 2), m + 2 == 2 * S ? 0.0 : -(x0 + (m + 2) * hh), a1, s1, p1);
    const M22 A0{0.0, a0, s0, -p0}, Ah{0.0, am, sm, -pm}, A1{0.0, a1, s1, -p1};
    const M22 K2 = mm(Ah, M22{1.0, hh * A0.b, hh * A0.c, 1.0 + hh * A0.d});
    const M22 K3 = mm(Ah, M22{1.0 + hh * K2.a, hh * K2.b, hh * K2.c, 1.0 + hh * K2.d});
    const M22 K4 = mm(A1, M22{1.0 + h * K3.a, h * K3.b, h * K3.c, 1.0 + h * K3.d});
    const double h6 = h / 6;
    M[j] = M22{1.0 + h6 * (2 * K2.a + 2 * K3.a + K4.a), h6 * (A0.b + 2 * K2.b + 2 * K3.b + K4.b),
               h6 * (A0.c + 2 * K2.c + 2 * K3.c + K4.c), 1.0 + h6 * (A0.d + 2 * K2.d + 2 * K3.d + K4.d)};
    s_at[j] = s0;
    p_at[j] = p0;
    Q = mm(M[j], Q);
    a0 = a1; s0 = s1; p0 = p1;
  }
  s_end = s0;
  p_end = p0;

  // ---- inclusive scan of the lane products (later steps on the left), over the wave, then over the four waves ----
  M22 P = Q;
#pragma unroll
  for (int dlt = 1; dlt < 64; dlt <<= 1) {
    const M22 R = shfl_up_m(P, dlt);
    if (lane >= dlt) P = mm(P, R);
  }
  if (lane == 63) wave_tot[wave] = P;
  __syncthreads();
  M22 E = shfl_up_m(P, 1);  // exclusive prefix inside the wave
  if (lane == 0) E = M22{1.0, 0.0, 0.0, 1.0};
  for (int v = wave - 1; v >= 0; --v) E = mm(E, wave_tot[v]);  // E (this wave's earlier lanes) x waves wave-1 ... 0

  // ---- re-walk the lane's own steps from its start value: boundary values into LDS ----
  double y1 = E.a * d.fs8_a_init + E.b, y2 = E.c * d.fs8_a_init + E.d;  // y(0) = (a_init, 1), fs8/fs8.py:79-82
#pragma unroll
  for (int j = 0; j < C; ++j) {
    bnd[C * tid + j] = d2{y2, s_at[j] * y1 - p_at[j] * y2};
    const double n1 = M[j].a * y1 + M[j].b * y2, n2 = M[j].c * y1 + M[j].d * y2;
    y1 = n1;
    y2 = n2;
  }
  if (tid == CF_GROWTH_TPB - 1) {
    bnd[S] = d2{y2, s_end * y1 - p_end * y2};
    resid[CF_MAX_FS8] = y1;  // delta(a = 1)
  }
  __syncthreads();

  // ---- data points, one lane each (wave 0): delta'(a_k) by cubic Hermite inside its step (weights tabulated at cf_create),
  //      theory, Alcock-Paczynski factor, residual ----
  const double delta0 = resid[CF_MAX_FS8], s8 = slot_get(d, CF_P_S8_D, th), ferr = slot_get(d, CF_P_FS8ERR_D, th);
  if (wave == 0 && lane < n) {
    const int k = d.fs8_order[lane], i = d.fs8_step_of[lane];
    const d4* pt = reinterpret_cast<const d4*>(d.fs8_pts) + 8 * lane;  // the datum's record, see fs8_points (cosmofit_api.hip)
    const d4 hw = pt[0], aux = pt[1];  // Hermite weights {h00, h h10, h01, h h11}; {a_k, nu(z_k), li, window flags}
    auto read_out = [&](const d4& wgt, int step) {
      const d2 b0 = bnd[step], b1 = bnd[step + 1];
      return wgt[0] * b0[0] + wgt[1] * b0[1] + wgt[2] * b1[0] + wgt[3] * b1[1];
    };
    double dprime;
    if (d.fs8_n_agrid > 0) {
      // as the scripts: delta' sampled at the nodes of their logarithmic a-grid, then interp_pchip at a_k (fs8/fs8.py:79-98 ->
      // interpolator.py:5-108) -- the four nodes around a_k are all the interpolant looks at
      const d4 xs = pt[6], st = pt[7];
      const double y0 = read_out(pt[2], (int)st[0]), y1 = read_out(pt[3], (int)st[1]), y2 = read_out(pt[4], (int)st[2]),
                   y3 = read_out(pt[5], (int)st[3]);
      dprime = pchip_window4(xs[0], xs[1], xs[2], xs[3], y0, y1, y2, y3, (int)aux[2], (int)aux[3], aux[0]);
    } else {
      dprime = read_out(hw, i);
    }
    NodeView T;
    T.p = aux_nodes + (w * d.n_aux + d.n_bao + k) * CF_BAO_NODES;
    T.base = d.bao_base[d.n_bao + k];
    T.G = d.n_grid; T.step = d.step; T.inv_step = d.inv_step; T.inv_last = d.inv_last; T.z_max = d.z_max;
    const double z = d.fs8_z[k];
    const double theory = (s8 / delta0) * aux[0] * dprime;
    const double Hz = wc.H0 * sqrt(e2_of_z<MODEL, FDE>(d, wc, z, MODEL == CF_EZ_PHYSICAL_D ? aux[1] : -1.0));
    const double q = Hz * hermite_tab(T, z) / d.fs8_fid[k];
    if (theory_out) theory_out[w * n + k] = theory;
    resid[k] = d.fs8_val[k] - theory / q;
  }
  __syncthreads();
  // ---- delta @ inv_cov @ delta: thread (j = lane, quarter = wave) sums the rows i = wave (mod 4) of column j ----
  double tj = 0.0;
  if (lane < n) {
    const double* col = d.fs8_inv_cov + lane;
#pragma unroll 4
    for (int i = wave; i < n; i += 4) tj += resid[i] * col[i * n];
  }
  part[wave * 64 + lane] = tj;
  __syncthreads();
  if (wave != 0) return;
  double r_own = lane < n ? (((part[lane] + part[64 + lane]) + part[128 + lane]) + part[192 + lane]) * resid[lane] : 0.0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) r_own += __shfl_xor(r_own, off);
  if (lane == 0) {
    const double c = r_own * (ferr * ferr);
    chi2_extra[w] = accumulate ? chi2_extra[w] + c : c;
    if (blocks_out) blocks_out[w] = c;
  }
}

// H(z) of one theta at arbitrary redshifts: the H_z(z, params) the post-fit blocks plot (ohd/cc.py:95-96, ohd/plot_predictions.py:7-21)
template <int MODEL, int FDE>
__global__ void hz_kernel(cf_dev_desc d, const double* __restrict__ theta, const double* __restrict__ z, int64_t n,
                          double* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const WalkerCosmo wc = make_cosmo(d, theta);
  out[i] = H_of_z<MODEL, FDE>(d, wc, z[i]);
}

#define CF_INSTANTIATE_GROWTH(M, F, C) \
  template __global__ void growth_kernel<M, F, C>(cf_dev_desc, const double*, int64_t, const d2*, double*, int, double*, double*);
#define CF_INSTANTIATE_WALKER(M, F)                                                                              \
  template __global__ void walker_kernel<M, F>(cf_dev_desc, const double*, int64_t, double*, double*, double*, d2*, d2*); \
  template __global__ void walker_fast_kernel<M, F>(cf_walker_args, const double*, int64_t, double*, d2*, double*, int, int);     \
  template __global__ void small_blocks_kernel<M, F, 16, 1>(cf_dev_desc, const double*, int64_t, const d2*, double*, double*, \
                                                            double*);                                                      \
  template __global__ void small_blocks_kernel<M, F, 64, 1>(cf_dev_desc, const double*, int64_t, const d2*, double*, double*, \
                                                            double*);                                                      \
  template __global__ void small_blocks_kernel<M, F, 64, 2>(cf_dev_desc, const double*, int64_t, const d2*, double*, double*, \
                                                            double*);                                                      \
  template __global__ void hz_kernel<M, F>(cf_dev_desc, const double*, const double*, int64_t, double*);                     \
  CF_INSTANTIATE_GROWTH(M, F, 1) CF_INSTANTIATE_GROWTH(M, F, 2) CF_INSTANTIATE_GROWTH(M, F, 4) CF_INSTANTIATE_GROWTH(M, F, 8)
CF_INSTANTIATE_WALKER(0, 0) CF_INSTANTIATE_WALKER(0, 1) CF_INSTANTIATE_WALKER(0, 2) CF_INSTANTIATE_WALKER(0, 3)
CF_INSTANTIATE_WALKER(1, 0) CF_INSTANTIATE_WALKER(1, 1) CF_INSTANTIATE_WALKER(1, 2) CF_INSTANTIATE_WALKER(1, 3)

// ------------------------------------------------------------------------------------------------
// Prior / output epilogue shared by every likelihood form.   sn/pantheon.py:80-97
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double epi_slot(const cf_dev_slot& sl, const double* __restrict__ th) {
  return sl.idx >= 0 ? th[sl.idx] * sl.scale : sl.fixed;
}
__device__ __forceinline__ double finalize_value(const cf_epilogue& d, const double* __restrict__ th, double chi2, int out_kind,
                                                 unsigned long long* nonfinite) {
  for (int g = 0; g < d.n_chi2_gauss; ++g) {
    double diff = th[d.chi2_gauss_idx[g]] - d.chi2_gauss_mean[g];
    chi2 += diff * diff / (d.chi2_gauss_sigma[g] * d.chi2_gauss_sigma[g]);
  }
  if (out_kind == CF_OUT_CHI2_D) return chi2;
  double lp = 0.0;
  if (out_kind == CF_OUT_LOGP_D) {
    if (d.has_bounds) {
      // four parameters' reads issued together, no short-circuit: with `&&` every parameter was a dependent load + branch in the
      // last arriver of the solve, the one workgroup the evaluation's latency waits for (all 16 at once cost 96 VGPRs)
      bool inbox = true;
      for (int k0 = 0; k0 < d.ndim; k0 += 4) {
#pragma unroll
        for (int k = k0; k < k0 + 4; ++k) {
          const double t = th[k < d.ndim ? k : 0];
          inbox = inbox & ((k >= d.ndim) | ((d.lo[k & (CF_MAX_NDIM - 1)] < t) & (t < d.hi[k & (CF_MAX_NDIM - 1)])));
        }
      }
      if (!inbox) return -INFINITY;  // the likelihood is not consulted outside the box, sn/pantheon.py:90-92
      lp = d.log_norm;
    }
    for (int g = 0; g < d.n_gauss; ++g) {
      double diff = th[d.gauss_idx[g]] - d.gauss_mean[g];
      lp = lp - 0.5 * (diff * diff) / (d.gauss_sigma[g] * d.gauss_sigma[g]);
    }
  }
  // hard wall of the CPL scripts, part of log L itself: bao/desi_fs_lya_cmb.py:118-121
  if (d.cpl_wall && epi_slot(d.w0, th) + epi_slot(d.wa, th) >= 0.0) return lp + -1e8;
  if (!isfinite(chi2)) {  // emcee aborts on NaN: map to -inf and count it
    atomicAdd(nonfinite, 1ull);
    return -INFINITY;
  }
  double ll = -0.5 * chi2 + d.logl_const;
  if (d.n_fs8 > 0)  // -0.5 (chi2 - 2 N ln f_err), fs8/fs8.py:123-125 (f_err fixed to 1 where a script has none)
    ll += d.n_fs8 * log(epi_slot(d.fs8err, th));
  if (d.n_cc > 0)  // Gaussian normalisation with rescaled errors, bao/desi_union3_cc_theta_star.py:135-139
    ll -= 0.5 * (d.n_cc * 1.8378770664093453 + d.cc_logdet +
                 (d.cc_f_inverse ? 2 : -2) * d.n_cc * log(epi_slot(d.fcc, th)));  // ohd/cc_pantheon.py:92
  return lp + ll;
}

// ------------------------------------------------------------------------------------------------
// Kernel B: blocked triangular solve + chi^2 on FP64 MFMA, one workgroup per 16-walker panel.
//
// v_mfma_f64_16x16x4_f64 operand maps (lane l):  A[i = l&15][k = l>>4],  B[k = l>>4][n = l&15],
// C/D register r: row = (l>>4) + 4r, col = l&15.  Hence register r of a solved 16x16 tile IS the
// B fragment of K-step r of that tile: Y tiles go back into the product with no lane movement.
//
// A panel is one dependent chain of block rows, so it stays on one CU; the workgroup runs NW = 4*KS
// waves (KS = 2: two per SIMD, the MFMA pipe takes one instruction per 64 cycles and a single wave's
// loads and LDS round trips leave it idle):
// in the update phase wave (wq = wave&3, g = wave>>2) owns the tiles wq, wq+4, wq+8, wq+12 of the
// 256-row block row and the K range [g, g+1) * (r0/KS); the KS partial right-hand sides meet in
// LDS, and the diagonal phase (rhs times the pre-inverted diagonal block) is spread over all NW
// waves by tile pairs (v, 2NW-1-v) of equal triangular work.
//
// Packed factor (built once on the host, cf_pack.h), two K-steps (8 columns) per 16-byte lane
// element, in the order the wave consumes it -- every load is one coalesced 1 KiB access:
//   update stream (b, wq, g): [q < 32b/KS][slot j < nt][lane] -> {-L[row][8 s2 + k], -L[row][8 s2 + 4 + k]}
//                             row = 16 (16b + wq + 4j) + (l&15), k = l>>4, s2 = g*32b/KS + q
//   diag stream   (b, v)    : [sl2 <= 2 mlmax + 1][slot j < 16/NW][lane] -> the same for inv(L_bb),
//                             tile of slot j = cf_diag_tile(NW, v, j)
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

template <int NT, int PF, int NTU>
__device__ __forceinline__ void update_loop(d4 (&acc)[NTU], const d2* __restrict__ A, const d2* __restrict__ Yp,
                                            int n_s2, int lane) {
  // Software pipeline of depth PF over K-step pairs (n_s2 is a multiple of PF).  Branch-free body:
  // stage p is consumed by its 2*NT MFMAs and then immediately refilled with the fragments of step
  // s2+p+PF.  The refill of the last PF steps reads past the wave's range (the next stream / Y
  // fragments that are not needed) -- both buffers carry slack for that, and the values are unused.
  d2 a[PF][NT];
  d2 yb[PF];
  const d2* Ap = A + lane;
  const d2* Yq = Yp + lane;
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int p = 0; p < PF; ++p) {
#pragma unroll
    for (int j = 0; j < NT; ++j) a[p][j] = Ap[(p * NT + j) * 64];
    yb[p] = Yq[p * 64];
    __builtin_amdgcn_sched_barrier(0);  // keep the fill in stage order: stage 0 must be the OLDEST load
  }
  Ap += PF * NT * 64;
  Yq += PF * 64;
  for (int s2 = 0; s2 < n_s2; s2 += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = mfma_f64(a[p][j].x, yb[p].x, acc[j]);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = mfma_f64(a[p][j].y, yb[p].y, acc[j]);
      // hipcc's scheduler otherwise hoists every refill to the top of the iteration and drains
      // them with vmcnt(0) at its end, which exposes one full memory latency per iteration
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; ++j) a[p][j] = Ap[(p * NT + j) * 64];
      yb[p] = Yq[p * 64];
      __builtin_amdgcn_sched_barrier(0);
    }
    Ap += PF * NT * 64;
    Yq += PF * 64;
  }
}


template <int KS, int TC>
__global__ void __launch_bounds__(64 * KS * TC)
trsm_chi2_kernel(const cf_epilogue* __restrict__ epi, int n_pad, int n_ld, int ndim, cf_dev_pack pk, const double* __restrict__ theta,
                 int64_t W, const double* __restrict__ delta, d2* __restrict__ ypk, const double* __restrict__ chi2_extra,
                 double* __restrict__ out, int out_kind, unsigned long long* nonfinite, double* __restrict__ chi2_sn_out) {
  constexpr int NW = TC * KS;                  // waves per workgroup
  constexpr int NTU = CF_BLOCK_TILES / TC;     // tiles per wave in the update phase
  constexpr int NTD = CF_BLOCK_TILES / NW;     // tiles per wave in the diagonal phase
  constexpr int CF_DIAG_PREFETCH = NW == 16 ? 2 : 4;
  constexpr int PANEL_FRAGS = CF_BLOCK_ROWS / 8 * 64;  // one 256x16 block as B fragments: 2048 d2 = 32 KB
  extern __shared__ __align__(16) d2 ldsP[];            // [KS][PANEL_FRAGS] partial right-hand sides
  __shared__ double chi_part[NW][16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wq = wave % TC, g = wave / TC;
  const int col = lane & 15, kq = lane >> 4;
  const int64_t panel = blockIdx.x;
  const int64_t w0 = panel * 16;
  const int T = n_pad / 16;
  d2* Yp = ypk + panel * (int64_t)(n_pad / 8) * 64;
  const double* dcol = delta + (w0 + col) * (int64_t)n_ld;
  double chi = 0.0;

  for (int b = 0; b < pk.n_blocks; ++b) {
    const int tiles_b = min(CF_BLOCK_TILES, T - b * CF_BLOCK_TILES);
    const int nt = tiles_b > wq ? (tiles_b - wq + TC - 1) / TC : 0;
    const int r0 = b * CF_BLOCK_ROWS;
    // diagonal-phase stream of this wave: start its first loads now, they land during the update
    int ml[NTD], ml_max = -1;
#pragma unroll
    for (int j = 0; j < NTD; ++j) {
      const int t = cf_diag_tile(NW, wave, j);
      ml[j] = (t >= 0 && t < tiles_b) ? t : -1;
      ml_max = max(ml_max, ml[j]);
    }
    const d2* D = pk.frags + pk.diag_off[b * NW + wave] * 64;
    d2 da[CF_DIAG_PREFETCH][NTD];
    if (ml_max >= 0) {
#pragma unroll
      for (int p = 0; p < CF_DIAG_PREFETCH; ++p)
#pragma unroll
        for (int j = 0; j < NTD; ++j) da[p][j] = D[((int64_t)p * NTD + j) * 64 + lane];
    }
    // residual tile in C layout (row = 4r + kq inside the tile); K-split groups > 0 start from zero
    d4 acc[NTU];
#pragma unroll
    for (int j = 0; j < NTU; ++j) {
      acc[j] = (d4){0.0, 0.0, 0.0, 0.0};
      if (j < nt && g == 0) {
        const double* p = dcol + r0 + 16 * (wq + TC * j) + kq;
        acc[j] = (d4){p[0], p[4], p[8], p[12]};
      }
    }
    // ---- off-diagonal update: acc += (-L[b, K range of g]) * Y[K range of g] ----
    if (b > 0 && nt > 0) {
      const d2* A = pk.frags + pk.upd_off[(b * TC + wq) * KS + g] * 64;
      const int n_s2 = r0 / 8 / KS;
      const d2* Yg = Yp + (int64_t)g * n_s2 * 64;
      constexpr int PF = (NW == 16 && NTU == 4) ? 2 : 4;  // 16 waves leave 128 VGPRs per lane
      if constexpr (NTU == 4) {
        switch (nt) {
          case 4: update_loop<4, PF, NTU>(acc, A, Yg, n_s2, lane); break;
          case 3: update_loop<3, PF, NTU>(acc, A, Yg, n_s2, lane); break;
          case 2: update_loop<2, PF, NTU>(acc, A, Yg, n_s2, lane); break;
          default: update_loop<1, PF, NTU>(acc, A, Yg, n_s2, lane); break;
        }
      } else {
        if (nt == 2) update_loop<2, PF, NTU>(acc, A, Yg, n_s2, lane);
        else update_loop<1, PF, NTU>(acc, A, Yg, n_s2, lane);
      }
    }
    // ---- publish this wave's partial right-hand side as B fragments in LDS ----
#pragma unroll
    for (int j = 0; j < NTU; ++j)
      if (j < nt) {
        const int t = wq + TC * j;
        ldsP[g * PANEL_FRAGS + (2 * t) * 64 + lane] = (d2){acc[j][0], acc[j][1]};
        ldsP[g * PANEL_FRAGS + (2 * t + 1) * 64 + lane] = (d2){acc[j][2], acc[j][3]};
      }
    __syncthreads();
    // ---- diagonal block through its inverse: y = inv(L_bb) * rhs (lower triangular) ----
    if (ml_max >= 0) {
      // two accumulation chains per tile (even / odd K-steps): the dependent-MFMA latency, not the
      // issue rate, bounds this short loop
      d4 y[NTD], y2[NTD];
#pragma unroll
      for (int j = 0; j < NTD; ++j) y[j] = y2[j] = (d4){0.0, 0.0, 0.0, 0.0};
      const int n_sl2 = 2 * ml_max + 2;
      for (int s0 = 0; s0 < n_sl2; s0 += CF_DIAG_PREFETCH) {
#pragma unroll
        for (int p = 0; p < CF_DIAG_PREFETCH; ++p) {
          const int sl2 = s0 + p;
          if (sl2 < n_sl2) {  // wave-uniform
            d2 tb = ldsP[sl2 * 64 + lane];
#pragma unroll
            for (int k = 1; k < KS; ++k) {
              const d2 o = ldsP[k * PANEL_FRAGS + sl2 * 64 + lane];
              tb.x += o.x;
              tb.y += o.y;
            }
            d2 ca[NTD];
#pragma unroll
            for (int j = 0; j < NTD; ++j) ca[j] = da[p][j];
            // the packed stream carries 8*NTD fragments of slack, so this prefetch never leaves it
#pragma unroll
            for (int j = 0; j < NTD; ++j) da[p][j] = D[((int64_t)(sl2 + CF_DIAG_PREFETCH) * NTD + j) * 64 + lane];
#pragma unroll
            for (int j = 0; j < NTD; ++j)
              if (sl2 <= 2 * ml[j] + 1) {
                y[j] = mfma_f64(ca[j].x, tb.x, y[j]);
                y2[j] = mfma_f64(ca[j].y, tb.y, y2[j]);
              }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NTD; ++j) y[j] += y2[j];
#pragma unroll
      for (int j = 0; j < NTD; ++j)
        if (ml[j] >= 0) {
          chi += y[j][0] * y[j][0] + y[j][1] * y[j][1] + y[j][2] * y[j][2] + y[j][3] * y[j][3];
          const int mt = b * CF_BLOCK_TILES + ml[j];
          Yp[(int64_t)(2 * mt) * 64 + lane] = (d2){y[j][0], y[j][1]};
          Yp[(int64_t)(2 * mt + 1) * 64 + lane] = (d2){y[j][2], y[j][3]};
        }
    }
    __syncthreads();  // Y of this block visible to the whole workgroup; ldsP reusable
  }

  // ---- chi^2 per walker column: over the 4 row groups of a wave, then over the NW waves ----
  chi += __shfl_xor(chi, 16, CF_WAVE);
  chi += __shfl_xor(chi, 32, CF_WAVE);
  if (lane < 16) chi_part[wave][lane] = chi;
  __syncthreads();
  if (tid < 16 && w0 + tid < W) {
    const int64_t w = w0 + tid;
    double c2 = 0.0;
#pragma unroll
    for (int k = 0; k < NW; ++k) c2 += chi_part[k][tid];
    if (chi2_sn_out) chi2_sn_out[w] = c2;  // the SN block alone (cf_eval_parts)
    if (chi2_extra) c2 += chi2_extra[w];
    out[w] = finalize_value(*epi, theta + w * ndim, c2, out_kind, nonfinite);  // scalar loads from the handle's device copy
  }
}

#define CF_INSTANTIATE_TRSM(KS, TC)                                                                                          \
  template __global__ void trsm_chi2_kernel<KS, TC>(const cf_epilogue*, int, int, int, cf_dev_pack, const double*, int64_t, \
                                                    const double*, d2*, const double*, double*, int, unsigned long long*, double*);
CF_INSTANTIATE_TRSM(2, 4)  // the shipped shape: 8 waves, two per SIMD

// ------------------------------------------------------------------------------------------------
// Inverse-GEMM solve: Y = X Delta with X = L^-1 inverted once on the host (cf_pack.h), a triangular
// GEMM with no dependency between row blocks.  One 256-thread workgroup per UNIT = (64-row block rb, panel of 16*NP walkers): the
// four waves split the K range [0, 64 (rb+1)), each keeps the 4 x NP accumulator tiles of the block in registers (an A fragment
// feeds NP MFMAs), the quarters meet in LDS, and the unit's share of chi^2 goes to partial[rb][walker].  The workgroup that
// arrives last for a panel adds the shares in a fixed order and applies the prior / output epilogue -> one launch, and results do
// not depend on timing.  B fragments are 16-byte loads straight from the row-major residual rows.
//
// SCHEDULING: a plain grid, blockIdx -> unit in the order panel group (its residual rows stay in the Infinity Cache while the
// group's row blocks pass) > row block, LARGEST FIRST > panel inside the group (consecutive workgroups = different XCDs on the same
// factor stream; with 8 | panels per group, panel px always lands on XCD px % 8, whose L2 keeps its residual rows).  The hardware
// dispatcher refills a CU as its workgroups retire, which IS longest-first list scheduling.  Round 4 built the alternative the
// round-3 review asked for -- a persistent grid pulling units from agent-scope queues, in three forms (units pulled at the
// hand-off; software-pipelined units whose next fragments are requested before the exchange and whose queue / arrival adds are
// never waited for; per-XCD queues that keep a panel on one XCD), each bit-identical and parity-green: 3-5 % SLOWER than this
// grid from 2048 walkers up, even at 1024, slower below (profiles/r04_queue_v*_sizes_raw.txt, the kernel:
// profiles/r04_work_queue_kernel_v3.patch).  Its in-kernel timeline says why scheduling is not where the time is
// (profiles/r04_solve_timeline_v3.txt): a wave spends 24 % of its life outside K loops, 12 us per unit, waiting at the exchange
// barrier for its workgroup's three other waves -- they sit on four SIMDs that each arbitrate four workgroups' waves and drift
// apart over a 50-100 us K loop -- and the kernel's ragged end (CUs finish 174-187 us into a 190 us kernel) is set by the
// (rb + 1)-sized units in flight when the queue runs dry, which a queue cannot shorten.
// ------------------------------------------------------------------------------------------------

// Workgroup barrier for LDS traffic only (no wait on global loads that have nothing to do with the exchange).  Neither this nor
// s_barrier itself orders a wave's global STORES: a hand-off to another agent needs the storing wave's own s_waitcnt vmcnt(0).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The last arriver's read-back of a panel's shares: `n` doubles at `src` (written by other workgroups with agent-scope
// write-through stores) into LDS.  Every thread first ISSUES its (up to eight) agent-scope loads, then stores them: written as
// one load + one LDS store per loop iteration, each iteration waited for its own round trip to memory (~1.2 us; the loads
// bypass this XCD's L2) -- seven serialised round trips, 9 of the small-batch kernel's 20 us (profiles/r03_small_batch_solve.txt).
// n <= 4096: two straight-line stages of eight loads, not a loop -- at a loop head hipcc waits for every load in flight (vmcnt(0)).
__device__ __forceinline__ void fetch_shares_stage(const double* src, int n, double* __restrict__ sh, int tid, int base) {
  double v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int idx = base + k * 256 + tid;
    v[k] = __hip_atomic_load(&src[idx < n ? idx : n - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // unconditional: no branch, no wait
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int idx = base + k * 256 + tid;
    if (idx < n) sh[idx] = v[k];
  }
}
__device__ __forceinline__ void fetch_shares_to_lds(const double* src, int n, double* __restrict__ sh, int tid) {
  fetch_shares_stage(src, n, sh, tid, 0);
  if (n > 8 * 256) fetch_shares_stage(src, n, sh, tid, 8 * 256);
}

// One 16 x 16 tile of Y = X Delta from the four K quarters' partial tiles (C layout: register r of lane l is row (l >> 4) + 4 r,
// column l & 15), and its share of chi^2 per walker column: the sum of squares over the tile's 16 rows -- four in registers,
// then across the four lane groups.  ONE function for the throughput kernel and the small-batch kernel: the same expression,
// hence the same FMA contraction, hence bit-identical shares.
__device__ __forceinline__ double tile_chi2_share(const d4& p0, const d4& p1, const d4& p2, const d4& p3) {
  const d4 y = ((p0 + p1) + p2) + p3;
  double v = y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3];
  v += __shfl_xor(v, 16, CF_WAVE);
  v += __shfl_xor(v, 32, CF_WAVE);
  return v;
}
// a row block's share from its four tiles' shares, in tile order
__device__ __forceinline__ double rowblock_share(double t0, double t1, double t2, double t3) { return ((t0 + t1) + t2) + t3; }

// The last arriver's epilogue when a panel's shares do not fit the LDS copy (more than ~109 row blocks: N > 6976): a loop of
// dependent agent-scope loads and the epilogue's fields as scalar loads from the handle's device copy.  Out of line: nothing of
// it may cost the production path a register.
__device__ __attribute__((noinline)) void panel_epilogue_from_memory(const cf_epilogue* epi, const double* theta, int ndim, const double* partial,
                                                                      int64_t w_pad, int n_rb, int64_t w, const double* chi2_extra, double* out,
                                                                      int out_kind, unsigned long long* nonfinite, double* chi2_sn_out) {
  double c2 = 0.0;
  for (int r = 0; r < n_rb; ++r) c2 += __hip_atomic_load(&partial[(int64_t)r * w_pad + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (chi2_sn_out) chi2_sn_out[w] = c2;
  if (chi2_extra) c2 += chi2_extra[w];
  out[w] = finalize_value(*epi, theta + w * ndim, c2, out_kind, nonfinite);
}

// The workgroup that arrived last for panel `px`: the panel's shares summed in row-block order, the prior / output epilogue, the
// completion word.  `sh`: 4096 doubles of LDS (the K-quarter exchange buffer, free at this point).
template <int NP>
__device__ __forceinline__ void panel_last_arriver(double* sh, const cf_epilogue* __restrict__ epi, int ndim, int n_rb,
                                                             const double* __restrict__ theta, int64_t W, int64_t w_pad, double* partial,
                                                             unsigned int* arrivals, const double* __restrict__ chi2_extra,
                                                             double* __restrict__ out, int out_kind, unsigned long long* nonfinite,
                                                             double* __restrict__ chi2_sn_out, int px, unsigned long long* done_flag,
                                                             unsigned long long done_seq) {
  constexpr int PW = 16 * NP;
  const int tid = threadIdx.x;
  const int64_t w0 = (int64_t)px * PW;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (tid == 0) __hip_atomic_store(&arrivals[px], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // The shares come back through LDS: 256 threads fetch the (row block, walker) entries side by side, then thread w adds its
  // walker's shares in row-block order -- the same additions as a loop of dependent loads, without its n_rb round
  // trips to memory (27 x ~0.7 us: two thirds of a one-panel solve; profiles/r02_epilogue_loads_ab.txt).  With them come one
  // load per lane of the epilogue's fields (the handle's device copy) and the panel's theta rows, parked in LDS for
  // finalize_value.  Straight-line code: at a loop head hipcc drains every load in flight.
  const int n_sh = n_rb * PW, n_th = PW * ndim;
  if (n_sh + CF_EPI_WORDS + PW * CF_MAX_NDIM <= 4096) {  // shares, epilogue words, theta rows
    unsigned long long* epi_lds = reinterpret_cast<unsigned long long*>(sh + n_sh);
    double* th_lds = sh + n_sh + CF_EPI_WORDS;
    unsigned long long dv = 0ull;
    double tv[NP];
    if (tid < CF_EPI_WORDS) dv = reinterpret_cast<const unsigned long long*>(epi)[tid];
#pragma unroll
    for (int k = 0; k < NP; ++k) {  // PW * CF_MAX_NDIM = 256 NP entries at most
      const int idx = k * 256 + tid;
      const int64_t ti = w0 * ndim + idx;
      tv[k] = (idx < n_th && ti < W * ndim) ? theta[ti] : 0.0;
    }
    auto stage = [&](int base) {
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int idx = base + k * 256 + tid, ic = idx < n_sh ? idx : n_sh - 1;
        v[k] = __hip_atomic_load(&partial[(int64_t)(ic / PW) * w_pad + w0 + ic % PW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int idx = base + k * 256 + tid;
        if (idx < n_sh) sh[idx] = v[k];
      }
    };
    stage(0);
    if (n_sh > 8 * 256) stage(8 * 256);
    if (tid < CF_EPI_WORDS) epi_lds[tid] = dv;
#pragma unroll
    for (int k = 0; k < NP; ++k)
      if (k * 256 + tid < n_th) th_lds[k * 256 + tid] = tv[k];
    lds_barrier();
    if (tid < PW && w0 + tid < W) {
      const cf_epilogue& el = *reinterpret_cast<const cf_epilogue*>(epi_lds);
      const int64_t w = w0 + tid;
      const double extra = chi2_extra ? chi2_extra[w] : 0.0;
      double c2 = 0.0;
      for (int r0 = 0; r0 < n_rb; r0 += 16) {  // in row-block order, 16 LDS reads in flight
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = sh[(r0 + k < n_rb ? r0 + k : n_rb - 1) * PW + tid];
#pragma unroll
        for (int k = 0; k < 16; ++k)
          if (r0 + k < n_rb) c2 += v[k];
      }
      if (chi2_sn_out) chi2_sn_out[w] = c2;  // the SN block alone (cf_eval_parts)
      if (chi2_extra) c2 += extra;
      out[w] = finalize_value(el, th_lds + tid * ndim, c2, out_kind, nonfinite);
    }
  } else if (tid < PW && w0 + tid < W) {
    panel_epilogue_from_memory(epi, theta, ndim, partial, w_pad, n_rb, w0 + tid, chi2_extra, out, out_kind, nonfinite, chi2_sn_out);
  }
  // the panel's completion word for a synchronous zero-copy host call (see tri_gemm_small_kernel): PW <= 32 lanes of wave 0 stored
  if (done_flag && tid < 64) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) __hip_atomic_store(&done_flag[px], done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

#ifdef CF_DIAG_CLOCK
// DIAGNOSTIC BUILD ONLY (tools/build_variant.sh clock -DCF_DIAG_CLOCK; tools/solve_clock.py): per workgroup the shader-clock and
// real-time (100 MHz) counters at entry, behind the hand-off and behind the last arriver's epilogue, where it ran (XCC_ID, HW_ID) --
// the clock the chip holds under this kernel is d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back
// (6)).  The stamps go to a buffer of their own; no output value depends on them.
__device__ unsigned long long cf_solve_clock[4096 * 8];  // [workgroup][mt0, rt0, mt1, rt1, mt2, rt2, units, panels]
extern "C" int cf_debug_solve_clock(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cf_solve_clock), sizeof(cf_solve_clock));
}
#define CF_CLOCK_STAMP(k)                                                                     \
  if (tid == 0 && blockIdx.x < 4096) {                                                        \
    cf_solve_clock[blockIdx.x * 8 + 2 * (k)] = __builtin_amdgcn_s_memtime();                  \
    cf_solve_clock[blockIdx.x * 8 + 2 * (k) + 1] = __builtin_amdgcn_s_memrealtime();          \
  }
#define CF_CLOCK_COUNT(k, v) \
  if (tid == 0 && blockIdx.x < 4096) cf_solve_clock[blockIdx.x * 8 + (k)] = (unsigned long long)(v)
// wave 0's cycles inside K loops, its K-step pairs and its units, per workgroup
__device__ unsigned long long cf_solve_kloop[4096 * 4];
__device__ unsigned long long cf_solve_phase[4096 * 16];  // [workgroup][K loop begin / end of waves 0-3 (8), first exchange barrier passed, shares formed, stores acknowledged, arrival add returned]
extern "C" int cf_debug_solve_phase(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cf_solve_phase), sizeof(cf_solve_phase));
}
#define CF_PHASE(k) \
  if (blockIdx.x < 4096) cf_solve_phase[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime()
extern "C" int cf_debug_solve_kloop(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cf_solve_kloop), sizeof(cf_solve_kloop));
}
#define CF_KLOOP_DECL unsigned long long kl_t = 0, kl_cyc = 0, kl_pairs = 0, kl_units = 0
#define CF_KLOOP_BEGIN kl_t = __builtin_amdgcn_s_memtime()
#define CF_KLOOP_END(nq)                                  \
  kl_cyc += __builtin_amdgcn_s_memtime() - kl_t;          \
  kl_pairs += (nq);                                       \
  ++kl_units
#define CF_KLOOP_STORE                                    \
  if (tid == 0 && blockIdx.x < 4096) {                    \
    cf_solve_kloop[blockIdx.x * 4] = kl_cyc;              \
    cf_solve_kloop[blockIdx.x * 4 + 1] = kl_pairs;        \
    cf_solve_kloop[blockIdx.x * 4 + 2] = kl_units;        \
  }
#else
#define CF_CLOCK_STAMP(k)
#define CF_CLOCK_COUNT(k, v)
#define CF_KLOOP_DECL
#define CF_KLOOP_BEGIN
#define CF_KLOOP_END(nq)
#define CF_KLOOP_STORE
#define CF_PHASE(k)
#endif

// The add is RELAXED unless CF_HANDOFF_RELEASE is defined (tests/test_gpu_handoff.py compares the two builds bit for bit): an
// agent-scope release lowers to buffer_wbl2 (write back this XCD's L2) + vmcnt(0) in EVERY workgroup; with write-through stores
// there is nothing for it to write back, and 3456 of them per launch were measured to cost 20 % of the kernel
// (profiles/r02_handoff_and_traffic_ab.txt).
#ifdef CF_HANDOFF_RELEASE
#define CF_ARRIVE_ORDER __ATOMIC_RELEASE
#else
#define CF_ARRIVE_ORDER __ATOMIC_RELAXED
#endif

template <int NP, int PF>
__global__ void __launch_bounds__(256, 4)  // four workgroups per CU: 128 VGPRs
tri_gemm_chi2_kernel(const cf_epilogue* __restrict__ epi, const d2* __restrict__ frags, int n_ld, int ndim, int n_rb,
                     const double* __restrict__ theta, int64_t W, const double* __restrict__ delta, int64_t w_pad, double* partial,
                     unsigned int* arrivals, const double* __restrict__ chi2_extra, double* __restrict__ out, int out_kind,
                     unsigned long long* nonfinite, double* __restrict__ chi2_sn_out, int panels_per_group, int snake, int nt_last,
                     int diag_skip, int split_levels, unsigned long long* done_flag, unsigned long long done_seq) {
  __shared__ __align__(16) d4 part[4][4][64];  // [wave][tile][lane] of one 16-walker panel: 32 KB
  __shared__ double chi_tile[4][16 * NP];
  __shared__ unsigned int arrived_before;
  constexpr int PW = 16 * NP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6);  // the wave's K quarter, in a scalar register: its address arithmetic too
  const int col = lane & 15, kq = lane >> 4;
  // HALF UNITS.  The `split_levels` lowest row blocks of a 32-walker panel are worked as two units of 16 walkers each (NP = 2 only;
  // the launcher asks for six levels at ~900-1800 walkers): a launch's ragged end -- at 1024-2048 walkers 8.5 % of it -- is as long as the
  // units in flight when the grid runs out, and those are the short row blocks' (profiles/NOTES_r04.md section 1).  A half unit computes its 16 walkers exactly as the
  // full unit would (the panels of a unit share nothing but the factor fragments), so the bits are the same; a panel then has
  // n_rb + split_levels arrivals.
  const int n_full_levels = n_rb - split_levels;
  const int per_group = panels_per_group * (n_rb + split_levels);
  const int grp = (int)blockIdx.x / per_group;
  int rem_id = (int)blockIdx.x % per_group;
  // `snake`: a grid that is resident all at once (<= 1024 workgroups) is placed statically, workgroups i, i + 256, i + 512, ... on the
  // same CU: in plain descending order some CUs then hold 60 row-block units and others 36 (512 / 1024 walkers; the mean is 47.25).
  // Alternate blocks of 256 workgroups run ascending instead (43 .. 51 per CU; a longest-first assignment computed on the host
  // would reach 46 .. 50): 256 walkers 46.4 -> 43.0 us per call, 512: 58.9 -> 55.5, 1024: 94.5 -> 87.3; a grid that arrives in
  // waves (4096 walkers) is better off descending (254 against 257.5 us).  profiles/r03_gemm_stamps_and_pairing.txt
  if (snake) {
    const int b = rem_id >> 8, len = (per_group - (b << 8)) < 256 ? per_group - (b << 8) : 256;
    if (b & 1) rem_id = (b << 8) + (len - 1 - (rem_id & 255));
  }
  int rb, px_in_group, half = -1;  // largest row blocks first; half: -1 a full unit, 0 / 1 the panel's first / second 16 walkers
  if (rem_id < panels_per_group * n_full_levels) {
    rb = n_rb - 1 - rem_id / panels_per_group;
    px_in_group = rem_id % panels_per_group;
  } else {
    const int r2 = rem_id - panels_per_group * n_full_levels;
    rb = split_levels - 1 - r2 / (2 * panels_per_group);
    px_in_group = (r2 % (2 * panels_per_group)) >> 1;
    half = r2 & 1;
  }
  const int px = grp * panels_per_group + px_in_group;
  const int np_u = half < 0 ? NP : 1;                                    // 16-walker panels of this unit
  const int64_t w0 = (int64_t)px * PW + (half > 0 ? 16 : 0);             // its first walker
  if ((int64_t)px * PW >= W) return;  // the last group may be partly empty
  // (the second half of a batch's last panel may hold no walker: it computes nothing and stores nothing, but it ARRIVES -- the
  // panel counts n_rb + split_levels arrivals whatever its fill)
  const bool empty_half = w0 >= W;
  CF_KLOOP_DECL;
  CF_CLOCK_STAMP(0);
  if (!empty_half) {
    const int nq = 2 * (rb + 1);  // K-step pairs per wave
    // the factor stream of (row block, K quarter) in closed form (cf_inv_stream_off): no dependent load in front of the first fragment
    const d2* A = frags + cf_inv_stream_off(rb, g) * 64 + lane;
    // B fragments: 16 bytes per lane from the walker's residual row; panel c is 16 rows (8 n_ld d2) further on
    const d2* Bq = reinterpret_cast<const d2*>(delta) + ((w0 + col) * (int64_t)n_ld + 8 * g * nq + 2 * kq) / 2;
    const int64_t bstride = half < 0 ? 8 * (int64_t)n_ld : 0;  // (a half unit's second-panel loads re-read its own rows: never multiplied)
    d4 acc[NP][4];
#pragma unroll
    for (int c = 0; c < NP; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[c][j] = (d4){0.0, 0.0, 0.0, 0.0};
    // PF-deep software pipeline over the nq K-step pairs.  Nothing is loaded past the wave's K range (but for row
    // block 0, covered by the buffers' slack): a load that nobody consumes still has to land before the wave may
    // retire, and it misses every cache.
    //
    // WHICH TILES A PAIR MULTIPLIES.  Every pair's four factor fragments are fetched (the loads' bookkeeping stays that of a
    // branch-free loop), but a tile whose fragment is all zeros is not multiplied: wave-uniform branches around IN-PLACE matrix
    // instructions (tiles j_lo .. j_hi - 1).  An accumulator that receives 0 x b keeps its value -- it is never -0: it starts at
    // +0 -- so the bits are those of the full product.
    //  * j_hi: the LAST row block's rows N .. n_ld - 1 are padding -- N = 1701: one of its four tiles is nothing else (1.8 % of the
    //    launch's matrix instructions, on its 128 longest units), N = 1820: two (3.3 %).
    //  * j_lo: the last 64 columns of the K range are the row block's diagonal 64 x 64 block, lower triangular: its pairs 2 m,
    //    2 m + 1 (columns 16 m .. 16 m + 15 of the block) meet only zeros in the tiles above tile m.  From row block 3 on those eight
    //    pairs are the END of wave 3's quarter, four PF-groups of two pairs: group m takes the tiles m .. 3 -- 24 NP of the wave's
    //    16 NP (rb + 1) matrix instructions; in row blocks 0 - 2 the block is spread over several waves, same rule.  2.7 % of the
    //    launch's matrix instructions.
    // (As straight-line variants per tile count the compiler kept a second set of 64 accumulator registers across the variants'
    // merge: 162-182 VGPRs.)
    d2 a[PF][4], bf[PF][NP];
    auto load_stage = [&](int p) {
#pragma unroll
      for (int j = 0; j < 4; ++j) a[p][j] = A[(p * 4 + j) * 64];
#pragma unroll
      for (int c = 0; c < NP; ++c) bf[p][c] = Bq[c * bstride + p * 4];
    };
    auto mfma_stage_full = [&](int p) {  // every tile, every panel: one straight line of independent accumulators
#pragma unroll
      for (int c = 0; c < NP; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[c][j] = mfma_f64(a[p][j].x, bf[p][c].x, acc[c][j]);
#pragma unroll
      for (int c = 0; c < NP; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[c][j] = mfma_f64(a[p][j].y, bf[p][c].y, acc[c][j]);
    };
    auto mfma_stage = [&](int p, int j_lo, int j_hi) {  // per accumulator the pair's two K steps in order
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j >= j_lo && j < j_hi) {
#pragma unroll
          for (int c = 0; c < NP; ++c)
            if (c < np_u) {
              acc[c][j] = mfma_f64(a[p][j].x, bf[p][c].x, acc[c][j]);
              acc[c][j] = mfma_f64(a[p][j].y, bf[p][c].y, acc[c][j]);
            }
        }
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      load_stage(p);  // unconditional (a guard costs hipcc its exact vmcnt bookkeeping); only row block 0 has nq < PF
      __builtin_amdgcn_sched_barrier(0);  // stage 0 must be the oldest load
    }
    A += PF * 4 * 64;
    Bq += PF * 4;
    const int n_groups = nq / PF, rem = nq - n_groups * PF;
    const int j_hi = rb == n_rb - 1 ? nt_last : 4;
    // the wave's group kg is 16-column group g n_groups + kg of the K range, the diagonal block starts at group 4 rb (PF = 2: a group
    // is two pairs = 16 columns); diag_skip = 0 multiplies everything (A/B)
    const int diag_off = (PF == 2 && diag_skip) ? g * n_groups - 4 * rb : -(1 << 20);
    CF_KLOOP_BEGIN;
    if (lane == 0) { CF_PHASE(2 * g); }
    // The guards are not free (every tile's instructions become a basic block of their own: with guarded stages everywhere the
    // launch took 4 % longer, profiles/r04_net_ab_guards_everywhere_raw.txt): the groups that multiply everything -- all but the
    // wave's diagonal groups, in a whole unit that is not of the last row block -- run the straight-line stage.
    const int n_fast = (j_hi < 4 || np_u < NP) ? 0 : (1 - diag_off < n_groups - 1 ? (1 - diag_off > 0 ? 1 - diag_off : 0) : n_groups - 1);
    for (int kg = 0; kg < n_fast; ++kg) {
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        mfma_stage_full(p);
        __builtin_amdgcn_sched_barrier(0);
        load_stage(p);
        __builtin_amdgcn_sched_barrier(0);
      }
      A += PF * 4 * 64;
      Bq += PF * 4;
    }
    for (int kg = n_fast; kg + 1 < n_groups; ++kg) {
      const int j_lo = diag_off + kg;  // <= 0 left of the diagonal block: every tile
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        mfma_stage(p, j_lo, j_hi);
        __builtin_amdgcn_sched_barrier(0);
        load_stage(p);
        __builtin_amdgcn_sched_barrier(0);
      }
      A += PF * 4 * 64;
      Bq += PF * 4;
    }
    if (n_groups > 0) {  // last full group: only the nq % PF pairs of the tail are still to be fetched
      const int j_lo = diag_off + n_groups - 1;
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        mfma_stage(p, j_lo, j_hi);
        __builtin_amdgcn_sched_barrier(0);
        if (p < rem) load_stage(p);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int p = 0; p < PF - 1; ++p)
      if (p < rem) mfma_stage(p, 0, j_hi);
    CF_KLOOP_END(nq);
    if (lane == 0) { CF_PHASE(2 * g + 1); }
    // the four K quarters meet in LDS, one 16-walker panel at a time; wave g owns tile g: y, then the column sums of y^2
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      if (c >= np_u) break;
      if (c > 0) lds_barrier();
#pragma unroll
      for (int j = 0; j < 4; ++j) part[g][j][lane] = acc[c][j];
      lds_barrier();
      if (c == 0 && tid == 0) { CF_PHASE(8); }
      const double v = tile_chi2_share(part[0][g][lane], part[1][g][lane], part[2][g][lane], part[3][g][lane]);
      if (lane < 16) chi_tile[g][c * 16 + lane] = v;
    }
  }
  lds_barrier();
  if (tid == 0) { CF_PHASE(9); }
  // Hand-off between workgroups on different XCDs (their L2s are not coherent), in the form MI355X_MICROARCH.md lists as
  // valid for gfx950 (Workgroup dispatch, XCD placement & inter-workgroup visibility: "Valid forms", first table row):
  // producer -- wave 0 stores the workgroup's shares with agent-scope (sc1, write-through) stores, drains them with
  // s_waitcnt vmcnt(0) (inline asm with a memory clobber: the compiler may not move the stores or the add across it),
  // then ONE lane bumps the panel's arrival counter with an agent-scope atomic add.  Consumer -- the workgroup whose
  // add returned n_rb - 1 came last: behind the workgroup barrier its lanes issue an agent-scope ACQUIRE fence
  // and read every share back with agent-scope (sc1) loads, add the row blocks in a fixed order (the result does not
  // depend on which workgroup it was) and re-arm the counter for the next launch (panel_last_arriver).
  if (g == 0) {
    if (lane < 16 * np_u && !empty_half)
      __hip_atomic_store(&partial[(int64_t)rb * w_pad + w0 + lane],
                         rowblock_share(chi_tile[0][lane], chi_tile[1][lane], chi_tile[2][lane], chi_tile[3][lane]), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores of every lane of this wave have reached memory
    if (tid == 0) { CF_PHASE(10); }
    if (lane == 0)
      arrived_before = __hip_atomic_fetch_add(&arrivals[px], 1u, CF_ARRIVE_ORDER, __HIP_MEMORY_SCOPE_AGENT);
  }
  lds_barrier();
  if (tid == 0) { CF_PHASE(11); }
  CF_CLOCK_STAMP(1);
  CF_KLOOP_STORE;
  CF_CLOCK_COUNT(6, ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4));  // XCC_ID, HW_ID
  CF_CLOCK_COUNT(7, arrived_before == (unsigned)(n_rb + split_levels) - 1u);
  if (arrived_before != (unsigned)(n_rb + split_levels) - 1u) return;
  panel_last_arriver<NP>(reinterpret_cast<double*>(part), epi, ndim, n_rb, theta, W, w_pad, partial, arrivals, chi2_extra, out, out_kind,
                         nonfinite, chi2_sn_out, px, done_flag, done_seq);
  CF_CLOCK_STAMP(2);
}

#define CF_INSTANTIATE_TRIGEMM(NP, PF)                                                                                              \
  template __global__ void tri_gemm_chi2_kernel<NP, PF>(const cf_epilogue*, const d2*, int, int, int, const double*, int64_t,       \
                                                        const double*, int64_t, double*, unsigned int*, const double*, double*, int, \
                                                        unsigned long long*, double*, int, int, int, int, int,                        \
                                                        unsigned long long*, unsigned long long);
CF_INSTANTIATE_TRIGEMM(1, 2)  // up to 512 walkers
CF_INSTANTIATE_TRIGEMM(2, 2)  // beyond: an A fragment feeds two MFMAs

// ------------------------------------------------------------------------------------------------
// The same solve for SMALL batches (W <= a few panels of 16 walkers: emcee's 16-walker half-steps of BASELINE configs[0],
// sn/pantheon.py:108-123; the serial callers of log_evidence.py:20-46).  With one panel the throughput kernel above is 27
// workgroups; the longest (row block 26) chains 4 tiles x 108 K-steps = 432 MFMAs of 64 cycles per wave, 13 us, while 229 CUs idle.
// Here the unit of work is ONE 16-row tile: a 256-thread workgroup per (panel, row block, tile), its four waves the four K
// quarters of the row block exactly as above -- 4 x 27 = 108 workgroups per panel, every wave alone on its SIMD with a chain of
// at most 108 dependent MFMAs, and a PF-deep prefetch of the wave's factor and residual fragments (one 1 KiB load each per
// K-step pair; the streams are short, so the loads in flight, not the matrix pipe, set the time).
// Every y element is accumulated in the throughput kernel's order (K quarters sequentially in the MFMA accumulator, quarters
// added in LDS, tile_chi2_share), the tile shares go to `partial4[panel][row block][tile][walker]`, and the workgroup that
// arrives last for a panel adds tiles, then row blocks, in the fixed order of rowblock_share + the row-block loop above: a walker's
// result is BIT-IDENTICAL to the throughput kernel's (tests/test_gpu_parity.py::test_config2_batch_invariance).
// Grid: blockIdx.x = panel * units_pad + unit, units_pad a multiple of 8 so that the panels of one (row block, tile) land on the
// same XCD and share its factor fragments in that L2; units in descending row-block order (longest chains first); over a
// sequence of small calls unit u stays on XCD u % 8, whose 4 MB L2 keeps its eighth of the 11.8 MB factor.
// ------------------------------------------------------------------------------------------------
template <int PF, bool FRAG, int TPW>
__global__ void __launch_bounds__(256)
tri_gemm_small_kernel(const cf_epilogue* __restrict__ epi, const d2* __restrict__ frags, int n_ld, int ndim, int n_rb,
                      const double* __restrict__ theta, int64_t W, const double* __restrict__ delta, double* partial4, unsigned int* arrivals,
                      const double* __restrict__ chi2_extra, double* __restrict__ out, int out_kind,
                      unsigned long long* nonfinite, double* __restrict__ chi2_sn_out, int units_pad,
                      unsigned long long* done_flag, unsigned long long done_seq) {
  static_assert(TPW == 1 || TPW == 2 || TPW == 4, "tiles per workgroup");
  __shared__ __align__(16) d4 part[4][TPW][64];  // [K quarter][tile][lane] partial tiles: 8 KB per tile
  __shared__ double chi_tile[TPW][16];
  // the last arriver's copy of the panel's shares [row block][tile][walker] and their sums over the tiles [row block][walker]:
  // dynamic LDS, 80 n_rb doubles (the launcher sizes it; none when the shares exceed 4096 doubles), so that the workgroups of a
  // many-panel batch fit four to a CU
  extern __shared__ double small_dyn[];
  double* const sh = small_dyn;
  double* const rbs = small_dyn + 64 * n_rb;
  __shared__ __align__(16) unsigned long long desc_lds[CF_EPI_WORDS];  // the last arriver's copy of the epilogue's fields (see below)
  __shared__ double th_lds[16 * CF_MAX_NDIM];                                           // ... and of the panel's theta rows
  __shared__ unsigned int arrived_before;
  const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const int unit = (int)blockIdx.x % units_pad, px = (int)blockIdx.x / units_pad;
  // TPW tiles per workgroup (units of TPW / 4 row block): with more than ~2 panels the one-tile units outnumber the CUs and share
  // their L1 bandwidth, and a unit of several tiles loads a residual fragment once for all of them
  constexpr int UPR = 4 / TPW;  // units per row block
  if (unit >= UPR * n_rb) return;
  const int rb = n_rb - 1 - unit / UPR, j = (unit % UPR) * TPW;  // first tile of the unit
  const int64_t w0 = (int64_t)px * 16;
  if (w0 >= W) return;
  const int nq = 2 * (rb + 1);  // K-step pairs per wave
  // K-step pair q of this wave's quarter: factor fragment of tile j (1 KiB, stride 4 KiB), residual fragment (16 B per lane)
  // (the stream offset in closed form, cf_inv_stream_off: one dependent load less in front of every wave's first fragment)
  const d2* A = frags + cf_inv_stream_off(rb, g) * 64 + j * 64 + lane;
  // FRAG: walker_fast_kernel left the panel's residuals in fragment order (delta_index): a K-step pair is 1 KiB in a row, like
  // the factor's -- the row form's 16 B per lane from 16 walker rows 8 n_ld bytes apart is sixteen cache lines per load, and those
  // gathers, not the matrix pipe (64 cycles per MFMA of a dependent chain, tools/mfma_chain_latency.hip) or the factor stream, set
  // the K loop's ~180 cycles per MFMA (profiles/r03_small_batch_solve.txt)
  constexpr int BS = FRAG ? 64 : 4;  // d2 per K-step pair
  const d2* Bq = FRAG ? reinterpret_cast<const d2*>(delta) + ((int64_t)px * (n_ld / 8) + (int64_t)g * nq) * 64 + lane
                      : reinterpret_cast<const d2*>(delta) + ((w0 + col) * (int64_t)n_ld + 8 * (int64_t)g * nq + 2 * kq) / 2;
  d4 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
  d2 a[PF][TPW], bf[PF];
  auto load_pair = [&](int p, int q) {
#pragma unroll
    for (int t = 0; t < TPW; ++t) a[p][t] = A[q * 256 + t * 64];
    bf[p] = Bq[q * BS];
  };
  auto mfma_pair = [&](int p) {  // per tile: the pair's two K steps in order, as the throughput kernel accumulates them
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = mfma_f64(a[p][t].x, bf[p].x, acc[t]);
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = mfma_f64(a[p][t].y, bf[p].y, acc[t]);
  };
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int p = 0; p < PF; ++p) {  // a short stream (nq < PF) re-reads its last pair: nothing outside the wave's range is touched
    load_pair(p, p < nq ? p : nq - 1);
    __builtin_amdgcn_sched_barrier(0);  // stage 0 must be the oldest load
  }
  A += PF * 256;
  Bq += PF * BS;
  const int n_groups = nq / PF, rem = nq - n_groups * PF;
  for (int grp = 0; grp + 1 < n_groups; ++grp) {  // branch-free body, as in the throughput kernel
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      mfma_pair(p);
      __builtin_amdgcn_sched_barrier(0);
      load_pair(p, p);
      __builtin_amdgcn_sched_barrier(0);
    }
    A += PF * 256;
    Bq += PF * BS;
  }
  if (n_groups > 0) {  // last full group: only the nq % PF pairs of the tail are still to be fetched
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      mfma_pair(p);
      __builtin_amdgcn_sched_barrier(0);
      if (p < rem) load_pair(p, p);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int p = 0; p < PF - 1; ++p)
    if (p < rem) mfma_pair(p);
  // The prior / output epilogue of the LAST ARRIVER reads a dozen of the epilogue's fields and the walkers' theta rows: as scalar
  // loads and dependent vector loads, each in its own basic block, they were 5 k cycles of the one workgroup the whole call waits
  // for.  Instead every workgroup issues, HERE, one load per lane of the handle's device copy of the epilogue block and of the
  // panel's theta rows -- they return behind the hand-off's own wait -- and the last arriver parks them in LDS, where
  // finalize_value reads them.
  constexpr int DW = CF_EPI_WORDS;
  static_assert(DW <= 256, "one epilogue word per thread");
  const int n_th = 16 * ndim;
  unsigned long long dv = 0ull;
  double tv = 0.0;
  {
    const int64_t ti = w0 * ndim + tid;
    if (tid < DW) dv = reinterpret_cast<const unsigned long long*>(epi)[tid];
    if (tid < n_th && ti < W * ndim) tv = theta[ti];
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t) part[g][t][lane] = acc[t];
  lds_barrier();
  // hand-off as in the throughput kernel (agent-scope write-through stores, vmcnt(0), one relaxed agent-scope add, the last
  // arriver's acquire fence); one counter per panel counts the 4 n_rb (row block, tile) shares, TPW per workgroup
  double* mine = partial4 + (int64_t)px * (4 * n_rb * 16);
  if (TPW == 1) {
    if (g == 0) {
      const double v = tile_chi2_share(part[0][0][lane], part[1][0][lane], part[2][0][lane], part[3][0][lane]);
      if (lane < 16) __hip_atomic_store(&mine[(rb * 4 + j) * 16 + lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {  // wave t forms tile t's share; wave 0 stores them all, so that its vmcnt(0) covers every store in front of the add
    if (g < TPW) {
      const double v = tile_chi2_share(part[0][g][lane], part[1][g][lane], part[2][g][lane], part[3][g][lane]);
      if (lane < 16) chi_tile[g][lane] = v;
    }
    lds_barrier();
    if (g == 0 && lane < 16 * TPW)
      __hip_atomic_store(&mine[(rb * 4 + j) * 16 + lane], chi_tile[lane >> 4][lane & 15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (g == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0)
      arrived_before = __hip_atomic_fetch_add(&arrivals[px], (unsigned)TPW, CF_ARRIVE_ORDER, __HIP_MEMORY_SCOPE_AGENT);
  }
  lds_barrier();
  if (arrived_before != 4u * (unsigned)n_rb - (unsigned)TPW) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (tid == 0) __hip_atomic_store(&arrivals[px], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int n_sh = 4 * n_rb * 16;
  const bool via_lds = n_sh <= 4096;
  if (tid < DW) desc_lds[tid] = dv;  // the descriptor and theta copies fetched before the hand-off (see above)
  if (tid < n_th) th_lds[tid] = tv;
  if (via_lds) fetch_shares_to_lds(mine, n_sh, sh, tid);
  const cf_epilogue& dl = *reinterpret_cast<const cf_epilogue*>(desc_lds);
  if (via_lds) {
    lds_barrier();
    // the (row block, walker) shares from their four tile shares, 256 threads side by side; thread w then adds its walker's
    // n_rb row-block shares in order -- the throughput kernel's additions, one short dependent chain instead of 4 n_rb
    for (int i = tid; i < n_rb * 16; i += 256) {
      const int r = i >> 4, wl = i & 15;
      rbs[i] = rowblock_share(sh[(r * 4 + 0) * 16 + wl], sh[(r * 4 + 1) * 16 + wl], sh[(r * 4 + 2) * 16 + wl], sh[(r * 4 + 3) * 16 + wl]);
    }
    lds_barrier();
  }
  else
    lds_barrier();  // desc_lds, th_lds
  if (tid < 16 && w0 + tid < W) {
    const int64_t w = w0 + tid;
    const double extra = chi2_extra ? chi2_extra[w] : 0.0;
    double c2 = 0.0;
    if (via_lds) {
      // the row-block shares added in order, 32 LDS reads in flight at a time (one read + one add per iteration waited an LDS
      // round trip each: 110 cycles x n_rb)
      for (int r0 = 0; r0 < n_rb; r0 += 32) {
        double v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = rbs[(r0 + k < n_rb ? r0 + k : n_rb - 1) * 16 + tid];
#pragma unroll
        for (int k = 0; k < 32; ++k)
          if (r0 + k < n_rb) c2 += v[k];
      }
    } else {
      for (int r = 0; r < n_rb; ++r) {
        double t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = __hip_atomic_load(&mine[(r * 4 + k) * 16 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c2 += rowblock_share(t[0], t[1], t[2], t[3]);
      }
    }
    if (chi2_sn_out) chi2_sn_out[w] = c2;  // the SN block alone (cf_eval_parts)
    if (chi2_extra) c2 += extra;
    out[w] = finalize_value(dl, th_lds + tid * ndim, c2, out_kind, nonfinite);
  }
  // A synchronous host call (zero-copy: `out` is the pinned staging block) does not wait for the END of this kernel -- its
  // teardown, the completion signal and the runtime's stream query are microseconds of a 40 us call -- but for this word in pinned
  // host memory: the panel's results have left (they were stored by lanes of this wave: vmcnt(0)), then ONE system-scope
  // release store of the call's sequence number.  Writes to the host travel in order.
  if (done_flag && tid < 64) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) __hip_atomic_store(&done_flag[px], done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
#define CF_INSTANTIATE_TRIGEMM_SMALL(PF, FRAG, TPW)                                                                                       \
  template __global__ void tri_gemm_small_kernel<PF, FRAG, TPW>(const cf_epilogue*, const d2*, int, int, int, const double*, int64_t,     \
                                                                const double*, double*, unsigned int*, const double*, double*, int,       \
                                                                unsigned long long*, double*, int, unsigned long long*, unsigned long long);
// one tile per workgroup with a 16-deep prefetch (up to 96 walkers), two tiles with an 8-deep one (97-160 walkers); FRAG = the
// per-walker kernel wrote the residuals in fragment order (the production path), row order for the accessor / generic paths
CF_INSTANTIATE_TRIGEMM_SMALL(16, true, 1)
CF_INSTANTIATE_TRIGEMM_SMALL(16, false, 1)
CF_INSTANTIATE_TRIGEMM_SMALL(8, true, 2)
CF_INSTANTIATE_TRIGEMM_SMALL(8, false, 2)

// ------------------------------------------------------------------------------------------------
// Likelihoods without an SN block: only the epilogue.
// ------------------------------------------------------------------------------------------------
extern "C" __global__ void finalize_kernel(cf_epilogue d, const double* __restrict__ theta, int64_t W,
                                           const double* __restrict__ chi2_extra, double* __restrict__ out,
                                           int out_kind, unsigned long long* nonfinite, unsigned long long* done_flag,
                                           unsigned long long done_seq) {
  const int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (w < W) out[w] = finalize_value(d, theta + w * d.ndim, chi2_extra ? chi2_extra[w] : 0.0, out_kind, nonfinite);
  if (done_flag) {  // completion word of this block's 256 walkers for a synchronous zero-copy host call
    // All four waves stored results into the pinned block.  s_barrier does NOT drain vmcnt: each storing wave waits for its own
    // stores to be acknowledged BEFORE the barrier, so that the flag (released by wave 0 behind the barrier) cannot overtake the
    // result stores of waves 1-3.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&done_flag[blockIdx.x], done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ------------------------------------------------------------------------------------------------
// Stand-alone interpolators (interpolator.py) on arbitrary (non-uniform) grids in global memory.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t searchsorted_left(const double* __restrict__ x, int64_t n, double v) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = lo + (hi - lo) / 2;
    if (x[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ double sgn(double v) { return (double)((v > 0) - (v < 0)); }

// Fritsch-Carlson slope at node i (interpolator.py:5-68), 3-point local stencil.
__device__ double pchip_slope_at(const double* __restrict__ x, const double* __restrict__ y, int64_t n, int64_t i) {
  if (n < 2) return 0.0;
  if (n == 2) return (y[1] - y[0]) / (x[1] - x[0]);
  if (i > 0 && i < n - 1) {
    double hl = x[i] - x[i - 1], hr = x[i + 1] - x[i];
    double dl = (y[i] - y[i - 1]) / hl, dr = (y[i + 1] - y[i]) / hr;
    if (dl != 0.0 && dr != 0.0 && dl * dr > 0.0) {
      double w1 = 2.0 * hr + hl, w2 = hr + 2.0 * hl;
      return (w1 + w2) / (w1 / dl + w2 / dr);
    }
    return 0.0;
  }
  // end points: non-centred three-point formula with the sign / overshoot guards
  double h0, h1, d0, d1;
  if (i == 0) {
    h0 = x[1] - x[0]; h1 = x[2] - x[1];
    d0 = (y[1] - y[0]) / h0; d1 = (y[2] - y[1]) / h1;
  } else {
    h0 = x[n - 1] - x[n - 2]; h1 = x[n - 2] - x[n - 3];
    d0 = (y[n - 1] - y[n - 2]) / h0; d1 = (y[n - 2] - y[n - 3]) / h1;
  }
  double e = ((2 * h0 + h1) * d0 - h0 * d1) / (h0 + h1);
  if (d0 == 0.0 || sgn(e) != sgn(d0)) return 0.0;
  if (sgn(d0) != sgn(d1) && fabs(e) > fabs(3 * d0)) return 3 * d0;
  return e;
}

// mode 0: Hermite with given slopes (exact=True); mode 1: PCHIP (slopes on the fly, clamped)
extern "C" __global__ void interp_kernel(const double* __restrict__ xq, int64_t nq, const double* __restrict__ x,
                                         const double* __restrict__ y, const double* __restrict__ yp, int64_t n,
                                         double* __restrict__ out, int mode) {
  const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (k >= nq) return;
  const double xi = xq[k];
  if (xi <= x[0]) {
    out[k] = mode == 0 ? y[0] + yp[0] * (xi - x[0]) : y[0];
    return;
  }
  if (xi >= x[n - 1]) {
    out[k] = mode == 0 ? y[n - 1] + yp[n - 1] * (xi - x[n - 1]) : y[n - 1];
    return;
  }
  const int64_t i = searchsorted_left(x, n, xi) - 1;
  const double d_i = mode == 0 ? yp[i] : pchip_slope_at(x, y, n, i);
  const double d_i1 = mode == 0 ? yp[i + 1] : pchip_slope_at(x, y, n, i + 1);
  const double h_i = x[i + 1] - x[i];
  const double t = (xi - x[i]) / h_i;
  const double t2 = t * t, t3 = t2 * t;
  const double h00 = 2 * t3 - 3 * t2 + 1, h10 = t3 - 2 * t2 + t, h01 = -2 * t3 + 3 * t2, h11 = t3 - t2;
  out[k] = h00 * y[i] + h10 * h_i * d_i + h01 * y[i + 1] + h11 * h_i * d_i1;
}

// Self-test hook: the in-kernel log10 on arbitrary inputs (tests/test_gpu_parity.py checks its ulp error).
// out[4 k .. 4 k + 3] = {sqrt_pos(a), sqrt(a), div_pos(a, b), a / b}: the positive-operand routines beside the library's (cf_selftest_pos_ops)
extern "C" __global__ void pos_ops_selftest_kernel(const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                                   double* __restrict__ out) {
  const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (k >= n) return;
  out[4 * k + 0] = sqrt_pos(a[k]);
  out[4 * k + 1] = sqrt(a[k]);
  out[4 * k + 2] = div_pos(a[k], b[k]);
  out[4 * k + 3] = a[k] / b[k];
}

// mode 0: log10_pos; mode 1: log10_tab with the table at `tab`; mode 2: exp_tab with the 2^(j/64) table at `tab`
extern "C" __global__ void log10_selftest_kernel(const double* __restrict__ x, int64_t n, double* __restrict__ out, int mode,
                                                 const cf_d2* __restrict__ tab) {
  const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (k >= n) return;
  if (mode == 2) out[k] = exp_tab(x[k], reinterpret_cast<const double*>(tab));
  else out[k] = mode == 0 ? log10_pos(x[k]) : log10_tab(x[k], reinterpret_cast<const d2*>(tab));
}

// Copy right-hand sides b[nrhs][n] into the padded residual layout Delta[nrhs_pad][n_pad].
extern "C" __global__ void pad_rhs_kernel(const double* __restrict__ b, int64_t nrhs, int64_t n, int64_t n_ld,
                                          double* __restrict__ delta) {
  const int64_t w = blockIdx.x;
  for (int64_t i = threadIdx.x; i < n_ld; i += blockDim.x) delta[w * n_ld + i] = (w < nrhs && i < n) ? b[w * n + i] : 0.0;
}
