// cosmofit_api.hip — host side of the C-ABI declared in include/cosmofit.h.
//
// Owns device memory (data vectors, packed factor, per-call workspace), one HIP stream per
// handle, and the launches of the kernels in cosmofit_kernels.hip.  There is NO CPU evaluation
// path here: without a HIP device every entry point returns CF_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <sys/prctl.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <utility>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cosmofit.h"
#include "cf_pack.h"
#include "cosmofit_device.h"

typedef double d2 __attribute__((ext_vector_type(2)));

// the solve kernel's prefetch reads up to 4 K-step pairs (4 KiB) past the end of a panel's Y range
#define CF_YPK_SLACK 8192
#define CF_DELTA_SLACK 4096  // the inverse-GEMM pipeline prefetches a few K-step pairs past the last residual row

// Tuning overrides, ONE environment string read once per process: CF_TUNE="key=value,key=value,..." (integers).  None of them
// changes a result; the table of keys is in DESIGN.md ("Knobs").  The three settings a user may need have their own variables
// (CF_HOST_WAIT, CF_ZEROCOPY_MAX here; COSMOFIT_LIB in the Python loader).
static long long cf_tune(const char* key, long long dflt) {
  static const std::string env = [] { const char* e = getenv("CF_TUNE"); return std::string(e ? e : ""); }();
  const size_t kl = strlen(key);
  for (size_t pos = 0; pos < env.size();) {
    const size_t end = std::min(env.find(',', pos), env.size());
    if (end - pos > kl + 1 && env.compare(pos, kl, key) == 0 && env[pos + kl] == '=') return atoll(env.c_str() + pos + kl + 1);
    pos = end + 1;
  }
  return dflt;
}

#define CF_BAO_NODES 6  // table nodes copied out per BAO datum (cosmofit_kernels.hip)
#define CF_SN_PARTS_MAX 4                         // workgroups per walker of a small batch (walker_fast_kernel)
#define CF_SN_REC_SLACK (512 * CF_SN_PARTS_MAX)   // spare SN records behind the last one
template <int MODEL, int FDE>
__global__ void walker_kernel(cf_dev_desc d, const double* theta, int64_t W, double* delta, double* dm_out, double* mucorr_out,
                              d2* bao_nodes, d2* table_out);
template <int MODEL, int FDE>
__global__ void walker_fast_kernel(cf_walker_args d, const double* theta, int64_t W, double* delta, d2* bao_nodes, double* theta_copy,
                                   int frag_b, int sn_parts);
template <int MODEL, int FDE, int LANES, int ROLES>
__global__ void small_blocks_kernel(cf_dev_desc d, const double* theta, int64_t W, const d2* bao_nodes, double* chi2_extra,
                                    double* blocks_out, double* bao_out);
template <int MODEL, int FDE, int C>
__global__ void growth_kernel(cf_dev_desc d, const double* theta, int64_t W, const d2* aux_nodes, double* chi2_extra, int accumulate,
                              double* blocks_out, double* theory_out);
template <int MODEL, int FDE>
__global__ void hz_kernel(cf_dev_desc d, const double* theta, const double* z, int64_t n, double* out);
#define CF_DECLARE_GROWTH(M, F, C) \
  extern template __global__ void growth_kernel<M, F, C>(cf_dev_desc, const double*, int64_t, const d2*, double*, int, double*, double*);
#define CF_DECLARE_WALKER(M, F)                                                                                            \
  extern template __global__ void walker_kernel<M, F>(cf_dev_desc, const double*, int64_t, double*, double*, double*, d2*, d2*); \
  extern template __global__ void walker_fast_kernel<M, F>(cf_walker_args, const double*, int64_t, double*, d2*, double*, int, int); \
  extern template __global__ void small_blocks_kernel<M, F, 16, 1>(cf_dev_desc, const double*, int64_t, const d2*, double*, \
                                                                   double*, double*);                                      \
  extern template __global__ void small_blocks_kernel<M, F, 64, 1>(cf_dev_desc, const double*, int64_t, const d2*, double*, \
                                                                   double*, double*);                                      \
  extern template __global__ void small_blocks_kernel<M, F, 64, 2>(cf_dev_desc, const double*, int64_t, const d2*, double*, \
                                                                   double*, double*);                                      \
  extern template __global__ void hz_kernel<M, F>(cf_dev_desc, const double*, const double*, int64_t, double*);             \
  CF_DECLARE_GROWTH(M, F, 1) CF_DECLARE_GROWTH(M, F, 2) CF_DECLARE_GROWTH(M, F, 4) CF_DECLARE_GROWTH(M, F, 8)
CF_DECLARE_WALKER(0, 0) CF_DECLARE_WALKER(0, 1) CF_DECLARE_WALKER(0, 2) CF_DECLARE_WALKER(0, 3)
CF_DECLARE_WALKER(1, 0) CF_DECLARE_WALKER(1, 1) CF_DECLARE_WALKER(1, 2) CF_DECLARE_WALKER(1, 3)

typedef void (*walker_fn)(cf_dev_desc, const double*, int64_t, double*, double*, double*, d2*, d2*);
static walker_fn pick_walker(int model, int fde) {
  static const walker_fn table[2][4] = {
      {walker_kernel<0, 0>, walker_kernel<0, 1>, walker_kernel<0, 2>, walker_kernel<0, 3>},
      {walker_kernel<1, 0>, walker_kernel<1, 1>, walker_kernel<1, 2>, walker_kernel<1, 3>}};
  return table[model][fde];
}
typedef void (*walker_fast_fn)(cf_walker_args, const double*, int64_t, double*, d2*, double*, int, int);
static walker_fast_fn pick_walker_fast(int model, int fde) {
  static const walker_fast_fn table[2][4] = {
      {walker_fast_kernel<0, 0>, walker_fast_kernel<0, 1>, walker_fast_kernel<0, 2>, walker_fast_kernel<0, 3>},
      {walker_fast_kernel<1, 0>, walker_fast_kernel<1, 1>, walker_fast_kernel<1, 2>, walker_fast_kernel<1, 3>}};
  return table[model][fde];
}
// The lean kernel arguments of the production per-walker kernel, and whether this descriptor may take it: register path of the
// table build (<= 4096 grid nodes), an SN block that fits sn_fast_loop.  CF_TUNE walker_generic=1 forces the generic kernel (A/B).
static bool walker_fast_ok(const cf_dev_desc& d) {
  static const bool off = cf_tune("walker_generic", 0) != 0;
  return !off && d.chunk_shift == 3 && d.ndim <= 64 && !d.sn_fixed_mu && !d.sn_dir && !d.sn_vel_mult && (!d.sn_lin || d.lin_in_rec);
}
// What finalize_value reads of the descriptor (cosmofit_device.h: cf_epilogue).
static cf_epilogue epilogue_of(const cf_dev_desc& d) {
  cf_epilogue e{};
  e.ndim = d.ndim; e.has_bounds = d.has_bounds; e.n_gauss = d.n_gauss; e.n_chi2_gauss = d.n_chi2_gauss;
  e.cpl_wall = d.cpl_wall; e.n_fs8 = d.n_fs8; e.n_cc = d.n_cc; e.cc_f_inverse = d.cc_f_inverse;
  e.log_norm = d.log_norm; e.logl_const = d.logl_const; e.cc_logdet = d.cc_logdet;
  e.w0 = d.slot[CF_P_W0_D]; e.wa = d.slot[CF_P_WA_D]; e.fs8err = d.slot[CF_P_FS8ERR_D]; e.fcc = d.slot[CF_P_FCC_D];
  for (int k = 0; k < CF_MAX_NDIM; ++k) { e.lo[k] = d.lo[k]; e.hi[k] = d.hi[k]; }
  for (int k = 0; k < CF_MAX_GAUSS; ++k) {
    e.gauss_idx[k] = d.gauss_idx[k]; e.gauss_mean[k] = d.gauss_mean[k]; e.gauss_sigma[k] = d.gauss_sigma[k];
    e.chi2_gauss_idx[k] = d.chi2_gauss_idx[k]; e.chi2_gauss_mean[k] = d.chi2_gauss_mean[k]; e.chi2_gauss_sigma[k] = d.chi2_gauss_sigma[k];
  }
  return e;
}
static cf_walker_args walker_args_of(const cf_dev_desc& d) {
  cf_walker_args a{};
  a.ndim = d.ndim; a.n_grid = d.n_grid; a.ez_model = d.ez_model; a.fde = d.fde;
  a.chunk_shift = d.chunk_shift; a.om_mode = d.om_mode;
  a.n_sn = d.n_sn; a.n_ld = d.n_ld;
  a.has_vstep = d.has_vstep; a.step_pm1 = d.step_pm1; a.lin_in_rec = d.lin_in_rec; a.n_aux = d.n_aux;
  a.z_max = d.z_max; a.step = d.step; a.c = d.c; a.inv_step = d.inv_step; a.inv_last = d.inv_last;
  a.or_h2 = d.or_h2; a.omnu_h2 = d.omnu_h2;
  for (int k = 0; k < CF_N_SLOTS; ++k) a.slot[k] = d.slot[k];
  a.sn_rec = d.sn_rec; a.log10_tab = d.log10_tab; a.nu_sw = d.nu_sw; a.ln_sw = d.ln_sw; a.exp2_tab = d.exp2_tab;
  a.bao_base = d.bao_base;
  return a;
}
typedef void (*small_blocks_fn)(cf_dev_desc, const double*, int64_t, const d2*, double*, double*, double*);
// lanes per walker: 16 or 64; roles (waves per walker, 64 lanes only): 1 or 2
static small_blocks_fn pick_small_blocks(int model, int fde, int lanes, int roles = 1) {
#define CF_SB_TABLE(L, R)                                                                                                                \
  {{small_blocks_kernel<0, 0, L, R>, small_blocks_kernel<0, 1, L, R>, small_blocks_kernel<0, 2, L, R>, small_blocks_kernel<0, 3, L, R>}, \
   {small_blocks_kernel<1, 0, L, R>, small_blocks_kernel<1, 1, L, R>, small_blocks_kernel<1, 2, L, R>, small_blocks_kernel<1, 3, L, R>}}
  static const small_blocks_fn narrow[2][4] = CF_SB_TABLE(16, 1);
  static const small_blocks_fn wide[2][4] = CF_SB_TABLE(64, 1);
  static const small_blocks_fn wide2[2][4] = CF_SB_TABLE(64, 2);
#undef CF_SB_TABLE
  return lanes == 64 ? (roles == 2 ? wide2[model][fde] : wide[model][fde]) : narrow[model][fde];
}
typedef void (*hz_fn)(cf_dev_desc, const double*, const double*, int64_t, double*);
static hz_fn pick_hz(int model, int fde) {
  static const hz_fn table[2][4] = {{hz_kernel<0, 0>, hz_kernel<0, 1>, hz_kernel<0, 2>, hz_kernel<0, 3>},
                                    {hz_kernel<1, 0>, hz_kernel<1, 1>, hz_kernel<1, 2>, hz_kernel<1, 3>}};
  return table[model][fde];
}
typedef void (*growth_fn)(cf_dev_desc, const double*, int64_t, const d2*, double*, int, double*, double*);
#define CF_GROWTH_ROW(M, C) {growth_kernel<M, 0, C>, growth_kernel<M, 1, C>, growth_kernel<M, 2, C>, growth_kernel<M, 3, C>}
static growth_fn pick_growth(int model, int fde, int steps) {  // steps = 256 C, C in {1, 2, 4, 8}
  static const growth_fn table[4][2][4] = {{CF_GROWTH_ROW(0, 1), CF_GROWTH_ROW(1, 1)}, {CF_GROWTH_ROW(0, 2), CF_GROWTH_ROW(1, 2)},
                                           {CF_GROWTH_ROW(0, 4), CF_GROWTH_ROW(1, 4)}, {CF_GROWTH_ROW(0, 8), CF_GROWTH_ROW(1, 8)}};
  const int c = steps / 256;
  return table[c == 1 ? 0 : c == 2 ? 1 : c == 4 ? 2 : 3][model][fde];
}
template <int KS, int TC>
__global__ void trsm_chi2_kernel(const cf_epilogue* epi, int n_pad, int n_ld, int ndim, cf_dev_pack pk, const double* theta, int64_t W,
                                 const double* delta, d2* ypk, const double* chi2_extra, double* out, int out_kind,
                                 unsigned long long* nonfinite, double* chi2_sn_out);
extern template __global__ void trsm_chi2_kernel<2, 4>(const cf_epilogue*, int, int, int, cf_dev_pack, const double*, int64_t, const double*,
                                                       d2*, const double*, double*, int, unsigned long long*, double*);
template <int NP, int PF>
__global__ void tri_gemm_chi2_kernel(const cf_epilogue* epi, const d2* frags, int n_ld, int ndim, int n_rb, const double* theta, int64_t W,
                                     const double* delta, int64_t w_pad, double* partial, unsigned int* arrivals, const double* chi2_extra,
                                     double* out, int out_kind, unsigned long long* nonfinite, double* chi2_sn_out, int panels_per_group,
                                     int snake, int nt_last, int diag_skip, int split_levels, unsigned long long* done_flag,
                                     unsigned long long done_seq);
template <int PF, bool FRAG, int TPW>
__global__ void tri_gemm_small_kernel(const cf_epilogue* epi, const d2* frags, int n_ld, int ndim, int n_rb, const double* theta, int64_t W,
                                      const double* delta, double* partial4, unsigned int* arrivals, const double* chi2_extra, double* out,
                                      int out_kind, unsigned long long* nonfinite, double* chi2_sn_out, int units_pad,
                                      unsigned long long* done_flag, unsigned long long done_seq);
#define CF_DECLARE_TRIGEMM_SMALL(PF, FRAG, TPW)                                                                                          \
  extern template __global__ void tri_gemm_small_kernel<PF, FRAG, TPW>(const cf_epilogue*, const d2*, int, int, int, const double*, int64_t, \
                                                                       const double*, double*, unsigned int*, const double*, double*, int, \
                                                                       unsigned long long*, double*, int, unsigned long long*, unsigned long long);
CF_DECLARE_TRIGEMM_SMALL(16, true, 1)
CF_DECLARE_TRIGEMM_SMALL(16, false, 1)
CF_DECLARE_TRIGEMM_SMALL(8, true, 2)
CF_DECLARE_TRIGEMM_SMALL(8, false, 2)
#define CF_DECLARE_TRIGEMM(NP, PF)                                                                                                       \
  extern template __global__ void tri_gemm_chi2_kernel<NP, PF>(const cf_epilogue*, const d2*, int, int, int, const double*, int64_t,     \
                                                               const double*, int64_t, double*, unsigned int*, const double*, double*,  \
                                                               int, unsigned long long*, double*, int, int, int, int, int,              \
                                                               unsigned long long*, unsigned long long);
CF_DECLARE_TRIGEMM(1, 2)
CF_DECLARE_TRIGEMM(2, 2)
extern "C" __global__ void finalize_kernel(cf_epilogue d, const double* theta, int64_t W, const double* chi2_extra,
                                           double* out, int out_kind, unsigned long long* nonfinite, unsigned long long* done_flag,
                                           unsigned long long done_seq);
extern "C" __global__ void interp_kernel(const double* xq, int64_t nq, const double* x, const double* y,
                                         const double* yp, int64_t n, double* out, int mode);
extern "C" __global__ void log10_selftest_kernel(const double* x, int64_t n, double* out, int mode, const cf_d2* tab);
extern "C" __global__ void pos_ops_selftest_kernel(const double* a, const double* b, int64_t n, double* out);
extern "C" __global__ void pad_rhs_kernel(const double* b, int64_t nrhs, int64_t n, int64_t n_ld, double* delta);

// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

// Shape of the blocked solve kernel's workgroup: ksplit x tclasses = 2 x 4 waves, two per SIMD (see cf_dev_pack).
static void default_shape(int& ks, int& tc) {
  ks = 2;
  tc = 4;
}
#define CF_PROBE_LIMIT 1e-11

static int pack_default(const double* L, int64_t n, int64_t ld, cf_host_pack& hp, double* probe_rel = nullptr) {
  int ks, tc;
  default_shape(ks, tc);
  int rc = cf_pack_cholesky(L, n, ld, hp, ks, tc);
  if (rc == 0 && probe_rel) *probe_rel = cf_pack_probe(hp, L, ld);
  return rc;
}

static std::string fmt_g(double v) {
  char buf[32];
  snprintf(buf, sizeof(buf), "%.3g", v);
  return buf;
}

static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
int cf_set_error(int code, const std::string& msg) { return fail(code, msg); }  // for cosmofit_ensemble.hip

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(CF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int ensure(size_t need) {
    if (need <= bytes) return 0;
    if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
    HIP_TRY(hipMalloc(&p, need));
    bytes = need;
    return 0;
  }
  template <class T> T* as() const { return (T*)p; }
};

struct PackedFactor {
  DevBuf frags, upd_off, diag_off;
  cf_dev_pack dev{};
  int64_t n = 0, n_pad = 0, bytes = 0;
  int upload(const cf_host_pack& hp) {
    n = hp.n;
    n_pad = hp.n_pad;
    bytes = (int64_t)(hp.frags.size() * sizeof(cf_d2));
    if (frags.ensure(hp.frags.size() * sizeof(cf_d2) + 16)) return CF_ERR_HIP;
    if (upd_off.ensure(hp.upd_off.size() * 8)) return CF_ERR_HIP;
    if (diag_off.ensure(hp.diag_off.size() * 8)) return CF_ERR_HIP;
    HIP_TRY(hipMemcpy(frags.p, hp.frags.data(), hp.frags.size() * sizeof(cf_d2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(upd_off.p, hp.upd_off.data(), hp.upd_off.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(diag_off.p, hp.diag_off.data(), hp.diag_off.size() * 8, hipMemcpyHostToDevice));
    dev.frags = frags.as<const cf_d2>();
    dev.upd_off = upd_off.as<const int64_t>();
    dev.diag_off = diag_off.as<const int64_t>();
    dev.n_blocks = hp.n_blocks;
    dev.ksplit = hp.ksplit;
    dev.tclasses = hp.tclasses;
    return 0;
  }
};

struct InversePack {
  DevBuf frags, off;
  cf_dev_invpack dev{};
  int64_t bytes = 0;
  int upload(const cf_host_invpack& hp) {
    bytes = (int64_t)(hp.frags.size() * sizeof(cf_d2));
    if (frags.ensure(hp.frags.size() * sizeof(cf_d2))) return CF_ERR_HIP;
    if (off.ensure(hp.off.size() * 8)) return CF_ERR_HIP;
    HIP_TRY(hipMemcpy(frags.p, hp.frags.data(), hp.frags.size() * sizeof(cf_d2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(off.p, hp.off.data(), hp.off.size() * 8, hipMemcpyHostToDevice));
    dev.frags = frags.as<const cf_d2>();
    dev.off = off.as<const int64_t>();
    dev.n_rowblocks = hp.n_rowblocks;
    return 0;
  }
};

// Pinned host staging: pageable caller buffers make hipMemcpyAsync go through the runtime's own
// bounce buffers; a handle-owned pinned block keeps the host round trip of a small batch short.
struct PinnedBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
  int ensure(size_t need) {
    if (need <= bytes) return 0;
    if (p) { (void)hipHostFree(p); p = nullptr; bytes = 0; }
    HIP_TRY(hipHostMalloc(&p, need, hipHostMallocDefault));
    bytes = need;
    return 0;
  }
};

#ifndef CF_ZEROCOPY_DEFAULT
#define CF_ZEROCOPY_DEFAULT 4096  // walkers; beyond, in-place access to the pinned block is no faster than two copy commands (profiles/r03_zerocopy_threshold.txt)
#endif
#define CF_DONE_FLAGS 256       // completion words per handle: panels of the largest zero-copy batch (4096 walkers / 16)
#define CF_SMALL_MAX_PANELS 16  // largest batch of the small-batch solve kernel: 256 walkers (the default switch is lower)
#ifndef CF_SMALL_DEFAULT
#define CF_SMALL_DEFAULT 160  // walkers: batches up to this size take the small-batch solve kernel (150 walkers: 39 against 44 us per call,
                             // 200: 45-47 against 48, 256: slower; profiles/r03_small_batch_solve.txt)
#endif
#ifndef CF_NP2_FROM
#define CF_NP2_FROM 512  // walkers: 32-walker panels in the throughput solve kernel beyond this batch size, 16-walker panels up to it
#endif

// A host thread that evaluates one replica's slice of a multi-device cf_eval (one per replica beyond the first).
struct cf_worker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  bool has_job = false, done = false, quit = false;
  const double* theta = nullptr;
  double* out = nullptr;
  int64_t W = 0;
  int out_kind = 0, rc = 0;
  std::string err;
};

struct cf_handle {
  int device = 0;
  int solve_mode = 0;
  PinnedBuf stage_in, stage_out;
  InversePack ipack;
  DevBuf partial, arrivals;  // inverse-GEMM solve: chi^2 shares per (row block, walker); arrival counters per panel
  DevBuf partial4;           // small-batch solve: shares per (panel, row block, tile, walker), CF_SMALL_MAX_PANELS panels
  DevBuf epi;                // device copy of `epi_host`: what the solve kernels' last arrivers read of the prior / output epilogue
  cf_epilogue epi_host{};
  hipStream_t stream = nullptr;  // host-buffer evaluations (cf_eval, cf_eval_parts)
  // the ONE workspace is shared by every evaluation of this handle: an evaluation launched on a different stream
  // than the previous one first waits (on the host) for that stream
  hipStream_t last_stream = nullptr;
  bool has_last = false;
  // timing ring: per evaluation 4 events (before the walker kernel, behind it, behind the small-block / growth kernels, behind the solve)
  std::vector<hipEvent_t> ev;
  int timing_slots = 0, timing_stride = 1;
  int64_t timed_calls = 0, eval_calls = 0;
  cf_dev_desc d{};
  PackedFactor pack;
  DevBuf z_cmb, z_hel, obs, sn_step, sn_rec, log10_tab;
  DevBuf bao_z, bao_val, bao_inv_cov, bao_qty, gl_x, gl_w, fixed_mu, cc_z, cc_h, cc_inv_cov, nu_grid, ln_grid, nu_sw, ln_sw, exp2_tab, sn_lin, sn_dir;
  DevBuf theta, out, delta, ypk, chi2_extra, nonfinite;
  DevBuf fs8_z, fs8_val, fs8_inv_cov, fs8_fid, fs8_step_of, fs8_order, fs8_tab, fs8_pts;
  DevBuf bao_nodes, bao_base;  // [max_walkers][n_bao][CF_BAO_NODES] table nodes for small_blocks_kernel; first node per datum
  PinnedBuf done_flags;           // [CF_DONE_FLAGS] words the evaluation's last kernel sets as each panel's results reach the staging block
  unsigned long long done_seq = 0;
  int done_armed = 0;             // panels whose word the current synchronous call waits for (0: wait for the stream)
  bool theta_on_host = false;     // set around a zero-copy evaluation: theta is the pinned staging block (reads cross the host link)
  bool has_small_blocks = false;  // BAO and / or CMB block present
  bool has_growth = false;        // growth-rate block present
  int64_t max_walkers = 0;
  double pack_probe_rel = 0.0;
  int cu_count = 0;
  char arch[64] = {0};
  // replicas on further devices (owned by the primary handle) and their worker threads
  std::vector<cf_handle*> peers;
  std::vector<std::unique_ptr<cf_worker>> workers;
  std::mutex mu;
};

// ------------------------------------------------------------------------------------------------
extern "C" int cf_abi_version(void) { return CF_ABI_VERSION; }
extern "C" const char* cf_last_error(void) { return g_err.c_str(); }

extern "C" int cf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// First of the CF_BAO_NODES table nodes copied out for a datum at redshift xi: its interval on the grid by the arithmetic of
// hermite_tab (cosmofit_kernels.hip), two nodes below.
static int32_t aux_node_base(const cf_dev_desc& d, double xi) {
  const int G = d.n_grid;
  int i = xi * d.inv_step >= (double)G ? G : (xi > 0.0 ? (int)(xi * d.inv_step) : 0);
  i = i > G - 2 ? G - 2 : (i < 0 ? 0 : i);
  if (i > 0 && (double)i * d.step >= xi) --i;
  if (i < G - 2 && (double)(i + 1) * d.step < xi) ++i;
  int b = i - 2;
  return b < 0 ? 0 : (b > G - CF_BAO_NODES ? G - CF_BAO_NODES : b);
}

// A per-node table in the order the 512 threads of walker_kernel fetch it (thread t owns nodes 8 t .. 8 t + 7):
// out[k * 512 + t] = table[min(8 t + k, n - 1)]   (cosmofit_device.h: nu_sw / ln_sw)
static std::vector<double> swizzle_for_walker_threads(const std::vector<double>& table) {
  std::vector<double> out(8 * 512);
  const size_t n = table.size();
  for (int k = 0; k < 8; ++k)
    for (int t = 0; t < 512; ++t) out[(size_t)k * 512 + t] = table[std::min((size_t)(8 * t + k), n - 1)];
  return out;
}

static int upload_vec(DevBuf& b, const double* src, int64_t n) {
  if (b.ensure((size_t)n * 8)) return CF_ERR_HIP;
  HIP_TRY(hipMemcpy(b.p, src, (size_t)n * 8, hipMemcpyHostToDevice));
  return 0;
}

// Reduction table of log10_tab (cosmofit_kernels.hip): 64 cells of [0.5, 1), {1 / centre, log10 centre}, each
// correctly rounded from extended precision.
static int upload_log10_table(DevBuf& b) {
  cf_d2 t[64];
  for (int j = 0; j < 64; ++j) {
    const long double c = 0.5L * (1.0L + ((long double)j + 0.5L) / 64.0L);
    t[j].x = (double)(1.0L / c);
    t[j].y = (double)log10l(c);
  }
  if (b.ensure(sizeof(t))) return CF_ERR_HIP;
  HIP_TRY(hipMemcpy(b.p, t, sizeof(t), hipMemcpyHostToDevice));
  return 0;
}

// Reduction table of exp_tab (cosmofit_kernels.hip): 2^(j/64), j < 64, correctly rounded from extended precision.
static int upload_exp2_table(DevBuf& b) {
  double t[64];
  for (int j = 0; j < 64; ++j) t[j] = (double)exp2l((long double)j / 64.0L);
  return upload_vec(b, t, 64);
}

static int ensure_workspace(cf_handle* h, int64_t W) {
  const int64_t w_pad = (W + 31) / 32 * 32;  // whole panels of the widest solve kernel (2 x 16 walkers)
  if (!h->done_flags.p) {  // completion words of synchronous zero-copy calls (every likelihood form)
    if (h->done_flags.ensure((size_t)CF_DONE_FLAGS * 8)) return CF_ERR_HIP;
    memset(h->done_flags.p, 0, (size_t)CF_DONE_FLAGS * 8);
  }
  if (w_pad <= h->max_walkers) return 0;
  // earlier evaluations may still be running on a caller's stream and use the buffers about to be replaced
  HIP_TRY(hipDeviceSynchronize());
  const int64_t n_pad = h->d.n_pad > 0 ? h->d.n_pad : 16, n_ld = h->d.n_ld > 0 ? h->d.n_ld : 64;
  if (h->theta.ensure((size_t)w_pad * (h->d.ndim > 0 ? h->d.ndim : 1) * 8)) return CF_ERR_HIP;
  if (h->out.ensure((size_t)w_pad * 8)) return CF_ERR_HIP;
  if (h->stage_in.ensure((size_t)w_pad * (h->d.ndim > 0 ? h->d.ndim : 1) * 8)) return CF_ERR_HIP;
  if (h->stage_out.ensure((size_t)w_pad * 8)) return CF_ERR_HIP;
  if (h->chi2_extra.ensure((size_t)w_pad * 8)) return CF_ERR_HIP;
  if (h->d.n_aux > 0 && h->bao_nodes.ensure((size_t)w_pad * h->d.n_aux * CF_BAO_NODES * sizeof(d2))) return CF_ERR_HIP;
  if (h->d.n_sn > 0 && h->solve_mode == CF_SOLVE_INVERSE_GEMM) {
    if (h->partial.ensure((size_t)w_pad * h->ipack.dev.n_rowblocks * 8)) return CF_ERR_HIP;
    if (h->arrivals.ensure((size_t)(w_pad / 16) * 4)) return CF_ERR_HIP;
    if (h->partial4.ensure((size_t)CF_SMALL_MAX_PANELS * 4 * h->ipack.dev.n_rowblocks * 16 * 8)) return CF_ERR_HIP;
    HIP_TRY(hipMemsetAsync(h->arrivals.p, 0, (size_t)(w_pad / 16) * 4, h->stream));  // the kernels re-arm them themselves
  }
  if (h->d.n_sn > 0) {
    if (h->delta.ensure((size_t)w_pad * n_ld * 8 + CF_DELTA_SLACK)) return CF_ERR_HIP;
    // columns of a partly filled last panel must hold finite numbers
    HIP_TRY(hipMemsetAsync(h->delta.p, 0, (size_t)w_pad * n_ld * 8 + CF_DELTA_SLACK, h->stream));
    if (h->solve_mode == CF_SOLVE_BLOCKED_TRSM) {  // solved panels Y, only the blocked solve stores them
      if (h->ypk.ensure((size_t)w_pad * n_pad * 8 + CF_YPK_SLACK)) return CF_ERR_HIP;
      HIP_TRY(hipMemsetAsync(h->ypk.p, 0, (size_t)w_pad * n_pad * 8 + CF_YPK_SLACK, h->stream));
    }
    // the evaluation may be launched on a caller's stream: the fills must have landed before it starts
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  h->max_walkers = w_pad;
  return 0;
}

static int validate_desc(const cf_desc* c) {
  if (c->abi_version != CF_ABI_VERSION || c->struct_size != (int32_t)sizeof(cf_desc))
    return fail(CF_ERR_INVALID, "cf_create: descriptor ABI version / size mismatch (got version " +
                                    std::to_string(c->abi_version) + ", size " + std::to_string(c->struct_size) +
                                    ", expected " + std::to_string(CF_ABI_VERSION) + ", " +
                                    std::to_string(sizeof(cf_desc)) + ")");
  if (c->ndim < 1 || c->ndim > CF_MAX_NDIM) return fail(CF_ERR_INVALID, "cf_create: ndim must be in 1..16");
  if (c->n_grid < 4 || c->n_grid > 8192)
    return fail(CF_ERR_INVALID, "cf_create: n_grid must be in 4..8192 (the {cum_dm, dh} table must fit the 160 KB LDS)");
  if (!(c->z_max > 0.0) || !std::isfinite(c->z_max)) return fail(CF_ERR_INVALID, "cf_create: z_max must be > 0");
  if (!(c->c_km_s > 0.0)) return fail(CF_ERR_INVALID, "cf_create: c_km_s must be > 0");
  if (c->ez_model != CF_EZ_LATE_FLAT && c->ez_model != CF_EZ_PHYSICAL) return fail(CF_ERR_INVALID, "cf_create: bad ez_model");
  if (c->fde < CF_FDE_LCDM || c->fde > CF_FDE_CPL) return fail(CF_ERR_INVALID, "cf_create: bad fde");
  if (c->n_bao < 0 || c->n_bao > CF_MAX_BAO) return fail(CF_ERR_INVALID, "cf_create: n_bao must be in 0..64");
  if (c->n_bao > 0) {
    if (c->n_grid < CF_BAO_NODES) return fail(CF_ERR_INVALID, "cf_create: a BAO block needs n_grid >= 6");
    if (!c->bao_z || !c->bao_val || !c->bao_qty || !c->bao_inv_cov)
      return fail(CF_ERR_INVALID, "cf_create: BAO block arrays must not be null");
    for (int k = 0; k < c->n_bao; ++k)
      if (c->bao_qty[k] < CF_BAO_DV || c->bao_qty[k] > CF_BAO_FAP) return fail(CF_ERR_INVALID, "cf_create: bad bao_qty code");
    if (c->rd_mode != CF_RD_PARAM && c->rd_mode != CF_RD_FIT) return fail(CF_ERR_INVALID, "cf_create: bad rd_mode");
    if (c->bao_dh_mode != CF_BAO_DH_PCHIP && c->bao_dh_mode != CF_BAO_DH_EXACT)
      return fail(CF_ERR_INVALID, "cf_create: bad bao_dh_mode");
    if (c->bao_dh_mode == CF_BAO_DH_PCHIP && c->n_grid < 3) return fail(CF_ERR_INVALID, "cf_create: PCHIP needs >= 3 grid nodes");
  }
  if (c->cmb_mode < CF_CMB_NONE || c->cmb_mode > CF_CMB_THETA_WB_WM) return fail(CF_ERR_INVALID, "cf_create: bad cmb_mode");
  if (c->cmb_mode != CF_CMB_NONE) {
    if (c->ez_model != CF_EZ_PHYSICAL)
      return fail(CF_ERR_INVALID, "cf_create: the CMB block needs CF_EZ_PHYSICAL (radiation + neutrinos up to z*)");
    if (c->n_gl < 1 || c->n_gl > CF_MAX_GL || !c->gl_x || !c->gl_w)
      return fail(CF_ERR_INVALID, "cf_create: the CMB block needs 1..256 Gauss-Legendre nodes");
  }
  if (c->solve_mode != CF_SOLVE_BLOCKED_TRSM && c->solve_mode != CF_SOLVE_INVERSE_GEMM && c->solve_mode != CF_SOLVE_AUTO)
    return fail(CF_ERR_INVALID, "cf_create: bad solve_mode");
  if (c->n_cc < 0 || c->n_cc > CF_MAX_CC) return fail(CF_ERR_INVALID, "cf_create: n_cc must be in 0..64");
  if (c->n_cc > 0 && (!c->cc_z || !c->cc_h || !c->cc_inv_cov))
    return fail(CF_ERR_INVALID, "cf_create: cosmic-chronometer arrays must not be null");
  if (c->ez_model == CF_EZ_PHYSICAL && !(c->nu_rho0 > 0.0))
    return fail(CF_ERR_INVALID, "cf_create: CF_EZ_PHYSICAL needs the neutrino constants (nu_rho0 > 0)");
  if (c->n_gauss > CF_MAX_GAUSS || c->n_chi2_gauss > CF_MAX_GAUSS || c->n_gauss < 0 || c->n_chi2_gauss < 0)
    return fail(CF_ERR_INVALID, "cf_create: at most 8 Gaussian terms of each kind");
  for (int s = 0; s < CF_P_NSLOTS; ++s)
    if (c->param[s].idx >= c->ndim) return fail(CF_ERR_INVALID, "cf_create: parameter slot index >= ndim");
  if (c->n_sn < 0 || c->n_sn > (1 << 20)) return fail(CF_ERR_INVALID, "cf_create: bad n_sn");
  if (c->n_sn > 0) {
    if (!c->sn_z_cmb || !c->sn_z_hel || !c->sn_obs || !c->sn_chol)
      return fail(CF_ERR_INVALID, "cf_create: SN block arrays must not be null");
    if (c->sn_chol_ld < c->n_sn) return fail(CF_ERR_INVALID, "cf_create: sn_chol_ld < n_sn");
  }
  if (c->n_fs8 < 0 || c->n_fs8 > CF_MAX_FS8) return fail(CF_ERR_INVALID, "cf_create: n_fs8 must be in 0..64");
  if (c->n_fs8 > 0) {
    if (!c->fs8_z || !c->fs8_val || !c->fs8_inv_cov || !c->fs8_fid) return fail(CF_ERR_INVALID, "cf_create: growth-rate arrays must not be null");
    if (!(c->fs8_a_init > 0.0 && c->fs8_a_init < 1.0)) return fail(CF_ERR_INVALID, "cf_create: fs8_a_init must be in (0, 1)");
    if (c->fs8_steps < 0 || c->fs8_steps > 2048) return fail(CF_ERR_INVALID, "cf_create: fs8_steps must be in 0..2048");
    if (c->fs8_n_agrid != 0 && (c->fs8_n_agrid < 4 || c->fs8_n_agrid > (1 << 20)))
      return fail(CF_ERR_INVALID, "cf_create: fs8_n_agrid must be 0 (direct read-out) or the number (>= 4) of points of the scripts' a-grid");
    if (c->n_grid < CF_BAO_NODES) return fail(CF_ERR_INVALID, "cf_create: a growth-rate block needs n_grid >= 6");
    for (int k = 0; k < c->n_fs8; ++k)
      if (!(1.0 / (1.0 + c->fs8_z[k]) >= c->fs8_a_init) || c->fs8_z[k] < 0.0)
        return fail(CF_ERR_INVALID, "cf_create: growth-rate data must lie in a_init <= a <= 1");
  }
  if (!std::isfinite(c->logl_const)) return fail(CF_ERR_INVALID, "cf_create: logl_const must be finite");
  if ((c->sn_vel_mode | 1) != 1 || (c->cc_f_mode | 1) != 1 || (c->prior_norm_mode | 1) != 1)
    return fail(CF_ERR_INVALID, "cf_create: sn_vel_mode, cc_f_mode and prior_norm_mode must be 0 or 1");
  if (c->sn_vel_mode == 1 && (c->sn_dir || c->sn_fixed_mu))
    return fail(CF_ERR_INVALID, "cf_create: sn_vel_mode 1 cannot be combined with sn_dir or sn_fixed_mu");
  if (c->om_mode != 0 && c->om_mode != 1) return fail(CF_ERR_INVALID, "cf_create: om_mode must be 0 or 1");
  if (c->rd_wm_mode != 0 && c->rd_wm_mode != 1) return fail(CF_ERR_INVALID, "cf_create: rd_wm_mode must be 0 or 1");
  if (c->n_devices < -1 || c->n_devices > 64 || (c->n_devices > 0 && !c->devices))
    return fail(CF_ERR_INVALID, "cf_create: n_devices must be -1 (all), 0 (cf_desc.device) or 1..64 with a devices array");
  if (!(c->probe_limit >= 0.0)) return fail(CF_ERR_INVALID, "cf_create: probe_limit must be >= 0 (0 = default)");
  return 0;
}

// Host-side preparation shared by the replicas of a multi-device handle: the factor is packed (and, for the
// inverse-GEMM solve, inverted in extended precision) once, then uploaded to every device.
struct HostPrep {
  bool packed = false;
  int solve_mode = 0;
  double probe_rel = 0.0;
  cf_host_invpack ip;
  cf_host_pack hp;
};

// Growth-rate data points for growth_kernel: the RK4 step that contains ln a_k, the data in step order (ascending a), and per
// datum in that order a record of CF_FS8_REC doubles:
//   [0..3]  cubic-Hermite weights {h00, h h10, h01, h h11} of ln a_k inside its RK4 step (direct read-out, fs8_n_agrid = 0)
//   [4] a_k   [5] massive-neutrino density at z_k   [6] li   [7] flags (1: the window starts at the grid's first node, 2: ends at its last)
//   [8..23] the same Hermite weights for the FOUR nodes x_0..x_3 of the scripts' logarithmic a-grid around a_k
//   [24..27] x_0..x_3   [28..31] the RK4 steps that contain ln x_0 .. ln x_3
// The scripts do not read delta' at a_k from the ODE solution directly: they sample it on a_span = np.logspace(log10 a_init, 0,
// fs8_n_agrid) and interpolate with interp_pchip (fs8/fs8.py:79-98).  With fs8_n_agrid > 0 the kernel does the same: delta' at
// the four grid nodes around a_k (nodes i - 1 .. i + 2 of the interval x_i < a_k <= x_{i+1}, shifted inside the grid at its
// ends; li = index of x_i in the window), then the Fritsch-Carlson slopes at x_i, x_{i+1} and the cubic of interpolator.py:5-108.
#define CF_FS8_REC 32
static void fs8_hermite_weights(double x0, double hstep, int steps, double ln_a, int& step, double* w) {
  int i = (int)((ln_a - x0) / hstep);
  i = i < 0 ? 0 : (i > steps - 1 ? steps - 1 : i);
  const double t = (ln_a - (x0 + i * hstep)) / hstep, t2 = t * t, t3 = t2 * t;
  step = i;
  w[0] = 2 * t3 - 3 * t2 + 1;
  w[1] = (t3 - 2 * t2 + t) * hstep;
  w[2] = -2 * t3 + 3 * t2;
  w[3] = (t3 - t2) * hstep;
}

static void fs8_points(const cf_dev_desc& d, const double* zs, int n, std::vector<int32_t>& step_of, std::vector<int32_t>& order,
                       std::vector<double>& pts) {
  const double x0 = std::log(d.fs8_a_init), hstep = -x0 / d.fs8_steps;
  std::vector<std::pair<int32_t, int32_t>> so;
  for (int k = 0; k < n; ++k) {
    int i = (int)((std::log(1.0 / (1.0 + zs[k])) - x0) / hstep);
    i = i < 0 ? 0 : (i > d.fs8_steps - 1 ? d.fs8_steps - 1 : i);
    so.emplace_back(i, k);
  }
  std::sort(so.begin(), so.end());
  step_of.resize((size_t)n);
  order.resize((size_t)n);
  pts.assign((size_t)n * CF_FS8_REC, 0.0);
  // np.logspace(lo, 0, na) = 10 ** np.linspace(lo, 0, na): node k at 10 ** (lo + k * (-lo / (na - 1))), the last one at 10 ** 0
  const int na = d.fs8_n_agrid;
  const double lo = std::log10(d.fs8_a_init), dl = na > 1 ? -lo / (na - 1) : 0.0;
  auto node = [&](int k) { return k >= na - 1 ? 1.0 : std::pow(10.0, lo + k * dl); };
  for (int s = 0; s < n; ++s) {
    const int k = order[s] = so[s].second;
    const double zp1 = 1.0 + zs[k], a = 1.0 / zp1;
    double* p = &pts[(size_t)s * CF_FS8_REC];
    int st;
    fs8_hermite_weights(x0, hstep, d.fs8_steps, std::log(a), st, p);
    step_of[s] = st;
    p[4] = a;
    if (d.ez_model == CF_EZ_PHYSICAL) {  // cmb/data_planck_act_compression.py:53-66
      const double r = d.nu_m0 / zp1, mz_sq = r * r;
      double ws = 0.0;
      for (int q = 0; q < 5; ++q) ws += std::sqrt(d.nu_qs_sq[q] + mz_sq) * d.nu_ws[q];
      p[5] = zp1 * zp1 * zp1 * zp1 * ws / d.nu_rho0;
    }
    if (na >= 4) {
      // interval of interp_pchip: i = searchsorted_left(x, a) - 1, clamped into the grid (outside it the interpolant is the end value,
      // which the cubic also returns at t = 0 / t = 1)                                   interpolator.py:80-94
      int i = (int)((std::log10(a) - lo) / dl);
      i = i < 0 ? 0 : (i > na - 2 ? na - 2 : i);
      while (i > 0 && node(i) >= a) --i;
      while (i < na - 2 && node(i + 1) < a) ++i;
      const int j0 = i - 1 < 0 ? 0 : (i - 1 > na - 4 ? na - 4 : i - 1);
      p[6] = (double)(i - j0);
      p[7] = (double)((j0 == 0 ? 1 : 0) | (j0 + 3 == na - 1 ? 2 : 0));
      for (int m = 0; m < 4; ++m) {
        const double x = node(j0 + m);
        int sm;
        fs8_hermite_weights(x0, hstep, d.fs8_steps, x >= 1.0 ? 0.0 : std::log(x), sm, p + 8 + 4 * m);
        p[24 + m] = x;
        p[28 + m] = (double)sm;
      }
    }
  }
}

// The calling thread's current device is the caller's business (a torch process has its own idea of it): every entry point switches
// to the handle's device for its own HIP calls and switches back on the way out.  One hipGetDevice per call when the two agree.
struct DeviceScope {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceScope(int dev) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) cur = -1;
    if (cur != dev) {
      err = hipSetDevice(dev);
      if (err == hipSuccess) prev = cur;
    }
  }
  ~DeviceScope() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};

static int create_one(const cf_desc* c, int device, HostPrep& prep, cf_handle** out) {
  cf_handle* h = new cf_handle();
  auto bail = [&](int code) { cf_destroy(h); return code; };
  h->device = device;
  DeviceScope on_device(h->device);
  if (on_device.err != hipSuccess) return bail(fail(CF_ERR_HIP, "hipSetDevice failed"));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, h->device) == hipSuccess) {
    h->cu_count = prop.multiProcessorCount;
    snprintf(h->arch, sizeof(h->arch), "%s", prop.gcnArchName);
  }
  if (strncmp(h->arch, "gfx950", 6) != 0)
    return bail(fail(CF_ERR_UNSUPPORTED, std::string("cf_create: device is ") + h->arch + ", this library is built for gfx950 only"));
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(CF_ERR_HIP, "hipStreamCreate failed"));

  cf_dev_desc& d = h->d;
  d.ndim = c->ndim;
  d.n_grid = c->n_grid;
  d.ez_model = c->ez_model;
  d.fde = c->fde;
  d.z_max = c->z_max;
  d.step = c->z_max / (double)(c->n_grid - 1);  // np.linspace step
  d.inv_step = 1.0 / d.step;
  d.inv_last = 1.0 / (c->z_max - (double)(c->n_grid - 2) * d.step);
  d.chunk_shift = 3;  // 8 grid nodes per thread of the 512-thread walker kernel, more for grids > 4096
  while ((512 << d.chunk_shift) < c->n_grid) d.chunk_shift++;
  d.c = c->c_km_s;
  for (int s = 0; s < CF_P_NSLOTS; ++s) {
    d.slot[s].idx = c->param[s].idx;
    d.slot[s].scale = c->param[s].scale;
    d.slot[s].fixed = c->param[s].fixed;
  }
  d.n_sn = (int32_t)c->n_sn;
  d.n_pad = (int32_t)((c->n_sn + 15) / 16 * 16);
  d.n_ld = (int32_t)((c->n_sn + 63) / 64 * 64);
  // a velocity step exists if its slot is a parameter or a non-zero constant (with directions: any of the three)
  d.has_vstep = c->param[CF_P_V].idx >= 0 || c->param[CF_P_V].fixed != 0.0;
  if (c->sn_dir)
    for (int s : {CF_P_V2, CF_P_V3}) d.has_vstep = d.has_vstep || c->param[s].idx >= 0 || c->param[s].fixed != 0.0;
  d.om_mode = c->om_mode;
  d.lin_in_rec = c->sn_lin_coef && !d.has_vstep;
  d.or_h2 = c->or_h2;
  d.omnu_h2 = c->omnu_h2;
  d.o_gamma_h2 = c->o_gamma_h2;
  d.nu_m0 = c->nu_m0;
  d.nu_rho0 = c->nu_rho0;
  for (int i = 0; i < 5; ++i) {
    d.nu_qs_sq[i] = c->nu_qs_sq[i];
    d.nu_ws[i] = c->nu_ws[i];
  }
  d.n_bao = c->n_bao;
  d.bao_dh_exact = c->bao_dh_mode == CF_BAO_DH_EXACT;
  d.rd_from_fit = c->rd_mode == CF_RD_FIT;
  d.rd_wm_late = c->rd_wm_mode == 1;
  for (int i = 0; i < 11; ++i) {
    d.rd_fit[i] = c->rd_fit[i];
    d.zstar_fit[i] = c->zstar_fit[i];
  }
  d.cmb_mode = c->cmb_mode;
  d.n_gl = c->cmb_mode ? c->n_gl : 0;
  for (int i = 0; i < 3; ++i) d.cmb_prior[i] = c->cmb_prior[i];
  for (int i = 0; i < 9; ++i) d.cmb_inv_cov[i] = c->cmb_inv_cov[i];
  d.n_cc = c->n_cc;
  d.cc_logdet = c->cc_logdet;
  h->has_small_blocks = c->n_bao > 0 || c->cmb_mode != CF_CMB_NONE || c->n_cc > 0;
  h->has_growth = c->n_fs8 > 0;
  d.n_fs8 = c->n_fs8;
  // 256 lanes per walker, 1 / 2 / 4 / 8 steps per lane: the request rounded up to 256, 512, 1024 (default), 2048.  Measured against
  // the scripts' own equation integrated to 1e-12 (tools/fs8_parity_probe.py, profiles/r03_fs8_parity.txt): 512 steps leave 4e-9 on
  // the theory and 1e-8 on chi^2, 2048 steps 5e-11 / 5e-11; the error goes as steps^-4
  d.fs8_steps = 256;
  while (d.fs8_steps < (c->fs8_steps > 0 ? c->fs8_steps : 1024)) d.fs8_steps *= 2;
  d.fs8_a_init = c->fs8_a_init;
  d.fs8_n_agrid = c->fs8_n_agrid;
  d.n_aux = c->n_bao + c->n_fs8;
  d.cpl_wall = c->cpl_wall;
  d.logl_const = c->logl_const;
  d.sn_vel_mult = c->sn_vel_mode == 1;
  d.cc_f_inverse = c->cc_f_mode == 1;
  d.has_bounds = c->bounds != nullptr;
  d.log_norm = 0.0;
  if (c->bounds) {
    double s = 0.0;
    for (int k = 0; k < c->ndim; ++k) {
      d.lo[k] = c->bounds[2 * k];
      d.hi[k] = c->bounds[2 * k + 1];
      if (!(d.hi[k] > d.lo[k])) return bail(fail(CF_ERR_INVALID, "cf_create: bounds must satisfy lo < hi"));
      s += std::log(d.hi[k] - d.lo[k]);  // normalization = -sum(log(hi-lo)), sn/pantheon.py:77
    }
    d.log_norm = c->prior_norm_mode == 1 ? 0.0 : -s;  // ohd/cc_cmb.py:70-73 returns 0.0 inside the box
  }
  d.n_gauss = c->n_gauss;
  for (int g = 0; g < c->n_gauss; ++g) {
    if (c->gauss[g].idx < 0 || c->gauss[g].idx >= c->ndim) return bail(fail(CF_ERR_INVALID, "cf_create: gauss idx"));
    d.gauss_idx[g] = c->gauss[g].idx;
    d.gauss_mean[g] = c->gauss[g].mean;
    d.gauss_sigma[g] = c->gauss[g].sigma;
  }
  d.n_chi2_gauss = c->n_chi2_gauss;
  for (int g = 0; g < c->n_chi2_gauss; ++g) {
    if (c->chi2_gauss[g].idx < 0 || c->chi2_gauss[g].idx >= c->ndim)
      return bail(fail(CF_ERR_INVALID, "cf_create: chi2_gauss idx"));
    d.chi2_gauss_idx[g] = c->chi2_gauss[g].idx;
    d.chi2_gauss_mean[g] = c->chi2_gauss[g].mean;
    d.chi2_gauss_sigma[g] = c->chi2_gauss[g].sigma;
  }

  if (c->n_sn > 0) {
    std::vector<double> step((size_t)c->n_sn);
    d.step_pm1 = 1;
    for (int64_t i = 0; i < c->n_sn; ++i) {
      step[i] = c->sn_step ? c->sn_step[i] : (c->sn_z_cmb[i] <= c->sn_z_turn ? 1.0 : -1.0);  // sn/pantheon.py:46
      if (step[i] != 1.0 && step[i] != -1.0) d.step_pm1 = 0;
    }
    int rc;
    if ((rc = upload_vec(h->z_cmb, c->sn_z_cmb, c->n_sn))) return bail(rc);
    if ((rc = upload_vec(h->z_hel, c->sn_z_hel, c->n_sn))) return bail(rc);
    if ((rc = upload_vec(h->obs, c->sn_obs, c->n_sn))) return bail(rc);
    if ((rc = upload_vec(h->sn_step, step.data(), c->n_sn))) return bail(rc);
    d.z_cmb = h->z_cmb.as<const double>();
    d.z_hel = h->z_hel.as<const double>();
    d.obs = h->obs.as<const double>();
    d.sn_step = h->sn_step.as<const double>();
    {
      // one record per SN for the production loop; spare records so that its look-ahead (one stride of at most CF_SN_PARTS_MAX x
      // 512 records) needs no bounds check
      std::vector<cf_d4> rec((size_t)d.n_ld + CF_SN_REC_SLACK, cf_d4{1.0, 1.0, 1.0, 0.0});
      for (int64_t i = 0; i < c->n_sn; ++i)
        rec[i] = cf_d4{d.has_vstep ? 1.0 + c->sn_z_cmb[i] : c->sn_z_cmb[i], d.lin_in_rec ? c->sn_lin_coef[i] : step[i],
                       1.0 + c->sn_z_hel[i], c->sn_obs[i]};
      if (h->sn_rec.ensure(rec.size() * sizeof(cf_d4))) return bail(CF_ERR_HIP);
      if (hipMemcpy(h->sn_rec.p, rec.data(), rec.size() * sizeof(cf_d4), hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(CF_ERR_HIP, "hipMemcpy(sn_rec) failed"));
      d.sn_rec = h->sn_rec.as<const cf_d4>();
      int rc2;
      if ((rc2 = upload_log10_table(h->log10_tab))) return bail(rc2);
      d.log10_tab = h->log10_tab.p;
    }
    for (int64_t i = 0; i < c->n_sn; ++i) {
      const double piv = c->sn_chol[i * c->sn_chol_ld + i];
      if (!(piv > 0.0) || !std::isfinite(piv))
        return bail(fail(CF_ERR_NOT_POSDEF, "cf_create: the Cholesky factor has a non-positive or non-finite pivot"));
    }
    if (!prep.packed) {
      const double limit = c->probe_limit > 0.0 ? c->probe_limit : CF_PROBE_LIMIT;
      prep.solve_mode = c->solve_mode;
      if (c->solve_mode != CF_SOLVE_BLOCKED_TRSM) {
        cf_pack_inverse(c->sn_chol, c->n_sn, c->sn_chol_ld, prep.ip);
        const double inv_probe = cf_invpack_probe(prep.ip, c->sn_chol, c->sn_chol_ld);
        if (inv_probe <= limit) {
          prep.solve_mode = CF_SOLVE_INVERSE_GEMM;
          prep.probe_rel = inv_probe;
        } else if (c->solve_mode == CF_SOLVE_INVERSE_GEMM) {
          return bail(fail(CF_ERR_ILL_CONDITIONED, "cf_create: the explicit inverse of the factor disagrees with row-by-row forward "
                                                       "substitution by " + fmt_g(inv_probe) + " relative (limit " +
                                                       fmt_g(limit) + "); use CF_SOLVE_BLOCKED_TRSM or CF_SOLVE_AUTO for "
                                                       "this covariance"));
        } else {
          prep.solve_mode = CF_SOLVE_BLOCKED_TRSM;  // CF_SOLVE_AUTO falls back to forward substitution by blocks
          prep.ip = cf_host_invpack();
        }
      }
      if (prep.solve_mode == CF_SOLVE_BLOCKED_TRSM) {
        if (pack_default(c->sn_chol, c->n_sn, c->sn_chol_ld, prep.hp, &prep.probe_rel) != 0)
          return bail(fail(CF_ERR_NOT_POSDEF, "cf_create: the Cholesky factor has a non-positive or non-finite pivot"));
        // the blocked solve inverts only 256-row diagonal blocks: its limit is the library's, not the caller's
        if (!(prep.probe_rel <= CF_PROBE_LIMIT))
          return bail(fail(CF_ERR_ILL_CONDITIONED, "cf_create: the blocked solve disagrees with row-by-row forward substitution by " +
                                                       fmt_g(prep.probe_rel) + " relative on a probe vector (limit 1e-11): "
                                                       "the factor's diagonal blocks are too ill-conditioned for 256-row block inverses"));
      }
      prep.packed = true;
    }
    h->solve_mode = prep.solve_mode;
    h->pack_probe_rel = prep.probe_rel;
    if (h->solve_mode == CF_SOLVE_INVERSE_GEMM) {
      if ((rc = h->ipack.upload(prep.ip))) return bail(rc);
    } else {
      if ((rc = h->pack.upload(prep.hp))) return bail(rc);
    }
  }
  if (c->n_bao > 0) {
    int rc;
    if ((rc = upload_vec(h->bao_z, c->bao_z, c->n_bao))) return bail(rc);
    if ((rc = upload_vec(h->bao_val, c->bao_val, c->n_bao))) return bail(rc);
    if ((rc = upload_vec(h->bao_inv_cov, c->bao_inv_cov, (int64_t)c->n_bao * c->n_bao))) return bail(rc);
    if (h->bao_qty.ensure((size_t)c->n_bao * 4)) return bail(CF_ERR_HIP);
    if (hipMemcpy(h->bao_qty.p, c->bao_qty, (size_t)c->n_bao * 4, hipMemcpyHostToDevice) != hipSuccess)
      return bail(fail(CF_ERR_HIP, "hipMemcpy(bao_qty) failed"));
    d.bao_z = h->bao_z.as<const double>();
    d.bao_val = h->bao_val.as<const double>();
    d.bao_inv_cov = h->bao_inv_cov.as<const double>();
    d.bao_qty = h->bao_qty.as<const int32_t>();
  }
  if (d.n_aux > 0) {
    // interval of each BAO / growth-rate redshift on the grid, by the arithmetic of hermite_tab (cosmofit_kernels.hip); the
    // copy of CF_BAO_NODES nodes starts two nodes below it, so that the kernel's own interval search, the Hermite pair
    // and the two 3-point PCHIP stencils all stay inside the copy
    std::vector<int32_t> base((size_t)d.n_aux);
    for (int k = 0; k < d.n_aux; ++k) base[k] = aux_node_base(d, k < c->n_bao ? c->bao_z[k] : c->fs8_z[k - c->n_bao]);
    if (h->bao_base.ensure(base.size() * 4)) return bail(CF_ERR_HIP);
    if (hipMemcpy(h->bao_base.p, base.data(), base.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
      return bail(fail(CF_ERR_HIP, "hipMemcpy(bao_base) failed"));
    d.bao_base = h->bao_base.as<const int32_t>();
  }
  if (c->n_fs8 > 0) {
    int rc;
    const int n = c->n_fs8;
    if ((rc = upload_vec(h->fs8_z, c->fs8_z, n))) return bail(rc);
    if ((rc = upload_vec(h->fs8_val, c->fs8_val, n))) return bail(rc);
    if ((rc = upload_vec(h->fs8_inv_cov, c->fs8_inv_cov, (int64_t)n * n))) return bail(rc);
    if ((rc = upload_vec(h->fs8_fid, c->fs8_fid, n))) return bail(rc);
    d.fs8_z = h->fs8_z.as<const double>();
    d.fs8_val = h->fs8_val.as<const double>();
    d.fs8_inv_cov = h->fs8_inv_cov.as<const double>();
    d.fs8_fid = h->fs8_fid.as<const double>();
    // RK4 step that contains ln a_k, the data in step order (ascending a), their Hermite weights
    const double x0 = std::log(c->fs8_a_init), hstep = -x0 / d.fs8_steps;
    std::vector<int32_t> step_of, order;
    std::vector<double> pts;
    fs8_points(d, c->fs8_z, n, step_of, order, pts);
    if (h->fs8_step_of.ensure((size_t)n * 4) || h->fs8_order.ensure((size_t)n * 4)) return bail(CF_ERR_HIP);
    if (hipMemcpy(h->fs8_step_of.p, step_of.data(), (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->fs8_order.p, order.data(), (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess)
      return bail(fail(CF_ERR_HIP, "hipMemcpy(fs8 order) failed"));
    d.fs8_step_of = h->fs8_step_of.as<const int32_t>();
    d.fs8_order = h->fs8_order.as<const int32_t>();
    // what the ODE's coefficients need that does not depend on theta, at the 2 S + 1 boundaries and midpoints of the steps:
    // a = exp(x), 1 + z, the massive-neutrino density ratio and nu 3 (1 + w_nu) / (1 + z)   cmb/data_planck_act_compression.py:53-83
    const int npts = 2 * d.fs8_steps + 1;
    std::vector<double> tab((size_t)(2 * (d.fs8_steps / 256)) * 257 * 4, 0.0);
    for (int m = 0; m < npts; ++m) {
      const double a = m == npts - 1 ? 1.0 : std::exp(x0 + m * (0.5 * hstep)), zp1 = 1.0 / a;
      double nu = 0.0, dnu = 0.0;
      if (c->ez_model == CF_EZ_PHYSICAL) {
        const double r = c->nu_m0 / zp1, mz_sq = r * r;
        double ws = 0.0, num = 0.0, den = 0.0;
        for (int i = 0; i < 5; ++i) {
          const double f = std::sqrt(c->nu_qs_sq[i] + mz_sq);
          ws += f * c->nu_ws[i];
          num += c->nu_ws[i] / f;
          den += c->nu_ws[i] * f;
        }
        nu = zp1 * zp1 * zp1 * zp1 * ws / c->nu_rho0;
        const double w_nu = (1.0 / 3) - (1.0 / 3) * mz_sq * num / den;
        dnu = nu * 3 * (1.0 + w_nu) / zp1;
      }
      // stored in the order the 256 lanes of growth_kernel fetch it: point m at [(m % 2C) * 257 + m / 2C], C = steps / 256
      const int c2 = 2 * (d.fs8_steps / 256);
      const size_t at = (size_t)(m % c2) * 257 + (size_t)(m / c2);
      tab[4 * at] = a; tab[4 * at + 1] = zp1; tab[4 * at + 2] = nu; tab[4 * at + 3] = dnu;
    }
    if ((rc = upload_vec(h->fs8_tab, tab.data(), (int64_t)tab.size()))) return bail(rc);
    d.fs8_tab = h->fs8_tab.as<const double>();
    if ((rc = upload_vec(h->fs8_pts, pts.data(), (int64_t)pts.size()))) return bail(rc);
    d.fs8_pts = h->fs8_pts.as<const double>();
  }
  if (c->ez_model == CF_EZ_PHYSICAL) {
    // massive-neutrino density at the grid nodes, cmb/data_planck_act_compression.py:53-66 -- independent of theta
    std::vector<double> nu((size_t)c->n_grid);
    for (int g = 0; g < c->n_grid; ++g) {
      const double z = g == c->n_grid - 1 ? c->z_max : (double)g * d.step;
      const double zp1 = 1.0 + z, r = c->nu_m0 / zp1, mz_sq = r * r;
      const double ws = std::sqrt(c->nu_qs_sq[0] + mz_sq) * c->nu_ws[0] + std::sqrt(c->nu_qs_sq[1] + mz_sq) * c->nu_ws[1] +
                        std::sqrt(c->nu_qs_sq[2] + mz_sq) * c->nu_ws[2] + std::sqrt(c->nu_qs_sq[3] + mz_sq) * c->nu_ws[3] +
                        std::sqrt(c->nu_qs_sq[4] + mz_sq) * c->nu_ws[4];
      const double zp1_2 = zp1 * zp1;
      nu[g] = zp1_2 * zp1_2 * ws / c->nu_rho0;
    }
    int rc;
    if ((rc = upload_vec(h->nu_grid, nu.data(), c->n_grid))) return bail(rc);
    d.nu_grid = h->nu_grid.as<const double>();
    if (d.chunk_shift == 3) {
      const std::vector<double> sw = swizzle_for_walker_threads(nu);
      if ((rc = upload_vec(h->nu_sw, sw.data(), (int64_t)sw.size()))) return bail(rc);
      d.nu_sw = h->nu_sw.as<const double>();
    }
  }
  if (c->fde == CF_FDE_WCDM || c->fde == CF_FDE_CPL) {
    // ln(1 + z) at the grid nodes, correctly rounded from extended precision: the power-law dark-energy forms become one exp
    std::vector<double> ln((size_t)c->n_grid);
    for (int g = 0; g < c->n_grid; ++g) {
      const double z = g == c->n_grid - 1 ? c->z_max : (double)g * d.step;
      ln[g] = (double)log1pl((long double)z);
    }
    int rc;
    if ((rc = upload_vec(h->ln_grid, ln.data(), c->n_grid))) return bail(rc);
    d.ln_grid = h->ln_grid.as<const double>();
    if (d.chunk_shift == 3) {
      const std::vector<double> sw = swizzle_for_walker_threads(ln);
      if ((rc = upload_vec(h->ln_sw, sw.data(), (int64_t)sw.size()))) return bail(rc);
      d.ln_sw = h->ln_sw.as<const double>();
      // 2^(j/64), correctly rounded from extended precision: reduction table of the table build's exp (exp_tab)
      if ((rc = upload_exp2_table(h->exp2_tab))) return bail(rc);
      d.exp2_tab = h->exp2_tab.as<const double>();
    }
  }
  if (c->cmb_mode != CF_CMB_NONE && (c->fde == CF_FDE_WCDM || c->fde == CF_FDE_CPL)) {
    // the dark-energy factor at the Gauss-Legendre nodes of the CMB distances goes through exp_tab / log10_tab (H_at_gl_node)
    int rc;
    if (!d.exp2_tab) {
      if ((rc = upload_exp2_table(h->exp2_tab))) return bail(rc);
      d.exp2_tab = h->exp2_tab.as<const double>();
    }
    if (!d.log10_tab) {
      if ((rc = upload_log10_table(h->log10_tab))) return bail(rc);
      d.log10_tab = h->log10_tab.p;
    }
  }
  if (c->n_cc > 0) {
    int rc;
    if ((rc = upload_vec(h->cc_z, c->cc_z, c->n_cc))) return bail(rc);
    if ((rc = upload_vec(h->cc_h, c->cc_h, c->n_cc))) return bail(rc);
    if ((rc = upload_vec(h->cc_inv_cov, c->cc_inv_cov, (int64_t)c->n_cc * c->n_cc))) return bail(rc);
    d.cc_z = h->cc_z.as<const double>();
    d.cc_h = h->cc_h.as<const double>();
    d.cc_inv_cov = h->cc_inv_cov.as<const double>();
  }
  if (c->n_sn > 0 && c->sn_fixed_mu) {
    int rc;
    if ((rc = upload_vec(h->fixed_mu, c->sn_fixed_mu, c->n_sn))) return bail(rc);
    d.sn_fixed_mu = h->fixed_mu.as<const double>();
  }
  if (c->n_sn > 0 && c->sn_lin_coef) {
    int rc;
    if ((rc = upload_vec(h->sn_lin, c->sn_lin_coef, c->n_sn))) return bail(rc);
    d.sn_lin = h->sn_lin.as<const double>();
  }
  if (c->n_sn > 0 && c->sn_dir) {
    int rc;
    if ((rc = upload_vec(h->sn_dir, c->sn_dir, 3 * c->n_sn))) return bail(rc);
    d.sn_dir = h->sn_dir.as<const double>();
  }
  if (c->cmb_mode != CF_CMB_NONE) {
    int rc;
    if ((rc = upload_vec(h->gl_x, c->gl_x, c->n_gl))) return bail(rc);
    if ((rc = upload_vec(h->gl_w, c->gl_w, c->n_gl))) return bail(rc);
    d.gl_x = h->gl_x.as<const double>();
    d.gl_w = h->gl_w.as<const double>();
  }
  if (h->nonfinite.ensure(8)) return bail(CF_ERR_HIP);
  if (hipMemset(h->nonfinite.p, 0, 8) != hipSuccess) return bail(fail(CF_ERR_HIP, "hipMemset failed"));
  // the register path of walker_kernel's table build (chunk_shift == 3) reads its theta-independent tables without a
  // run-time fallback: a descriptor that needs one and does not carry it must never reach a launch
  if (d.chunk_shift == 3 && ((d.ez_model == CF_EZ_PHYSICAL_D && !d.nu_sw) ||
                             ((d.fde == CF_FDE_WCDM_D || d.fde == CF_FDE_CPL_D) && (!d.ln_sw || !d.exp2_tab))))
    return bail(fail(CF_ERR_INVALID, "cf_create: internal: the table build's neutrino / ln(1 + z) tables are missing"));
  // the prior / output epilogue's view of the descriptor: by value for finalize_kernel, through its device copy for the solve kernels
  h->epi_host = epilogue_of(d);
  if (h->epi.ensure(sizeof(cf_epilogue))) return bail(CF_ERR_HIP);
  if (hipMemcpy(h->epi.p, &h->epi_host, sizeof(cf_epilogue), hipMemcpyHostToDevice) != hipSuccess)
    return bail(fail(CF_ERR_HIP, "hipMemcpy of the epilogue block failed"));
  *out = h;
  return CF_OK;
}

static int eval_host_single(cf_handle* h, const double* theta, int64_t W, double* out, int out_kind, bool worker_thread = false);

// Worker thread of one replica: waits for a slice, evaluates it on its device, reports the status.
static void worker_main(cf_handle* replica, cf_worker* w) {
  (void)hipSetDevice(replica->device);  // this thread's device for its whole life: the evaluations' DeviceScope then has nothing to do
  std::unique_lock<std::mutex> lk(w->m);
  for (;;) {
    w->cv.wait(lk, [&] { return w->has_job || w->quit; });
    if (w->quit) return;
    w->has_job = false;
    lk.unlock();
    const int rc = eval_host_single(replica, w->theta, w->W, w->out, w->out_kind, true);
    const std::string err = rc ? g_err : std::string();
    lk.lock();
    w->rc = rc;
    w->err = err;
    w->done = true;
    w->cv.notify_all();
  }
}

extern "C" int cf_create(const cf_desc* c, cf_handle** out) {
  if (!c || !out) return fail(CF_ERR_INVALID, "cf_create: null argument");
  *out = nullptr;
  int rc = validate_desc(c);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(CF_ERR_NO_DEVICE, "cf_create: no HIP device visible (this library has no CPU path)");
  std::vector<int> devs;
  if (c->n_devices == 0) devs.push_back(c->device);
  else if (c->n_devices < 0) for (int i = 0; i < ndev; ++i) devs.push_back(i);
  else devs.assign(c->devices, c->devices + c->n_devices);
  for (int dv : devs)
    if (dv < 0 || dv >= ndev) return fail(CF_ERR_INVALID, "cf_create: device ordinal out of range");
  HostPrep prep;
  cf_handle* primary = nullptr;
  if ((rc = create_one(c, devs[0], prep, &primary))) return rc;
  for (size_t k = 1; k < devs.size(); ++k) {
    cf_handle* r = nullptr;
    if ((rc = create_one(c, devs[k], prep, &r))) {
      const std::string keep = g_err;
      cf_destroy(primary);
      return fail(rc, keep);
    }
    primary->peers.push_back(r);
    primary->workers.emplace_back(new cf_worker());
    cf_worker* w = primary->workers.back().get();
    w->th = std::thread(worker_main, r, w);
  }
  *out = primary;
  return CF_OK;
}

extern "C" void cf_destroy(cf_handle* h) {
  if (!h) return;
  for (auto& w : h->workers) {
    {
      std::lock_guard<std::mutex> lk(w->m);
      w->quit = true;
    }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
  }
  for (cf_handle* r : h->peers) cf_destroy(r);
  DeviceScope on_device(h->device);
  (void)hipDeviceSynchronize();  // evaluations launched on callers' streams may still use the workspace
  for (auto& e : h->ev)
    if (e) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

extern "C" int cf_get_info(cf_handle* h, cf_info* info) {
  if (!h || !info) return fail(CF_ERR_INVALID, "cf_get_info: null argument");
  std::lock_guard<std::mutex> lk(h->mu);
  memset(info, 0, sizeof(*info));
  info->n_sn = h->d.n_sn;
  info->n_sn_pad = h->d.n_pad;
  info->packed_chol_bytes = h->pack.bytes + h->ipack.bytes;
  info->solve_mode = h->solve_mode;
  info->workspace_bytes = (int64_t)(h->theta.bytes + h->out.bytes + h->delta.bytes + h->ypk.bytes + h->partial.bytes +
                                    h->arrivals.bytes + h->bao_nodes.bytes + h->chi2_extra.bytes);
  info->max_walkers = h->max_walkers;
  info->device = h->device;
  info->cu_count = h->cu_count;
  info->pack_probe_rel = h->pack_probe_rel;
  info->n_devices = 1 + (int32_t)h->peers.size();
  info->devices[0] = h->device;
  for (size_t k = 0; k < h->peers.size() && k + 1 < 16; ++k) info->devices[k + 1] = h->peers[k]->device;
  snprintf(info->gcn_arch, sizeof(info->gcn_arch), "%s", h->arch);
  unsigned long long nf = 0;
  DeviceScope on_device(h->device);
  HIP_TRY(on_device.err);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(&nf, h->nonfinite.p, 8, hipMemcpyDeviceToHost));
  info->nonfinite_count = (int64_t)nf;
  return CF_OK;
}

extern "C" int cf_enable_timing(cf_handle* h, int slots) {
  if (!h) return fail(CF_ERR_INVALID, "cf_enable_timing: null handle");
  if (slots < 0 || slots > 4096) return fail(CF_ERR_INVALID, "cf_enable_timing: slots must be in 0..4096");
  std::lock_guard<std::mutex> lk(h->mu);
  DeviceScope on_device(h->device);
  HIP_TRY(on_device.err);
  while ((int)h->ev.size() < 4 * slots) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    h->ev.push_back(e);
  }
  h->timing_slots = slots;
  h->timed_calls = 0;
  h->eval_calls = 0;  // the sampling phase restarts with the ring
  return CF_OK;
}

// Sample the kernel timing: events are recorded on every `stride`-th evaluation only (each evaluation with events pays for
// four event records on its stream; sampled timing keeps the timed loop undisturbed).
extern "C" int cf_set_timing_stride(cf_handle* h, int stride) {
  if (!h || stride < 1) return fail(CF_ERR_INVALID, "cf_set_timing_stride: stride must be >= 1");
  std::lock_guard<std::mutex> lk(h->mu);
  h->timing_stride = stride;
  h->eval_calls = 0;
  return CF_OK;
}

extern "C" int64_t cf_timed_calls(cf_handle* h) { return h ? h->timed_calls : 0; }

// t[3] = {walker_kernel, small-block + growth kernels, solve kernel (or the bare epilogue)} of timed evaluation `call`
extern "C" int cf_kernel_ms3(cf_handle* h, int64_t call, float t[3]) {
  if (!h || !t) return fail(CF_ERR_INVALID, "cf_kernel_ms3: null argument");
  if (h->timing_slots == 0 || call < 0 || call >= h->timed_calls || call < h->timed_calls - h->timing_slots)
    return fail(CF_ERR_INVALID, "cf_kernel_ms3: that call is not in the timing ring");
  const int slot = (int)(call % h->timing_slots);
  hipEvent_t* e = &h->ev[4 * slot];
  HIP_TRY(hipEventSynchronize(e[3]));
  for (int k = 0; k < 3; ++k) HIP_TRY(hipEventElapsedTime(&t[k], e[k], e[k + 1]));
  return CF_OK;
}

// t[2] = {per-walker kernels (walker_kernel + small blocks + growth), solve kernel}
extern "C" int cf_kernel_ms(cf_handle* h, int64_t call, float t[2]) {
  float t3[3];
  if (!t) return fail(CF_ERR_INVALID, "cf_kernel_ms: null argument");
  const int rc = cf_kernel_ms3(h, call, t3);
  if (rc) return rc;
  t[0] = t3[0] + t3[1];
  t[1] = t3[2];
  return CF_OK;
}

extern "C" int cf_last_kernel_ms(cf_handle* h, float t[2]) {
  if (!h || !t) return fail(CF_ERR_INVALID, "cf_last_kernel_ms: null argument");
  if (h->timed_calls == 0) return fail(CF_ERR_INVALID, "cf_last_kernel_ms: call cf_enable_timing(h, slots>0) before evaluating");
  return cf_kernel_ms(h, h->timed_calls - 1, t);
}

// Blocked solve (fallback): one 512-thread workgroup per 16-walker panel, trsm_chi2_kernel<2, 4>.
static int launch_trsm(const cf_epilogue* epi, int n_pad, int n_ld, int ndim, const cf_dev_pack& pk, const double* d_theta, int64_t W,
                       const double* delta, d2* ypk, const double* chi2_extra, double* d_out, int out_kind, unsigned long long* nf,
                       hipStream_t st, double* chi2_sn_out = nullptr) {
  if (pk.ksplit != 2 || pk.tclasses != 4) return fail(CF_ERR_INVALID, "bad solve shape");
  const size_t lds = (size_t)2 * (CF_BLOCK_ROWS / 8 * 64) * sizeof(d2);
  static thread_local int attr_device = -1;  // > 64 KB of dynamic LDS must be allowed once per device
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (lds > 64 * 1024 && attr_device != dev) {
    HIP_TRY(hipFuncSetAttribute((const void*)(&trsm_chi2_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_device = dev;
  }
  const unsigned panels = (unsigned)((W + 15) / 16);
  hipLaunchKernelGGL((trsm_chi2_kernel<2, 4>), dim3(panels), dim3(512), lds, st, epi, n_pad, n_ld, ndim, pk, d_theta, W, delta, ypk, chi2_extra, d_out, out_kind, nf, chi2_sn_out);
  return 0;
}

// Inverse-GEMM solve.  Panels of 16 NP walkers per unit of work: NP = 2 halves the factor traffic per flop and is the throughput
// shape; NP = 1 keeps the units of a batch of <= 512 walkers short.
struct TriGemmArgs {
  const cf_epilogue* epi;
  const d2* frags;
  int n_ld, ndim, n_rb;
  int nt_last;  // 16-row tiles of the last row block that hold rows of the factor (the others are padding)
  int diag_skip;  // 1: wave 3 skips the all-zero tiles of the diagonal block's K-step pairs
  const double* theta;
  int64_t W;
  const double* delta;
  int64_t w_pad;
  double* partial;
  unsigned int* arrivals;
  const double* chi2_extra;
  double* out;
  int out_kind;
  unsigned long long* nonfinite;
  double* chi2_sn_out;
  double* partial4;
  unsigned long long* done_flag;  // pinned host words the last arrivers set to done_seq, or null
  unsigned long long done_seq;
  bool frag_b;  // `delta` holds the panels' residuals in the small-batch kernel's fragment order (walker_fast_kernel, frag_b)
};

template <int NP, int PF>
static int launch_tri_gemm_t(const TriGemmArgs& a, hipStream_t st) {
  const int panels = (int)((a.W + 16 * NP - 1) / (16 * NP));
  // Panel groups: the grid runs group by group (a group = `ppg` panels x all row blocks), so that a group's residual rows
  // stay in the 256 MB Infinity Cache while its 27 row blocks pass over them.  Up to 8192 walkers (256 panels of 32: 113 MB
  // of residual rows) one group is best (W = 4096: one group 226 us, two 229 us, four 249 us -- the factor streams are
  // re-read per group); beyond that the rows no longer fit and every row block would stream them from HBM again
  // (W = 65536: 0.9 GB x 14 passes), so larger batches run in groups of 128 panels = 4096 walkers (measured at W = 65536,
  // profiles/r02_panel_groups.txt: one group 3.66 ms = 0.66 of peak, groups of 256 panels 3.05 ms = 0.79, of 128 panels
  // 2.96 ms = 0.82).  CF_TUNE gemm_group=<panels> overrides (multiples of 8 so that a panel stays on one XCD; 0 = one group).
  static const int env_group = (int)cf_tune("gemm_group", -1);
  const int max_group = env_group >= 0 ? env_group : (panels > 256 ? 128 : 0);
  int ppg = panels;
  if (max_group > 0 && panels > max_group) {
    const int n_groups = (panels + max_group - 1) / max_group;
    ppg = ((panels + n_groups - 1) / n_groups + 7) / 8 * 8;
  }
  const int n_groups = (panels + ppg - 1) / ppg;
  // the lowest row blocks of a 32-walker panel as two 16-walker units each (the kernel: HALF UNITS).  Measured with the final kernel
  // over 0 / 6 / 12 levels (profiles/r04_half_units_final_kernel_raw.txt): 6 levels gain 2 % at 1024-1536 walkers (0.643 -> 0.654,
  // 0.698 -> 0.715), nothing at 640-768 and 2048, and cost 3 % at 3072 (the factor fragments feed one panel instead of two; every
  // stage of a half unit runs the guarded form).  CF_TUNE gemm_split=<levels> forces a number.
  static const int split_env = (int)cf_tune("gemm_split", -1);
  const int split_levels = NP == 2 ? std::max(0, std::min(a.n_rb - 1, split_env >= 0 ? split_env : (panels >= 28 && panels <= 56 ? 6 : 0))) : 0;
  const int n_wgs = n_groups * ppg * (a.n_rb + split_levels);
  // order of the row blocks inside the grid: descending; for a grid that is resident all at once, alternate blocks of 256 workgroups
  // ascending (see the kernel).  CF_TUNE gemm_order=0|1 forces one.
  static const int order_env = (int)cf_tune("gemm_order", -1);
  const int snake = order_env >= 0 ? order_env : (n_wgs <= 1024 + 256 ? 1 : 0);  // (a few half units beyond the resident 1024 change nothing)
  hipLaunchKernelGGL((tri_gemm_chi2_kernel<NP, PF>), dim3((unsigned)n_wgs), dim3(256), 0, st, a.epi, a.frags, a.n_ld, a.ndim, a.n_rb, a.theta,
                     a.W, a.delta, a.w_pad, a.partial, a.arrivals, a.chi2_extra, a.out, a.out_kind, a.nonfinite, a.chi2_sn_out, ppg, snake,
                     a.nt_last, a.diag_skip, split_levels, a.done_flag, a.done_seq);
  return 0;
}

// Small batches: one workgroup per (panel, row block, 16-row tile), see tri_gemm_small_kernel.
template <int PF, bool FRAG, int TPW>
static int launch_tri_gemm_small_t(const TriGemmArgs& a, hipStream_t st) {
  const int panels = (int)((a.W + 15) / 16);
  const int units_pad = ((4 / TPW) * a.n_rb + 7) / 8 * 8;
  // dynamic LDS: the last arriver's shares + row-block sums -- padded so that at most TWO workgroups fit a CU.  With several
  // panels the grid outnumbers the CUs, and left to itself the dispatcher stacks three or four of these bandwidth-bound workgroups
  // on some CUs while others hold one (75 walkers: 33.7-34.6 -> 31.8-32.2 us per call, 150 walkers with two tiles per workgroup:
  // 44 -> 39 us; profiles/r03_small_batch_solve.txt).  CF_TUNE small_lds_cap=0 turns the padding off.
  static const bool cap2 = cf_tune("small_lds_cap", 1) != 0;
  // the kernel's static LDS and the CU's LDS size as the runtime reports them; 0 = a query failed -> no padding (a guess could push
  // static + dynamic LDS past the 64 KB a workgroup may have, which would only show as a launch failure)
  static const size_t static_lds = [] {
    hipFuncAttributes fa;
    return hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&tri_gemm_small_kernel<PF, FRAG, TPW>)) == hipSuccess ? fa.sharedSizeBytes
                                                                                                                          : (size_t)0;
  }();
  static const size_t cu_lds = [] {
    int dev = 0;
    hipDeviceProp_t pr;
    return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? (size_t)pr.maxSharedMemoryPerMultiProcessor
                                                                                                 : (size_t)0;
  }();
  size_t dyn_lds = 64 * a.n_rb <= 4096 ? (size_t)80 * a.n_rb * 8 : 0;
  const size_t third_of_cu = cu_lds / 3 + 1024;  // a third of a CU's LDS and a little: three workgroups no longer fit
  if (cap2 && static_lds > 0 && cu_lds > 0 && static_lds + dyn_lds < third_of_cu && third_of_cu <= 64 * 1024)
    dyn_lds = third_of_cu - static_lds;
  hipLaunchKernelGGL((tri_gemm_small_kernel<PF, FRAG, TPW>), dim3((unsigned)(panels * units_pad)), dim3(256), dyn_lds, st, a.epi, a.frags, a.n_ld,
                     a.ndim, a.n_rb, a.theta, a.W, a.delta, a.partial4, a.arrivals, a.chi2_extra, a.out, a.out_kind, a.nonfinite,
                     a.chi2_sn_out, units_pad, a.done_flag, a.done_seq);
  return 0;
}

// largest batch of the small-batch solve kernel (CF_TUNE small_max=<walkers>; 0 = never; at most 16 x CF_SMALL_MAX_PANELS)
static int64_t small_batch_max() {
  static const int64_t v = [] {
    const long long m = cf_tune("small_max", CF_SMALL_DEFAULT);
    return (int64_t)(m < 0 ? 0 : (m > 16 * CF_SMALL_MAX_PANELS ? 16 * CF_SMALL_MAX_PANELS : m));
  }();
  return v;
}

// walkers per panel of the throughput solve kernel for W walkers: 16 up to 512 walkers (48 against 53 us per call), 32 beyond
// (528-576 walkers 64 against 67-69 us, 768: 71.7 against 77.1, 10 % at 4096; profiles/r03_gemm_stamps_and_pairing.txt).
// CF_TUNE gemm_np=1|2 forces one.
static int tri_gemm_panel_width(int64_t W) {
  static const int forced = (int)cf_tune("gemm_np", 0);
  return forced == 1 ? 16 : forced == 2 ? 32 : (W > CF_NP2_FROM ? 32 : 16);
}

static int launch_tri_gemm(const TriGemmArgs& a, hipStream_t st) {
  if (a.W <= small_batch_max() && a.partial4) {
    // measured: one tile per workgroup up to 96 walkers (75: 32.0 against 34.0 us), two beyond (128: 37 against 38.5)
    static const int tpw_env = (int)cf_tune("small_tpw", 0);
    const int tpw = (tpw_env == 1 || tpw_env == 2) ? tpw_env : (a.W <= 96 ? 1 : 2);
    if (tpw == 2) return a.frag_b ? launch_tri_gemm_small_t<8, true, 2>(a, st) : launch_tri_gemm_small_t<8, false, 2>(a, st);
    return a.frag_b ? launch_tri_gemm_small_t<16, true, 1>(a, st) : launch_tri_gemm_small_t<16, false, 1>(a, st);
  }
  if (a.frag_b) return fail(CF_ERR_INVALID, "internal: fragment-ordered residuals handed to the throughput solve kernel");
  return tri_gemm_panel_width(a.W) == 32 ? launch_tri_gemm_t<2, 2>(a, st) : launch_tri_gemm_t<1, 2>(a, st);
}

// One evaluation of W walkers on stream `st`: per-walker kernel (distance table, residuals; the small-blocks / growth kernels of a
// joint likelihood), then the solve + chi^2 + epilogue (or the bare epilogue for likelihoods without an SN block).
// ev: 4 timing events or null.
static int launch_eval(cf_handle* h, const double* th, int64_t Wc, double* out, int out_kind, hipStream_t st,
                       double* dm_out, double* mucorr_out, double* blocks_out, double* bao_out, double* chi2_sn_out,
                       hipEvent_t* ev, double* fs8_block_out = nullptr, double* fs8_theory_out = nullptr) {
  const cf_dev_desc& d = h->d;
  unsigned long long* nf = h->nonfinite.as<unsigned long long>();
  double* delta = h->delta.as<double>();
  d2* bao_nodes = h->bao_nodes.as<d2>();
  if (ev) HIP_TRY(hipEventRecord(ev[0], st));
  if (ev && !(h->d.n_sn > 0 || h->has_small_blocks || h->has_growth)) HIP_TRY(hipEventRecord(ev[1], st));
  const bool walker_work = d.n_sn > 0 || h->has_small_blocks || h->has_growth;
  double* extra = (h->has_small_blocks || h->has_growth) ? h->chi2_extra.as<double>() : nullptr;
  // a small batch of the production path: walker_fast_kernel writes the residuals in the fragment order the small-batch solve
  // kernel loads them in (one contiguous 1 KiB load per K-step pair instead of a 16-row gather: sn_fast_loop, FRAG)
  static const bool frag_env = cf_tune("small_frag", 1) != 0;
  const bool small_solve = d.n_sn > 0 && h->solve_mode == CF_SOLVE_INVERSE_GEMM && h->partial4.p && Wc <= small_batch_max();
  const bool frag_b = frag_env && small_solve && !dm_out && !mucorr_out && walker_fast_ok(d);
  if (walker_work) {
    // skewed {cum, dh} table: one spare 16-byte slot per 2^chunk_shift nodes
    const size_t lds = ((size_t)d.n_grid + (d.n_grid >> d.chunk_shift) + 2) * 16;
    if (!dm_out && !mucorr_out && walker_fast_ok(d)) {  // the production form: lean kernel arguments, theta row across the lanes
      // a zero-copy evaluation reads theta from the pinned host block: the walker kernel leaves a copy of each row in device
      // memory and every later kernel of the evaluation (small blocks, growth, the solve's prior / output epilogue) reads THAT --
      // the epilogue's dependent theta reads were microseconds each across the host link
      double* th_copy = h->theta_on_host ? h->theta.as<double>() : nullptr;
      // a small batch leaves most of the chip idle: several workgroups per walker, each with the walker's table and a share of
      // its SNe (CF_TUNE sn_parts=1|2|4 overrides)
      static const int parts_env = (int)cf_tune("sn_parts", 0);
      int sn_parts = parts_env > 0 ? parts_env : (Wc <= 64 ? 4 : (Wc <= 160 ? 2 : 1));  // measured: W = 16 26.5 -> 25.3 us, W = 64 31.5 -> 30.1 us per call
      if (sn_parts > CF_SN_PARTS_MAX || d.n_sn == 0) sn_parts = 1;
      hipLaunchKernelGGL(pick_walker_fast(d.ez_model, d.fde), dim3((unsigned)(Wc * sn_parts)), dim3(512), lds, st, walker_args_of(d), th,
                         Wc, delta, bao_nodes, th_copy, frag_b ? 1 : 0, sn_parts);
      if (th_copy) th = th_copy;
    } else
      hipLaunchKernelGGL(pick_walker(d.ez_model, d.fde), dim3((unsigned)Wc), dim3(512), lds, st, d, th, Wc, delta, dm_out, mucorr_out,
                         bao_nodes, (d2*)nullptr);
    if (ev) HIP_TRY(hipEventRecord(ev[1], st));  // between walker_kernel and the small-block / growth kernels
    if (h->has_small_blocks) {
      // sixteen lanes per walker, sixteen walkers per workgroup (4096 walkers = one wave per SIMD, issue-bound); below that the
      // chip is not full and the time is the serial chain of a lane: a whole wave per walker then (4 instead of 13 Gauss-Legendre
      // nodes per lane).  Measured on the w0waCDM joint likelihood: 16 walkers 50.7 -> 45.1 us per call, 256: 77 -> 67, 2048: 193 ->
      // 185, 4096: 329 -> 334 (profiles/r03_small_blocks_lanes_ab.txt).  The sums do not depend on the width (VirtualLaneSum): a
      // walker's result is the same bits either way.  CF_TUNE sb_wide_max=<walkers> moves the switch.
      static const int64_t wide_max = cf_tune("sb_wide_max", 2048);
      // ... and TWO waves per walker: the BAO / cosmic-chronometer blocks beside the powers and the CMB integrals (small_blocks_kernel,
      // ROLES): 16 walkers 44.6 -> 37.9 us per call, 256: 67 -> 60, 2048: 184 -> 182.  CF_TUNE sb_roles_max=<walkers> (0 = never).
      static const int64_t roles_max = cf_tune("sb_roles_max", 2048);
      const int lanes = Wc <= wide_max ? 64 : 16, roles = (lanes == 64 && Wc <= roles_max) ? 2 : 1, per_wg = 256 / (lanes * roles);
      hipLaunchKernelGGL(pick_small_blocks(d.ez_model, d.fde, lanes, roles), dim3((unsigned)((Wc + per_wg - 1) / per_wg)), dim3(256), 0, st, d, th, Wc,
                         (const d2*)bao_nodes, extra, blocks_out, bao_out);
    }
    if (h->has_growth)  // 256 lanes per walker: the growth ODE as a scan of 2 x 2 step matrices, the f sigma_8 quadratic form
      hipLaunchKernelGGL(pick_growth(d.ez_model, d.fde, d.fs8_steps), dim3((unsigned)Wc), dim3(256),
                         (size_t)(2 * (d.fs8_steps + 1) + 16 + CF_MAX_FS8 + 2 + 256) * 8, st, d, th, Wc, (const d2*)bao_nodes, extra,
                         h->has_small_blocks ? 1 : 0, fs8_block_out, fs8_theory_out);
  }
  if (ev) HIP_TRY(hipEventRecord(ev[2], st));
  if (d.n_sn > 0 && h->solve_mode == CF_SOLVE_INVERSE_GEMM) {
    static const bool trim = cf_tune("gemm_trim", 1) != 0;  // 0: the padded tiles of the last row block are computed (A/B)
    const int n_rb_ = h->ipack.dev.n_rowblocks, nt_last = trim ? std::max(1, std::min(4, (d.n_pad - 64 * (n_rb_ - 1)) / 16)) : 4;
    static const int diag_skip = cf_tune("gemm_diag_skip", 1) != 0;  // 0: the zero tiles of the diagonal blocks are multiplied (A/B)
    TriGemmArgs a{h->epi.as<const cf_epilogue>(), h->ipack.dev.frags, d.n_ld, d.ndim, n_rb_, nt_last, diag_skip, th, Wc, delta,
                  h->max_walkers, h->partial.as<double>(), h->arrivals.as<unsigned int>(), extra, out,
                  out_kind, nf, chi2_sn_out, h->partial4.as<double>(), nullptr, 0ull, frag_b};
    // a synchronous zero-copy call: the solve kernel's last arrivers set one word per panel in pinned host memory and the host
    // waits for those instead of the end of the kernel (launch_tri_gemm's own choice of kernel and panel width decides how many)
    if (h->theta_on_host && h->done_flags.p) {
      a.done_flag = (unsigned long long*)h->done_flags.p;
      a.done_seq = h->done_seq;
      const bool small = Wc <= small_batch_max() && a.partial4;
      const int pw = small ? 16 : tri_gemm_panel_width(Wc);
      h->done_armed = (int)((Wc + pw - 1) / pw);
      if (h->done_armed > CF_DONE_FLAGS) { a.done_flag = nullptr; h->done_armed = 0; }
    }
    int rc = launch_tri_gemm(a, st);
    if (rc) return rc;
  } else if (d.n_sn > 0) {
    int rc = launch_trsm(h->epi.as<const cf_epilogue>(), d.n_pad, d.n_ld, d.ndim, h->pack.dev, th, Wc, delta, h->ypk.as<d2>(), extra, out,
                         out_kind, nf, st, chi2_sn_out);
    if (rc) return rc;
  } else {
    unsigned long long* fin_flag = nullptr;
    if (h->theta_on_host && h->done_flags.p && (Wc + 255) / 256 <= CF_DONE_FLAGS) {
      fin_flag = (unsigned long long*)h->done_flags.p;
      h->done_armed = (int)((Wc + 255) / 256);
    }
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((Wc + 255) / 256)), dim3(256), 0, st, h->epi_host, th, Wc, (const double*)extra, out,
                       out_kind, nf, fin_flag, h->done_seq);
  }
  if (ev) HIP_TRY(hipEventRecord(ev[3], st));
  return 0;
}

// The ONE workspace of a handle is shared by all its evaluations: an evaluation on another stream than the previous one first
// waits, on the HOST, for the previous one (a rare switch -- host calls run on the handle's own stream, a device-resident sampler
// on its own -- so no event is recorded per evaluation: a record is a marker packet and ~2-4 us of every call).  The caller's
// stream handle is only ever COMPARED here, never passed back to HIP: its owner may have destroyed it since, so the
// wait for a foreign stream's evaluation is a device-wide synchronise; the library's own stream is waited for directly.
// Consequence, documented in cosmofit.h: the first evaluation after a stream switch blocks the host and cannot be captured.
static int order_behind_last(cf_handle* h, hipStream_t st) {
  if (!h->has_last || h->last_stream == st) return 0;
  if (h->last_stream == h->stream)
    HIP_TRY(hipStreamSynchronize(h->stream));
  else
    HIP_TRY(hipDeviceSynchronize());
  h->has_last = false;  // drained: nothing of this handle is in flight
  return 0;
}

// Launch one evaluation of W walkers, ordered on `st` (and behind the handle's previous evaluation, whatever stream that ran on).
static int launch_path(cf_handle* h, const double* d_theta, int64_t W, double* d_out, int out_kind, hipStream_t st,
                       double* dm_out, double* mucorr_out, double* blocks_out, double* bao_out,
                       double* chi2_sn_out = nullptr, double* fs8_block_out = nullptr, double* fs8_theory_out = nullptr) {
  { int rc0 = order_behind_last(h, st); if (rc0) return rc0; }
  const bool timed = h->timing_slots && (h->eval_calls++ % h->timing_stride) == 0;
  const int slot = timed ? (int)(h->timed_calls % h->timing_slots) : 0;
  hipEvent_t* ev = timed ? &h->ev[4 * slot] : nullptr;
  int rc = launch_eval(h, d_theta, W, d_out, out_kind, st, dm_out, mucorr_out, blocks_out, bao_out, chi2_sn_out, ev, fs8_block_out,
                       fs8_theory_out);
  if (rc) return rc;
  if (ev) h->timed_calls++;
  h->last_stream = st;
  h->has_last = true;
  HIP_TRY(hipGetLastError());
  return 0;
}

static int check_eval_args(cf_handle* h, const void* theta, int64_t W, const void* out, int out_kind, const char* fn) {
  if (!h || !theta || !out) return fail(CF_ERR_INVALID, std::string(fn) + ": null argument");
  if (W < 0 || W > ((int64_t)1 << 24)) return fail(CF_ERR_INVALID, std::string(fn) + ": W out of range");
  if (out_kind < CF_OUT_CHI2 || out_kind > CF_OUT_LOGP) return fail(CF_ERR_INVALID, std::string(fn) + ": bad out_kind");
  return 0;
}

extern "C" int cf_eval_device(cf_handle* h, const double* d_theta, int64_t W, double* d_out, int32_t out_kind,
                              void* hip_stream) {
  int rc = check_eval_args(h, d_theta, W, d_out, out_kind, "cf_eval_device");
  if (rc) return rc;
  if (!h->peers.empty())
    return fail(CF_ERR_INVALID, "cf_eval_device: this handle spans several devices; device-resident evaluation needs one handle per device");
  if (W == 0) return CF_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  DeviceScope on_device(h->device);
  HIP_TRY(on_device.err);
  if ((rc = ensure_workspace(h, W))) return rc;
  // exactly the caller's stream; NULL is HIP's default (null) stream, which is what torch reports as 0
  return launch_path(h, d_theta, W, d_out, out_kind, (hipStream_t)hip_stream, nullptr, nullptr, nullptr, nullptr);
}

// Wait for the handle's stream from a synchronous host call.  hipStreamSynchronize is no option either way: it SPINS a core for
// the whole wait on this runtime (8 replica threads "blocked" in it burn 23 ms of CPU per 3.85 ms call, profiles/r04_host_wait.txt),
// and where it does sleep it adds tens of microseconds to a call of 0.03-0.3 ms.  So the wait is a hipStreamQuery poll:
//   * the CALLING thread (replica 0; the only thread of a one-device handle) spins for the first 300 us -- every batch a sampler
//     waits for with nothing else to do (up to ~4096 walkers) ends inside that window, at no added latency;
//   * a WORKER thread of a multi-device handle spins 50 us only: its slice is a large batch by construction (cf_eval sends batches
//     of <= 32 walkers to replica 0 alone), eight spinning threads are eight cores taken from the sampler's process;
//   * after that both sleep between polls, 1/16 of the time waited so far, at most 50 us: a 3.8 ms slice is noticed <= 50 us
//     (1.3 %) late for ~2 % of a core instead of 100 %.
// CF_HOST_WAIT=spin polls without ever sleeping (round-3 behaviour), CF_HOST_WAIT=block calls hipStreamSynchronize.
static int wait_stream(hipStream_t st, bool worker_thread) {
  static const int mode = [] { const char* e = getenv("CF_HOST_WAIT"); return !e ? 0 : !strcmp(e, "spin") ? 1 : !strcmp(e, "block") ? 2 : 0; }();
  if (mode == 2) {
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
  }
  const auto t0 = std::chrono::steady_clock::now();
  const auto spin = std::chrono::microseconds(worker_thread ? 50 : 300);
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) return fail(CF_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(q));
    const auto waited = std::chrono::steady_clock::now() - t0;
    if (mode == 1 || waited < spin) continue;
    // (the default timer slack of a thread, 50 us, would turn every short sleep into 60-110 us: 4.14 against 3.78 ms per call of an
    // eight-replica handle; with 1 us of slack a 10 us sleep is 12-15 us)
    static thread_local const bool slack_set = [] { return prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0) == 0; }();
    (void)slack_set;
    std::this_thread::sleep_for(std::min<std::chrono::nanoseconds>(waited / 16, std::chrono::microseconds(50)));
  }
}

// Spin on the words the evaluation's last kernel sets (pinned host memory) for up to ~1 ms; false = not seen (the caller
// then waits for the stream as usual).  CF_DONE_FLAG=0 disables the short cut (A/B).
// THE EARLY-RETURN CONTRACT.  When every word has been seen, cf_eval returns while the tail of its last kernel (teardown, the
// completion signal, the runtime's stream bookkeeping) is still draining on h->stream.  What has been ordered: every result of the
// call is in `stage_out` (the storing wave drained its stores, then released the word at system scope) and has been copied to the
// caller's `out`.  What the next call may touch meanwhile: only h->stream-ordered work (queued behind the tail) and the pinned
// staging blocks -- stage_in is no longer read by the finished evaluation (theta is read from the walker kernel's device copy
// after the first kernel) and stage_out / the words are written only before the release.  Ordering against evaluations on a
// CALLER's stream does not rely on this path: order_behind_last() drains h->stream at the switch.  A GPU fault in the drained tail
// is reported by the next HIP call of the handle (launch_path ends in hipGetLastError), i.e. by the following evaluation.
// The spin holds h->mu (one caller per handle at a time is the documented threading model) and is bounded.
static bool wait_done_flags(cf_handle* h) {
  static const bool on = [] { const char* e = getenv("CF_DONE_FLAG"); return !e || atoi(e) != 0; }();
  if (!on) return false;
  const volatile unsigned long long* f = (const volatile unsigned long long*)h->done_flags.p;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned it = 0;; ++it) {
    bool all = true;
    for (int p = 0; p < h->done_armed; ++p) all = all && f[p] == h->done_seq;
    if (all) {
      std::atomic_thread_fence(std::memory_order_acquire);
      return true;
    }
    if ((it & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(1)) return false;
  }
}

// One replica, host buffers: stage through the handle's pinned block, run on the handle's stream, wait.
static int eval_host_single(cf_handle* h, const double* theta, int64_t W, double* out, int out_kind, bool worker_thread) {
  int rc;
  std::lock_guard<std::mutex> lk(h->mu);
  DeviceScope on_device(h->device);
  HIP_TRY(on_device.err);
  if ((rc = ensure_workspace(h, W))) return rc;
  memcpy(h->stage_in.p, theta, (size_t)W * h->d.ndim * 8);
  // Small batches are latency-bound: the kernels then read theta from, and write the results to, the pinned (device-visible,
  // coherent) staging block in place -- a few cache lines over the host link instead of two copy commands on the critical
  // path.  CF_ZEROCOPY_MAX = largest such batch (walkers; 0 = always copy).
  static const int64_t zc_max = [] { const char* e = getenv("CF_ZEROCOPY_MAX"); return e ? atoll(e) : (long long)CF_ZEROCOPY_DEFAULT; }();
  if (W <= zc_max) {
    h->theta_on_host = true;
    h->done_armed = 0;
    ++h->done_seq;
    rc = launch_path(h, (const double*)h->stage_in.p, W, (double*)h->stage_out.p, out_kind, h->stream, nullptr, nullptr, nullptr, nullptr);
    h->theta_on_host = false;
    if (rc) return rc;
    if (h->done_armed > 0 && wait_done_flags(h)) {  // the small-batch kernel has signalled every panel: the stream drains in the background
      memcpy(out, h->stage_out.p, (size_t)W * 8);
      return CF_OK;
    }
  } else {
    HIP_TRY(hipMemcpyAsync(h->theta.p, h->stage_in.p, (size_t)W * h->d.ndim * 8, hipMemcpyHostToDevice, h->stream));
    if ((rc = launch_path(h, h->theta.as<const double>(), W, h->out.as<double>(), out_kind, h->stream, nullptr, nullptr,
                          nullptr, nullptr)))
      return rc;
    HIP_TRY(hipMemcpyAsync(h->stage_out.p, h->out.p, (size_t)W * 8, hipMemcpyDeviceToHost, h->stream));
  }
  if ((rc = wait_stream(h->stream, worker_thread))) return rc;
  memcpy(out, h->stage_out.p, (size_t)W * 8);
  return CF_OK;
}

// Rows of a W-walker batch that replica k of n evaluates: contiguous, whole 32-walker panels, near-equal.
extern "C" void cf_split_rows(int64_t W, int32_t n, int32_t k, int64_t* begin, int64_t* end) {
  const int64_t panels = (W + 31) / 32, base = panels / n, extra = panels % n;
  const int64_t p0 = k * base + (k < extra ? k : extra), p1 = p0 + base + (k < extra ? 1 : 0);
  *begin = p0 * 32 < W ? p0 * 32 : W;
  *end = p1 * 32 < W ? p1 * 32 : W;
}

extern "C" int cf_eval(cf_handle* h, const double* theta, int64_t W, double* out, int32_t out_kind) {
  int rc = check_eval_args(h, theta, W, out, out_kind, "cf_eval");
  if (rc) return rc;
  if (W == 0) return CF_OK;
  const int n = 1 + (int)h->peers.size();
  if (n == 1 || W <= 32) return eval_host_single(h, theta, W, out, out_kind);
  // several devices: replica k > 0 gets its slice through its worker thread, this thread evaluates slice 0
  const int ndim = h->d.ndim;
  for (int k = 1; k < n; ++k) {
    int64_t b, e;
    cf_split_rows(W, n, k, &b, &e);
    cf_worker* w = h->workers[k - 1].get();
    std::lock_guard<std::mutex> lk(w->m);
    w->theta = theta + b * ndim;
    w->out = out + b;
    w->W = e - b;
    w->out_kind = out_kind;
    w->done = w->W == 0;
    w->rc = 0;
    w->has_job = w->W > 0;
    if (w->has_job) w->cv.notify_all();
  }
  int64_t b0, e0;
  cf_split_rows(W, n, 0, &b0, &e0);
  rc = e0 > b0 ? eval_host_single(h, theta + b0 * ndim, e0 - b0, out + b0, out_kind) : 0;
  std::string err = rc ? g_err : std::string();
  for (int k = 1; k < n; ++k) {  // always wait for every worker: they write into the caller's buffers
    cf_worker* w = h->workers[k - 1].get();
    std::unique_lock<std::mutex> lk(w->m);
    w->cv.wait(lk, [&] { return w->done; });
    if (w->rc && !rc) {
      rc = w->rc;
      err = w->err;
    }
  }
  return rc ? fail(rc, err) : CF_OK;
}

extern "C" int cf_eval_parts(cf_handle* h, const double* theta, int64_t W, double* dm_obs, double* mu_corr,
                             double* delta, double* chi2_blocks, double* bao_theory, double* fs8_theory) {
  if (!h || !theta) return fail(CF_ERR_INVALID, "cf_eval_parts: null argument");
  if (W <= 0 || W > (1 << 20)) return fail(CF_ERR_INVALID, "cf_eval_parts: W out of range");
  if (bao_theory && h->d.n_bao == 0) return fail(CF_ERR_INVALID, "cf_eval_parts: this likelihood has no BAO block");
  if (fs8_theory && h->d.n_fs8 == 0) return fail(CF_ERR_INVALID, "cf_eval_parts: this likelihood has no growth-rate block");
  if ((dm_obs || mu_corr || delta) && h->d.n_sn == 0) return fail(CF_ERR_INVALID, "cf_eval_parts: this likelihood has no SN block");
  std::lock_guard<std::mutex> lk(h->mu);
  DeviceScope on_device(h->device);
  HIP_TRY(on_device.err);
  int rc;
  if ((rc = ensure_workspace(h, W))) return rc;
  const int64_t n = h->d.n_sn, n_ld = h->d.n_ld, nb = h->d.n_bao, nf = h->d.n_fs8;
  DevBuf dm, mc, blk, bt, snb, fsb, ft;
  if (snb.ensure((size_t)W * 8) || fsb.ensure((size_t)W * 8)) return CF_ERR_HIP;
  HIP_TRY(hipMemsetAsync(snb.p, 0, (size_t)W * 8, h->stream));
  HIP_TRY(hipMemsetAsync(fsb.p, 0, (size_t)W * 8, h->stream));
  // the SN accessor path (reference-order mu_corr / mu_theory) is selected by a non-null dm / mu_corr buffer
  if (n > 0 && dm.ensure((size_t)W * n * 8)) return CF_ERR_HIP;
  if (n > 0 && mu_corr && mc.ensure((size_t)W * n * 8)) return CF_ERR_HIP;
  if (blk.ensure((size_t)W * 8 * 8)) return CF_ERR_HIP;
  HIP_TRY(hipMemsetAsync(blk.p, 0, (size_t)W * 8 * 8, h->stream));
  if (nb > 0 && bt.ensure((size_t)W * nb * 8)) return CF_ERR_HIP;
  if (nf > 0 && ft.ensure((size_t)W * nf * 8)) return CF_ERR_HIP;
  HIP_TRY(hipMemcpyAsync(h->theta.p, theta, (size_t)W * h->d.ndim * 8, hipMemcpyHostToDevice, h->stream));
  if ((rc = launch_path(h, h->theta.as<const double>(), W, h->out.as<double>(), CF_OUT_CHI2, h->stream,
                        dm.as<double>(), mc.as<double>(), blk.as<double>(), bt.as<double>(), snb.as<double>(), fsb.as<double>(),
                        ft.as<double>())))
    return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (dm_obs) HIP_TRY(hipMemcpy(dm_obs, dm.p, (size_t)W * n * 8, hipMemcpyDeviceToHost));
  if (mu_corr) HIP_TRY(hipMemcpy(mu_corr, mc.p, (size_t)W * n * 8, hipMemcpyDeviceToHost));
  if (delta && n > 0)
    HIP_TRY(hipMemcpy2D(delta, (size_t)n * 8, h->delta.p, (size_t)n_ld * 8, (size_t)n * 8, (size_t)W,
                        hipMemcpyDeviceToHost));
  if (bao_theory) HIP_TRY(hipMemcpy(bao_theory, bt.p, (size_t)W * nb * 8, hipMemcpyDeviceToHost));
  if (fs8_theory) HIP_TRY(hipMemcpy(fs8_theory, ft.p, (size_t)W * nf * 8, hipMemcpyDeviceToHost));
  if (chi2_blocks) {
    std::vector<double> sn((size_t)W), b8((size_t)W * 8), fs((size_t)W);
    HIP_TRY(hipMemcpy(sn.data(), snb.p, (size_t)W * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(b8.data(), blk.p, (size_t)W * 8 * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(fs.data(), fsb.p, (size_t)W * 8, hipMemcpyDeviceToHost));
    for (int64_t w = 0; w < W; ++w) {
      // sn (written by the solve kernel before the other blocks are added), bao, cmb, the CMB distance vector, cc, fs8, then
      // the two fitting formulae the blocks evaluated: z* and r_drag (0 where no block needs them)
      double* o = chi2_blocks + 10 * w;
      const double* b = &b8[8 * (size_t)w];
      o[0] = sn[w];
      o[1] = b[0]; o[2] = b[1]; o[3] = b[2]; o[4] = b[3]; o[5] = b[4]; o[6] = b[5];
      o[7] = fs[w];
      o[8] = b[6]; o[9] = b[7];
    }
  }
  return CF_OK;
}

// The distance table of W walkers: cum_dm[W*G] and dh[W*G] on the reference's grid linspace(0, z_max, G)  -- what
// DM_z(params, z) of the scripts interpolates (sn/pantheon.py:34-40).  Only the table build of walker_kernel runs.
extern "C" int cf_eval_table(cf_handle* h, const double* theta, int64_t W, double* cum_dm, double* dh) {
  if (!h || !theta || !cum_dm || !dh) return fail(CF_ERR_INVALID, "cf_eval_table: null argument");
  if (W <= 0 || W > 4096) return fail(CF_ERR_INVALID, "cf_eval_table: W must be in 1..4096");
  std::lock_guard<std::mutex> lk(h->mu);
  DeviceScope on_device(h->device);
  HIP_TRY(on_device.err);
  int rc;
  if ((rc = ensure_workspace(h, W))) return rc;
  if ((rc = order_behind_last(h, h->stream))) return rc;
  cf_dev_desc d = h->d;  // a copy without the SN / BAO consumers of the table: only the build runs
  const int G = d.n_grid;
  d.n_sn = 0;
  d.n_bao = 0;
  d.n_fs8 = 0;
  d.n_aux = 0;  // no table-node copies either: their buffer is not passed
  d.n_ld = 0;
  DevBuf tab;
  if (tab.ensure((size_t)W * G * sizeof(d2))) return CF_ERR_HIP;
  HIP_TRY(hipMemcpyAsync(h->theta.p, theta, (size_t)W * d.ndim * 8, hipMemcpyHostToDevice, h->stream));
  const size_t lds = ((size_t)G + (G >> d.chunk_shift) + 2) * 16;
  hipLaunchKernelGGL(pick_walker(d.ez_model, d.fde), dim3((unsigned)W), dim3(512), lds, h->stream, d, h->theta.as<const double>(), W,
                     (double*)nullptr, (double*)nullptr, (double*)nullptr, (d2*)nullptr, tab.as<d2>());
  HIP_TRY(hipGetLastError());
  std::vector<cf_d2> host((size_t)W * G);
  HIP_TRY(hipMemcpyAsync(host.data(), tab.p, host.size() * sizeof(cf_d2), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (size_t i = 0; i < host.size(); ++i) {
    cum_dm[i] = host[i].x;
    dh[i] = host[i].y;
  }
  return CF_OK;
}

extern "C" int cf_eval_fs8_at(cf_handle* h, const double* theta, const double* z, int64_t n, double* out) {
  if (!h || !theta || (n > 0 && (!z || !out))) return fail(CF_ERR_INVALID, "cf_eval_fs8_at: null argument");
  if (n < 0) return fail(CF_ERR_INVALID, "cf_eval_fs8_at: n must be >= 0");
  if (!h->has_growth) return fail(CF_ERR_INVALID, "cf_eval_fs8_at: the handle has no growth-rate block");
  for (int64_t k = 0; k < n; ++k)
    if (!(z[k] >= 0.0) || !(1.0 / (1.0 + z[k]) >= h->d.fs8_a_init))
      return fail(CF_ERR_INVALID, "cf_eval_fs8_at: redshifts must lie in a_init <= a <= 1");
  if (n == 0) return CF_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  DeviceScope on_device(h->device);
  HIP_TRY(on_device.err);
  int rc;
  if ((rc = ensure_workspace(h, 1))) return rc;
  if ((rc = order_behind_last(h, h->stream))) return rc;
  HIP_TRY(hipMemcpyAsync(h->theta.p, theta, (size_t)h->d.ndim * 8, hipMemcpyHostToDevice, h->stream));
  const int B = CF_MAX_FS8;
  DevBuf dz, dbase, dstep, dorder, dpts, dval, dinv, dfid, nodes, dout, extra;
  if (dz.ensure(B * 8) || dbase.ensure(B * 4) || dstep.ensure(B * 4) || dorder.ensure(B * 4) || dpts.ensure(B * CF_FS8_REC * 8) ||
      dval.ensure(B * 8) || dinv.ensure((size_t)B * B * 8) || dfid.ensure(B * 8) ||
      nodes.ensure((size_t)B * CF_BAO_NODES * sizeof(d2)) || dout.ensure(B * 8) || extra.ensure(64))
    return CF_ERR_HIP;
  HIP_TRY(hipMemsetAsync(dval.p, 0, B * 8, h->stream));
  HIP_TRY(hipMemsetAsync(dinv.p, 0, (size_t)B * B * 8, h->stream));
  const std::vector<double> ones((size_t)B, 1.0);  // the Alcock-Paczynski factor divides the residual only, not theory_out
  HIP_TRY(hipMemcpyAsync(dfid.p, ones.data(), B * 8, hipMemcpyHostToDevice, h->stream));
  for (int64_t k0 = 0; k0 < n; k0 += B) {
    const int m = (int)(n - k0 < B ? n - k0 : B);
    // a copy of the descriptor whose only block is a growth-rate block at the requested redshifts (data 0, inverse
    // covariance 0: only theory_out is used); E(z), sigma_8 and a_init stay the handle's
    cf_dev_desc d = h->d;
    d.n_sn = 0; d.n_ld = 0; d.n_bao = 0; d.n_cc = 0; d.cmb_mode = 0;
    d.n_fs8 = m; d.n_aux = m;
    std::vector<int32_t> step_of, order, base((size_t)m);
    std::vector<double> pts;
    fs8_points(d, z + k0, m, step_of, order, pts);
    for (int k = 0; k < m; ++k) base[k] = aux_node_base(d, z[k0 + k]);
    HIP_TRY(hipMemcpyAsync(dz.p, z + k0, (size_t)m * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(dbase.p, base.data(), (size_t)m * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(dstep.p, step_of.data(), (size_t)m * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(dorder.p, order.data(), (size_t)m * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(dpts.p, pts.data(), (size_t)m * CF_FS8_REC * 8, hipMemcpyHostToDevice, h->stream));
    d.fs8_z = dz.as<const double>(); d.bao_base = dbase.as<const int32_t>(); d.fs8_step_of = dstep.as<const int32_t>();
    d.fs8_order = dorder.as<const int32_t>(); d.fs8_pts = dpts.as<const double>(); d.fs8_val = dval.as<const double>();
    d.fs8_inv_cov = dinv.as<const double>(); d.fs8_fid = dfid.as<const double>();
    const size_t lds = ((size_t)d.n_grid + (d.n_grid >> d.chunk_shift) + 2) * 16;
    hipLaunchKernelGGL(pick_walker(d.ez_model, d.fde), dim3(1), dim3(512), lds, h->stream, d, h->theta.as<const double>(), (int64_t)1,
                       (double*)nullptr, (double*)nullptr, (double*)nullptr, nodes.as<d2>(), (d2*)nullptr);
    hipLaunchKernelGGL(pick_growth(d.ez_model, d.fde, d.fs8_steps), dim3(1), dim3(256),
                       (size_t)(2 * (d.fs8_steps + 1) + 16 + CF_MAX_FS8 + 2 + 256) * 8, h->stream, d, h->theta.as<const double>(),
                       (int64_t)1, (const d2*)nodes.as<d2>(), extra.as<double>(), 0, (double*)nullptr, dout.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out + k0, dout.p, (size_t)m * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));  // the host vectors and the staging buffers are reused by the next chunk
  }
  return CF_OK;
}

extern "C" int cf_eval_hz(cf_handle* h, const double* theta, const double* z, int64_t n, double* out) {
  if (!h || !theta || (n > 0 && (!z || !out))) return fail(CF_ERR_INVALID, "cf_eval_hz: null argument");
  if (n < 0) return fail(CF_ERR_INVALID, "cf_eval_hz: n must be >= 0");
  if (n == 0) return CF_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  DeviceScope on_device(h->device);
  HIP_TRY(on_device.err);
  int rc;
  if ((rc = ensure_workspace(h, 1))) return rc;
  if ((rc = order_behind_last(h, h->stream))) return rc;
  DevBuf dz, dout;
  if (dz.ensure((size_t)n * 8) || dout.ensure((size_t)n * 8)) return CF_ERR_HIP;
  HIP_TRY(hipMemcpyAsync(h->theta.p, theta, (size_t)h->d.ndim * 8, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(dz.p, z, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(pick_hz(h->d.ez_model, h->d.fde), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d,
                     h->theta.as<const double>(), dz.as<const double>(), n, dout.as<double>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CF_OK;
}

extern "C" int cf_eval_bao_at(cf_handle* h, const double* theta, const double* z, const int32_t* qty, int64_t n, double* out) {
  if (!h || !theta || (n > 0 && (!z || !qty || !out))) return fail(CF_ERR_INVALID, "cf_eval_bao_at: null argument");
  if (n < 0) return fail(CF_ERR_INVALID, "cf_eval_bao_at: n must be >= 0");
  for (int64_t k = 0; k < n; ++k) {
    if (!std::isfinite(z[k])) return fail(CF_ERR_INVALID, "cf_eval_bao_at: z must be finite");
    if (qty[k] < 0 || qty[k] > 3) return fail(CF_ERR_INVALID, "cf_eval_bao_at: qty must be 0 (D_V), 1 (D_M), 2 (D_H) or 3 (F_AP)");
  }
  if (n == 0) return CF_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  DeviceScope on_device(h->device);
  HIP_TRY(on_device.err);
  int rc;
  if ((rc = ensure_workspace(h, 1))) return rc;
  if ((rc = order_behind_last(h, h->stream))) return rc;
  HIP_TRY(hipMemcpyAsync(h->theta.p, theta, (size_t)h->d.ndim * 8, hipMemcpyHostToDevice, h->stream));
  DevBuf dz, dq, dbase, dval, dinv, nodes, dout, extra;
  const int B = CF_MAX_BAO;
  if (dz.ensure(B * 8) || dq.ensure(B * 4) || dbase.ensure(B * 4) || dval.ensure(B * 8) || dinv.ensure((size_t)B * B * 8) ||
      nodes.ensure((size_t)B * CF_BAO_NODES * sizeof(d2)) || dout.ensure(B * 8) || extra.ensure(64))
    return CF_ERR_HIP;
  HIP_TRY(hipMemsetAsync(dval.p, 0, B * 8, h->stream));
  HIP_TRY(hipMemsetAsync(dinv.p, 0, (size_t)B * B * 8, h->stream));
  for (int64_t k0 = 0; k0 < n; k0 += B) {
    const int m = (int)(n - k0 < B ? n - k0 : B);
    // a copy of the descriptor whose only block is a BAO block at the requested points (data value 0, inverse covariance 0:
    // the quadratic form is not used); the E(z) model, the D_H convention and the sound horizon stay the handle's
    cf_dev_desc d = h->d;
    d.n_sn = 0; d.n_ld = 0; d.n_fs8 = 0; d.n_cc = 0; d.cmb_mode = 0;
    d.n_bao = m; d.n_aux = m;
    std::vector<int32_t> base((size_t)m);
    for (int k = 0; k < m; ++k) base[k] = aux_node_base(d, z[k0 + k]);
    HIP_TRY(hipMemcpyAsync(dz.p, z + k0, (size_t)m * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(dq.p, qty + k0, (size_t)m * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(dbase.p, base.data(), (size_t)m * 4, hipMemcpyHostToDevice, h->stream));
    d.bao_z = dz.as<const double>(); d.bao_qty = dq.as<const int32_t>(); d.bao_base = dbase.as<const int32_t>();
    d.bao_val = dval.as<const double>(); d.bao_inv_cov = dinv.as<const double>();
    const size_t lds = ((size_t)d.n_grid + (d.n_grid >> d.chunk_shift) + 2) * 16;
    hipLaunchKernelGGL(pick_walker(d.ez_model, d.fde), dim3(1), dim3(512), lds, h->stream, d, h->theta.as<const double>(), (int64_t)1,
                       (double*)nullptr, (double*)nullptr, (double*)nullptr, nodes.as<d2>(), (d2*)nullptr);
    hipLaunchKernelGGL(pick_small_blocks(d.ez_model, d.fde, 16), dim3(1), dim3(256), 0, h->stream, d, h->theta.as<const double>(), (int64_t)1,
                       (const d2*)nodes.as<d2>(), extra.as<double>(), (double*)nullptr, dout.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out + k0, dout.p, (size_t)m * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));  // `base` and the staging buffers are reused by the next chunk
  }
  return CF_OK;
}

// ------------------------------------------------------------------------------------------------
// Stand-alone operators
// ------------------------------------------------------------------------------------------------
static int interp_common(const double* xq, int64_t nq, const double* x, const double* y, const double* yp, int64_t n,
                         double* out, int mode) {
  if (!xq || !x || !y || !out || (mode == 0 && !yp)) return fail(CF_ERR_INVALID, "cf_interp: null argument");
  if (nq < 0 || n < 2) return fail(CF_ERR_INVALID, "cf_interp: need n >= 2 nodes");
  if (mode == 1 && n < 3) return fail(CF_ERR_INVALID, "cf_interp_pchip: need n >= 3 nodes");
  if (cf_device_count() == 0) return fail(CF_ERR_NO_DEVICE, "cf_interp: no HIP device visible (this library has no CPU path)");
  if (nq == 0) return CF_OK;
  DevBuf dxq, dx, dy, dyp, dout;
  if (dxq.ensure((size_t)nq * 8) || dx.ensure((size_t)n * 8) || dy.ensure((size_t)n * 8) || dout.ensure((size_t)nq * 8))
    return CF_ERR_HIP;
  HIP_TRY(hipMemcpy(dxq.p, xq, (size_t)nq * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dy.p, y, (size_t)n * 8, hipMemcpyHostToDevice));
  if (mode == 0) {
    if (dyp.ensure((size_t)n * 8)) return CF_ERR_HIP;
    HIP_TRY(hipMemcpy(dyp.p, yp, (size_t)n * 8, hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(interp_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, dxq.as<const double>(), nq,
                     dx.as<const double>(), dy.as<const double>(), dyp.as<const double>(), n, dout.as<double>(), mode);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout.p, (size_t)nq * 8, hipMemcpyDeviceToHost));
  return CF_OK;
}

extern "C" int cf_interp_hermite(const double* xq, int64_t nq, const double* x, const double* y, const double* y_prime,
                                 int64_t n, double* out) {
  return interp_common(xq, nq, x, y, y_prime, n, out, 0);
}

extern "C" int cf_interp_pchip(const double* xq, int64_t nq, const double* x, const double* y, int64_t n, double* out) {
  return interp_common(xq, nq, x, y, nullptr, n, out, 1);
}

extern "C" int cf_solve_triangular(const double* L, int64_t n, int64_t ld, const double* b, int64_t nrhs, double* out) {
  if (!L || !b || !out) return fail(CF_ERR_INVALID, "cf_solve_triangular: null argument");
  if (n < 1 || ld < n || nrhs < 0 || n > (1 << 20)) return fail(CF_ERR_INVALID, "cf_solve_triangular: bad sizes");
  if (cf_device_count() == 0)
    return fail(CF_ERR_NO_DEVICE, "cf_solve_triangular: no HIP device visible (this library has no CPU path)");
  if (nrhs == 0) return CF_OK;
  cf_host_pack hp;
  double probe = 0.0;
  if (pack_default(L, n, ld, hp, &probe) != 0)
    return fail(CF_ERR_NOT_POSDEF, "cf_solve_triangular: non-positive or non-finite diagonal entry");
  if (!(probe <= CF_PROBE_LIMIT))
    return fail(CF_ERR_ILL_CONDITIONED, "cf_solve_triangular: blocked solve loses " + std::to_string(probe) + " relative on this factor");
  PackedFactor pf;
  int rc;
  if ((rc = pf.upload(hp))) return rc;
  const int64_t w_pad = (nrhs + 15) / 16 * 16, n_pad = hp.n_pad, n_ld = (n + 63) / 64 * 64;
  DevBuf db, delta, ypk, dout, nf, dth, epi0;
  if (db.ensure((size_t)nrhs * n * 8) || delta.ensure((size_t)w_pad * n_ld * 8) || ypk.ensure((size_t)w_pad * n_pad * 8 + CF_YPK_SLACK) ||
      dout.ensure((size_t)w_pad * 8) || nf.ensure(8) || dth.ensure(8) || epi0.ensure(sizeof(cf_epilogue)))
    return CF_ERR_HIP;
  HIP_TRY(hipMemset(epi0.p, 0, sizeof(cf_epilogue)));  // no prior, no Gaussian terms: out_kind chi^2 returns y . y
  HIP_TRY(hipMemcpy(db.p, b, (size_t)nrhs * n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(nf.p, 0, 8));
  hipLaunchKernelGGL(pad_rhs_kernel, dim3((unsigned)w_pad), dim3(256), 0, 0, db.as<const double>(), nrhs, n, n_ld,
                     delta.as<double>());
  if ((rc = launch_trsm(epi0.as<const cf_epilogue>(), (int)n_pad, (int)n_ld, 0, pf.dev, dth.as<const double>(), nrhs, delta.as<const double>(), ypk.as<d2>(), nullptr,
                        dout.as<double>(), (int)CF_OUT_CHI2, nf.as<unsigned long long>(), (hipStream_t)0, nullptr)))
    return rc;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout.p, (size_t)nrhs * 8, hipMemcpyDeviceToHost));
  return CF_OK;
}

// Host-only self-test of the inverse-GEMM packing (explicit inverse): chi^2 of one right-hand side by
// replaying the fragment streams, and the create-time probe value.  CPU test-suite only.
extern "C" int cf_selftest_invpack_host(const double* L, int64_t n, int64_t ld, const double* b, double* chi2_out,
                                        double* probe_out) {
  if (!L || !b || !chi2_out) return fail(CF_ERR_INVALID, "cf_selftest_invpack_host: null argument");
  for (int64_t i = 0; i < n; ++i)
    if (!(L[i * ld + i] > 0.0) || !std::isfinite(L[i * ld + i])) return fail(CF_ERR_NOT_POSDEF, "cf_selftest_invpack_host: bad pivot");
  cf_host_invpack ip;
  cf_pack_inverse(L, n, ld, ip);
  *chi2_out = cf_invpack_replay_host(ip, b);
  if (probe_out) *probe_out = cf_invpack_probe(ip, L, ld);
  return CF_OK;
}

// Self-tests of the in-kernel log10 routines (device): out[k] = log10_pos(x[k]) / log10_tab(x[k]).
static int selftest_log10(const double* x, int64_t n, double* out, int mode, const char* fn) {
  if (!x || !out || n < 0) return fail(CF_ERR_INVALID, std::string(fn) + ": bad argument");
  if (cf_device_count() == 0) return fail(CF_ERR_NO_DEVICE, std::string(fn) + ": no HIP device visible");
  if (n == 0) return CF_OK;
  DevBuf dx, dout, tab;
  if (dx.ensure((size_t)n * 8) || dout.ensure((size_t)n * 8)) return CF_ERR_HIP;
  int rc;
  if (mode == 2) {
    if ((rc = upload_exp2_table(tab))) return rc;
  } else if ((rc = upload_log10_table(tab))) {
    return rc;
  }
  HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(log10_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx.as<const double>(), n,
                     dout.as<double>(), mode, tab.as<const cf_d2>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  return CF_OK;
}
extern "C" int cf_selftest_log10(const double* x, int64_t n, double* out) { return selftest_log10(x, n, out, 0, "cf_selftest_log10"); }
extern "C" int cf_selftest_log10_tab(const double* x, int64_t n, double* out) {
  return selftest_log10(x, n, out, 1, "cf_selftest_log10_tab");
}
extern "C" int cf_selftest_exp_tab(const double* x, int64_t n, double* out) {
  return selftest_log10(x, n, out, 2, "cf_selftest_exp_tab");
}
extern "C" int cf_selftest_pos_ops(const double* a, const double* b, int64_t n, double* out) {
  if (!a || !b || !out || n < 0) return fail(CF_ERR_INVALID, "cf_selftest_pos_ops: bad argument");
  if (cf_device_count() == 0) return fail(CF_ERR_NO_DEVICE, "cf_selftest_pos_ops: no HIP device visible");
  if (n == 0) return CF_OK;
  DevBuf da, db, dout;
  if (da.ensure((size_t)n * 8) || db.ensure((size_t)n * 8) || dout.ensure((size_t)n * 32)) return CF_ERR_HIP;
  HIP_TRY(hipMemcpy(da.p, a, (size_t)n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db.p, b, (size_t)n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(pos_ops_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, da.as<const double>(),
                     db.as<const double>(), n, dout.as<double>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * 32, hipMemcpyDeviceToHost));
  return CF_OK;
}

// ------------------------------------------------------------------------------------------------
// Host-only self-test of the packing logic (CPU test-suite; never on the evaluation path).
// Returns || L^-1 b ||^2 computed by replaying the packed fragment streams on the host.
// ------------------------------------------------------------------------------------------------
extern "C" int cf_selftest_pack_host(const double* L, int64_t n, int64_t ld, const double* b, double* chi2_out,
                                     int64_t* packed_bytes) {
  if (!L || !b || !chi2_out) return fail(CF_ERR_INVALID, "cf_selftest_pack_host: null argument");
  cf_host_pack hp;
  if (pack_default(L, n, ld, hp) != 0) return fail(CF_ERR_NOT_POSDEF, "cf_selftest_pack_host: bad pivot");
  *chi2_out = cf_pack_replay_host(hp, b);
  if (packed_bytes) *packed_bytes = (int64_t)(hp.frags.size() * sizeof(cf_d2));
  return CF_OK;
}
