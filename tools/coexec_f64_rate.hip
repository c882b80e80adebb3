// Micro-benchmark: what do v_mfma_f64_16x16x4_f64 and v_fma_f64 sustain on every SIMD of the chip, alone and side
// by side?  Every wave works from registers for a fixed window of s_memrealtime and reports how many instructions
// it got through, so the aggregate does not depend on how fairly a SIMD arbitrates between its waves; the in-kernel
// clock is s_memtime / s_memrealtime.  (An earlier version of this measurement looped over four accumulators held
// in an array: hipcc moved them between VGPRs and AGPRs around every iteration -- 64 v_accvgpr moves per 4 MFMAs --
// and the "ceiling" it reported, 47 TFLOP/s, was that of the copies.)
//   hipcc --offload-arch=gfx950 -O3 tools/coexec_f64_rate.hip -o /tmp/cx && /tmp/cx
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

// roles: 0 = everyone MFMA, 1 = everyone VALU, 2 = even blocks MFMA / odd blocks VALU, 3 = waves 0,1 MFMA / 2,3 idle ...
// Every wave works until `window` ticks of s_memrealtime (100 MHz) have passed since ITS start and reports how many
// instructions it got through: aggregate rates are then independent of how fairly the SIMD arbitrates between waves.
__global__ void __launch_bounds__(256) k(double* out, unsigned long long* counts, int roles, unsigned long long window, double a0) {
  const bool mf = roles == 0 || (roles == 2 && (blockIdx.x & 1) == 0);
  double s = 0;
  unsigned long long n = 0;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (mf) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-3, b = a0 - threadIdx.x * 1e-3;
    while (__builtin_amdgcn_s_memrealtime() - t0 < window) {
      for (int it = 0; it < 8; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      n += 32;
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double f[16];
    for (int i = 0; i < 16; ++i) f[i] = a0 + i + threadIdx.x;
    const double m = 1.0 + a0 * 1e-9, c = a0 * 1e-9;
    while (__builtin_amdgcn_s_memrealtime() - t0 < window) {
      for (int it = 0; it < 16; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) f[i] = __builtin_fma(f[i], m, c);
      n += 256;
    }
    for (int i = 0; i < 16; ++i) s += f[i];
  }
  asm volatile("" ::"v"(s));
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    counts[2 * (blockIdx.x * 4 + threadIdx.x / 64)] = n;
    counts[2 * (blockIdx.x * 4 + threadIdx.x / 64) + 1] = (c1 - c0) * 1000ull / (t1 - t0 ? t1 - t0 : 1);  // clock in 0.1 MHz
  }
}

static void run(int blocks, int roles, double window_ms = 4.0) {
  double* out;
  unsigned long long* st;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
  (void)hipMalloc(&st, (size_t)blocks * 4 * 16);
  const unsigned long long window = (unsigned long long)(window_ms * 1e5);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, st, roles, 1000ull, 1.0);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, st, roles, window, 1.0);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)blocks * 8);
  (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  double nm = 0, nv = 0, wm = 0, wv = 0, mn = 1e30, mx = 0;
  std::vector<double> clk;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < 4; ++w) {
      bool mf = roles == 0 || (roles == 2 && (b & 1) == 0);
      double c = (double)h[2 * (b * 4 + w)];
      clk.push_back((double)h[2 * (b * 4 + w) + 1] * 1e-4);
      if (mf) { nm += c; wm += 1; mn = std::min(mn, c); mx = std::max(mx, c); } else { nv += c; wv += 1; }
    }
  const double sec = window_ms * 1e-3;
  std::sort(clk.begin(), clk.end());
  printf("blocks=%4d roles=%d window %.1f ms wall %.3f ms clock %.2f GHz", blocks, roles, window_ms, ms, clk[clk.size() / 2]);
  if (wm > 0) printf(" | MFMA %.1f waves/SIMD: %.1f TFLOP/s (%.1f ns per MFMA per SIMD; per-wave count min %.0f max %.0f)", wm / 1024, nm * 2048 / sec / 1e12, sec * 1e9 / (nm / 1024), mn, mx);
  if (wv > 0) printf(" | VALU %.1f waves/SIMD: %.1f TFLOP/s (%.2f ns per v_fma_f64 per SIMD)", wv / 1024, nv * 128 / sec / 1e12, sec * 1e9 / (nv / 1024));
  if (wm > 0 && wv > 0) printf(" | sum %.1f TFLOP/s", (nm * 2048 + nv * 128) / sec / 1e12);
  printf("\n");
  (void)hipFree(out);
  (void)hipFree(st);
}

int main() {
  printf("-- alone\n");
  run(256, 0); run(512, 0); run(1024, 0);  // 1, 2, 4 waves per SIMD (more do not fit beside each other: 2 rounds)
  run(256, 1); run(512, 1); run(1024, 1);
  printf("-- mixed (half the blocks each)\n");
  run(512, 2); run(1024, 2);
  return 0;
}
