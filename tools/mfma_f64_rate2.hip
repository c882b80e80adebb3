// Micro-benchmark 2: what lets v_mfma_f64_16x16x4_f64 approach its 64-cycle pipe rate?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_rate2.hip -o /tmp/m2 && /tmp/m2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

// MODE 0: plain; 1: distinct A/B per MFMA; 2: s_nop 15 between; 3: two v_fma_f64 between MFMAs; 4: s_setprio 3
template <int NACC, int MODE>
__global__ void __launch_bounds__(256) k(double* out, unsigned long long* stamps, int iters, double a0, double b0) {
  d4 acc[NACC];
  double a[NACC], b[NACC], f[NACC];
  for (int i = 0; i < NACC; ++i) {
    acc[i] = (d4){0, 0, 0, 0};
    a[i] = a0 + threadIdx.x * 1e-3 + i;
    b[i] = b0 - threadIdx.x * 1e-3 - i;
    f[i] = 1.0 + i;
  }
  if (MODE == 4) __builtin_amdgcn_s_setprio(3);
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const int s = (MODE == 1) ? i : 0;
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc[i], 0, 0, 0);
      if (MODE == 2) asm volatile("s_nop 15");
      if (MODE == 3) {
        f[i] = __builtin_fma(f[i], 1.0000001, 1e-9);
        f[i] = __builtin_fma(f[i], 0.9999999, 1e-9);
      }
    }
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + f[i];
  asm volatile("" ::"v"(s));
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    stamps[2 * (blockIdx.x * 4 + threadIdx.x / 64)] = c1 - c0;
    stamps[2 * (blockIdx.x * 4 + threadIdx.x / 64) + 1] = r1 - r0;
  }
}

template <int NACC, int MODE>
void run(int blocks, int iters = 20000) {
  double* out;
  unsigned long long* st;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
  (void)hipMalloc(&st, (size_t)blocks * 4 * 16);
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, out, st, 100, 1.0, 2.0);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, out, st, iters, 1.0, 2.0);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)blocks * 8);
  (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc, clk;
  for (int i = 0; i < blocks * 4; ++i) {
    cyc.push_back((double)h[2 * i]);
    clk.push_back((double)h[2 * i] / ((double)h[2 * i + 1] * 10.0));
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(clk.begin(), clk.end());
  double wps = blocks / 256.0;
  double per_wave = cyc[cyc.size() / 2] / ((double)iters * NACC);
  printf("mode=%d NACC=%d waves/SIMD=%.0f: wall %.3f ms | per wave %.1f cyc/MFMA -> per SIMD %.1f cyc/MFMA (slowest wave %.1f), clock %.3f GHz\n",
         MODE, NACC, wps, ms, per_wave, per_wave / wps, cyc.back() / ((double)iters * NACC) / wps, clk[clk.size() / 2]);
  (void)hipFree(out);
  (void)hipFree(st);
}

int main() {
  printf("-- plain, waves/SIMD sweep (NACC=4)\n");
  run<4, 0>(256); run<4, 0>(512); run<4, 0>(768); run<4, 0>(1024); run<4, 0>(1280); run<4, 0>(1536); run<4, 0>(2048);
  printf("-- distinct A/B registers per MFMA\n");
  run<4, 1>(256); run<4, 1>(512); run<4, 1>(1024);
  printf("-- s_nop 15 after each MFMA\n");
  run<4, 2>(256); run<4, 2>(512); run<4, 2>(1024);
  printf("-- two v_fma_f64 after each MFMA\n");
  run<4, 3>(256); run<4, 3>(512); run<4, 3>(1024);
  printf("-- s_setprio 3\n");
  run<4, 4>(512); run<4, 4>(1024);
  printf("-- NACC=8\n");
  run<8, 0>(512); run<8, 0>(1024);
  return 0;
}
