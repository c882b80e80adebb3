// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 and the clock the chip holds
// under it (pins the FP64 matrix peak that bench.py's roofline is compared with).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) k(double* out, unsigned long long* stamps, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  asm volatile("" ::"v"(s));
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    stamps[2 * (blockIdx.x * 4 + threadIdx.x / 64)] = c1 - c0;
    stamps[2 * (blockIdx.x * 4 + threadIdx.x / 64) + 1] = r1 - r0;
  }
}

template <int NACC>
void run(int blocks, int iters = 20000) {
  double* out;
  unsigned long long* st;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
  (void)hipMalloc(&st, (size_t)blocks * 4 * 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, st, 100, 1.0, 2.0);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, st, iters, 1.0, 2.0);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)blocks * 8);
  (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc, clk;
  for (int i = 0; i < blocks * 4; ++i) { cyc.push_back((double)h[2 * i]); clk.push_back((double)h[2 * i] / ((double)h[2 * i + 1] * 10.0)); }  // realtime ticks = 100 MHz -> 10 ns
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  double flops = (double)iters * NACC * 2048.0 * 4 * blocks;
  double wps = blocks / 256.0;
  printf("NACC=%d waves/SIMD=%.0f: %.3f ms, %6.2f TFLOP/s | per wave: %.1f cycles per MFMA (median), clock %.3f GHz (median)\n",
         NACC, wps, ms, flops / ms / 1e9, cyc[cyc.size() / 2] / ((double)iters * NACC), clk[clk.size() / 2]);
  (void)hipFree(out); (void)hipFree(st);
}

int main() {
  run<1>(256); run<2>(256); run<4>(256); run<8>(256);
  run<4>(512); run<4>(1024); run<4>(2048);
  run<1>(1024); run<2>(1024);
  // sustained: long run to let the clock settle
  run<4>(1024, 400000);
  return 0;
}
