// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (pins the FP64 matrix peak that
// bench.py's roofline uses).  hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_rate.hip -o /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) k(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd, int blocks) {
  double* out;
  int threads = 256 * waves_per_simd;
  hipMalloc(&out, (size_t)blocks * threads * 8);
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 100, 1.0, 2.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double n_mfma_per_simd = (double)iters * NACC * waves_per_simd * ((double)blocks / 256.0);
  double flops = (double)iters * NACC * 2048.0 * (threads / 64) * blocks;
  printf("NACC=%d waves/SIMD=%d blocks=%d: %.3f ms, %.2f TFLOP/s, %.1f ns per MFMA per SIMD (= %.1f cycles @2.4GHz)\n", NACC,
         waves_per_simd, blocks, ms, flops / ms / 1e9, ms * 1e6 / n_mfma_per_simd, ms * 1e6 / n_mfma_per_simd * 2.4);
  hipFree(out);
}

int main() {
  run<1>(1, 256); run<2>(1, 256); run<4>(1, 256); run<8>(1, 256);
  run<1>(2, 256); run<4>(2, 256);
  run<4>(1, 512); run<4>(1, 1024);
  return 0;
}
