// Micro-benchmark: sustained v_fma_f64 rate with a SCALAR (SGPR) multiplicand, i.e. the inner loop of a
// "64 walkers per wave" GEMM  acc[r] += X[r][k] * y[k][lane]  (X wave-uniform -> s_load, y one vector load per k).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_f64_rate.hip -o /tmp/v && /tmp/v
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int R>
__global__ void __launch_bounds__(256) k(const double* __restrict__ X, const double* __restrict__ Y, double* out, int K) {
  double acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.0;
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  // wave-uniform base: different waves read different (cached) rows of X
  const double* Xs = X + (size_t)__builtin_amdgcn_readfirstlane(wave_global % 64) * (size_t)K * R;
  const double* Yl = Y + lane;
  for (int kk = 0; kk < K; ++kk) {
    const double y = Yl[(size_t)kk * 64];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = __builtin_fma(Xs[(size_t)kk * R + r], y, acc[r]);
  }
  double s = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) s += acc[r] * acc[r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int R>
void run(int blocks, int K) {
  double *X, *Y, *out;
  (void)hipMalloc(&X, (size_t)64 * K * R * 8);
  (void)hipMalloc(&Y, (size_t)K * 64 * 8);
  (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
  std::vector<double> hx((size_t)64 * K * R, 1e-3), hy((size_t)K * 64, 0.5);
  (void)hipMemcpy(X, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(Y, hy.data(), hy.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k<R>, dim3(blocks), dim3(256), 0, 0, X, Y, out, K);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(k<R>, dim3(blocks), dim3(256), 0, 0, X, Y, out, K);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  double flops = 2.0 * 64 * R * (double)K * blocks * 4;
  double fma_instr_per_simd = (double)R * K * blocks * 4 / 1024.0;
  printf("R=%2d waves/SIMD=%.0f K=%d: %.3f ms  %.1f TFLOP/s  (%.2f cycles per v_fma_f64 per SIMD @2.4GHz)\n", R, blocks / 256.0, K, ms,
         flops / ms / 1e9, ms * 1e-3 * 2.4e9 / fma_instr_per_simd);
  (void)hipFree(X); (void)hipFree(Y); (void)hipFree(out);
}

int main() {
  run<8>(256, 4096); run<8>(512, 4096); run<8>(1024, 4096); run<8>(2048, 4096);
  run<16>(256, 4096); run<16>(512, 4096); run<16>(1024, 4096); run<16>(2048, 4096);
  run<32>(256, 2048); run<32>(512, 2048); run<32>(1024, 2048);
  return 0;
}
