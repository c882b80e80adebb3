// Micro-benchmark 2: v_fma_f64 with scalar multiplicands, software-pipelined (the scalars and the vector of
// step k+1 are requested before the FMAs of step k), y through LDS or global.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int R, bool YLDS>
__global__ void __launch_bounds__(256) k(const double* __restrict__ X, const double* __restrict__ Y, double* out, int K) {
  __shared__ double ylds[64 * 64];  // a 64-step window of y, re-read cyclically (4 workgroups per CU must fit)
  double acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.0;
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const double* Xs = X + (size_t)__builtin_amdgcn_readfirstlane(wave_global % 64) * (size_t)(K + 2) * R;
  const double* Yl = Y + lane;
  if (YLDS) {
    for (int i = threadIdx.x; i < 64 * 64; i += 256) ylds[i] = Y[i];
    __syncthreads();
  }
  double xa[R], xb[R], ya, yb;
#pragma unroll
  for (int r = 0; r < R; ++r) xa[r] = Xs[r];
  ya = YLDS ? ylds[lane] : Yl[0];
  for (int kk = 0; kk < K; kk += 2) {
    // step kk's operands were requested one stage ago: wait for them FIRST (scalar loads return out of order,
    // so the only wait is lgkmcnt(0) and it must not see the next request), then request step kk+1
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < R; ++r) xb[r] = Xs[(size_t)(kk + 1) * R + r];
    yb = YLDS ? ylds[((kk + 1) & 63) * 64 + lane] : Yl[(size_t)(kk + 1) * 64];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = __builtin_fma(xa[r], ya, acc[r]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < R; ++r) xa[r] = Xs[(size_t)(kk + 2) * R + r];
    ya = YLDS ? ylds[((kk + 2) & 63) * 64 + lane] : Yl[(size_t)(kk + 2) * 64];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = __builtin_fma(xb[r], yb, acc[r]);
    __builtin_amdgcn_sched_barrier(0);
  }
  double s = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) s += acc[r] * acc[r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int R, bool YLDS>
void run(int blocks, int K) {
  double *X, *Y, *out;
  (void)hipMalloc(&X, (size_t)64 * (K + 2) * R * 8);
  (void)hipMalloc(&Y, (size_t)(K + 2) * 64 * 8);
  (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
  std::vector<double> hx((size_t)64 * (K + 2) * R, 1e-3), hy((size_t)(K + 2) * 64, 0.5);
  (void)hipMemcpy(X, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(Y, hy.data(), hy.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k<R, YLDS>), dim3(blocks), dim3(256), 0, 0, X, Y, out, K);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k<R, YLDS>), dim3(blocks), dim3(256), 0, 0, X, Y, out, K);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  double flops = 2.0 * 64 * R * (double)K * blocks * 4;
  double fma_instr_per_simd = (double)R * K * blocks * 4 / 1024.0;
  printf("R=%2d y-from-%s waves/SIMD=%.0f: %.3f ms  %.1f TFLOP/s  (%.2f cycles per v_fma_f64 per SIMD @2.4GHz)\n", R, YLDS ? "LDS" : "global",
         blocks / 256.0, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / fma_instr_per_simd);
  (void)hipFree(X); (void)hipFree(Y); (void)hipFree(out);
}

int main() {
  run<16, false>(256, 4096); run<16, false>(512, 4096); run<16, false>(1024, 4096); run<16, false>(2048, 4096);
  run<16, true>(256, 4096); run<16, true>(512, 4096); run<16, true>(1024, 4096);
  run<8, true>(512, 4096); run<8, true>(1024, 4096);
  run<24, true>(256, 4096); run<24, true>(512, 4096);
  return 0;
}
