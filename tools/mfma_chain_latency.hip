// Micro-benchmark: cycles per v_mfma_f64_16x16x4_f64 when each MFMA accumulates into the result of the previous one (a dependent
// chain, as one (tile, K quarter) accumulator of the solve), against 2 / 4 / 8 independent chains interleaved in one wave.
// One wave per workgroup, one workgroup per CU; s_memtime around 1024 MFMAs per chain.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_chain_latency.hip -o /tmp/mcl && /tmp/mcl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ void chain_kernel(double a0, double b0, unsigned long long* cycles, double* sink) {
  double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  d4 acc[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 128; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < CH; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < CH; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int CH>
static void run(const char* what) {
  const int blocks = 256;
  unsigned long long* cyc;
  double* sink;
  hipMalloc(&cyc, blocks * sizeof(*cyc));
  hipMalloc(&sink, blocks * 64 * sizeof(double));
  for (int rep = 0; rep < 3; ++rep) chain_kernel<CH><<<blocks, 64>>>(1e-3, 2e-3, cyc, sink);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * sizeof(*cyc), hipMemcpyDeviceToHost);
  unsigned long long mn = ~0ull, mx = 0;
  for (auto v : h) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
  // s_memtime ticks are shader cycles
  printf("%-24s %d MFMAs per chain: cycles min %llu max %llu -> %.1f cycles per chain step, %.1f per MFMA\n", what, 1024, mn, mx,
         mn / 1024.0, mn / 1024.0 / CH);
  hipFree(cyc);
  hipFree(sink);
}

int main() {
  run<1>("1 chain (dependent)");
  run<2>("2 chains interleaved");
  run<4>("4 chains interleaved");
  run<8>("8 chains interleaved");
  return 0;
}
