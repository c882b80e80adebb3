// Host-side packing code (csrc/cf_pack.h: both factor packings, the threaded long-double inversion, the probes) under
// AddressSanitizer + UBSan or ThreadSanitizer on the CPU.  GPU sanitizers are not available on the pool; this covers the
// code that indexes the fragment streams the kernels read, for sizes on and off every tile / block boundary.
//
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all -pthread tools/pack_sanitize.cpp -o /tmp/ps_asan
//   g++ -O1 -g -std=c++17 -fsanitize=thread -pthread tools/pack_sanitize.cpp -o /tmp/ps_tsan
//
// Prints one line per size: n, worst probe of the blocked pack and of the inverse pack; exit code 1 if either exceeds 1e-11.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../cosmology-model-fit_amd/csrc/cf_pack.h"

// SPD covariance diag(sigma^2) + A A^T (rank 8), its Cholesky factor with NaN above the diagonal: only L[i][j <= i]
// may be read (cho_factor(..., lower=True)[0] semantics, sn/pantheon.py:14).
static std::vector<double> make_factor(int64_t n, int64_t ld, unsigned seed) {
  std::mt19937_64 rng(seed);
  std::normal_distribution<double> g(0.0, 1.0);
  std::uniform_real_distribution<double> u(0.1, 0.3);
  const int r = 8;
  std::vector<double> A((size_t)n * r), C((size_t)n * n, 0.0), L((size_t)n * ld, std::nan(""));
  for (auto& a : A) a = 0.05 * g(rng);
  for (int64_t i = 0; i < n; ++i) {
    for (int64_t j = 0; j <= i; ++j) {
      double s = 0.0;
      for (int k = 0; k < r; ++k) s += A[i * r + k] * A[j * r + k];
      C[i * n + j] = s;
    }
    const double sg = u(rng);
    C[i * n + i] += sg * sg;
  }
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j <= i; ++j) {
      double s = C[i * n + j];
      for (int64_t k = 0; k < j; ++k) s -= L[i * ld + k] * L[j * ld + k];
      L[i * ld + j] = i == j ? std::sqrt(s) : s / L[j * ld + j];
    }
  return L;
}

int main(int argc, char** argv) {
  std::vector<int64_t> sizes = {1, 2, 15, 16, 17, 63, 64, 65, 255, 256, 257, 300, 511, 513};
  if (argc > 1) {
    sizes.clear();
    for (int i = 1; i < argc; ++i) sizes.push_back(std::atoll(argv[i]));
  }
  int bad = 0;
  for (int64_t n : sizes) {
    const int64_t ld = n + 3;  // a leading dimension larger than n: rows are not contiguous
    const std::vector<double> L = make_factor(n, ld, 1000u + (unsigned)n);
    cf_host_pack pk;
    if (cf_pack_cholesky(L.data(), n, ld, pk) != 0) {
      std::printf("n = %lld: cf_pack_cholesky failed\n", (long long)n);
      return 2;
    }
    const double p_blocked = cf_pack_probe(pk, L.data(), ld);
    cf_host_invpack ip;
    cf_pack_inverse(L.data(), n, ld, ip);
    const double p_inverse = cf_invpack_probe(ip, L.data(), ld);
    std::printf("n = %4lld  n_pad = %4lld  blocked probe %.2e  inverse probe %.2e\n", (long long)n, (long long)pk.n_pad, p_blocked,
                p_inverse);
    if (!(p_blocked < 1e-11) || !(p_inverse < 1e-11)) bad = 1;
  }
  return bad;
}
