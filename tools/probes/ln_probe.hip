// Development probe: the LayerNorm row kernels on the train step's shape (M = 15 968 rows of 768), with and without the
// hidden-dropout hash, next to a plain 2-read / 2-write streaming kernel of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -Iinclude -Issak_amd/csrc -x hip tools/probes/ln_probe.hip ssak_amd/csrc/api.cpp -o tools/probes/ln_probe.run
#include "../../ssak_amd/csrc/norm_act.hip"

#include <cstdio>
#include <vector>

__global__ void stream22_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ c, uint4* __restrict__ d, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const uint4 x = a[i], y = b[i];
    c[i] = make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
    d[i] = make_uint4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

template <typename F>
static float timeit(F f, int n = 50) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) f();
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / n * 1e3f;
}

int main() {
  const int M = 15968, C = 768;
  const size_t n = (size_t)M * C;
  bf16 *y, *res, *r, *out, *g1, *dr, *dy;
  float *gamma, *beta, *mean, *rstd, *dgamma, *dbeta, *partial;
  hipMalloc(&y, n * 2); hipMalloc(&res, n * 2); hipMalloc(&r, n * 2); hipMalloc(&out, n * 2); hipMalloc(&g1, n * 2); hipMalloc(&dr, n * 2); hipMalloc(&dy, n * 2);
  hipMalloc(&gamma, C * 4); hipMalloc(&beta, C * 4); hipMalloc(&mean, M * 4); hipMalloc(&rstd, M * 4); hipMalloc(&dgamma, C * 4); hipMalloc(&dbeta, C * 4);
  hipMalloc(&partial, (size_t)LN_BWD_BLOCKS * 3 * C * 4);
  std::vector<unsigned short> h(n);
  srand(1);
  for (size_t i = 0; i < n; ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15));
  hipMemcpy(y, h.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(res, h.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(g1, h.data(), n * 2, hipMemcpyHostToDevice);
  std::vector<float> ones(C, 1.f);
  hipMemcpy(gamma, ones.data(), C * 4, hipMemcpyHostToDevice); hipMemset(beta, 0, C * 4);
  DropSpec none, drop;
  drop.seed = 1234; drop.stream = 17; drop.p = 0.1f;
  const double fwd_bytes = 4.0 * n * 2, bwd_bytes = 5.0 * n * 2;
  float t;
  t = timeit([&] { stream22_kernel<<<2048, 256>>>((const uint4*)y, (const uint4*)res, (uint4*)r, (uint4*)out, (long)(n / 8)); });
  printf("stream 2 reads + 2 writes           %7.1f us  %5.2f TB/s\n", t, fwd_bytes / t / 1e6);
  t = timeit([&] { k_layernorm_fwd(y, res, gamma, beta, r, out, mean, rstd, M, C, 1e-5f, none, none, 0); });
  printf("ln_fwd  no dropout                  %7.1f us  %5.2f TB/s\n", t, fwd_bytes / t / 1e6);
  t = timeit([&] { k_layernorm_fwd(y, res, gamma, beta, r, out, mean, rstd, M, C, 1e-5f, drop, none, 0); });
  printf("ln_fwd  hidden dropout on y         %7.1f us  %5.2f TB/s\n", t, fwd_bytes / t / 1e6);
  t = timeit([&] { k_layernorm_bwd(g1, nullptr, r, mean, rstd, gamma, nullptr, dr, dy, dgamma, dbeta, partial, M, C, none, none, 0); });
  printf("ln_bwd  no dropout (g1, r -> dr, dy) %7.1f us  %5.2f TB/s (4 streams)\n", t, 4.0 * n * 2 / t / 1e6);
  t = timeit([&] { k_layernorm_bwd(g1, y, r, mean, rstd, gamma, nullptr, dr, dy, dgamma, dbeta, partial, M, C, drop, none, 0, DropSpec(), dbeta); });
  printf("ln_bwd  g1 + g2, dropout, dy colsum %7.1f us  %5.2f TB/s (5 streams)\n", t, bwd_bytes / t / 1e6);
  return 0;
}
