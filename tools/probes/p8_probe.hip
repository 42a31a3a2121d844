// Development probe: per-workgroup wall-clock stamps of gemm_p8_kernel (start / pipeline primed / main loop done /
// epilogue issued / stores drained) on one shape.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude
//   -Issak_amd/csrc -DP8_STAMPS tools/probes/p8_probe.hip -o build/p8_probe ; ./build/p8_probe M N K
#include "../../ssak_amd/csrc/gemm_p8.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

void ssak_set_error(const char*, ...) {}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 15968, N = argc > 2 ? atoi(argv[2]) : 2304, K = argc > 3 ? atoi(argv[3]) : 768;
  bf16 *A, *B, *C;
  hipMalloc(&A, (size_t)M * K * 2);
  hipMalloc(&B, (size_t)N * K * 2);
  hipMalloc(&C, (size_t)M * N * 2);
  std::vector<unsigned short> h((size_t)std::max(M, N) * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (rand() & 0xff);
  hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
#ifndef P8_MH
#define P8_MH 4
#endif
  GemmParams p{};
  p.A = A; p.B = B; p.C = C;
  p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N;
  p.nb2 = 1; p.alpha = 1.f; p.split_k = 1; p.nz = 1;
  p.tiles_m = (M + 64 * P8_MH - 1) / (64 * P8_MH); p.tiles_n = (N + 255) / 256;
  p.kt_per_split = (K + 63) / 64;
  p.ext_a = (uint32_t)((size_t)M * K * 2); p.ext_b = (uint32_t)((size_t)N * K * 2);
  const int nblk = p.tiles_m * p.tiles_n;
  unsigned long long* st;
  hipMalloc(&st, (size_t)nblk * 64);
  p.slab = (float*)st;
  #ifndef P8_MH
#define P8_MH 4
#endif
  auto kern = gemm_p8_kernel<P8_MH, false, false>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    kern<<<nblk, P8_THREADS, P8_LDS>>>(p);
    hipEventRecord(e1);
    hipDeviceSynchronize();
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> hs((size_t)nblk * 8);
  hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long t00 = ~0ull, tend = 0;
  for (int b = 0; b < nblk; ++b) { t00 = std::min(t00, hs[b * 8]); tend = std::max(tend, hs[b * 8 + 4]); }
  printf("M=%d N=%d K=%d blocks=%d  event %.1f us  stamps span %.1f us (100 MHz clock)\n", M, N, K, nblk, ms * 1e3, (tend - t00) * 0.01);
  double s_pro = 0, s_main = 0, s_epi = 0, s_drain = 0;
  for (int b = 0; b < nblk; ++b) {
    const unsigned long long* s = &hs[b * 8];
    s_pro += (s[1] - s[0]) * 0.01; s_main += (s[2] - s[1]) * 0.01; s_epi += (s[3] - s[2]) * 0.01; s_drain += (s[4] - s[3]) * 0.01;
  }
  printf("mean per workgroup: prologue %.2f us, main loop %.2f us, epilogue issue %.2f us, store drain %.2f us\n", s_pro / nblk,
         s_main / nblk, s_epi / nblk, s_drain / nblk);
  // timeline of a few workgroups + per-CU occupancy
  for (int b = 0; b < nblk; b += std::max(1, nblk / 24)) {
    const unsigned long long* s = &hs[b * 8];
    printf("wg %4d xcc %llu hwid %08llx: start %.2f primed %.2f main_end %.2f epi %.2f end %.2f\n", b, s[5] >> 32, s[5] & 0xffffffffu,
           (s[0] - t00) * 0.01, (s[1] - t00) * 0.01, (s[2] - t00) * 0.01, (s[3] - t00) * 0.01, (s[4] - t00) * 0.01);
  }
  return 0;
}
