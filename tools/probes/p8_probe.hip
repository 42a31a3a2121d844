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
  const int mode = argc > 4 ? atoi(argv[4]) : 0;  // 0 bf16, 1 GELU + saved pre-activation, 2 fp32 out, 3 x GELU'(aux), 4 bf16 + dropout
  bf16 *A, *B, *C;
  hipMalloc(&A, (size_t)M * K * 2);
  hipMalloc(&B, (size_t)N * K * 2);
  hipMalloc(&C, (size_t)M * N * 4);
  bf16* AUX;
  hipMalloc(&AUX, (size_t)M * N * 2);
  hipMemset(AUX, 0, (size_t)M * N * 2);
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
  if (mode == 1) { p.epilogue = SSAK_EPI_GELU; p.aux_out = AUX; }
  if (mode == 2) p.out_f32 = 1;
  if (mode == 3) { p.epilogue = SSAK_EPI_MUL_GELU_GRAD; p.aux_in = AUX; }
  if (mode == 4 || mode == 1) { p.drop_thresh = 6554; p.drop_scale = 1.f / 0.9f; p.drop_seed = 1234; p.drop_stream = 3; }
  const int ntiles = p.tiles_m * p.tiles_n;
  const int nblk = std::min(ntiles, 256);  // persistent: one workgroup per CU; stamps 1-3 are those of its LAST tile
  unsigned long long* st;
  hipMalloc(&st, (size_t)nblk * 256);
  hipMemset(st, 0, (size_t)nblk * 256);
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
  std::vector<unsigned long long> hs((size_t)nblk * 32);
  hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long t00 = ~0ull;
  for (int b = 0; b < nblk; ++b) t00 = std::min(t00, hs[b * 32]);
  printf("M=%d N=%d K=%d tiles=%d workgroups=%d mode=%d  event %.1f us\n", M, N, K, ntiles, nblk, mode, ms * 1e3);
  const int rounds = std::min(4, (ntiles + nblk - 1) / nblk);
  for (int r = 0; r < rounds; ++r) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0; int n = 0;
    for (int b = 0; b < nblk; ++b) {
      if (b + r * nblk >= ntiles) continue;
      const unsigned long long* s = &hs[(b * 4 + r) * 8];
      s0 += (s[0] - t00) * 0.01; s1 += (s[1] - t00) * 0.01; s2 += (s[2] - t00) * 0.01; s3 += (s[3] - t00) * 0.01; s4 += (s[4] - t00) * 0.01; s5 += (s[5] - t00) * 0.01; ++n;
    }
    printf("  round %d (%3d wgs): mean tile start %.2f  primed %.2f  main loop done %.2f  epilogue issued: wave 0 %.2f all %.2f  drained %.2f us\n", r, n, s0 / n, s1 / n, s2 / n, s3 / n, s4 / n, s5 / n);
  }
  return 0;
}
