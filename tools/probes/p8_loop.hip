// Development probe: the persistent 256 x 256 GEMM kernel (lean bf16 epilogue) timed on the train step's shapes, for A/B runs
// of main-loop variants selected with -D flags in gemm_p8.hip.  Prints us, TFLOP/s and an order-independent checksum of C
// (two builds that compute the same thing print the same checksum).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Issak_amd/csrc -mllvm -amdgpu-mfma-vgpr-form [-DP8_VARIANT=n]
//         -x hip tools/probes/p8_loop.hip ssak_amd/csrc/api.cpp -o tools/probes/p8_loop_<tag>.bin
#include "../../ssak_amd/csrc/gemm_p8.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void checksum_kernel(const unsigned short* c, long n, unsigned long long* out) {
  unsigned long long s = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += (unsigned long long)c[i] * (unsigned long long)((i % 8191) + 1);
  atomicAdd(out, s);
}

template <int EPI, bool B_KM>
static void run_epi(const char* name, int M, int N, int K) {
  bf16 *A, *B, *C, *AUX;
  float* CS;
  hipMalloc(&A, (size_t)M * K * 2);
  hipMalloc(&B, (size_t)N * K * 2);
  hipMalloc(&C, (size_t)M * N * 2);
  hipMalloc(&AUX, (size_t)M * N * 2);
  hipMalloc(&CS, (size_t)((M + 63) / 64) * N * 4);
  std::vector<unsigned short> h((size_t)std::max(M, N) * K);
  srand(1);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15));
  hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  hipMemset(AUX, 0x3c, (size_t)M * N * 2);
  GemmParams p{};
  p.A = A; p.B = B; p.C = C;
  p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = B_KM ? N : K; p.ldc = N;
  p.nb2 = 1; p.alpha = 0.01f; p.split_k = 1; p.nz = 1;
  p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256;
  p.kt_per_split = (K + 63) / 64;
  p.ext_a = (uint32_t)((size_t)M * K * 2); p.ext_b = (uint32_t)((size_t)N * K * 2);
  p.epilogue = EPI == 3 ? SSAK_EPI_GELU_SAVE_GRAD : EPI == 4 ? SSAK_EPI_MUL_AUX : SSAK_EPI_NONE;
  if (EPI == 3) { p.aux_out = AUX; p.drop_thresh = 6554; p.drop_scale = 1.f / 0.9f; p.drop_seed = 1234; p.drop_stream = 3; }
  if (EPI == 4) { p.aux_in = AUX; p.colsum = CS; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) launch_p8<4, false, B_KM, EPI>(p, 0);
  hipDeviceSynchronize();
  const int iters = 30;
  hipEventRecord(e0);
  for (int it = 0; it < iters; ++it) launch_p8<4, false, B_KM, EPI>(p, 0);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  unsigned long long* cs;
  hipMalloc(&cs, 8);
  hipMemset(cs, 0, 8);
  checksum_kernel<<<1024, 256>>>((const unsigned short*)C, (long)M * N, cs);
  unsigned long long hcs = 0;
  hipMemcpy(&hcs, cs, 8, hipMemcpyDeviceToHost);
  printf("%-18s M=%6d N=%5d K=%5d          %8.1f us  %7.1f TF/s  checksum %016llx\n", name, M, N, K, ms * 1e3, 2.0 * M * N * (double)K / ms / 1e9, hcs);
  hipFree(A); hipFree(B); hipFree(C); hipFree(AUX); hipFree(CS); hipFree(cs);
}

static void run(const char* name, int M, int N, int K, int nb) {
  bf16 *A, *B, *C;
  hipMalloc(&A, (size_t)nb * M * K * 2);
  hipMalloc(&B, (size_t)N * K * 2);
  hipMalloc(&C, (size_t)nb * M * N * 2);
  std::vector<unsigned short> h((size_t)std::max((size_t)nb * M, (size_t)N) * K);
  srand(1);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15));  // random mantissas and signs
  hipMemcpy(A, h.data(), (size_t)nb * M * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  GemmParams p{};
  p.A = A; p.B = B; p.C = C;
  p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N;
  p.nb2 = 1; p.alpha = 1.f; p.split_k = 1; p.nz = nb;
  p.sa1 = (long)M * K; p.sc1 = (long)M * N;
  p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256;
  p.kt_per_split = (K + 63) / 64;
  p.ext_a = (uint32_t)std::min<size_t>((size_t)nb * M * K * 2, 0xffffffffu); p.ext_b = (uint32_t)((size_t)N * K * 2);
  if (nb > 1) p.ext_a = (uint32_t)((size_t)M * K * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) launch_p8<4, false, false, P8_EPI_PLAIN_BF16>(p, 0);
  hipDeviceSynchronize();
  const int iters = 30;
  hipEventRecord(e0);
  for (int it = 0; it < iters; ++it) launch_p8<4, false, false, P8_EPI_PLAIN_BF16>(p, 0);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  unsigned long long* cs;
  hipMalloc(&cs, 8);
  hipMemset(cs, 0, 8);
  checksum_kernel<<<1024, 256>>>((const unsigned short*)C, (long)nb * M * N, cs);
  unsigned long long hcs = 0;
  hipMemcpy(&hcs, cs, 8, hipMemcpyDeviceToHost);
  printf("%-10s M=%6d N=%5d K=%5d nb=%3d  %8.1f us  %7.1f TF/s  checksum %016llx\n", name, M, N, K, nb, ms * 1e3, 2.0 * nb * M * N * (double)K / ms / 1e9, hcs);
  hipFree(A); hipFree(B); hipFree(C); hipFree(cs);
}

// B-direct form against the LDS form, same tile height, same operands: time and checksum of both
template <int MH, bool B_KM, int EPI = P8_EPI_PLAIN_BF16>
static void run_bd(const char* name, int M, int N, int K, int nb) {
  bf16 *A, *B, *BF, *C;
  const int nkt = (K + 63) / 64, tiles_n = (N + 255) / 256, nb64 = tiles_n * 4;
  const size_t bf_bytes = (size_t)nb64 * nkt * 8192;
  hipMalloc(&A, (size_t)nb * M * K * 2);
  hipMalloc(&B, (size_t)N * K * 2);
  hipMalloc(&BF, bf_bytes);
  hipMalloc(&C, (size_t)nb * M * N * 2);
  std::vector<unsigned short> h((size_t)std::max((size_t)nb * M, (size_t)N) * K);
  srand(1);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15));
  hipMemcpy(A, h.data(), (size_t)nb * M * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  GemmParams p{};
  p.A = A; p.B = B; p.C = C;
  p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = B_KM ? N : K; p.ldc = N;
  p.nb2 = 1; p.alpha = 1.f; p.split_k = 1; p.nz = nb;
  p.sa1 = (long)M * K; p.sc1 = (long)M * N;
  p.tiles_m = (M + 64 * MH - 1) / (64 * MH); p.tiles_n = tiles_n;
  p.kt_per_split = nkt;
  p.ext_a = (uint32_t)((size_t)M * K * 2); p.ext_b = (uint32_t)((size_t)N * K * 2);
  bf16* AUX = nullptr;
  float* CS = nullptr;
  if (EPI != P8_EPI_PLAIN_BF16) {
    hipMalloc(&AUX, (size_t)M * N * 2);
    hipMalloc(&CS, (size_t)((M + 63) / 64) * N * 4);
    hipMemset(AUX, 0x3c, (size_t)M * N * 2);
    p.alpha = 0.01f;
    p.epilogue = EPI == 3 ? SSAK_EPI_GELU_SAVE_GRAD : SSAK_EPI_MUL_AUX;
    if (EPI == 3) { p.aux_out = AUX; p.drop_thresh = 6554; p.drop_scale = 1.f / 0.9f; p.drop_seed = 1234; p.drop_stream = 3; }
    if (EPI == 4) { p.aux_in = AUX; p.colsum = CS; p.fq_a = FQ_STEP; p.fq_b = -FQ_ZERO * FQ_STEP; }
  }
  hipEvent_t e0, e1, e2, e3;
  hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
  unsigned long long* cs;
  hipMalloc(&cs, 16);
  hipMemset(cs, 0, 16);
  const int iters = 30;
  float ms[2], msf = 0.f;
  for (int v = 0; v < 2; ++v) {
    GemmParams q = p;
    if (v == 1) {
      hipEventRecord(e2);
      const void* Bs[1] = {B}; void* outs[1] = {BF};
      const long ldbs[1] = {p.ldb}; const int Ns[1] = {N}, Ks[1] = {K}, kms[1] = {B_KM ? 1 : 0};
      k_gemm_fragment_b_batched(1, Bs, ldbs, Ns, Ks, kms, outs, 0);
      hipEventRecord(e3);
      q.B = BF;
      q.ext_b = (uint32_t)bf_bytes;
    }
    hipMemset(C, 0, (size_t)nb * M * N * 2);
    for (int it = 0; it < 3; ++it) v ? launch_p8bd<MH, EPI>(q, 0) : launch_p8<MH, false, B_KM, EPI>(q, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < iters; ++it) v ? launch_p8bd<MH, EPI>(q, 0) : launch_p8<MH, false, B_KM, EPI>(q, 0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms[v], e0, e1);
    ms[v] /= iters;
    if (v) hipEventElapsedTime(&msf, e2, e3);
    checksum_kernel<<<1024, 256>>>((const unsigned short*)C, (long)nb * M * N, cs + v);
  }
  unsigned long long hcs[2] = {0, 0};
  hipMemcpy(hcs, cs, 16, hipMemcpyDeviceToHost);
  const double fl = 2.0 * nb * M * N * (double)K / 1e9;
  printf("%-10s MH=%d %s M=%6d N=%5d K=%5d nb=%2d  lds %7.1f us %7.1f TF | b-direct %7.1f us %7.1f TF (%+5.1f %%) frag %6.1f us  %s\n", name, MH,
         B_KM ? "km" : "kc", M, N, K, nb, ms[0] * 1e3, fl / ms[0], ms[1] * 1e3, fl / ms[1], 100.0 * (ms[0] / ms[1] - 1.0), msf * 1e3,
         hcs[0] == hcs[1] ? "checksum equal" : "CHECKSUM DIFFERS");
  hipFree(A); hipFree(B); hipFree(BF); hipFree(C); hipFree(cs);
}

// The same comparison with COLD operands: `copies` distinct A / B / fragment buffers visited round-robin (the train step touches
// every layer's weights once per step, and its activations come from the kernel before)
template <int MH, bool B_KM>
static void run_bd_cold(const char* name, int M, int N, int K, int copies) {
  const int nkt = (K + 63) / 64, tiles_n = (N + 255) / 256, nb64 = tiles_n * 4;
  const size_t bf_bytes = (size_t)nb64 * nkt * 8192;
  std::vector<bf16*> A(copies), B(copies), BF(copies);
  bf16* C;
  hipMalloc(&C, (size_t)M * N * 2);
  std::vector<unsigned short> h((size_t)std::max(M, N) * K);
  srand(1);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15));
  for (int c = 0; c < copies; ++c) {
    hipMalloc(&A[c], (size_t)M * K * 2);
    hipMalloc(&B[c], (size_t)N * K * 2);
    hipMalloc(&BF[c], bf_bytes);
    hipMemcpy(A[c], h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
    hipMemcpy(B[c], h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
    const void* Bs[1] = {B[c]}; void* outs[1] = {BF[c]};
    const long ldbs[1] = {B_KM ? N : K}; const int Ns[1] = {N}, Ks[1] = {K}, kms[1] = {B_KM ? 1 : 0};
    k_gemm_fragment_b_batched(1, Bs, ldbs, Ns, Ks, kms, outs, 0);
  }
  GemmParams p{};
  p.C = C;
  p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = B_KM ? N : K; p.ldc = N;
  p.nb2 = 1; p.alpha = 1.f; p.split_k = 1; p.nz = 1;
  p.tiles_m = (M + 64 * MH - 1) / (64 * MH); p.tiles_n = tiles_n;
  p.kt_per_split = nkt;
  p.ext_a = (uint32_t)((size_t)M * K * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[2];
  for (int v = 0; v < 2; ++v) {
    auto go = [&](int c) {
      GemmParams q = p;
      q.A = A[c];
      q.B = v ? BF[c] : B[c];
      q.ext_b = v ? (uint32_t)bf_bytes : (uint32_t)((size_t)N * K * 2);
      v ? launch_p8bd<MH, P8_EPI_PLAIN_BF16>(q, 0) : launch_p8<MH, false, B_KM, P8_EPI_PLAIN_BF16>(q, 0);
    };
    for (int c = 0; c < copies; ++c) go(c);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < copies; ++c) go(c);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms[v], e0, e1);
    ms[v] /= 3 * copies;
  }
  const double fl = 2.0 * M * N * (double)K / 1e9;
  printf("%-10s MH=%d %s M=%6d N=%5d K=%5d cold x%2d  lds %7.1f us %7.1f TF | b-direct %7.1f us %7.1f TF (%+5.1f %%)\n", name, MH, B_KM ? "km" : "kc", M, N, K,
         copies, ms[0] * 1e3, fl / ms[0], ms[1] * 1e3, fl / ms[1], 100.0 * (ms[0] / ms[1] - 1.0));
  for (int c = 0; c < copies; ++c) { hipFree(A[c]); hipFree(B[c]); hipFree(BF[c]); }
  hipFree(C);
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'c') {
    run_bd_cold<3, false>("qkv", 15968, 2304, 768, 12);
    run_bd_cold<3, false>("out", 15968, 768, 768, 12);
    run_bd_cold<3, false>("ffn_down", 15968, 768, 3072, 12);
    run_bd_cold<3, true>("dx 768", 15968, 768, 768, 12);
    run_bd_cold<3, true>("dx qkv", 15968, 768, 2304, 12);
    run_bd_cold<3, true>("dx w1", 15968, 768, 3072, 12);
    run_bd_cold<3, false>("ffn_down", 15968, 768, 3072, 1);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'b') {
    run_bd<4, false>("ffn_up", 15968, 3072, 768, 1);
    run_bd<3, false>("ffn_up", 15968, 3072, 768, 1);
    run_bd<4, false, 3>("ffn_up sg", 15968, 3072, 768, 1);
    run_bd<3, false, 3>("ffn_up sg", 15968, 3072, 768, 1);
    run_bd<4, true, 4>("ffn_dx2 ma", 15968, 3072, 768, 1);
    run_bd<3, true, 4>("ffn_dx2 ma", 15968, 3072, 768, 1);
    run_bd<3, true>("ffn_dx2", 15968, 3072, 768, 1);
    run_bd<3, true>("ffn_dx", 15968, 768, 3072, 1);
    run_bd<3, false>("conv1", 15999, 512, 1536, 32);
    run_bd<4, true>("ffn_dx", 15968, 768, 3072, 1);
    run_bd<4, true>("ffn_dx2", 15968, 3072, 768, 1);
    run_bd<3, false>("qkv", 15968, 2304, 768, 1);
    run_bd<3, false>("out", 15968, 768, 768, 1);
    run_bd<3, false>("ffn_down", 15968, 768, 3072, 1);
    run_bd<4, false>("ffn_down", 15968, 768, 3072, 1);
    run_bd<3, true>("dx 768", 15968, 768, 768, 1);
    run_bd<3, true>("dx qkv", 15968, 768, 2304, 1);
    run_bd<4, false>("conv1", 15999, 512, 1536, 32);
    run_bd<3, false>("odd", 1000, 300, 72 * 5 + 8, 1);
    run_bd<4, false>("4096^3", 4096, 4096, 4096, 1);
    run_bd<4, false>("8192^3", 8192, 8192, 8192, 1);
    return 0;
  }
  run_epi<P8_EPI_PLAIN_BF16, false>("ffn_up plain", 15968, 3072, 768);
  run_epi<SSAK_EPI_GELU_SAVE_GRAD, false>("ffn_up save_grad", 15968, 3072, 768);
  run_epi<P8_EPI_PLAIN_BF16, true>("ffn_dx plain", 15968, 3072, 768);
  run_epi<SSAK_EPI_MUL_AUX, true>("ffn_dx mul_aux", 15968, 3072, 768);
  run("ffn_up", 15968, 3072, 768, 1);
  run("qkv", 15968, 2304, 768, 1);
  run("ffn_down", 15968, 768, 3072, 1);
  run("conv1", 15999, 512, 1536, 32);
  run("4096^3", 4096, 4096, 4096, 1);
  run("8192^3", 8192, 8192, 8192, 1);
  return 0;
}
