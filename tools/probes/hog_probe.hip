// Development probe: how a persistent one-workgroup-per-CU GEMM behaves when another kernel (standing in for an RCCL
// all-reduce running on its own stream during the backward) already holds some of the CUs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/probes/hog_probe.hip -Lssak_amd/lib -lssak_hip -Wl,-rpath,$PWD/ssak_amd/lib -o tools/probes/hog_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ssak_hip.h"

// one workgroup per CU (1024 threads, 64 KB LDS: nothing of a 136 KB GEMM workgroup fits beside it), spinning for `us`
__global__ __launch_bounds__(1024) void hog(float* out, long ticks) {
  extern __shared__ float sm[];
  const long t0 = wall_clock64();
  float a = threadIdx.x;
  while (wall_clock64() - t0 < ticks) a = a * 1.0001f + 0.5f;
  sm[threadIdx.x] = a;
  if (a == 12345.f) out[blockIdx.x] = sm[(threadIdx.x + 1) & 1023];
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 15968, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  const int hog_wgs = argc > 4 ? atoi(argv[4]) : 32;
  const float hog_us = argc > 5 ? atof(argv[5]) : 1000.f;
  void *A, *B, *C;
  float* dummy;
  hipMalloc(&A, (size_t)M * K * 2);
  hipMalloc(&B, (size_t)N * K * 2);
  hipMalloc(&C, (size_t)M * N * 2);
  hipMalloc(&dummy, 4096);
  hipMemset(A, 0, (size_t)M * K * 2);
  hipMemset(B, 0, (size_t)N * K * 2);
  hipStream_t s1, s2;
  hipStreamCreate(&s1);
  hipStreamCreate(&s2);
  hipFuncSetAttribute((const void*)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  ssak_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.M = M, d.N = N, d.K = K, d.lda = K, d.ldb = K, d.ldc = N, d.nb1 = d.nb2 = 1, d.alpha = 1.f, d.split_k = 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const long ticks = (long)(hog_us * 100.0);  // wall_clock64 runs at 100 MHz
  for (int mode = 0; mode < 2; ++mode) {
    float best = 1e9, worst = 0;
    for (int it = 0; it < 8; ++it) {
      hipDeviceSynchronize();
      if (mode == 1) hog<<<hog_wgs, 1024, 65536, s2>>>(dummy, ticks);
      if (mode == 1) {
        // give the hog time to be resident before the GEMM is enqueued
        const long t0 = clock();
        while (clock() - t0 < CLOCKS_PER_SEC / 5000) {
        }
      }
      hipEventRecord(e0, s1);
      for (int r = 0; r < 4; ++r)
        if (ssak_gemm_bf16(&d, A, B, C, nullptr, nullptr, nullptr, nullptr, 0, s1)) {
          printf("gemm failed: %s\n", ssak_last_error());
          return 1;
        }
      hipEventRecord(e1, s1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
      worst = ms > worst ? ms : worst;
    }
    printf("%s: 4 GEMMs %d x %d x %d  best %.1f us  worst %.1f us  (per GEMM %.1f)\n",
           mode ? "with a hog on some CUs" : "alone                 ", M, N, K, best * 1e3, worst * 1e3, best * 250.f);
  }
  return 0;
}
