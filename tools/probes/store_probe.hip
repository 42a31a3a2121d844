// Development probe: cost of global_store_dwordx4 patterns as a GEMM epilogue issues them.  512-thread workgroups, each wave
// stores a 128 x 64 bf16 sub-tile (16 KiB) of a row-major [M, ld] matrix in one of several lane->address mappings.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_probe.hip -o tools/probes/store_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned short* C, long ld, int tiles_n, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  for (int it = 0; it < iters; ++it) {
    const int t = blockIdx.x + it * gridDim.x;
    const int tm = t / tiles_n, tn = t % tiles_n;
    unsigned short* base = C + (long)(tm * 256 + wr * 128) * ld + tn * 256 + wc * 64;
    const u32x4 v = {(unsigned)lane, (unsigned)t, 3u, 4u};
    if (MODE == 0) {  // 16 rows x 64 B per instruction (lane = row lm + 16 * lq, 16 B at column 16*lq; second store +8)
      const int lm = lane & 15, lq = lane >> 4;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        unsigned short* p = base + (long)(16 * i + lm) * ld + 16 * lq;
        *reinterpret_cast<u32x4*>(p) = v;
        *reinterpret_cast<u32x4*>(p + 8) = v;
      }
    } else if (MODE == 1) {  // 8 rows x 128 B per instruction
      const int r = lane >> 3, c = lane & 7;
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<u32x4*>(base + (long)(8 * i + r) * ld + 8 * c) = v;
    } else if (MODE == 2) {  // 16 rows x 32 B per instruction, 8-byte stores (the raw accumulator layout)
      const int lm = lane & 15, lq = lane >> 4;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<u32x2*>(base + (long)(16 * i + lm) * ld + 16 * j + 4 * lq) = (u32x2){v[0], v[1]};
    } else if (MODE == 3) {  // fp32 accumulator layout: 16 rows x 64 B per instruction, 4 per 16 rows (C is fp32 here)
      const int lm = lane & 15, lq = lane >> 4;
      float* b = reinterpret_cast<float*>(C) + (long)(tm * 256 + wr * 128) * ld + tn * 256 + wc * 64;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(b + (long)(16 * i + lm) * ld + 16 * j + 4 * lq) = v;
    } else if (MODE == 4) {  // fp32, 4 rows x 256 B per instruction
      const int r = lane >> 4, c = lane & 15;
      float* b = reinterpret_cast<float*>(C) + (long)(tm * 256 + wr * 128) * ld + tn * 256 + wc * 64;
#pragma unroll
      for (int i = 0; i < 32; ++i) *reinterpret_cast<u32x4*>(b + (long)(4 * i + r) * ld + 4 * c) = v;
    }
  }
}

int main(int argc, char** argv) {
  const int M = 15968 / 256 * 256, N = 3072, iters = 3, grid = argc > 1 ? atoi(argv[1]) : 248;
  unsigned short* C;
  hipMalloc(&C, (size_t)M * N * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[] = {"bf16 16 rows x 64 B (2 x 16 B per lane)", "bf16 8 rows x 128 B", "bf16 16 rows x 32 B (8 B per lane)",
                         "fp32 16 rows x 64 B", "fp32 4 rows x 256 B"};
  for (int mode = 0; mode < 5; ++mode) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      switch (mode) {
        case 0: k<0><<<grid, 512>>>(C, N, N / 256, iters); break;
        case 1: k<1><<<grid, 512>>>(C, N, N / 256, iters); break;
        case 2: k<2><<<grid, 512>>>(C, N, N / 256, iters); break;
        case 3: k<3><<<grid, 512>>>(C, N, N / 256, iters); break;
        case 4: k<4><<<grid, 512>>>(C, N, N / 256, iters); break;
      }
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double bytes = (double)grid * iters * 256 * 256 * (mode >= 3 ? 4 : 2);
    printf("%-44s grid %3d: %.1f us for %d tiles per workgroup = %.2f us per 256x256 tile, %.2f TB/s\n", names[mode], grid, best * 1e3, iters,
           best * 1e3 / iters, bytes / (best * 1e-3) / 1e12);
  }
  return 0;
}
