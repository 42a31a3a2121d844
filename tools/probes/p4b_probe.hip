// Development probe (round 4): main loop of a FOUR-wave 256 x 256-tile bf16 GEMM, one wave per SIMD, 128 x 128 register
// tiles (0.25 ds_read_b128 per MFMA instead of the 0.375 of the 8-wave kernel), K-contiguous operands.
//   * LDS: two 64 KB buffers, each one 64-deep K tile: A [256 rows][128 B] | B [256 rows][128 B], 16-byte chunk c of row r at
//     slot c ^ ((r >> 1) & 7) (conflict-free for ds_read_b128 under the real service groups); filled by LDS-DMA in whole
//     128-byte lines (8 rows per wave-instruction), swizzle on the source side.
//   * ONE raw s_barrier and ONE vmcnt wait per K tile, between its two 32-deep slices: at that point every wave has the
//     fragments of slice 1 in registers (buffer t is free -> tile t + 2 is staged into it right after the barrier, sixteen
//     LDS-DMA instructions per wave spread between the MFMAs) and waits for its share of tile t + 1 (issued a whole tile ago).
//   * fragments of the next slice are read while the MFMAs of the current one run (two register sets).
// Round 2's prototype of this shape (p4_probe.hip) lost to the 8-wave kernel -- its __syncthreads() drained the LDS-DMA
// pipeline (s_waitcnt vmcnt(0)) on every other K step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Issak_amd/csrc tools/probes/p4b_probe.hip -o tools/probes/p4b_probe.bin
#include "../../ssak_amd/csrc/common.h"
#include "../../ssak_amd/csrc/gemm_common.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

void ssak_set_error(const char*, ...) {}

namespace {

constexpr int P4_OP = 32768;       // one operand of one K tile
constexpr int P4_BUF = 2 * P4_OP;  // A | B
#ifndef P4B_DMA_S1
#define P4B_DMA_S1 16  // LDS-DMA instructions issued in the slice right after the barrier (the rest in the next slice)
#endif

#define P4_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int NI>
__global__ __launch_bounds__(256) void gemm_p4b_probe(const bf16* A, const bf16* B, bf16* C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  constexpr int BM = 32 * NI;
  const int tiles_n = N / 256;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int bm0 = (id / tiles_n) * BM, bn0 = (id % tiles_n) * 256;
  const int nkt = K / 64;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((size_t)M * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)((size_t)N * K * 2), 0x00020000);
  // LDS-DMA: instruction j of this wave fills bytes [(4 j + wave) KiB, + 1 KiB) of an operand = rows 32 j + 8 wave + (lane >> 3)
  const int drow = 8 * wave + (lane >> 3);
  const int dchunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  uint32_t va = (uint32_t)(((long)(bm0 + drow) * K + dchunk * 8) * 2);
  uint32_t vb = (uint32_t)(((long)(bn0 + drow) * K + dchunk * 8) * 2);
  const uint32_t jstride = (uint32_t)(32 * K * 2);
  auto dma = [&](char* buf, int q) {  // q = 0..15: A instructions 0..NA-1, then B
    constexpr int NA = BM / 32;
    if (q < NA) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(buf + (4 * q + wave) * 1024), 16, va + q * jstride, 0, 0, 0);
    } else if (q < NA + 8) {
      const int j = q - NA;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(buf + P4_OP + (4 * j + wave) * 1024), 16, vb + j * jstride, 0, 0, 0);
    }
  };
  constexpr int NQ = BM / 32 + 8;
  // fragment reads: row 16 i + lm of this wave's panel, 16-byte chunk 4 kk + lq
  const int lm = lane & 15, lq = lane >> 4;
  const int fo_a = (wr * 16 * NI + lm) * 128 + ((lq ^ ((lm >> 1) & 7)) << 4);
  const int fo_b = P4_OP + (wc * 128 + lm) * 128 + ((lq ^ ((lm >> 1) & 7)) << 4);

  f32x4 acc[NI][8];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  char* const buf0 = smem;
  char* const buf1 = smem + P4_BUF;
  // prologue: tiles 0 and 1 in flight
#pragma unroll
  for (int q = 0; q < NQ; ++q) dma(buf0, q);
  va += 128, vb += 128;
  if (nkt > 1) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) dma(buf1, q);
  }
  va += 128, vb += 128;
  if (nkt > 1) wait_vmcnt<NQ>(); else wait_vmcnt<0>();
  P4_FENCE();
  __builtin_amdgcn_s_barrier();
  P4_FENCE();
  bf16x8 fa0[NI], fb0[8], fa1[NI], fb1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) fb0[j] = *reinterpret_cast<const bf16x8*>(buf0 + fo_b + j * 2048);
#pragma unroll
  for (int i = 0; i < NI; ++i) fa0[i] = *reinterpret_cast<const bf16x8*>(buf0 + fo_a + i * 2048);

  for (int t = 0; t < nkt; ++t) {
    char* const cur = (t & 1) ? buf1 : buf0;
    char* const nxt = (t & 1) ? buf0 : buf1;
    // ---- slice 0: MFMAs on set 0, fragments of slice 1 (same buffer) into set 1
#pragma unroll
    for (int j = 0; j < 8; ++j) fb1[j] = *reinterpret_cast<const bf16x8*>(cur + (fo_b ^ 64) + j * 2048);
#pragma unroll
    for (int i = 0; i < NI; ++i) fa1[i] = *reinterpret_cast<const bf16x8*>(cur + (fo_a ^ 64) + i * 2048);
    if (P4B_DMA_S1 < NQ && t >= 1 && t + 1 < nkt) {
#pragma unroll
      for (int q = P4B_DMA_S1; q < NQ; ++q) dma(nxt, q);  // the rest of tile t + 1 (its head went out in slice 1 of tile t - 1)
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j], fa0[i], acc[i][j], 0, 0, 0);
    // schedule: one fragment read per two MFMAs (the sixteen reads are out by MFMA 32), LDS-DMA one per two MFMAs after them
#pragma unroll
    for (int g = 0; g < 8 + NI; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#pragma unroll
    for (int g = 0; g < NQ - P4B_DMA_S1; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    P4_FENCE();
    // ---- between the slices: my reads of `cur` are complete, my share of tile t + 1 has landed
    if (P4B_DMA_S1 < NQ) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    P4_FENCE();
    __builtin_amdgcn_s_barrier();
    P4_FENCE();
    // ---- slice 1: MFMAs on set 1, fragments of tile t + 1 slice 0 into set 0, tile t + 2 staged into `cur`
    if (t + 1 < nkt) {
#pragma unroll
      for (int j = 0; j < 8; ++j) fb0[j] = *reinterpret_cast<const bf16x8*>(nxt + fo_b + j * 2048);
#pragma unroll
      for (int i = 0; i < NI; ++i) fa0[i] = *reinterpret_cast<const bf16x8*>(nxt + fo_a + i * 2048);
    }
    if (t + 2 < nkt) {
#pragma unroll
      for (int q = 0; q < (P4B_DMA_S1 < NQ ? P4B_DMA_S1 : NQ); ++q) dma(cur, q);
    }
    va += 128, vb += 128;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j], fa1[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 8 + NI; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    P4_FENCE();
  }
  wait_vmcnt<0>();
  // epilogue (timing + check only): lane (lm, lq) of block (i, j) holds row 16 i + lm, columns 16 j + 4 lq ..
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bf16x4 o = {(bf16)acc[i][j][0], (bf16)acc[i][j][1], (bf16)acc[i][j][2], (bf16)acc[i][j][3]};
      const int row = bm0 + wr * 16 * NI + 16 * i + lm;
      if (row < M) *reinterpret_cast<bf16x4*>(C + (long)row * N + bn0 + wc * 128 + 16 * j + 4 * lq) = o;
    }
}

}  // namespace

static unsigned short tobf(float f) {
  union { float f; unsigned u; } c;
  c.f = f;
  return (unsigned short)((c.u + 0x7fff + ((c.u >> 16) & 1)) >> 16);
}
static float tof(unsigned short h) {
  union { float f; unsigned u; } c;
  c.u = (unsigned)h << 16;
  return c.f;
}

template <int NI>
static void run(int M, int N, int K) {
  bf16 *A, *B, *C;
  hipMalloc(&A, (size_t)M * K * 2);
  hipMalloc(&B, (size_t)N * K * 2);
  hipMalloc(&C, (size_t)M * N * 2);
  std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K);
  srand(1);
  for (auto& v : ha) v = tobf((float)rand() / RAND_MAX * 2.f - 1.f);  // uniform [-1, 1): the chip clocks lower on random data
  for (auto& v : hb) v = tobf((float)rand() / RAND_MAX * 2.f - 1.f);
  hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)gemm_p4b_probe<NI>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P4_BUF);
  const int grid = ((M + 32 * NI - 1) / (32 * NI)) * (N / 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int it = 0; it < 5; ++it) gemm_p4b_probe<NI><<<grid, 256, 2 * P4_BUF>>>(A, B, C, M, N, K);
  hipDeviceSynchronize();
  const int iters = 30;
  hipEventRecord(e0);
  for (int it = 0; it < iters; ++it) gemm_p4b_probe<NI><<<grid, 256, 2 * P4_BUF>>>(A, B, C, M, N, K);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  std::vector<unsigned short> hc((size_t)M * N);
  hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int s = 0; s < 400; ++s) {
    const int m = s < 8 ? M - 1 - s : rand() % M, n = s < 8 ? N - 1 - 37 * s : rand() % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)tof(ha[(size_t)m * K + k]) * tof(hb[(size_t)n * K + k]);
    const float got = tof(hc[(size_t)m * N + n]);
    if (fabs(got - ref) > 0.01 * fabs(ref) + 0.02 * sqrt((double)K)) ++bad;
  }
  printf("p4b<%d> M=%6d N=%5d K=%5d grid %4d: %8.1f us  %7.1f TF/s   spot check %d / 400 bad\n", NI, M, N, K, grid, ms * 1e3,
         2.0 * M * N * K / (ms * 1e-3) / 1e12, bad);
  hipFree(A);
  hipFree(B);
  hipFree(C);
}

int main(int argc, char** argv) {
  run<8>(4096, 4096, 4096);
  run<8>(8192, 8192, 8192);
  run<8>(15968, 3072, 768);
  run<8>(15968, 768, 3072);
  run<8>(15968, 2304, 768);
  run<6>(15968, 768, 768);
  run<6>(15968, 768, 3072);
  run<6>(15968, 2304, 768);
  return 0;
}
