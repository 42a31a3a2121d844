// Development probe (round 4): HAND-SCHEDULED main loop of a four-wave 256 x 256-tile bf16 GEMM, one wave per SIMD, 128 x 128
// register tiles (accumulators in the 256 AccVGPRs, two fragment sets in VGPRs), K-contiguous operands.  Every instruction of
// the loop is an `asm volatile` statement, so the emitted order IS the written order (the compiler only allocates registers):
// MFMAs back to back with at most one ds_read_b128 / LDS-DMA between two of them, ONE raw s_barrier and ONE vmcnt wait per
// 64-deep K tile.  (p4b_probe.hip is the same loop from builtins: hipcc shuffles the 256 accumulators through v_accvgpr
// moves -- ~10 per MFMA -- which is why this shape "lost as compiler-scheduled HIP" in round 2.)
//   * LDS: two 64 KB buffers, each one K tile: A [256 rows][128 B] | B [256 rows][128 B], 16-byte chunk c of row r at slot
//     c ^ ((r >> 1) & 7); filled by LDS-DMA in whole 128-byte lines (8 rows per wave-instruction), swizzle on the source side.
//   * K tile t: slice 0 = MFMAs on fragment set 0 while set 1 (slice 1, same buffer) is read; then
//     s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: buffer t is free and tile t + 1 has landed; slice 1 = MFMAs on set 1 while set 0
//     of tile t + 1 is read from the other buffer and tile t + 2 is staged into buffer t (16 LDS-DMA per wave).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Issak_amd/csrc tools/probes/p4c_probe.hip -o tools/probes/p4c_probe.bin
#include "../../ssak_amd/csrc/common.h"
#include "../../ssak_amd/csrc/gemm_common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

void ssak_set_error(const char*, ...) {}

namespace {

constexpr int P4_OP = 32768;       // one operand of one K tile
constexpr int P4_BUF = 2 * P4_OP;  // A | B
#ifndef P4C_RSTEP
#define P4C_RSTEP 4  // slice 1: one fragment read after every RSTEP-th MFMA
#endif
#ifndef P4C_DSTEP
#define P4C_DSTEP 4  // slice 1: one LDS-DMA per group of DSTEP MFMAs (M0 write in the gap before it)
#endif
#ifndef P4C_R0STEP
#define P4C_R0STEP 2  // slice 0: one fragment read after every R0STEP-th MFMA
#endif

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int I>
using IC = std::integral_constant<int, I>;
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(IC<Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

#define MFMA(ACC, FB, FA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(FB), "v"(FA))
#ifndef P4C_ABL
#define P4C_ABL 0  // ablations (wrong results, upper bounds): 1 no LDS-DMA in the loop, 2 no fragment reads, 4 no barrier, 8 no vmcnt wait
#endif
#if P4C_ABL & 2
#define DSREAD(DST, ADDR, OFF) asm volatile("" : "+v"(DST) : "v"(ADDR))
#else
#define DSREAD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "i"(OFF) : "memory")
#endif

__device__ __forceinline__ void dma16(uint32_t lds_dst, uint32_t voff, u32x4 rsrc) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_dst), "v"(voff), "s"(rsrc) : "memory");
}

template <int NI>
__global__ __launch_bounds__(256) void gemm_p4c_probe(const bf16* A, const bf16* B, bf16* C, int M, int N, int K, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef P4C_STAMP
  const unsigned long long cE = __builtin_amdgcn_s_memtime();
#endif
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  constexpr int BM = 32 * NI;
  constexpr int NA = BM / 32, NQ = NA + 8;  // LDS-DMA instructions per wave and K tile: A, then B
  const int tiles_n = N / 256;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int bm0 = (id / tiles_n) * BM, bn0 = (id % tiles_n) * 256;
  const int nkt = K / 64;  // even, >= 4 (host)
  // raw buffer descriptors (base, stride 0, num_records = bytes, DATA_FORMAT 32): rows beyond the matrix read as zeros
  auto make_rsrc = [](const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)p;
    return (u32x4){(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) & 0xffffu,
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
  };
  const u32x4 ra = make_rsrc(A, (uint32_t)((size_t)M * K * 2));
  const u32x4 rb = make_rsrc(B, (uint32_t)((size_t)N * K * 2));
  // LDS-DMA: instruction j of this wave fills bytes [(4 j + wave) KiB, + 1 KiB) of an operand = rows 32 j + 8 wave + (lane >> 3).
  // One VGPR offset per instruction (row part, constant for the output tile: the range check that zero-fills rows beyond the
  // matrix sees it) + ONE scalar offset for the K tile (stays inside the row).
  const int drow = 8 * wave + (lane >> 3);
  const int dchunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  uint32_t vo[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int row = (q < NA ? bm0 + 32 * q : bn0 + 32 * (q - NA)) + drow;
    vo[q] = (uint32_t)(((long)row * K + dchunk * 8) * 2);
  }
#ifdef P4C_PF
  // L2 prefetch, P4C_PF K tiles ahead: one dword per 128-byte line (lane = row), A rows 64 w .. and B rows 64 w .. of the tile
  const uint32_t pf_a = (wave * 64 + lane < BM) ? (uint32_t)(((long)(bm0 + wave * 64 + lane) * K) * 2) : 0x80000000u;
  const uint32_t pf_b = (uint32_t)(((long)(bn0 + wave * 64 + lane) * K) * 2);
  uint32_t pf_sink = 0;
#endif
  typedef __attribute__((address_space(3))) char lds_char;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)smem;
  const uint32_t wbase = lds0 + wave * 1024;  // + buffer + q * 4096 (B: + 32768 - NA * 4096 more)
  // fragment reads: row 16 i + lm of this wave's panel, 16-byte chunk 4 kk + lq: [buffer][slice]
  const int lm = lane & 15, lq = lane >> 4;
  uint32_t fo_a[2][2], fo_b[2][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      fo_a[b][kk] = lds0 + b * P4_BUF + (wr * 16 * NI + lm) * 128 + (((4 * kk + lq) ^ ((lm >> 1) & 7)) << 4);
      fo_b[b][kk] = lds0 + b * P4_BUF + P4_OP + (wc * 128 + lm) * 128 + (((4 * kk + lq) ^ ((lm >> 1) & 7)) << 4);
    }

  f32x4 acc[NI][8];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // prologue: tiles 0 and 1 in flight
  uint32_t koff = 0;
  static_for<2 * NQ>([&vo, &ra, &rb, wbase](auto qq) {
    constexpr int b = decltype(qq)::value / NQ, q = decltype(qq)::value % NQ;
    const uint32_t dst = wbase + b * P4_BUF + (q < NA ? q * 4096 : P4_OP + (q - NA) * 4096);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(vo[q]), "s"(q < NA ? ra : rb), "s"(b * 128u) : "memory");
  });
  koff = 256;
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NQ) : "memory");
  asm volatile("s_barrier" ::: "memory");
  u32x4 fa0[NI], fb0[8], fa1[NI], fb1[8];
  if (P4C_ABL & 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) fb0[i] = fb1[i] = (u32x4){(unsigned)lane * 0x3f813f81u + i, 0x3f803f80u, 0x3f813f80u, 0x3f803f81u};
#pragma unroll
    for (int i = 0; i < NI; ++i) fa0[i] = fa1[i] = (u32x4){(unsigned)lane * 0x3f813f81u + i, 0x3f803f80u, 0xbf813f80u, 0x3f803f81u};
  }
  static_for<8>([&fb0, &fo_b](auto j) { DSREAD(fb0[j], fo_b[0][0], j * 2048); });
  static_for<NI>([&fa0, &fo_a](auto i) { DSREAD(fa0[i], fo_a[0][0], i * 2048); });

  // one K tile in buffer P.  DMA: stage tile t + 2 into this buffer; READ_NEXT: read set 0 of tile t + 1 from the other one.
  // Every gap between two MFMAs carries at most ONE other instruction: a fragment read, the M0 write of an LDS-DMA or the
  // LDS-DMA itself (one statement = M0 write, MFMA, LDS-DMA: the compiler does not preserve M0 between statements).
#ifdef P4C_PF
#define PF_CAP , pf_a, pf_b, &pf_sink
#else
#define PF_CAP
#endif
  auto ktile = [&acc, &fa0, &fb0, &fa1, &fb1, &vo, &ra, &rb, &koff, &fo_a, &fo_b, wbase PF_CAP](auto par_c, auto dma_c, auto rn_c) {
    constexpr int P = decltype(par_c)::value;
    constexpr bool DMA = decltype(dma_c)::value && !(P4C_ABL & 1), READ_NEXT = decltype(rn_c)::value;
    const uint32_t wb = wbase + P * P4_BUF;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // set 0 is in registers
    // ---- slice 0: 8 NI MFMAs on set 0; set 1 <- slice 1 of this tile
    static_for<8 * NI>([&acc, &fa0, &fb0, &fa1, &fb1, &fo_a, &fo_b, &ra, &rb, &koff PF_CAP](auto mc) {
      constexpr int m = decltype(mc)::value, i = m / 8, j = m % 8;
      MFMA(acc[i][j], fb0[j], fa0[i]);
#ifdef P4C_PF
      if constexpr (DMA && m == 8 * NI - 12) asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(pf_sink) : "v"(pf_a), "s"(ra), "s"(koff + (P4C_PF - 2) * 128) : "memory");
      if constexpr (DMA && m == 8 * NI - 10) asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(pf_sink) : "v"(pf_b), "s"(rb), "s"(koff + (P4C_PF - 2) * 128) : "memory");
#endif
      if constexpr (m % P4C_R0STEP == 0 && m / P4C_R0STEP < 8 + NI) {
        constexpr int r = m / P4C_R0STEP;
        if constexpr (r == 0) DSREAD(fa1[0], fo_a[P][1], 0);
        else if constexpr (r <= 8) DSREAD(fb1[r - 1], fo_b[P][1], (r - 1) * 2048);
        else DSREAD(fa1[r - 8], fo_a[P][1], (r - 8) * 2048);
      }
      if constexpr (m == 8 * NI - 4) {
        if constexpr (P4C_ABL & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef P4C_PF
        else if constexpr (DMA) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");  // (the two prefetch loads just issued stay in flight)
#endif
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // my reads of this buffer are done, my share of tile t + 1 has landed
      }
      if constexpr (m == 8 * NI - 3 && !(P4C_ABL & 4)) asm volatile("s_barrier" ::: "memory");
    });
    // ---- slice 1: 8 NI MFMAs on set 1; set 0 <- slice 0 of tile t + 1 (other buffer); tile t + 2 -> this buffer.
    // groups of four MFMAs: [M0 write | LDS-DMA | fragment read | -]
    static_for<8 * NI>([&acc, &fa0, &fb0, &fa1, &fb1, &vo, &ra, &rb, &koff, &fo_a, &fo_b, wb](auto mc) {
      constexpr int m = decltype(mc)::value, i = m / 8, j = m % 8;
      constexpr int DS = (8 * NI / NQ < P4C_DSTEP) ? 8 * NI / NQ : P4C_DSTEP;
      constexpr int g = m / DS, ph = m % DS;
      if constexpr (DMA && ph == 1 && g < NQ) {
        constexpr int q = g;
        constexpr int imm = q < NA ? q * 4096 : P4_OP + (q - NA) * 4096;
        asm volatile("s_add_u32 m0, %3, %4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tbuffer_load_dwordx4 %5, %6, %7 offen lds"
                     : "+a"(acc[i][j])
                     : "v"(fb1[j]), "v"(fa1[i]), "s"(wb), "i"(imm), "v"(vo[q]), "s"(q < NA ? ra : rb), "s"(koff)
                     : "memory", "scc");
      } else {
        MFMA(acc[i][j], fb1[j], fa1[i]);
      }
      if constexpr (READ_NEXT && m % DS == 0 && m / DS < 8 + NI) {
        constexpr int r = m / DS;
        if constexpr (r == 0) DSREAD(fa0[0], fo_a[P ^ 1][0], 0);
        else if constexpr (r <= 8) DSREAD(fb0[r - 1], fo_b[P ^ 1][0], (r - 1) * 2048);
        else DSREAD(fa0[r - 8], fo_a[P ^ 1][0], (r - 8) * 2048);
      }
    });
    koff += 128;
  };
  using T = std::true_type;
  using F = std::false_type;
#ifdef P4C_STAMP
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int t = 0; t + 2 < nkt; t += 2) {
    ktile(IC<0>{}, T{}, T{});
    ktile(IC<1>{}, T{}, T{});
  }
  ktile(IC<0>{}, F{}, T{});
  ktile(IC<1>{}, F{}, F{});
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results are read by compiler-generated code
#ifdef P4C_STAMP
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
#endif
  // epilogue (timing + check only): lane (lm, lq) of block (i, j) holds row 16 i + lm, columns 16 j + 4 lq ..
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bf16x4 o = {(bf16)acc[i][j][0], (bf16)acc[i][j][1], (bf16)acc[i][j][2], (bf16)acc[i][j][3]};
      const int row = bm0 + wr * 16 * NI + 16 * i + lm;
      if (row < M) *reinterpret_cast<bf16x4*>(C + (long)row * N + bn0 + wc * 128 + 16 * j + 4 * lq) = o;
    }
#ifdef P4C_STAMP
  {
    const unsigned long long c2 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long c3 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
      stamps[8 * blockIdx.x] = c1 - c0;
      stamps[8 * blockIdx.x + 1] = r1 - r0;
      stamps[8 * blockIdx.x + 2] = c0 - cE;
      stamps[8 * blockIdx.x + 3] = c2 - c1;
      stamps[8 * blockIdx.x + 4] = c3 - c2;
    }
  }
#endif
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

static int g_sets = 1, g_iters = 200, g_rot = 3;  // g_rot: bit 0 rotate A, bit 1 rotate B
template <int NI>
static void run(int M, int N, int K) {
  bf16 *A, *B, *C;
  hipMalloc(&A, (size_t)M * K * 2 * g_sets);
  hipMalloc(&B, (size_t)N * K * 2 * g_sets);
  hipMalloc(&C, (size_t)M * N * 2);
  std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K);
  srand(1);
  for (auto& v : ha) v = tobf((float)rand() / RAND_MAX * 2.f - 1.f);  // uniform [-1, 1): the chip clocks lower on random data
  for (auto& v : hb) v = tobf((float)rand() / RAND_MAX * 2.f - 1.f);
  for (int s = 0; s < g_sets; ++s) {
    hipMemcpy(A + (size_t)s * M * K, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(B + (size_t)s * N * K, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
  }
  hipMemset(C, 0xff, (size_t)M * N * 2);
  hipFuncSetAttribute((const void*)gemm_p4c_probe<NI>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P4_BUF);
  const int grid = ((M + 32 * NI - 1) / (32 * NI)) * (N / 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  unsigned long long* stamps;
  hipMalloc(&stamps, (size_t)grid * 64);
  for (int it = 0; it < 5; ++it) gemm_p4c_probe<NI><<<grid, 256, 2 * P4_BUF>>>(A, B, C, M, N, K, stamps);
  hipDeviceSynchronize();
  const int iters = g_iters;
  hipEventRecord(e0);
  for (int it = 0; it < iters; ++it)
    gemm_p4c_probe<NI><<<grid, 256, 2 * P4_BUF>>>(A + (size_t)((g_rot & 1) ? it % g_sets : 0) * M * K, B + (size_t)((g_rot & 2) ? it % g_sets : 0) * N * K, C, M, N, K, stamps);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  std::vector<unsigned short> hc((size_t)M * N);
  hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int s = 0; s < 600; ++s) {
    const int m = s < 8 ? M - 1 - s : rand() % M, n = s < 8 ? N - 1 - 37 * s : rand() % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)tof(ha[(size_t)m * K + k]) * tof(hb[(size_t)n * K + k]);
    const float got = tof(hc[(size_t)m * N + n]);
    if (!(fabs(got - ref) <= 0.01 * fabs(ref) + 0.02 * sqrt((double)K))) ++bad;
  }
  printf("p4c<%d> M=%6d N=%5d K=%5d grid %4d: %8.1f us  %7.1f TF/s   spot check %d / 600 bad", NI, M, N, K, grid, ms * 1e3,
         2.0 * M * N * K / (ms * 1e-3) / 1e12, bad);
#ifdef P4C_STAMP
  {
    std::vector<unsigned long long> hs((size_t)grid * 8);
    hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, ghz, pro, epi, drain;
    for (int b = 0; b < grid; ++b) {
      cyc.push_back((double)hs[8 * b] / (K / 64));
      ghz.push_back((double)hs[8 * b] / ((double)hs[8 * b + 1] * 10.0));
      pro.push_back((double)hs[8 * b + 2]);
      epi.push_back((double)hs[8 * b + 3]);
      drain.push_back((double)hs[8 * b + 4]);
    }
    for (auto* v : {&cyc, &ghz, &pro, &epi, &drain}) std::sort(v->begin(), v->end());
    printf("   | loop %.0f cyc per K tile (MFMA-bound %d) at %.2f GHz; prologue %.0f, epilogue issue %.0f, store drain %.0f cyc (medians)", cyc[grid / 2], 16 * 16 * NI,
           ghz[grid / 2], pro[grid / 2], epi[grid / 2], drain[grid / 2]);
  }
#endif
  printf("\n");
  hipFree(stamps);
  hipFree(A);
  hipFree(B);
  hipFree(C);
}

int main(int argc, char** argv) {
  if (argc > 2) g_sets = atoi(argv[2]);
  if (argc > 3) g_iters = atoi(argv[3]);
  if (argc > 4) g_rot = atoi(argv[4]);
  if (argc > 1) {  // short list for A/B runs of variants
    run<8>(4096, 4096, 4096);
    run<8>(15968, 3072, 768);
    run<6>(15968, 768, 3072);
    run<6>(15968, 2304, 768);
    return 0;
  }
  run<8>(4096, 4096, 4096);
  run<8>(8192, 8192, 8192);
  run<8>(15968, 3072, 768);
  run<8>(15968, 768, 3072);
  run<8>(15968, 2304, 768);
  run<6>(15968, 768, 768);
  run<6>(15968, 768, 3072);
  run<6>(15968, 2304, 768);
  run<8>(256, 256, 256);
  run<8>(512, 512, 384);
  return 0;
}
