// (256|192|128) x 256-tile bf16 MFMA GEMM with a HAND-SCHEDULED main loop: four waves, one per SIMD, each owning a (128|96|64) x 128
// register tile -- 0.25 ds_read_b128 per MFMA where the 8-wave kernel of gemm_p8.hip reads 0.375, accumulators in the AccVGPRs.
// Same contract as gemm_p8.hip (GemmParams, the LDS-free fused epilogues of gemm_common.h); the dispatcher (gemm.hip: plan_gemm)
// sends the K-contiguous-A products with K a multiple of 128 and N a multiple of 256 here.
//
// Why hand-scheduled: with 256 accumulator registers + two fragment sets live, hipcc's own schedule of the same loop from
// builtins shuffles the accumulators through ~10 v_accvgpr moves per MFMA (tools/probes/p4b_probe.hip) -- "this shape lost as
// compiler-scheduled HIP" in round 2, and that prototype's __syncthreads() also drained the LDS-DMA pipeline.  Here every
// instruction of the K loop is an `asm volatile` statement: the emitted order IS the written order, the compiler only
// allocates registers.  What the probes said (tools/probes/p4c_probe.hip, in-kernel s_memtime stamps, profiles/r04_gemm_p4_*):
//   * fragment reads between MFMAs cost no cycles (2 118 cycles per 64-deep K tile with reads + barrier vs 2 048 MFMA-bound);
//   * an LDS-DMA costs ~27 cycles when its statement also carries the M0 write, an s_nop and two address adds, and ~10 when
//     every gap between two MFMAs holds at most ONE other instruction -- so: one VGPR offset per DMA (constant per output
//     tile), the K offset in an SGPR (soffset), the M0 write one gap ahead of its DMA (one statement: M0 write, MFMA, DMA --
//     the compiler does not preserve M0 between statements);
//   * the chip is power-limited in this loop (1.7-1.9 GHz on random data): cycles saved come back partly as a lower clock.
//
// Structure of one 64-deep K tile t (buffer P = t & 1; LDS = two 64 KB buffers, A [rows][128 B] | B [256][128 B], 16-byte
// chunk c of row r at slot c ^ ((r >> 1) & 7), filled by LDS-DMA in whole 128-byte lines, swizzle on the source side):
//   slice 0: 8 NI MFMAs on fragment set 0 | set 1 <- slice 1 of buffer P (one read per two MFMAs)
//            s_waitcnt vmcnt lgkmcnt(0); s_barrier        -- buffer P is free, tile t + 1 has landed (issued a tile ago)
//   slice 1: 8 NI MFMAs on set 1 | set 0 <- slice 0 of buffer P ^ 1 | tile t + 2 -> buffer P: groups of MFMAs [M0 | DMA | read | -]
// ONE barrier and ONE counted wait per K tile.  The first K tile's slice 0 multiplies into a zero constant (no accumulator
// clear); the last two K tiles stage the NEXT output tile's first two K tiles (after their own barriers both buffers are
// free), so the pipeline fill of a tile hides under the tail of the previous one and the epilogue's stores drain under the
// next main loop (counted wait: everything older than the stores).
// K-major B (the dX products: B = a weight read along its rows): image [64 k-rows][512 B], 16-byte chunk c of k-row r at
// slot c ^ (((r & 3) | ((r >> 3) & 1) << 2) << 1), fragments by ds_read_b64_tr_b16 (two per 16 x 32 operand).
#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "common.h"
#include "gemm_common.h"
#include "kernels.h"

namespace {

constexpr int P4_THREADS = 256;
constexpr int P4_OP = 32768;        // one operand of one K tile
constexpr int P4_BUF = 2 * P4_OP;   // A | B
constexpr int P4_PIPE = 2 * P4_BUF;
constexpr int P4_BIAS = 4 * 1024;   // per output-tile parity: 1 KiB per wave (128 fp32 from lanes 0-31; the other lanes' LDS-DMA writes zeros)
constexpr int P4_LDS = P4_PIPE + 2 * P4_BIAS;

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

// (operands of an asm statement inside a generic lambda must be captured explicitly: implicit capture does not see them)
#define P4_MFMA(ACC, FB, FA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(FB), "v"(FA))
#define P4_MFMA_Z(ACC, FB, FA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(ACC) : "v"(FB), "v"(FA))
#define P4_READ(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "i"(OFF) : "memory")
// K-major operand: the 16 x 32 fragment is two transposing reads, k-rows +0..3 and +4..7 (2 KiB further down), into the two
// halves of ONE four-register operand -- which an asm operand cannot name; see the K-major instantiation below
#define P4_READ_TR(DST, ADDR, OFF) static_assert(!B_KM, "K-major B: fragment reads not built yet")

struct P4Tile {
  int bm0, bn0, z, z1, z2;
};
typedef u32x4 Frag;  // 8 bf16 = one MFMA operand

// NI: 16-row groups per wave (tile = 32 NI x 256).  EPI: as gemm_p8_kernel.
template <int NI, bool B_KM, int EPI>
__global__ __launch_bounds__(P4_THREADS) void gemm_p4_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 32 * NI;
  constexpr int NA = NI, NQ = NA + 8;  // LDS-DMA instructions per wave and K tile: A (32 rows each), then B
  // epilogues that only store a fixed number of 16-byte rows per 16-row group may leave their stores in flight
  constexpr bool EPI_EARLY = EPI == P8_EPI_PLAIN_BF16 || EPI == P8_EPI_GELU_ONLY || EPI == SSAK_EPI_GELU_SAVE_GRAD;
  constexpr int EPI_STORES = (EPI == SSAK_EPI_GELU_SAVE_GRAD ? 6 : 4) * NI;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int per_z = p.tiles_m * p.tiles_n;
  const int ntiles = per_z * p.nz;
  const int nkt = p.K / BK;  // even, >= 4 (launcher)
  auto decode = [&](int t) __attribute__((always_inline)) {
    P4Tile c;
    const int id = xcd_remap(t, ntiles);
    const int zs = id / per_z, rem = id % per_z;
    c.z = zs;
    c.z1 = zs / p.nb2;
    c.z2 = zs % p.nb2;
    c.bm0 = rem / p.tiles_n * BM;
    c.bn0 = rem % p.tiles_n * 256;
    return c;
  };
  typedef __attribute__((address_space(3))) char lds_char;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)smem;
  const uint32_t wbase = lds0 + wave * 1024;  // LDS-DMA destination of this wave: + buffer + instruction * 4 KiB
  // ---- LDS-DMA source offsets.  K-contiguous operand: instruction j of this wave fills rows 32 j + 8 wave + (lane >> 3) (128 B
  // each); K-major B: k-rows 8 j + 2 wave + (lane >> 5) (512 B each).  One VGPR offset per instruction, constant for the output
  // tile (the descriptor's range check zero-fills rows beyond the matrix from it); the K tile's offset is scalar.
  const int drow = 8 * wave + (lane >> 3);
  const int dchunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  const int bkrow = 2 * wave + (lane >> 5);  // (+ 8 j: bits 0, 1, 3 of the k-row -- the swizzle's inputs -- do not depend on j... bit 3 does)
  uint32_t vo[NQ];
  u32x4 ra_v, rb_v;  // descriptors as computed; uni4() right before a K loop hands them to the asm statements in SGPRs
  uint32_t bias_vo = 0x80000000u;
  auto uni4 = [](u32x4 v) __attribute__((always_inline)) {
    return (u32x4){(uint32_t)__builtin_amdgcn_readfirstlane((int)v[0]), (uint32_t)__builtin_amdgcn_readfirstlane((int)v[1]),
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)v[2]), (uint32_t)__builtin_amdgcn_readfirstlane((int)v[3])};
  };
  auto make_rsrc = [](const void* ptr, uint32_t bytes) __attribute__((always_inline)) {
    const uint64_t a = (uint64_t)(uintptr_t)ptr;
    return (u32x4){(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) & 0xffffu,
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
  };
  const u32x4 rbias = uni4(make_rsrc(p.bias, p.bias ? (uint32_t)((((long)p.nb2 - 1) * p.bias_s2 + p.N) * 4) : 0u));
  auto setup = [&](const P4Tile& c, bool real) __attribute__((always_inline)) {
    ra_v = make_rsrc(p.A + c.z1 * p.sa1 + c.z2 * p.sa2, p.ext_a);
    rb_v = make_rsrc(p.B + c.z1 * p.sb1 + c.z2 * p.sb2, p.ext_b);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      uint32_t o;
      if (q < NA) {
        o = (uint32_t)(((long)(c.bm0 + 32 * q + drow) * p.lda + dchunk * 8) * 2);
      } else if (!B_KM) {
        o = (uint32_t)(((long)(c.bn0 + 32 * (q - NA) + drow) * p.ldb + dchunk * 8) * 2);
      } else {
        const int kr = 8 * (q - NA) + bkrow;
        const int ch = (lane & 31) ^ (((kr & 3) | (((kr >> 3) & 1) << 2)) << 1);
        o = (uint32_t)(((long)kr * p.ldb + c.bn0 + ch * 8) * 2);
      }
      vo[q] = real ? o : 0x80000000u;  // no next tile: out of range = zeros into slots nobody reads, no memory traffic
    }
    // bias of the wave's 128 columns: lanes 0-31, 16 B each
    bias_vo = (real && lane < 32) ? (uint32_t)((c.z2 * p.bias_s2 + c.bn0 + wc * 128 + 4 * lane) * 4) : 0x80000000u;
  };
  const uint32_t kstep_b = B_KM ? (uint32_t)(64 * p.ldb * 2) : 128u;  // byte advance of B per K tile
  // ---- fragment read addresses [buffer][slice]
  const int lm = lane & 15, lq = lane >> 4;
  uint32_t fo_a[2][2];
  uint32_t fo_b[2][B_KM ? 8 : 2];  // K-contiguous: [buffer][slice]; K-major: [buffer][column group j] (slice = + 16 KiB)
#pragma unroll
  for (int b = 0; b < 2; ++b) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fo_a[b][kk] = lds0 + b * P4_BUF + (wr * 16 * NI + lm) * 128 + (((4 * kk + lq) ^ ((lm >> 1) & 7)) << 4);
    if (!B_KM) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        fo_b[b][kk] = lds0 + b * P4_BUF + P4_OP + (wc * 128 + lm) * 128 + (((4 * kk + lq) ^ ((lm >> 1) & 7)) << 4);
    } else {
      // lane = 16 g + 4 q4 + pp supplies k-row 8 g + q4 (+ 4: second read, + 32: slice 1), columns 16 j + 4 pp .. of the wave's 128
      const int g = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
      const int kr = 8 * g + q4;
      const int sw = ((kr & 3) | (((kr >> 3) & 1) << 2)) << 1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ch = wc * 16 + 2 * j + (pp >> 1);
        fo_b[b][j] = lds0 + b * P4_BUF + P4_OP + kr * 512 + ((ch ^ sw) << 4) + (pp & 1) * 8;
      }
    }
  }
  const uint32_t bias_lds0 = lds0 + P4_PIPE + wave * 1024;

  // one LDS-DMA with its M0 write: the form outside the MFMA stream (prologue)
  auto dma_plain = [](uint32_t dst, uint32_t voff, u32x4 rsrc, uint32_t soff) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  };

  f32x4 acc[2][NI][4];  // [column half][16-row group][16-column group]: a half is what gemm_epilogue_direct takes
  Frag fa0[NI], fb0[8], fa1[NI], fb1[8];
  uint32_t koff_a = 0, koff_b = 0;  // scalar byte offsets of the K tile being staged

  // ---- one K tile in buffer P.
  //   ZERO: first K tile of an output tile (slice 0 multiplies into 0).  STAGE: issue the NQ LDS-DMA of the tile two ahead (or of
  //   the next output tile: the caller has re-pointed vo / koff) into this buffer, BIAS: and the bias slice first.
  //   READ_NEXT: read set 0 of the following K tile from the other buffer.  mid_keep: vector-memory operations younger than
  //   the tile that must have landed (the previous epilogue's stores).
  auto ktile = [&acc, &fa0, &fb0, &fa1, &fb1, &vo, &bias_vo, &koff_a, &koff_b, &fo_a, &fo_b, wbase, kstep_b](
                   auto par_c, auto zero_c, auto stage_c, auto rn_c, auto bias_c, bool keep_stores, uint32_t bias_dst, const u32x4 ra,
                   const u32x4 rb, const u32x4 rbias) __attribute__((always_inline)) {
    constexpr int P = decltype(par_c)::value;
    constexpr bool ZERO = decltype(zero_c)::value, STAGE = decltype(stage_c)::value, READ_NEXT = decltype(rn_c)::value,
                   BIAS = decltype(bias_c)::value;
    const uint32_t wb = wbase + P * P4_BUF;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // set 0 is in registers
    // ---- slice 0
    static_for<8 * NI>([&acc, &fa0, &fb0, &fa1, &fb1, &fo_a, &fo_b, keep_stores](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value, i = m / 8, j = m % 8;
      if constexpr (ZERO) P4_MFMA_Z(acc[j / 4][i][j % 4], fb0[j], fa0[i]);
      else P4_MFMA(acc[j / 4][i][j % 4], fb0[j], fa0[i]);
      if constexpr (m % 2 == 0 && m / 2 < 8 + NI) {
        constexpr int r = m / 2;
        if constexpr (r == 0) P4_READ(fa1[0], fo_a[P][1], 0);
        else if constexpr (r <= 8) {
          if constexpr (B_KM) P4_READ_TR(fb1[r - 1], fo_b[P][r - 1], 16384);
          else P4_READ(fb1[r - 1], fo_b[P][1], (r - 1) * 2048);
        } else P4_READ(fa1[r - 8], fo_a[P][1], (r - 8) * 2048);
      }
      if constexpr (m == 8 * NI - 4) {
        // my reads of this buffer are done; my share of the next K tile has landed (everything but the previous epilogue's stores)
        if (keep_stores) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(EPI_STORES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      if constexpr (m == 8 * NI - 3) asm volatile("s_barrier" ::: "memory");
    });
    // ---- slice 1: groups of DS MFMAs: [fragment read | M0 write | LDS-DMA | ...]
    static_for<8 * NI>([&acc, &fa0, &fb0, &fa1, &fb1, &vo, ra, rb, rbias, &bias_vo, &koff_a, &koff_b, &fo_a, &fo_b, wb, bias_dst](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value, i = m / 8, j = m % 8;
      constexpr int DS = (8 * NI / (NQ + (BIAS ? 1 : 0)) < 4) ? 8 * NI / (NQ + (BIAS ? 1 : 0)) : 4;
      constexpr int g = m / DS, ph = m % DS;
      constexpr int q = BIAS ? g - 1 : g;  // LDS-DMA of this group (-1: the bias slice)
      if constexpr (STAGE && ph == 1 && q >= (BIAS ? -1 : 0) && q < NQ) {
        if constexpr (q < 0) {
          asm volatile("s_mov_b32 m0, %3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tbuffer_load_dwordx4 %4, %5, 0 offen lds"
                       : "+a"(acc[j / 4][i][j % 4])
                       : "v"(fb1[j]), "v"(fa1[i]), "s"(bias_dst), "v"(bias_vo), "s"(rbias)
                       : "memory");
        } else {
          constexpr int imm = q < NA ? q * 4096 : P4_OP + (q - NA) * 4096;
          asm volatile("s_add_u32 m0, %3, %4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tbuffer_load_dwordx4 %5, %6, %7 offen lds"
                       : "+a"(acc[j / 4][i][j % 4])
                       : "v"(fb1[j]), "v"(fa1[i]), "s"(wb), "i"(imm), "v"(vo[q]), "s"(q < NA ? ra : rb), "s"(q < NA ? koff_a : koff_b)
                       : "memory", "scc");
        }
      } else {
        P4_MFMA(acc[j / 4][i][j % 4], fb1[j], fa1[i]);
      }
      if constexpr (READ_NEXT) {
        // fragment reads of the next K tile: one per DS MFMAs in the DMA's free gaps, the rest after the last DMA
        constexpr int r_dense = m / DS;                       // reads placed so far if one per group
        constexpr int tail0 = DS * (NQ + (BIAS ? 1 : 0));     // first MFMA after the DMA groups
        constexpr bool in_groups = m < tail0 && ph == 0 && r_dense < 8 + NI;
        constexpr int r_tail = (NQ + (BIAS ? 1 : 0)) + (m - tail0);
        constexpr bool in_tail = m >= tail0 && r_tail < 8 + NI;
        if constexpr (in_groups || in_tail) {
          constexpr int r = in_groups ? r_dense : r_tail;
          if constexpr (r == 0) P4_READ(fa0[0], fo_a[P ^ 1][0], 0);
          else if constexpr (r <= 8) {
            if constexpr (B_KM) P4_READ_TR(fb0[r - 1], fo_b[P ^ 1][r - 1], 0);
            else P4_READ(fb0[r - 1], fo_b[P ^ 1][0], (r - 1) * 2048);
          } else P4_READ(fa0[r - 8], fo_a[P ^ 1][0], (r - 8) * 2048);
        }
      }
    });
    koff_a += 128;
    koff_b += kstep_b;
  };
  using T = std::true_type;
  using F = std::false_type;

  bool primed = false;
  int par = 0;  // output-tile parity: which bias slot
  P4Tile cur = decode(min((int)blockIdx.x, ntiles - 1));
  for (int t = blockIdx.x; t < ntiles;) {
    const P4Tile c = cur;
    const uint32_t bias_lds = bias_lds0 + par * P4_BIAS;
    const bool was_primed = primed;
    if (!primed) {
      setup(c, true);
      const u32x4 ra = uni4(ra_v), rb = uni4(rb_v);
      dma_plain(bias_lds, bias_vo, rbias, 0);
      static_for<2 * NQ>([&vo, ra, rb, &dma_plain, wbase, kstep_b](auto qq) __attribute__((always_inline)) {
        constexpr int b = decltype(qq)::value / NQ, q = decltype(qq)::value % NQ;
        dma_plain(wbase + b * P4_BUF + (q < NA ? q * 4096 : P4_OP + (q - NA) * 4096), vo[q], q < NA ? ra : rb, q < NA ? b * 128u : b * kstep_b);
      });
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NQ) : "memory");  // bias + K tile 0 (this wave's share)
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NQ + EPI_STORES < 63 ? NQ + EPI_STORES : 63) : "memory");  // everything older than K tile 1 and the stores
    }
    asm volatile("s_barrier" ::: "memory");
    static_for<8>([&fb0, &fo_b](auto j) __attribute__((always_inline)) {
      if constexpr (B_KM) P4_READ_TR(fb0[j], fo_b[0][j], 0);
      else P4_READ(fb0[j], fo_b[0][0], j * 2048);
    });
    static_for<NI>([&fa0, &fo_a](auto i) __attribute__((always_inline)) { P4_READ(fa0[i], fo_a[0][0], i * 2048); });
    koff_a = 256;
    koff_b = 2 * kstep_b;
    {
      const u32x4 ra = uni4(ra_v), rb = uni4(rb_v);
      ktile(IC<0>{}, T{}, T{}, T{}, F{}, was_primed, 0u, ra, rb, rbias);
      ktile(IC<1>{}, F{}, T{}, T{}, F{}, false, 0u, ra, rb, rbias);
      for (int kt = 4; kt < nkt; kt += 2) {
        ktile(IC<0>{}, F{}, T{}, T{}, F{}, false, 0u, ra, rb, rbias);
        ktile(IC<1>{}, F{}, T{}, T{}, F{}, false, 0u, ra, rb, rbias);
      }
    }
    // bias of this tile into registers (the next tile's slice lands in the other slot)
    BiasRegs<4> bias_regs[2];
    {
      const char* bl = smem + P4_PIPE + par * P4_BIAS + wave * 1024;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + (64 * h + 16 * j + 4 * lq) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) bias_regs[h].v[j][r] = b4[r];
        }
    }
    // the next output tile: its first two K tiles are staged by the last two K tiles of this one
    const int t_next = t + (int)gridDim.x;
    const bool full_rows = c.bm0 + BM <= p.M;  // (a tile with rows beyond M skips some epilogue stores: their count is not fixed)
    const bool stage_next = EPI_EARLY && t_next < ntiles && full_rows;
    if (t_next < ntiles) cur = decode(t_next);
    setup(cur, stage_next);
    koff_a = 0;
    koff_b = 0;
    const uint32_t bias_lds_next = bias_lds0 + (par ^ 1) * P4_BIAS;
    {
      const u32x4 ra = uni4(ra_v), rb = uni4(rb_v);
      ktile(IC<0>{}, F{}, T{}, T{}, T{}, false, bias_lds_next, ra, rb, rbias);
      ktile(IC<1>{}, F{}, T{}, F{}, F{}, false, 0u, ra, rb, rbias);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results are read by compiler-generated code
    primed = stage_next;

    const long coff = c.z1 * p.sc1 + c.z2 * p.sc2;
    (void)coff;
#pragma unroll
    for (int h = 0; h < 2; ++h)
      gemm_epilogue_direct<NI, EPI>(p, acc[h], bias_regs[h], c.bm0, c.bn0, wr * 16 * NI, wc * 128 + 64 * h, lane, c.z, c.z1, c.z2, 0);
    if (!primed) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // dummies (and whatever the epilogue left) before LDS is re-staged
    t = t_next;
    par ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int p4_num_cu(int* out) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    SSAK_HIP(hipGetDevice(&dev));
    SSAK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  }
  *out = n_cu;
  return SSAK_OK;
}

template <int NI, bool B_KM, int EPI>
int launch_p4(const GemmParams& p, hipStream_t st) {
  auto kern = gemm_p4_kernel<NI, B_KM, EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    SSAK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, P4_LDS));
    attr_done = true;
  }
  const long ntiles = (long)p.tiles_m * p.tiles_n * p.nz;
  int n_cu = 0;
  if (int rc = p4_num_cu(&n_cu)) return rc;
  // one timing slot per (instantiation, N, K), as launch_p8
  static std::mutex slot_mu;
  static std::map<std::pair<int, int>, int> slots;
  int slot;
  {
    std::lock_guard<std::mutex> lock(slot_mu);
    auto it = slots.find({p.N, p.K});
    if (it == slots.end()) {
      char nm[112];
      snprintf(nm, sizeof(nm), "gemm_p4_kernel<%d, %s, %d> (N = %d, K = %d)", NI, B_KM ? "true" : "false", EPI, p.N, p.K);
      it = slots.emplace(std::make_pair(p.N, p.K), ssak_prof_register(nm, SSAK_BOUND_MFMA)).first;
    }
    slot = it->second;
  }
  ProfScope prof_scope(slot, 2.0 * p.M * p.N * (double)p.K * p.nz, st);
  kern<<<dim3((unsigned)std::min<long>(ntiles, n_cu)), P4_THREADS, P4_LDS, st>>>(p);  // one persistent workgroup per CU
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <int NI>
int dispatch_p4(const GemmParams& p, int b_km, hipStream_t st) {
  if (p.epilogue == SSAK_EPI_GELU_SAVE_GRAD) {
    if (!b_km) return launch_p4<NI, false, SSAK_EPI_GELU_SAVE_GRAD>(p, st);
    ssak_set_error("gemm_p4: GELU_SAVE_GRAD is built for K-contiguous operands");
    return SSAK_ERR_INVALID;
  }
  if (b_km) {
    ssak_set_error("gemm_p4: the K-major B form is not built");
    return SSAK_ERR_INVALID;
  }
  if (p.epilogue == SSAK_EPI_MUL_AUX) return launch_p4<NI, false, SSAK_EPI_MUL_AUX>(p, st);
  const bool no_extras = !p.drop_thresh && !p.colsum;
  if (no_extras && p.epilogue == SSAK_EPI_NONE && !p.out_f32 && !p.accumulate) return launch_p4<NI, false, P8_EPI_PLAIN_BF16>(p, st);
  if (no_extras && p.epilogue == SSAK_EPI_GELU && !p.aux_out && !p.out_f32) return launch_p4<NI, false, P8_EPI_GELU_ONLY>(p, st);
  return launch_p4<NI, false, -1>(p, st);
}

}  // namespace

// true when the four-wave kernel can run this product as planned (tile height bm, no split-K): everything it does not cover
// stays on gemm_p8.hip
bool ssak_gemm_p4_supports(const void* params, int bm, int a_km, int b_km) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  if (b_km) return false;  // (K-major B: not built yet)
  if (a_km || p.split_k != 1 || p.dynamic || p.kperm_n2) return false;
  if (bm != 256 && bm != 192 && bm != 128) return false;
  if (p.K % 128 != 0 || p.K < 256 || p.N % 256 != 0) return false;
  if ((p.ldc & 7) || ((p.sc1 | p.sc2) & 7)) return false;
  const bool fq = p.epilogue == SSAK_EPI_GELU_SAVE_GRAD || p.epilogue == SSAK_EPI_MUL_AUX;
  if (fq && ((p.ldc | p.sc1 | p.sc2) & 15)) return false;
  if ((((uintptr_t)p.aux_in | (uintptr_t)p.aux_out | (uintptr_t)p.C) & 15) != 0) return false;
  if (p.out_f32 && p.epilogue != SSAK_EPI_NONE) return false;
  return true;
}

int ssak_gemm_p4_launch(const void* params, int bm, int b_km, hipStream_t st) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  if (bm == 256) return dispatch_p4<8>(p, b_km, st);
  if (bm == 192) return dispatch_p4<6>(p, b_km, st);
  return dispatch_p4<4>(p, b_km, st);
}
