// 128 x 256-tile bf16 MFMA GEMM for TWO CO-RESIDENT workgroups per CU: four waves each, 64 x 128 wave tiles (128 accumulators in
// the AccVGPRs), 32-deep K tiles in three LDS stages (72 KB + bias slots = 76 KB per workgroup), the K tile a hand-scheduled
// `asm volatile` sequence like gemm_p4.hip's.  Same contract as gemm_p8.hip / gemm_p4.hip (GemmParams, the LDS-free fused
// epilogues of gemm_common.h); the dispatcher (gemm.hip: plan_gemm) sends the epilogue-heavy K-contiguous products here.
//
// Why: the products whose VALU / store epilogue is as long as their 12-K-tile main loop (feed-forward up:
// GELU + GELU' + dropout + two store streams; feed-forward dX: the 8-bit factor; the K = 768 projections) ran at 0.25-0.34 of
// the matrix roof on both persistent kernels, because every wave of a CU reached the epilogue at the same moment -- the
// eight-wave kernel's two waves per SIMD share its barriers, the four-wave kernel has one wave per SIMD -- and every CU of
// the chip with them: a 98-147 MB store burst at ~5 TB/s with the matrix pipes idle.  Two INDEPENDENT workgroups per CU drift
// apart: one's epilogue (VALU, stores) runs under the other's main loop, and each one's barrier / LDS / DMA latencies are
// covered by the other's MFMAs without any hand-made phase alternation.
//
// One 32-deep K tile t of a wave (fragment set c = t & 1 already in registers, read during tile t - 1):
//     s_waitcnt vmcnt(6) lgkmcnt(0) ; s_barrier     -- set c is in registers; everybody's share of tile t + 1 has landed and
//                                                      everybody has finished reading tile t's stage
//     32 MFMAs on set c, one other instruction per gap: 12 ds_read_b128 of tile t + 1 into set c ^ 1 (stage (t + 1) % 3),
//     6 LDS-DMA of tile t + 3 into stage t % 3 (M0 write in the same statement, one gap ahead), address bookkeeping
// LDS stage: A [128 rows][64 B] | B [256 rows][64 B]; 16-byte chunk c of row r at slot c ^ ((4 - (r >> 2)) & 3): conflict-free for
// ds_read_b128 under its real lane groups (enumerated: gemm_c4 header of DESIGN.md), filled by LDS-DMA 16 rows x 64 B per
// wave-instruction with the swizzle on the source side.  The last three K tiles of an output tile stage the first three of the
// next one (and its bias slice); the epilogue's stores drain under the next main loop behind counted waits.
#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "common.h"
#include "gemm_common.h"
#include "kernels.h"

namespace {

constexpr int C4_THREADS = 256;
constexpr int C4_NI = 4;                     // 16-row groups per wave
constexpr int C4_BM = 128, C4_BN = 256, C4_BK = 32;
constexpr int C4_NA = 2, C4_NB = 4;          // LDS-DMA instructions per wave and K tile (1 KiB = 16 rows x 64 B each)
constexpr int C4_A_SZ = C4_BM * 64, C4_B_SZ = C4_BN * 64, C4_NST = 3;
constexpr int C4_LDS_B = C4_NST * C4_A_SZ, C4_LDS_BIAS = C4_LDS_B + C4_NST * C4_B_SZ;  // 24 KB | 48 KB | bias: 2 slots x 4 waves x 512 B
constexpr int C4_LDS = C4_LDS_BIAS + 2 * 4 * 512;
static_assert(2 * C4_LDS <= 160 * 1024, "two workgroups per CU");

template <int I>
using IC4 = std::integral_constant<int, I>;
template <class F, int... Is>
__device__ __forceinline__ void c4_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(IC4<Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void c4_static_for(F&& f) {
  c4_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

#define C4_MFMA(ACC, FB, FA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(FB), "v"(FA))
#define C4_MFMA_Z(ACC, FB, FA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(ACC) : "v"(FB), "v"(FA))
#define C4_READ(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "i"(OFF) : "memory")

struct C4Tile {
  int bm0, bn0, z, z1, z2;
  int k0;  // K tile this output tile's loop starts at (it wraps around): the K rotation of gemm_p4.hip
};
typedef u32x4 C4Frag;  // 8 bf16 = one MFMA operand

template <int EPI>
__global__ __launch_bounds__(C4_THREADS, 2) void gemm_c4_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NI = C4_NI, NA = C4_NA, NB = C4_NB, A_SZ = C4_A_SZ, B_SZ = C4_B_SZ;
  // epilogues that store at least a fixed number of 16-byte rows per 16-row group may leave their stores in flight (gemm_p4.hip)
  static constexpr int EPI_STORES = (EPI == SSAK_EPI_GELU_SAVE_GRAD ? 6 : 4) * C4_NI;
  static constexpr int TILE_DMA = C4_NA + C4_NB;  // LDS-DMA per wave and K tile
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int per_z = p.tiles_m * p.tiles_n;
  const int ntiles = per_z * p.nz;
  const int nkt = p.K / C4_BK;  // even, >= 6 (launcher)
  auto decode = [&](int t) __attribute__((always_inline)) {
    C4Tile c;
    const int id = xcd_remap(t, ntiles);
    const int zs = id / per_z, rem = id % per_z;
    c.z = zs;
    c.z1 = zs / p.nb2;
    c.z2 = zs % p.nb2;
    c.bm0 = rem / p.tiles_n * C4_BM;
    c.bn0 = rem % p.tiles_n * C4_BN;
    // K rotation per row panel (gemm_p4.hip: the workgroups of a round do not miss on the same cold weight K tile in lockstep);
    // in units of two K tiles so that a 128-byte line's two halves stay neighbours in time
    c.k0 = 2 * ((rem / p.tiles_n * 7 + zs * 3) % (nkt / 2));
    return c;
  };
  typedef __attribute__((address_space(3))) char lds_char;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)smem;
  const uint32_t wbase = lds0 + wave * 1024;  // LDS-DMA destination of this wave: + stage + instruction * 4 KiB
  // ---- LDS-DMA source offsets: instruction q of this wave fills rows 16 (wave + 4 q) + (lane >> 2) (64 B each); lane & 3 is the
  // 16-byte slot, the chunk it holds is slot ^ ((4 - (row >> 2)) & 3)
  const int drow = 16 * wave + (lane >> 2);
  const int dchunk = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
  uint32_t voa[NA], vob[NB];
  u32x4 ra_v, rb_v;
  uint32_t bias_vo = 0x80000000u;
  auto make_rsrc = [](const void* ptr, uint32_t bytes) __attribute__((always_inline)) {
    const uint64_t a = (uint64_t)(uintptr_t)ptr;
    return (u32x4){(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) & 0xffffu,
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
  };
  auto uni4 = [](u32x4 v) __attribute__((always_inline)) {
    return (u32x4){(uint32_t)__builtin_amdgcn_readfirstlane((int)v[0]), (uint32_t)__builtin_amdgcn_readfirstlane((int)v[1]),
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)v[2]), (uint32_t)__builtin_amdgcn_readfirstlane((int)v[3])};
  };
  const u32x4 rbias = uni4(make_rsrc(p.bias, p.bias ? (uint32_t)((((long)p.nb2 - 1) * p.bias_s2 + p.N) * 4) : 0u));
  // (`real` false: no tile to stage -- out of range = zeros into slots nobody reads, no memory traffic)
  auto setup_a = [&](const C4Tile& c, bool real) __attribute__((always_inline)) {
    ra_v = make_rsrc(p.A + c.z1 * p.sa1 + c.z2 * p.sa2, p.ext_a);
#pragma unroll
    for (int q = 0; q < NA; ++q) voa[q] = real ? (uint32_t)(((long)(c.bm0 + 64 * q + drow) * p.lda + dchunk * 8) * 2) : 0x80000000u;
  };
  auto setup_b = [&](const C4Tile& c, bool real) __attribute__((always_inline)) {
    rb_v = make_rsrc(p.B + c.z1 * p.sb1 + c.z2 * p.sb2, p.ext_b);
#pragma unroll
    for (int q = 0; q < NB; ++q) vob[q] = real ? (uint32_t)(((long)(c.bn0 + 64 * q + drow) * p.ldb + dchunk * 8) * 2) : 0x80000000u;
    // bias of the wave's 128 columns: two 4-byte-per-lane instructions (64 columns each)
    bias_vo = real ? (uint32_t)((c.z2 * p.bias_s2 + c.bn0 + wc * 128 + lane) * 4) : 0x80000000u;
  };
  // ---- fragment read addresses inside stage 0: row 16 i + lm of this wave's panel, 16-byte chunk lq
  const int lm = lane & 15, lq = lane >> 4;
  const uint32_t fo_a = lds0 + (wr * 16 * NI + lm) * 64 + ((lq ^ ((4 - (lm >> 2)) & 3)) << 4);
  const uint32_t fo_b = lds0 + C4_LDS_B + (wc * 128 + lm) * 64 + ((lq ^ ((4 - (lm >> 2)) & 3)) << 4);
  const uint32_t bias_lds0 = lds0 + C4_LDS_BIAS + wave * 512;

  // LDS-DMA forms outside the MFMA stream (prologue)
  auto dma_plain = [](uint32_t dst, uint32_t voff, u32x4 rsrc, uint32_t soff) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  };
  // (the second half's memory offset rides in soffset: an instruction offset would move the LDS address as well)
  auto dma_bias = [](uint32_t dst, uint32_t voff, u32x4 rsrc) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_add_u32 m0, %0, 256\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(dst), "v"(voff), "s"(rsrc), "s"(256u) : "memory", "scc");
  };

  f32x4 acc[2][NI][4];  // [column half][16-row group][16-column group]: a half is what gemm_epilogue_direct takes
  C4Frag fa0[NI], fb0[8], fa1[NI], fb1[8];

  // ---- one K tile on fragment set (FA, FB); set (NA_, NB_) <- the next K tile.
  //   s_cur: byte-offset index (0 .. 2) of the LDS stage holding this tile (in: this tile's; out: the next tile's).
  //   ZERO: first K tile of an output tile.  READ_NEXT: read the following K tile's fragments.  BIAS: stage the next output tile's
  //   bias slice too.  The DMA stages A / B of the K tile three ahead into THIS tile's stage (the caller has pointed voa / vob /
  //   the descriptors / koff at it: this or the next output tile).  extra_vm: LDS-DMA / stores younger than what the barrier needs
  //   beyond the following tile's (the previous epilogue's stores right after an early-staged start).
  auto ktile = [&acc, &voa, &vob, &bias_vo, fo_a, fo_b, wbase](auto zero_c, auto rn_c, auto bias_c, C4Frag(&FA)[NI], C4Frag(&FB)[8], C4Frag(&FAn)[NI],
                                                                  C4Frag(&FBn)[8], int& s_cur, uint32_t koff_a, uint32_t koff_b, bool keep_stores,
                                                                  uint32_t bias_dst, const u32x4 ra, const u32x4 rb, const u32x4 rbias) __attribute__((always_inline)) {
    constexpr bool ZERO = decltype(zero_c)::value, READ_NEXT = decltype(rn_c)::value, BIAS = decltype(bias_c)::value;
    // my share of the next K tile has landed (everything but the tile after it and, right after an early-staged start, the
    // previous epilogue's stores); this tile's fragments are in registers
    if (keep_stores) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(TILE_DMA + EPI_STORES < 63 ? TILE_DMA + EPI_STORES : 63) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(TILE_DMA) : "memory");
    asm volatile("s_barrier" ::: "memory");
    const int s_nxt = s_cur == 2 ? 0 : s_cur + 1;
    const uint32_t a_rd = fo_a + (uint32_t)s_nxt * A_SZ, b_rd = fo_b + (uint32_t)s_nxt * B_SZ;
    const uint32_t wb_a = wbase + (uint32_t)s_cur * A_SZ, wb_b = wbase + C4_LDS_B + (uint32_t)s_cur * B_SZ;
    c4_static_for<8 * NI>([&acc, &FA, &FB, &FAn, &FBn, &voa, &vob, ra, rb, rbias, &bias_vo, a_rd, b_rd, wb_a, wb_b, koff_a, koff_b, bias_dst](
                              auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value, i = m / 8, j = m % 8;
      // LDS-DMA in the odd gaps 1, 5, 9, ...: B first (4), then A (2), then the bias slice (2 dword-wide instructions)
      constexpr int g = (m % 4 == 1) ? m / 4 : -1;
      if constexpr (g >= 0 && g < NB) {
        if constexpr (ZERO) {
          asm volatile("s_add_u32 m0, %3, %4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\tbuffer_load_dwordx4 %5, %6, %7 offen lds"
                       : "=a"(acc[j / 4][i][j % 4])
                       : "v"(FB[j]), "v"(FA[i]), "s"(wb_b), "i"(g * 4096), "v"(vob[g]), "s"(rb), "s"(koff_b)
                       : "memory", "scc");
        } else {
          asm volatile("s_add_u32 m0, %3, %4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tbuffer_load_dwordx4 %5, %6, %7 offen lds"
                       : "+a"(acc[j / 4][i][j % 4])
                       : "v"(FB[j]), "v"(FA[i]), "s"(wb_b), "i"(g * 4096), "v"(vob[g]), "s"(rb), "s"(koff_b)
                       : "memory", "scc");
        }
      } else if constexpr (g >= NB && g < NB + NA) {
        if constexpr (ZERO) {
          asm volatile("s_add_u32 m0, %3, %4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\tbuffer_load_dwordx4 %5, %6, %7 offen lds"
                       : "=a"(acc[j / 4][i][j % 4])
                       : "v"(FB[j]), "v"(FA[i]), "s"(wb_a), "i"((g - NB) * 4096), "v"(voa[g - NB]), "s"(ra), "s"(koff_a)
                       : "memory", "scc");
        } else {
          asm volatile("s_add_u32 m0, %3, %4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tbuffer_load_dwordx4 %5, %6, %7 offen lds"
                       : "+a"(acc[j / 4][i][j % 4])
                       : "v"(FB[j]), "v"(FA[i]), "s"(wb_a), "i"((g - NB) * 4096), "v"(voa[g - NB]), "s"(ra), "s"(koff_a)
                       : "memory", "scc");
        }
      } else if constexpr (BIAS && (g == NB + NA || g == NB + NA + 1)) {
        static_assert(!ZERO, "the bias slice is staged by the last K tile");
        asm volatile("s_add_u32 m0, %3, %4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tbuffer_load_dword %5, %6, %7 offen lds"
                     : "+a"(acc[j / 4][i][j % 4])
                     : "v"(FB[j]), "v"(FA[i]), "s"(bias_dst), "i"((g - NB - NA) * 256), "v"(bias_vo), "s"(rbias), "s"((uint32_t)((g - NB - NA) * 256))
                     : "memory", "scc");
      } else {
        if constexpr (ZERO) C4_MFMA_Z(acc[j / 4][i][j % 4], FB[j], FA[i]);
        else C4_MFMA(acc[j / 4][i][j % 4], FB[j], FA[i]);
      }
      if constexpr (READ_NEXT && m % 2 == 0 && m / 2 < 8 + NI) {
        // fragment reads of the next K tile in the even gaps: A group 0, the eight B groups, A groups 1 .. 3 (MFMA order: i outer)
        constexpr int r = m / 2;
        if constexpr (r == 0) C4_READ(FAn[0], a_rd, 0);
        else if constexpr (r <= 8) C4_READ(FBn[r - 1], b_rd, (r - 1) * 1024);
        else C4_READ(FAn[r - 8], a_rd, (r - 8) * 1024);
      }
    });
    s_cur = s_nxt;
  };
  using T = std::true_type;
  using F = std::false_type;

  bool primed = false;
  int par = 0;    // output-tile parity: which bias slot
  int s_cur = 0;  // LDS stage of the current K tile (cycles through the three stages, across output tiles)
  C4Tile cur = decode(min((int)blockIdx.x, ntiles - 1));
  for (int t = blockIdx.x; t < ntiles;) {
    const C4Tile c = cur;
    const uint32_t bias_lds = bias_lds0 + par * 2048;
    const bool was_primed = primed;
    auto koff_of = [nkt](int k, int k0) __attribute__((always_inline)) { return (uint32_t)(k + k0 >= nkt ? k + k0 - nkt : k + k0) * 64u; };
    if (!primed) {
      setup_a(c, true);
      setup_b(c, true);
      const u32x4 ra = uni4(ra_v), rb = uni4(rb_v);
      dma_bias(bias_lds, bias_vo, rbias);
      int s = s_cur;
      for (int k = 0; k < 3; ++k) {  // B then A per K tile: the order and counts the tail of a previous tile leaves in flight
        const uint32_t ko = koff_of(k, c.k0);
        c4_static_for<NB>([&vob, rb, &dma_plain, wbase, s, ko](auto q) __attribute__((always_inline)) { dma_plain(wbase + C4_LDS_B + s * B_SZ + q * 4096, vob[q], rb, ko); });
        c4_static_for<NA>([&voa, ra, &dma_plain, wbase, s, ko](auto q) __attribute__((always_inline)) { dma_plain(wbase + s * A_SZ + q * 4096, voa[q], ra, ko); });
        s = s == 2 ? 0 : s + 1;
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * TILE_DMA) : "memory");  // bias, K tile 0 (this wave's share)
    } else {
      // everything older than K tiles 1 and 2 and the stores: the bias slice and K tile 0
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * TILE_DMA + EPI_STORES < 63 ? 2 * TILE_DMA + EPI_STORES : 63) : "memory");
    }
    asm volatile("s_barrier" ::: "memory");
    {
      const uint32_t a0 = fo_a + (uint32_t)s_cur * A_SZ, b0 = fo_b + (uint32_t)s_cur * B_SZ;
      c4_static_for<8>([&fb0, b0](auto j) __attribute__((always_inline)) { C4_READ(fb0[j], b0, j * 1024); });
      c4_static_for<NI>([&fa0, a0](auto i) __attribute__((always_inline)) { C4_READ(fa0[i], a0, i * 1024); });
    }
    // the next output tile: the tail of this one stages its first three K tiles
    const int t_next = t + (int)gridDim.x;
    const bool full_rows = c.bm0 + C4_BM <= p.M;  // (a tile with rows beyond M skips some epilogue stores: their count is not fixed)
    const bool stage_next = t_next < ntiles && full_rows;
    if (t_next < ntiles) cur = decode(t_next);
    const uint32_t bias_lds_next = bias_lds0 + (par ^ 1) * 2048;
    u32x4 ra = uni4(ra_v), rb = uni4(rb_v);
    // per K tile, before its MFMAs: which K tile it stages (three ahead), of this output tile or of the next one
    int kd = 2, k0d = c.k0;
    uint32_t koff = 0;
    auto pre = [&]() __attribute__((always_inline)) {
      if (++kd == nkt) {
        kd = 0;
        k0d = cur.k0;
        setup_a(cur, stage_next);
        setup_b(cur, stage_next);
        ra = uni4(ra_v);
        rb = uni4(rb_v);
      }
      koff = koff_of(kd, k0d);
    };
    pre();
    ktile(T{}, T{}, F{}, fa0, fb0, fa1, fb1, s_cur, koff, koff, was_primed, 0u, ra, rb, rbias);
    pre();  // (the previous epilogue's stores are still younger than what this tile's barrier needs: two K tiles to drain)
    ktile(F{}, T{}, F{}, fa1, fb1, fa0, fb0, s_cur, koff, koff, was_primed, 0u, ra, rb, rbias);
    for (int k = 2; k + 2 < nkt; k += 2) {
      pre();
      ktile(F{}, T{}, F{}, fa0, fb0, fa1, fb1, s_cur, koff, koff, false, 0u, ra, rb, rbias);
      pre();
      ktile(F{}, T{}, F{}, fa1, fb1, fa0, fb0, s_cur, koff, koff, false, 0u, ra, rb, rbias);
    }
    pre();
    ktile(F{}, T{}, F{}, fa0, fb0, fa1, fb1, s_cur, koff, koff, false, 0u, ra, rb, rbias);
    pre();
    ktile(F{}, F{}, T{}, fa1, fb1, fa0, fb0, s_cur, koff, koff, false, bias_lds_next, ra, rb, rbias);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results are read by compiler-generated code
    primed = stage_next;
    // the epilogue's lane index, recomputed here: as a kernel-entry value its derived addresses (C pointer, bias slot) were spilled
    // around the tile loop, and a scratch reload's compiler-inserted vmcnt(0) waits out the LDS-DMA just primed
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    // this tile's bias slice, one 64-column half at a time (the last K tile staged the NEXT tile's into the other slot): both
    // halves up front were 32 live values too many next to the feed-forward epilogues' own
    const int wm0 = wr * 16 * NI;
    const char* bl = smem + C4_LDS_BIAS + par * 2048 + wave * 512;
    auto load_bias_half = [bl, lane_e](int h, BiasRegs<4>& br) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + (64 * h + 16 * j + 4 * (lane_e >> 4)) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) br.v[j][r] = b4[r];
      }
    };
    if constexpr (EPI == SSAK_EPI_MUL_AUX) {
      // the factor codes of both halves in flight before any arithmetic (the fragment registers are free now)
      FqCodes<NI> codes[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) load_fq_codes<NI>(p, codes[h], c.bm0, c.bn0, wm0, wc * 128 + 64 * h, lane_e, c.z1, c.z2);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        BiasRegs<4> br;
        load_bias_half(h, br);
        gemm_epilogue_direct<NI, EPI>(p, acc[h], br, c.bm0, c.bn0, wm0, wc * 128 + 64 * h, lane_e, c.z, c.z1, c.z2, 0, &codes[h]);
      }
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        BiasRegs<4> br;
        load_bias_half(h, br);
        gemm_epilogue_direct<NI, EPI>(p, acc[h], br, c.bm0, c.bn0, wm0, wc * 128 + 64 * h, lane_e, c.z, c.z1, c.z2, 0);
      }
    }
    if (!primed) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // dummies (and whatever the epilogue left) before LDS is re-staged
      asm volatile("s_barrier" ::: "memory");           // ... by a wave that may be ahead: every wave's dummies have landed
    }
    t = t_next;
    par ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int c4_num_cu(int* out) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    SSAK_HIP(hipGetDevice(&dev));
    SSAK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  }
  *out = n_cu;
  return SSAK_OK;
}

template <int EPI>
int launch_c4(const GemmParams& p, hipStream_t st) {
  auto kern = gemm_c4_kernel<EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    SSAK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C4_LDS));
    attr_done = true;
  }
  const long ntiles = (long)p.tiles_m * p.tiles_n * p.nz;
  int n_cu = 0;
  if (int rc = c4_num_cu(&n_cu)) return rc;
  // one timing slot per (instantiation, N, K), as launch_p8
  static std::mutex slot_mu;
  static std::map<std::pair<int, int>, int> slots;
  int slot;
  {
    std::lock_guard<std::mutex> lock(slot_mu);
    auto it = slots.find({p.N, p.K});
    if (it == slots.end()) {
      char nm[112];
      snprintf(nm, sizeof(nm), "gemm_c4_kernel<%d> (N = %d, K = %d)", EPI, p.N, p.K);
      it = slots.emplace(std::make_pair(p.N, p.K), ssak_prof_register(nm, SSAK_BOUND_MFMA)).first;
    }
    slot = it->second;
  }
  ProfScope prof_scope(slot, 2.0 * p.M * p.N * (double)p.K * p.nz, st);
  kern<<<dim3((unsigned)std::min<long>(ntiles, 2L * n_cu)), C4_THREADS, C4_LDS, st>>>(p);  // two persistent workgroups per CU
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

}  // namespace

// true when the co-resident kernel can run this product (128-row tiles, no split-K, static tile order)
bool ssak_gemm_c4_supports(const void* params, int a_km, int b_km) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  if (a_km || b_km || p.split_k != 1 || p.dynamic) return false;
  const bool plain = p.epilogue == SSAK_EPI_NONE && !p.drop_thresh && !p.colsum && !p.out_f32 && !p.accumulate;
  const bool gelu = p.epilogue == SSAK_EPI_GELU && !p.aux_out && !p.drop_thresh && !p.colsum && !p.out_f32;
  if (!(plain || gelu || p.epilogue == SSAK_EPI_GELU_SAVE_GRAD || p.epilogue == SSAK_EPI_MUL_AUX)) return false;
  if (p.kperm_n2) return false;  // (Toeplitz K order: the conv stack stays on gemm_p4.hip)
  if (p.K % 64 != 0 || p.K < 192 || p.N % 256 != 0) return false;
  if ((p.ldc & 7) || ((p.sc1 | p.sc2) & 7)) return false;
  const bool fq = p.epilogue == SSAK_EPI_GELU_SAVE_GRAD || p.epilogue == SSAK_EPI_MUL_AUX;
  if (fq && ((p.ldc | p.sc1 | p.sc2) & 15)) return false;
  if ((((uintptr_t)p.aux_in | (uintptr_t)p.aux_out | (uintptr_t)p.C) & 15) != 0) return false;
  if (p.out_f32) return false;
  return true;
}

int ssak_gemm_c4_launch(const void* params, hipStream_t st) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  if (p.epilogue == SSAK_EPI_GELU_SAVE_GRAD) return launch_c4<SSAK_EPI_GELU_SAVE_GRAD>(p, st);
  if (p.epilogue == SSAK_EPI_MUL_AUX) return launch_c4<SSAK_EPI_MUL_AUX>(p, st);
  if (p.epilogue == SSAK_EPI_GELU) return launch_c4<P8_EPI_GELU_ONLY>(p, st);
  return launch_c4<P8_EPI_PLAIN_BF16>(p, st);
}
