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
// LDS: SA stages of A (32 NI rows x 128 B) | two stages of B (256 rows x 128 B) | two bias slots of 1 KiB per wave (128 fp32 from
// lanes 0-31; the other lanes' LDS-DMA writes zeros)

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

struct P4Tile {
  int bm0, bn0, z, z1, z2;
  int k0;  // K tile this output tile's loop starts at (it wraps around): see decode()
};
typedef u32x4 Frag;  // 8 bf16 = one MFMA operand

// NI: 16-row groups per wave (tile = 32 NI x 256).  SA: LDS stages of the A operand (3 where they fit: the activations are read
// once, from HBM, and want two K tiles in flight; B -- a weight every row panel re-reads, from L2 / the Infinity Cache -- has 2).
// EPI: as gemm_p8_kernel.
template <int NI, int SA, int EPI>
__global__ __launch_bounds__(P4_THREADS) void gemm_p4_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 32 * NI;
  constexpr int NA = NI, NB = 8;           // LDS-DMA instructions per wave and K tile: A (32 rows each), B
  constexpr int A_SZ = BM * 128, B_SZ = 32768;
  constexpr int LDS_B = SA * A_SZ, LDS_BIAS = LDS_B + 2 * B_SZ;
  // epilogues that store at least a fixed number of 16-byte rows per 16-row group may leave their stores in flight: the counted
  // waits of the next tile use that LOWER bound of what is younger than the primed K tiles (MUL_AUX: its column-sum stores are not
  // counted, its factor-code loads are consumed -- waited for, with everything older -- inside the epilogue)
  constexpr bool EPI_EARLY = EPI == P8_EPI_PLAIN_BF16 || EPI == P8_EPI_GELU_ONLY || EPI == SSAK_EPI_GELU_SAVE_GRAD || EPI == SSAK_EPI_MUL_AUX;
  constexpr int EPI_STORES = (EPI == SSAK_EPI_GELU_SAVE_GRAD ? 6 : 4) * NI;
  constexpr int MID_KEEP = SA == 3 ? NA : 0;  // LDS-DMA younger than what a K tile's barrier needs: the A tile staged last
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int per_z = p.tiles_m * p.tiles_n;
  const int ntiles = per_z * p.nz;
  const int nkt = p.K / BK;  // >= 3 (launcher)
  // byte offset of the K tile visited at step k (gemm_common.h: kperm_*; identity without a Toeplitz A)
  const int kpp = p.kperm_p, kpn = p.kperm_n2;
  auto decode = [&](int t) __attribute__((always_inline)) {
    P4Tile c;
    const int id = xcd_remap(t, ntiles);
    const int zs = id / per_z, rem = id % per_z;
    c.z = zs;
    c.z1 = zs / p.nb2;
    c.z2 = zs % p.nb2;
    c.bm0 = rem / p.tiles_n * BM;
    c.bn0 = rem % p.tiles_n * 256;
    // K ROTATION: row panel r starts its K loop at K tile (7 r) mod nkt and wraps around.  B is a weight that every row panel
    // streams, from HBM the first time in a train step; with every workgroup on the same K tile at the same moment each K tile
    // of it is a fresh miss for ALL of them at once, 2 us of latency against the ~1 us a B stage is staged ahead (cold
    // operands: +10 % on the K >= 2304 products, tools/bench_gemm_cold.py).  Rotated, one workgroup takes the miss and the
    // others find the lines in L2 / the Infinity Cache.  (The column tiles of a row panel keep one start: they share A.)
    // fp32 summation order differs per row panel; every panel's own order is fixed, results are deterministic.
    c.k0 = kpn ? 0 : (rem / p.tiles_n * 7 + zs * 3) % nkt;
    return c;
  };
  auto koff_of = [kpp, kpn](int k) __attribute__((always_inline)) {
    const int kt = k < 2 * kpn ? (k >> 1) + ((k & 1) ? kpp : 0) : k - kpn;
    return (uint32_t)kt * 128u;
  };
  typedef __attribute__((address_space(3))) char lds_char;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)smem;
  const uint32_t wbase = lds0 + wave * 1024;  // LDS-DMA destination of this wave: + stage + instruction * 4 KiB
  // ---- LDS-DMA source offsets: instruction j of this wave fills rows 32 j + 8 wave + (lane >> 3) (128 B each, chunk swizzle on
  // the source side).  One VGPR offset per instruction, constant for the output tile (the descriptor's range check zero-fills
  // rows beyond the matrix from it); the K tile's offset is scalar (soffset).
  const int drow = 8 * wave + (lane >> 3);
  const int dchunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  uint32_t voa[NA], vob[NB];
  u32x4 ra_v, rb_v;  // descriptors as computed; uni4() right before use hands them to the asm statements in SGPRs
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
  auto setup_a = [&](const P4Tile& c, bool real) __attribute__((always_inline)) {
    ra_v = make_rsrc(p.A + c.z1 * p.sa1 + c.z2 * p.sa2, p.ext_a);
#pragma unroll
    for (int q = 0; q < NA; ++q) voa[q] = real ? (uint32_t)(((long)(c.bm0 + 32 * q + drow) * p.lda + dchunk * 8) * 2) : 0x80000000u;
  };
  auto setup_b = [&](const P4Tile& c, bool real) __attribute__((always_inline)) {
    rb_v = make_rsrc(p.B + c.z1 * p.sb1 + c.z2 * p.sb2, p.ext_b);
#pragma unroll
    for (int q = 0; q < NB; ++q) vob[q] = real ? (uint32_t)(((long)(c.bn0 + 32 * q + drow) * p.ldb + dchunk * 8) * 2) : 0x80000000u;
    // bias of the wave's 128 columns: lanes 0-31, 16 B each
    bias_vo = (real && lane < 32) ? (uint32_t)((c.z2 * p.bias_s2 + c.bn0 + wc * 128 + 4 * lane) * 4) : 0x80000000u;
  };
  // ---- fragment read addresses inside stage 0 [slice]: row 16 i + lm of this wave's panel, 16-byte chunk 4 slice + lq
  const int lm = lane & 15, lq = lane >> 4;
  uint32_t fo_a[2], fo_b[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    fo_a[kk] = lds0 + (wr * 16 * NI + lm) * 128 + (((4 * kk + lq) ^ ((lm >> 1) & 7)) << 4);
    fo_b[kk] = lds0 + LDS_B + (wc * 128 + lm) * 128 + (((4 * kk + lq) ^ ((lm >> 1) & 7)) << 4);
  }
  const uint32_t bias_lds0 = lds0 + LDS_BIAS + wave * 1024;

  // one LDS-DMA with its M0 write: the form outside the MFMA stream (prologue)
  auto dma_plain = [](uint32_t dst, uint32_t voff, u32x4 rsrc, uint32_t soff) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  };

  f32x4 acc[2][NI][4];  // [column half][16-row group][16-column group]: a half is what gemm_epilogue_direct takes
  Frag fa0[NI], fb0[8], fa1[NI], fb1[8];

  // ---- one K tile.  s_sa / s_sb: byte offsets of the LDS stages holding its A / B (in: this tile's; out: the next tile's);
  //   a_cur / b_cur: this lane's slice-1 fragment addresses in them (in / out likewise) -- the few scalar and vector
  //   instructions that advance them sit in free gaps of slice 0, not in a block between two K tiles.
  //   ZERO: first K tile of an output tile (slice 0 multiplies into 0).  READ_NEXT: read set 0 of the following K tile.
  //   BIAS: stage the next output tile's bias slice too.  Slice 1 always stages B of the K tile two ahead into this tile's B
  //   stage and A of the K tile SA ahead into its A stage (the caller has pointed vob / voa / the descriptors / koff at them:
  //   this or the next output tile).  keep_stores: the previous epilogue's stores are still in flight behind this tile's operands.
  auto ktile = [&acc, &fa0, &fb0, &fa1, &fb1, &voa, &vob, &bias_vo, &fo_a, &fo_b, wbase](
                   auto zero_c, auto rn_c, auto bias_c, uint32_t& s_sa, uint32_t& s_sb, uint32_t& a_cur, uint32_t& b_cur, uint32_t koff_a,
                   uint32_t koff_b, bool keep_stores, uint32_t bias_dst, const u32x4 ra, const u32x4 rb, const u32x4 rbias, int* ticket_ctr,
                   uint32_t& ticket) __attribute__((always_inline)) {
    constexpr bool ZERO = decltype(zero_c)::value, READ_NEXT = decltype(rn_c)::value, BIAS = decltype(bias_c)::value;
    uint32_t n_sa = 0, n_sb = 0, a_nxt = 0, b_nxt = 0, a_cur_n = 0, b_cur_n = 0, wb_a = 0, wb_b = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // set 0 is in registers
    // ---- slice 0
    static_for<8 * NI>([&acc, &fa0, &fb0, &fa1, &fb1, &fo_a, &fo_b, &a_cur, &b_cur, &s_sa, &s_sb, &n_sa, &n_sb, &a_nxt, &b_nxt, &a_cur_n, &b_cur_n,
                        &wb_a, &wb_b, wbase, keep_stores, ticket_ctr, &ticket](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value, i = m / 8, j = m % 8;
      if constexpr (ZERO) P4_MFMA_Z(acc[j / 4][i][j % 4], fb0[j], fa0[i]);
      else P4_MFMA(acc[j / 4][i][j % 4], fb0[j], fa0[i]);
      if constexpr (m % 2 == 0 && m / 2 < 8 + NI) {
        constexpr int r = m / 2;
        if constexpr (r == 0) P4_READ(fa1[0], a_cur, 0);
        else if constexpr (r <= 8) P4_READ(fb1[r - 1], b_cur, (r - 1) * 2048);
        else P4_READ(fa1[r - 8], a_cur, (r - 8) * 2048);
      }
      // bookkeeping for slice 1 and for the next K tile, one instruction or two per free gap
      if constexpr (m == 1) n_sa = s_sa + A_SZ == SA * A_SZ ? 0u : s_sa + A_SZ;
      if constexpr (m == 3) n_sb = s_sb ^ B_SZ;
      if constexpr (m == 5) a_nxt = fo_a[0] + n_sa;
      if constexpr (m == 7) b_nxt = fo_b[0] + n_sb;
      if constexpr (m == 9) a_cur_n = fo_a[1] + n_sa;
      if constexpr (m == 11) b_cur_n = fo_b[1] + n_sb;
      if constexpr (m == 13) wb_a = wbase + s_sa;
      if constexpr (m == 15) wb_b = wbase + LDS_B + s_sb;
      if constexpr (m == 8 * NI - 4) {
        // my reads of these stages are done; my share of the next K tile has landed (everything but the youngest A tile and,
        // right after an early-staged start, the previous epilogue's stores)
        if (keep_stores) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(MID_KEEP + EPI_STORES < 63 ? MID_KEEP + EPI_STORES : 63) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(MID_KEEP) : "memory");
      }
      if constexpr (m == 8 * NI - 3) asm volatile("s_barrier" ::: "memory");
      if constexpr (ZERO && m == 8 * NI - 2) {
        // ticket tile order: one lane requests the NEXT tile's ticket here, behind the wait and ahead of this K tile's LDS-DMA --
        // the next K tile's counted wait (everything older than its youngest A tile) then covers the returning atomic, so no
        // wait count anywhere has to know about it; the value is read two K tiles later (pre())
        if (ticket_ctr != nullptr && threadIdx.x == 0) {
          const uint32_t one = 1;
          asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ticket) : "v"(ticket_ctr), "v"(one) : "memory");
        }
      }
    });
    // ---- slice 1: groups of DS MFMAs: [fragment read | M0 write | LDS-DMA | ...]; DMA order: (bias,) B, then A
    static_for<8 * NI>([&acc, &fa0, &fb0, &fa1, &fb1, &voa, &vob, ra, rb, rbias, &bias_vo, a_nxt, b_nxt, wb_a, wb_b, koff_a, koff_b, bias_dst](
                           auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value, i = m / 8, j = m % 8;
      constexpr int NG = NA + NB + (BIAS ? 1 : 0);
      constexpr int DS = (8 * NI / NG < 4) ? 8 * NI / NG : 4;
      constexpr int g = m / DS, ph = m % DS;
      constexpr int q = BIAS ? g - 1 : g;  // LDS-DMA of this group: -1 the bias slice, 0 .. NB - 1 B, NB .. A
      if constexpr (ph == 1 && g < NG) {
        if constexpr (q < 0) {
          asm volatile("s_mov_b32 m0, %3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tbuffer_load_dwordx4 %4, %5, 0 offen lds"
                       : "+a"(acc[j / 4][i][j % 4])
                       : "v"(fb1[j]), "v"(fa1[i]), "s"(bias_dst), "v"(bias_vo), "s"(rbias)
                       : "memory");
        } else if constexpr (q < NB) {
          asm volatile("s_add_u32 m0, %3, %4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tbuffer_load_dwordx4 %5, %6, %7 offen lds"
                       : "+a"(acc[j / 4][i][j % 4])
                       : "v"(fb1[j]), "v"(fa1[i]), "s"(wb_b), "i"(q * 4096), "v"(vob[q]), "s"(rb), "s"(koff_b)
                       : "memory", "scc");
        } else {
          asm volatile("s_add_u32 m0, %3, %4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tbuffer_load_dwordx4 %5, %6, %7 offen lds"
                       : "+a"(acc[j / 4][i][j % 4])
                       : "v"(fb1[j]), "v"(fa1[i]), "s"(wb_a), "i"((q - NB) * 4096), "v"(voa[q - NB]), "s"(ra), "s"(koff_a)
                       : "memory", "scc");
        }
      } else {
        P4_MFMA(acc[j / 4][i][j % 4], fb1[j], fa1[i]);
      }
      if constexpr (READ_NEXT) {
        // fragment reads of the next K tile: one per group in the DMA's free gaps, the rest after the last DMA
        constexpr int tail0 = DS * NG;  // first MFMA after the DMA groups
        constexpr bool in_groups = m < tail0 && ph == 0 && g < 8 + NI;
        constexpr int r_tail = NG + (m - tail0);
        constexpr bool in_tail = m >= tail0 && r_tail < 8 + NI;
        if constexpr (in_groups || in_tail) {
          constexpr int r = in_groups ? g : r_tail;
          if constexpr (r == 0) P4_READ(fa0[0], a_nxt, 0);
          else if constexpr (r <= 8) P4_READ(fb0[r - 1], b_nxt, (r - 1) * 2048);
          else P4_READ(fa0[r - 8], a_nxt, (r - 8) * 2048);
        }
      }
    });
    s_sa = n_sa;
    s_sb = n_sb;
    a_cur = a_cur_n;
    b_cur = b_cur_n;
  };
  using T = std::true_type;
  using F = std::false_type;

  bool primed = false;
  int par = 0;                  // output-tile parity: which bias slot
  uint32_t s_sa = 0, s_sb = 0;  // byte offsets of the LDS stages of the current K tile's A (cycles through SA stages, across output tiles) / B
  // Tile order.  Static (tile_ctr == null, the single-GPU default): workgroup b takes logical ids b, b + grid, ...  Dynamic
  // (ssak_gemm_desc.dynamic_tiles; data-parallel runs, as gemm_p8.hip): every tile is drawn from a per-XCD ticket counter
  // (x = blockIdx.x & 7, ticket k -> logical id 8 k + x), so that a workgroup whose CU is held by another stream's kernel --
  // the RCCL all-reduce of the previous layer's gradients -- takes fewer tiles instead of finishing last.  The ticket of the
  // next tile is requested inside K tile 0 (ktile), published through LDS by pre() at K tile 2 and read at K tile 3 (one
  // barrier in between); only the first ticket of a launch is waited for on the spot.
  int* const tile_ctr = p.tile_ctr ? p.tile_ctr + (blockIdx.x & 7) : nullptr;
  uint32_t ticket = 0;
  volatile int* const ticket_lds = reinterpret_cast<volatile int*>(smem + LDS_BIAS + 2 * 4096);
  int t_first = blockIdx.x;
  if (tile_ctr) {
    if (threadIdx.x == 0) *ticket_lds = atomicAdd(tile_ctr, 1);
    __syncthreads();
    t_first = 8 * __builtin_amdgcn_readfirstlane(*ticket_lds) + (int)(blockIdx.x & 7);
    __syncthreads();
  }
  P4Tile cur = decode(min(t_first, ntiles - 1));
  for (int t = t_first; t < ntiles;) {
    const P4Tile c = cur;
    const uint32_t bias_lds = bias_lds0 + par * 4096;
    const bool was_primed = primed;
    if (!primed) {
      setup_a(c, true);
      setup_b(c, true);
      const u32x4 ra = uni4(ra_v), rb = uni4(rb_v);
      dma_plain(bias_lds, bias_vo, rbias, 0);
      // A0 B0 A1 B1 (A2): the same order and counts the tail of a previous tile leaves in flight
      uint32_t sa = s_sa, sb = s_sb;
      for (int k = 0; k < 2; ++k) {
        static_for<NA>([&voa, ra, &dma_plain, wbase, sa, &koff_of, k, &c, nkt](auto q) __attribute__((always_inline)) { dma_plain(wbase + sa + q * 4096, voa[q], ra, koff_of((k + c.k0) % nkt)); });
        static_for<NB>([&vob, rb, &dma_plain, wbase, sb, &koff_of, k, &c, nkt](auto q) __attribute__((always_inline)) { dma_plain(wbase + LDS_B + sb + q * 4096, vob[q], rb, koff_of((k + c.k0) % nkt)); });
        sa = sa + A_SZ == SA * A_SZ ? 0 : sa + A_SZ;
        sb ^= B_SZ;
      }
      if (SA == 3) static_for<NA>([&voa, ra, &dma_plain, wbase, sa, &koff_of, &c, nkt](auto q) __attribute__((always_inline)) { dma_plain(wbase + sa + q * 4096, voa[q], ra, koff_of((2 + c.k0) % nkt)); });
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NA + NB + (SA == 3 ? NA : 0)) : "memory");  // bias, A0, B0 (this wave's share)
    } else {
      // everything older than B1, the youngest A tile and the stores: bias, A0, B0 (and A1)
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NB + NA + EPI_STORES < 63 ? NB + NA + EPI_STORES : 63) : "memory");
    }
    asm volatile("s_barrier" ::: "memory");
    {
      const uint32_t a0 = fo_a[0] + s_sa, b0 = fo_b[0] + s_sb;
      static_for<8>([&fb0, b0](auto j) __attribute__((always_inline)) { P4_READ(fb0[j], b0, j * 2048); });
      static_for<NI>([&fa0, a0](auto i) __attribute__((always_inline)) { P4_READ(fa0[i], a0, i * 2048); });
    }
    uint32_t a_cur = fo_a[1] + s_sa, b_cur = fo_b[1] + s_sb;
    // the next output tile: the tail of this one stages its first K tiles
    int t_next = t + (int)gridDim.x;  // (ticket order: known from K tile 3 on, pre())
    const bool full_rows = c.bm0 + BM <= p.M;  // (a tile with rows beyond M skips some epilogue stores: their count is not fixed)
    bool stage_next = EPI_EARLY && t_next < ntiles && full_rows;
    if (!tile_ctr && t_next < ntiles) cur = decode(t_next);
    const uint32_t bias_lds_next = bias_lds0 + (par ^ 1) * 4096;
    u32x4 ra = uni4(ra_v), rb = uni4(rb_v);
    BiasRegs<4> bias_regs[2];
    // per K tile, before its MFMAs: which K tiles its slice 1 stages -- B of K tile k + 2 and A of K tile k + SA, of this output
    // tile or of the next one (then the offsets / descriptors are re-pointed first)
    int ka = SA - 1, kb = 1;  // (+ 1 in pre(): K tiles SA and 2 for k = 0)
    int k0a = c.k0, k0b = c.k0;  // K rotation of the output tile whose A / B is being staged
    uint32_t koff_a = 0, koff_b = 0;
    int kcount = 0;
    auto pre = [&]() __attribute__((always_inline)) {
      if (tile_ctr) {
        if (kcount == 2) {
          // (wave 0 passed K tile 1's counted wait: its ticket has returned)
          if (threadIdx.x == 0) {
            typedef __attribute__((address_space(3))) int lds_int;
            const uint32_t a = (uint32_t)(uintptr_t)(lds_int*)ticket_lds;
            asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(ticket) : "memory");
          }
        } else if (kcount == 3) {
          t_next = 8 * __builtin_amdgcn_readfirstlane(*ticket_lds) + (int)(blockIdx.x & 7);
          stage_next = EPI_EARLY && t_next < ntiles && full_rows;
          if (t_next < ntiles) cur = decode(t_next);
        }
      }
      ++kcount;
      if (++ka == nkt) {
        ka = 0;
        k0a = cur.k0;
        setup_a(cur, stage_next);
        ra = uni4(ra_v);
      }
      if (++kb == nkt) {
        kb = 0;
        k0b = cur.k0;
        // (this tile's bias slice into registers first: the next tile's lands in the other slot)
        const char* bl = smem + LDS_BIAS + par * 4096 + wave * 1024;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + (64 * h + 16 * j + 4 * lq) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) bias_regs[h].v[j][r] = b4[r];
          }
        setup_b(cur, stage_next);
        rb = uni4(rb_v);
      }
      const int pa = ka + k0a >= nkt ? ka + k0a - nkt : ka + k0a, pb = kb + k0b >= nkt ? kb + k0b - nkt : kb + k0b;
      koff_a = kpn ? koff_of(pa) : (uint32_t)pa * 128u;
      koff_b = kpn ? koff_of(pb) : (uint32_t)pb * 128u;
    };
    pre();
    ktile(T{}, T{}, F{}, s_sa, s_sb, a_cur, b_cur, koff_a, koff_b, was_primed, 0u, ra, rb, rbias, tile_ctr, ticket);
    for (int k = 1; k + 1 < nkt; ++k) {
      pre();
      ktile(F{}, T{}, F{}, s_sa, s_sb, a_cur, b_cur, koff_a, koff_b, false, 0u, ra, rb, rbias, nullptr, ticket);
    }
    pre();
    ktile(F{}, F{}, T{}, s_sa, s_sb, a_cur, b_cur, koff_a, koff_b, false, bias_lds_next, ra, rb, rbias, nullptr, ticket);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results are read by compiler-generated code
    primed = stage_next;
    if constexpr (EPI == SSAK_EPI_MUL_AUX) {
      // the factor codes of both halves in flight before any arithmetic (the fragment registers are free now)
      FqCodes<NI> codes[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) load_fq_codes<NI>(p, codes[h], c.bm0, c.bn0, wr * 16 * NI, wc * 128 + 64 * h, lane, c.z1, c.z2);
#pragma unroll
      for (int h = 0; h < 2; ++h)
        gemm_epilogue_direct<NI, EPI>(p, acc[h], bias_regs[h], c.bm0, c.bn0, wr * 16 * NI, wc * 128 + 64 * h, lane, c.z, c.z1, c.z2, 0, &codes[h]);
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        gemm_epilogue_direct<NI, EPI>(p, acc[h], bias_regs[h], c.bm0, c.bn0, wr * 16 * NI, wc * 128 + 64 * h, lane, c.z, c.z1, c.z2, 0);
    }
    if (!primed) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // dummies (and whatever the epilogue left) before LDS is re-staged
    t = t_next;
    par ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // the last workgroup out leaves the counters at zero for the next launch on this stream (agent-scope atomics only, as gemm_p8.hip)
  if (p.tile_ctr && threadIdx.x == 0) {
    if (atomicAdd(p.tile_ctr + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i) atomicExch(p.tile_ctr + i, 0);
    }
  }
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

template <int NI, int EPI>
int launch_p4(const GemmParams& p, hipStream_t st) {
  constexpr int SA = NI <= 6 ? 3 : 2;  // three A stages where 160 KB of LDS hold them next to two B stages and the bias slots
  constexpr int P4_LDS = SA * NI * 32 * 128 + 2 * 32768 + 2 * 4096 + 64;  // (+ the ticket word)
  static_assert(P4_LDS <= 160 * 1024, "LDS");
  auto kern = gemm_p4_kernel<NI, SA, EPI>;
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
      snprintf(nm, sizeof(nm), "gemm_p4_kernel<%d, %d, %d> (N = %d, K = %d)", NI, SA, EPI, p.N, p.K);
      it = slots.emplace(std::make_pair(p.N, p.K), ssak_prof_register(nm, SSAK_BOUND_MFMA)).first;
    }
    slot = it->second;
  }
  GemmParams q = p;
  q.tile_ctr = nullptr;
  if (ntiles > n_cu && n_cu % 8 == 0 && p.dynamic)
    if (int rc = ssak_gemm_ticket_slot(st, &q.tile_ctr)) return rc;
  ProfScope prof_scope(slot, 2.0 * p.M * p.N * (double)p.K * p.nz, st);
  kern<<<dim3((unsigned)std::min<long>(ntiles, n_cu)), P4_THREADS, P4_LDS, st>>>(q);  // one persistent workgroup per CU
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <int NI>
int dispatch_p4(const GemmParams& p, int b_km, hipStream_t st) {
  if (b_km) {
    ssak_set_error("gemm_p4: built for K-contiguous operands");
    return SSAK_ERR_INVALID;
  }
  if (p.epilogue == SSAK_EPI_GELU_SAVE_GRAD) return launch_p4<NI, SSAK_EPI_GELU_SAVE_GRAD>(p, st);
  if (p.epilogue == SSAK_EPI_MUL_AUX) return launch_p4<NI, SSAK_EPI_MUL_AUX>(p, st);
  const bool no_extras = !p.drop_thresh && !p.colsum;
  if (no_extras && p.epilogue == SSAK_EPI_NONE && !p.out_f32 && !p.accumulate) return launch_p4<NI, P8_EPI_PLAIN_BF16>(p, st);
  if (no_extras && p.epilogue == SSAK_EPI_GELU && !p.aux_out && !p.out_f32) return launch_p4<NI, P8_EPI_GELU_ONLY>(p, st);
  if constexpr (NI == 8) {
    // the general epilogue form on a 256-row tile needs scratch inside the K loop (ssak_gemm_p4_supports never plans it: not built)
    ssak_set_error("gemm_p4: the general epilogue form is built for 128- and 192-row tiles");
    return SSAK_ERR_INVALID;
  } else {
    return launch_p4<NI, -1>(p, st);
  }
}

}  // namespace

// true when the four-wave kernel can run this product as planned (tile height bm, no split-K): everything it does not cover
// stays on gemm_p8.hip
bool ssak_gemm_p4_supports(const void* params, int bm, int a_km, int b_km) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  if (a_km || b_km || p.split_k != 1) return false;
  // the feed-forward up-projection's epilogue (GELU + GELU' + dropout hash + 8-bit codes: ~22 VALU instructions per element) is
  // VALU-issue-bound, and a lone wave per SIMD issues a VALU instruction every ~4.3 cycles where two waves share 2: the
  // eight-wave kernel runs it ~100 us per step faster (same-box A/B, profiles/r04_ab_ffn_up_kernel.log)
  if (p.epilogue == SSAK_EPI_GELU_SAVE_GRAD) return false;
  // ... and the general epilogue form (dropout / column sums / GELU with a saved pre-activation / fp32 out: every mode compiled into
  // one body) does not fit next to 256 accumulators without scratch on 256-row tiles: those stay on the eight-wave kernel too
  const bool lean = p.epilogue == SSAK_EPI_MUL_AUX ||
                    (!p.drop_thresh && !p.colsum && !p.out_f32 && !p.accumulate && (p.epilogue == SSAK_EPI_NONE || (p.epilogue == SSAK_EPI_GELU && !p.aux_out)));
  if (!lean && bm == 256) return false;
  if (p.dynamic && p.K < 6 * 64) return false;  // (the ticket of the next tile travels through K tiles 0 .. 3 ahead of the tail's staging)
  if (bm != 256 && bm != 192 && bm != 128) return false;
  if (p.K % 64 != 0 || p.K < 192 || p.N % 256 != 0) return false;
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
