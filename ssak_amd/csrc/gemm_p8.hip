// 256x256-tile bf16 MFMA GEMM with a phase-interleaved LDS-DMA pipeline (gfx950).  Same contract as the kernels
// of gemm.hip (GemmParams, fused epilogue); selected by the dispatcher for the large encoder products.
//
// Why another kernel: the 128x128 two-stage kernel keeps ONE 32 KB tile of LDS-DMA in flight per workgroup and
// drains it (vmcnt 0) before every K step, so each step pays the L2/HBM round trip (rocprofv3: MFMA busy 29 %,
// SQ_WAIT_ANY 48 %); its 64x64 wave tiles also read 0.5 ds_read_b128 per MFMA.  Here
//   * one 8-wave workgroup per CU owns a 256x256 tile: half the L2->LDS bytes per FLOP of a 128x128 tile;
//   * waves are 2 (M) x 4 (N), 128x64 of output each: 0.375 ds_read_b128 per MFMA, 128 accumulator registers;
//   * a 64-deep K tile is staged as FOUR 16 KB pieces, cut along what one phase consumes:
//       AT/AB = the top / bottom 64 rows of both wave rows' A panels, BL/BR = the left / right 32 columns of all
//       four wave columns' B panels;
//   * a K tile is TWO load / matrix phase pairs, one output half (64 x 64 per wave, 8 * MH MFMAs) each:
//       LA: read AT,BL,BR -> C(top, left | right)      DMA AB(t+1)                    wait for AB(t)
//       LB: read AB       -> C(bottom, left | right)   DMA AT(t+2), BL(t+2), BR(t+2)  wait for AT(t+1), BL(t+1), BR(t+1)
//     every wait is a counted `s_waitcnt vmcnt(8)`: four pieces (64 KB per CU) stay in flight across the raw s_barriers and
//     each piece has two phases (>= 1000 MFMA cycles) to land; vmcnt never drains in the loop.  (Round 1 ran FOUR pairs per K
//     tile, a quadrant each; tools/probes/p8_loop.hip showed that loop bound by its load phases -- 2 LDS-DMA instructions queueing
//     behind the CU's one address unit + up to 12 fragment reads + two barriers took 300-380 cycles against the partner's 256
//     cycles of MFMAs (the loop without its DMA: +32 %; without its barriers: -11 %).  With half as many, twice as long phases
//     the fixed costs are paid half as often and a ~450-cycle load phase hides under 512 cycles of MFMAs: +4 % (conv1) to
//     +13 % (K = 3072 feed-forward) on the train step's shapes, bit-identical results.)
// Hazards (G0 = wave row 0, G1 = wave row 1, one barrier behind): a piece waited for in phase w (before the phase's
// first barrier) is read in phase >= w+1; a slot last read in load phase r is re-staged in load phase r+1 -- the fragment
// reads are drained (lgkmcnt(0)) BEFORE the barrier that closes a load phase, so the lagging wave row has finished with the
// slot when the leading one issues the DMA.  Past the last K
// tile the same DMA instructions are issued with out-of-range offsets (the DMA then writes zeros into slots nobody
// reads any more), which keeps the vmcnt arithmetic uniform.
//
// LDS images per piece (as the 128x128 kernel): K-contiguous [128 rows][128 B], 16-B chunk c of row r at slot
// c ^ ((r >> 1) & 7) (ds_read_b128, conflict-free); K-major [64 k-rows][256 B], chunk ch of k-row kr at slot
// ch ^ km_swz(kr) (ds_read_b64_tr_b16).  The DMA writes linearly, so the swizzles are applied to the source address.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>

#include "common.h"
#include "gemm_common.h"
#include "kernels.h"

namespace {

constexpr int P8_THREADS = 512;
constexpr int P8_PIECE = 16384;          // bytes per piece
constexpr int P8_BUF = 4 * P8_PIECE;     // AT, AB, BL, BR
constexpr int P8_PIPE = 2 * P8_BUF;      // 128 KiB of operand pieces
constexpr int P8_LDS = P8_PIPE + 8 * 1024 + 64;  // + 1 KiB per wave: the tile's bias slice, staged by DMA with the operands; + the next tile's ticket

// Staging state of one operand (two pieces: half 0 = AT / BL, half 1 = AB / BR).  SEG = rows of a piece taken from one
// wave row / column (16 * MH for A, 32 for B), SPAN = that wave row's / column's extent in the tile (2 * SEG), NSEG = wave
// rows / columns (2 for A, 4 for B).
// NV = wave-instructions (1 KiB each) a piece really has; every wave always issues two per piece (index j * 8 + wave),
// those with index >= NV as zero-writing dummies into the slot's unused tail, so the vmcnt arithmetic never changes.
template <bool KM, int SEG, int SPAN, int NV, int NSEG>
struct P8Stager {
  static constexpr uint32_t OOB = 0x80000000u;
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t off[2][2];  // [half][instr]: byte offset of this lane's chunk in the current k-tile (OOB: outside the matrix)
  int kofs[2];         // KC: first k of the chunk inside the k-tile; KM: its k-row
  uint32_t kstep;
  int wave;

  __device__ __forceinline__ void init(const bf16* base, long ld, int row0, int rows_total, int kt0, uint32_t extent_bytes) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)extent_bytes, 0x00020000);
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    kstep = (uint32_t)((KM ? (long)BK * ld : (long)BK) * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int S = (j * 8 + wave) * 64 + lane;  // 16-B slot of the piece this lane fills
      const bool used = j * 8 + wave < NV;
      if (!KM) {
        const int r = S >> 3, pc = S & 7;
        const int c = pc ^ ((r >> 1) & 7);
        kofs[j] = c * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int gr = row0 + (r / SEG) * SPAN + h * SEG + (r % SEG);
          off[h][j] = (used && gr < rows_total) ? (uint32_t)(((long)gr * ld + c * 8) * 2 + (long)kt0 * kstep) : OOB;
        }
      } else {
        const int kr = S >> 4, pc = S & 15;
        const int ch = pc ^ km_swz<128>(kr);
        const int pr = ch * 8;  // first piece row of the chunk
        kofs[j] = kr;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int gr = row0 + (pr / SEG) * SPAN + h * SEG + (pr % SEG);
          // K-major pieces keep 256-byte k-rows whatever SEG is: chunks beyond the piece's NSEG * SEG columns stay empty
          off[h][j] = (pr < NSEG * SEG && gr < rows_total) ? (uint32_t)(((long)kr * ld + gr) * 2 + (long)kt0 * kstep) : OOB;
        }
      }
    }
  }
  // stage piece `h` of k-tile kt (absolute index; kt >= kt_end: zero-writing dummies) and advance that piece's offsets
  template <int H>
  __device__ __forceinline__ void issue(char* lds_piece, int kt, int kt_end, int K, int perm_p = 0, int perm_n2 = 0) {
#if defined(P8_ABL) && (P8_ABL & 1)
    if (kt >= 2) return;  // ablation: no LDS-DMA inside the main loop
#endif
#if defined(P8_ABL) && (P8_ABL & 32)
    if (NSEG == 4 && kt >= 2) return;  // ablation: no LDS-DMA of the B operand inside the main loop
#endif
    // k-rows / k-chunks of this tile that exist (uniform): all 64, a K tail, or none (dummy past the last tile)
    const int klim = kt < kt_end ? min(K - kt * BK, BK) : 0;
    // advance to the next K tile of the visiting order: step kt -> kt + 1 moves by + p (even step of a pair), 1 - p (odd) or 1
    // tile (past the pairs).  An offset that starts out of range (OOB = 2^31) stays there: the moves never sum below zero and
    // the total advance is < 2 GB.
    const int dkt = kt < 2 * perm_n2 ? ((kt & 1) ? 1 - perm_p : perm_p) : 1;
    const uint32_t adv = (uint32_t)dkt * kstep;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t o = kofs[j] < klim ? off[H][j] : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds_piece + (j * 8 + wave) * 1024), 16, o, 0, 0, 0);
      off[H][j] += adv;
    }
  }
};

// per-lane fragment offsets inside a piece; w0 = first piece row of this wave's panel, NF 16-row groups
template <bool KM, int NF, bool IS_B = false>
struct P8Frag {
  int off[KM ? NF : 1];
  __device__ __forceinline__ void init(int w0, int lane) {
    if (!KM) {
      const int r = w0 + (lane & 15);
      off[0] = r * 128 + (((lane >> 4) ^ ((r >> 1) & 7)) << 4);
    } else {
      const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
      const int kr = 8 * g + q;
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int ch = (w0 >> 3) + 2 * i + (p >> 1);
        off[i] = kr * 256 + ((ch ^ km_swz<128>(kr)) << 4) + (p & 1) * 8;
      }
    }
  }
  __device__ __forceinline__ bf16x8 read(const char* piece, int i, int kk) const {
#if defined(P8_ABL) && (P8_ABL & 2)
    return __builtin_bit_cast(bf16x8, (f32x4){(float)i, (float)kk, 1.f, 2.f});  // ablation: no fragment reads
#endif
#if defined(P8_ABL) && (P8_ABL & 16)
    if (IS_B) return __builtin_bit_cast(bf16x8, (f32x4){(float)i, (float)kk, 1.f, 2.f});  // ablation: no B fragment reads
#endif
    if (!KM) {
      return *reinterpret_cast<const bf16x8*>(piece + ((off[0] ^ (kk << 6)) + i * 2048));
    } else {
      // inline asm on purpose: for the ds_read_tr builtin the compiler cannot tell the read apart from the LDS-DMA writes
      // in flight and puts `s_waitcnt vmcnt(0)` in front of every fragment read, which drains the whole pipeline four
      // times per K tile.  Ordering is by the counted waits + barriers of the phase structure; results are consumed
      // after the explicit lgkmcnt(0) of P8_MFMA.
      typedef __attribute__((address_space(3))) char lds_char;
      const uint32_t a = (uint32_t)(uintptr_t)(lds_char*)(piece + off[KM ? i : 0] + kk * 32 * 256);
      s16x4 lo, hi;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(hi) : "v"(a) : "memory");
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      return __builtin_bit_cast(bf16x8, v);
    }
  }
};

// development aid (tools/probes/p8_probe.hip): per-workgroup wall-clock stamps written through p.slab
#ifdef P8_STAMPS
#define P8_STAMP(i)                                                                                   \
  do {                                                                                                \
    if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(p.slab)[(blockIdx.x * 4 + p8_round) * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define P8_STAMP(i)
#endif

#define P8_FENCE() __builtin_amdgcn_sched_barrier(0)
#define P8_BARRIER()                \
  do {                              \
    P8_FENCE();                     \
    __builtin_amdgcn_s_barrier();   \
    P8_FENCE();                     \
  } while (0)
// the 4 * MH MFMAs of one output quadrant: 16-row groups I0..I0+MH-1 x column groups J0, J0+1, both 32-deep halves
#if defined(P8_ABL) && (P8_ABL & 4)
#define P8_MFMA_BODY(I0, J0, FB) acc[(I0)][(J0)] += (f32x4){(float)FB[0][0][0], (float)fa[0][0][0], 0.f, 0.f}; /* ablation: no MFMAs */
#else
#define P8_MFMA_BODY(I0, J0, FB)                                                                              \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int i = 0; i < MH; ++i)           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[(I0) + i][(J0) + j] =                               \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[j][kk], fa[i][kk], acc[(I0) + i][(J0) + j], 0, 0, 0);
#endif
#if defined(P8_ABL) && (P8_ABL & 8)
#define P8_LOOP_BARRIER() P8_FENCE() /* ablation: no barriers inside the main loop */
#else
#define P8_LOOP_BARRIER() P8_BARRIER()
#endif
#define P8_MFMA(I0, J0, FB)                                                                                   \
  do {                                                                                                        \
    P8_LOOP_BARRIER();                                                                                        \
    __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0) */                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                            \
    P8_MFMA_BODY(I0, J0, FB)                                                                                  \
    __builtin_amdgcn_s_setprio(0);                                                                            \
    P8_LOOP_BARRIER();                                                                                        \
  } while (0)

// MH = 16-row groups per quadrant: the tile is (64 * MH) x 256, i.e. 256 / 192 / 128 rows -- picked by the host so
// that the tile count fills whole rounds of 256 workgroups (M = 15968: 192-row tiles give 84 x 3 = 252 tiles for N = 768).
// Tile coordinates of one unit of work (uniform across the workgroup).
struct P8Tile {
  int bm0, bn0, z, z1, z2, split, kt0, kt1, g;
};

// Grouped launch: up to 8 independent products of the same K, layouts and epilogue (the weight gradients of one or two
// encoder layers) share ONE persistent launch, so that together they fill a round of workgroups without split-K slabs.
// The table travels in the kernel arguments (scalar loads with a dynamic index).
struct P8Problem {
  const bf16* A;
  const bf16* B;
  void* C;
  long lda, ldb, ldc;
  int M, N;
  uint32_t ext_a, ext_b;
  int tile0, tiles_n;  // first tile id of this problem, its tile columns
};
// CAP = 48 for the grouped instantiations (twelve encoder layers x 4 weight gradients: on one GPU the whole backward's weight
// gradients run as ONE multi-round launch, which fills the chip where a pair of layers -- 216 tiles -- leaves 40 CUs idle), 1 for
// the others (the table travels by value in the kernel arguments).
constexpr int P8_GROUP_CAP = 48;
template <int CAP>
struct P8GroupT {
  int n, total_tiles;
  P8Problem pr[CAP];
};
typedef P8GroupT<P8_GROUP_CAP> P8Group;
static_assert(sizeof(GemmParams) + sizeof(P8Group) <= 4096, "kernel arguments are limited to 4 KiB");

// EPI: -1 = every epilogue mode of the general-purpose form; SSAK_EPI_GELU_SAVE_GRAD / SSAK_EPI_MUL_AUX = the feed-forward pair
// (gemm_common.h: gemm_epilogue_direct), instantiated for the layouts the encoder uses them with.
template <int MH, bool A_KM, bool B_KM, bool GROUPED = false, int EPI = -1>
__global__ __launch_bounds__(P8_THREADS) void gemm_p8_kernel(const GemmParams p, const P8GroupT<GROUPED ? P8_GROUP_CAP : 1> grp) {
  constexpr int BM = 64 * MH, SEGA = 16 * MH;
  // (Balanced load phases for the 192-row form -- the right B fragments of K tile t+1 read in LB(t) into a second register set
  // and BR staged from LA, so that both load phases carry 2 MH + 4 fragment reads and 4 LDS-DMA instructions instead of
  // 2 MH + 8 / 2 and 2 MH / 6 -- were built in round 3, bit-identical, and are 3-10 % SLOWER: profiles/r03_gemm_balanced_probe.log.)
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef P8_STAMPS
  int p8_round = 0;  // stamps: [workgroup][tile round < 4][8]
#endif
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int per_z = p.tiles_m * p.tiles_n;
  const int ntiles = GROUPED ? grp.total_tiles : per_z * p.nz * p.split_k;
  const int nkt = (p.K + BK - 1) / BK;
  auto decode = [&](int t) {
    P8Tile c;
    const int id = xcd_remap(t, ntiles);
    if (GROUPED) {
      int gi = 0;
      for (int k = 1; k < grp.n; ++k) gi = id >= grp.pr[k].tile0 ? k : gi;
      const int rem = id - grp.pr[gi].tile0;
      c.g = gi;
      c.bm0 = rem / grp.pr[gi].tiles_n * BM;
      c.bn0 = rem % grp.pr[gi].tiles_n * 256;
      c.z = c.z1 = c.z2 = c.split = 0;
      c.kt0 = 0;
      c.kt1 = nkt;
      return c;
    }
    c.g = 0;
    const int zs = id / per_z, rem = id % per_z;
    // (Row-major over the tiles: an XCD's 32 concurrent workgroups then share 2.7 row panels x all column panels.  An 8 x 4
    // "supertile" walk for the wide feed-forward outputs -- the same 8 row panels revisited with the next 4 column panels each
    // round, so that activations are fetched once per XCD and only a third of the weight matrix is live per round -- was
    // measured in round 3 and is SLOWER: feed-forward up 1 166 -> 1 268 us per step, dX 1 121 -> 1 160
    // (profiles/r03_ab_supertile_walk_slower.log): with 8 distinct row panels in flight per XCD a row panel is shared by 4
    // workgroups instead of 12.)
    const int tm = rem / p.tiles_n, tn = rem % p.tiles_n;
    c.split = zs % p.split_k;
    c.z = zs / p.split_k;
    c.z1 = c.z / p.nb2;
    c.z2 = c.z % p.nb2;
    c.bm0 = tm * BM;
    c.bn0 = tn * 256;
    c.kt0 = c.split * p.kt_per_split;
    c.kt1 = min(nkt, c.kt0 + p.kt_per_split);
    return c;
  };

  const int kpp = GROUPED ? 0 : p.kperm_p, kpn = GROUPED ? 0 : p.kperm_n2;  // K-tile visiting order (gemm_common.h)
  P8Stager<A_KM, SEGA, 2 * SEGA, A_KM ? 16 : 4 * MH, 2> sa;
  P8Stager<B_KM, 32, 64, 16, 4> sb;
  P8Frag<A_KM, MH> fra;
  P8Frag<B_KM, 2, true> frb;
  fra.init(wr * SEGA, lane);
  frb.init(wc * 32, lane);
  // piece slots of buffer b: AT = 0, AB = 1, BL = 2, BR = 3
  char* const buf0 = smem;
  char* const buf1 = smem + P8_BUF;
  // stage the first two K tiles of tile c (all eight piece slots): BL(0) AT(0) BR(0) AB(0) BL(1) AT(1) BR(1) AB(1)
  // bias of the wave's 64 columns: one more LDS-DMA at the head of the priming sequence (lanes 0-15, 16 B each; the rest
  // out of range = zeros).  A register load here would sit in the compiler's scoreboard across the persistent loop and
  // make it drain vmcnt -- stores included -- in the middle of the next tile's first K step.
  const __amdgpu_buffer_rsrc_t bias_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.bias, 0, p.bias ? (int)((((long)p.nb2 - 1) * p.bias_s2 + p.N) * 4) : 0, 0x00020000);
  char* const bias_lds = smem + P8_PIPE + wave * 1024;
  auto prime = [&](const P8Tile& c) {
    {
      const long col = c.z2 * p.bias_s2 + c.bn0 + wc * 64 + 4 * lane;
      const uint32_t o = lane < 16 ? (uint32_t)(col * 4) : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(bias_rsrc, (lds_void*)bias_lds, 16, o, 0, 0, 0);
    }
    if (GROUPED) {
      const P8Problem& q = grp.pr[c.g];
      sa.init(q.A, q.lda, c.bm0, q.M, c.kt0, q.ext_a);
      sb.init(q.B, q.ldb, c.bn0, q.N, c.kt0, q.ext_b);
    } else {
      sa.init(p.A + c.z1 * p.sa1 + c.z2 * p.sa2, p.lda, c.bm0, p.M, c.kt0, p.ext_a);
      sb.init(p.B + c.z1 * p.sb1 + c.z2 * p.sb2, p.ldb, c.bn0, p.N, c.kt0, p.ext_b);
    }
    sb.template issue<0>(buf0 + 2 * P8_PIECE, c.kt0, c.kt1, p.K, kpp, kpn);
    sa.template issue<0>(buf0 + 0 * P8_PIECE, c.kt0, c.kt1, p.K, kpp, kpn);
    sb.template issue<1>(buf0 + 3 * P8_PIECE, c.kt0, c.kt1, p.K, kpp, kpn);
    sa.template issue<1>(buf0 + 1 * P8_PIECE, c.kt0, c.kt1, p.K, kpp, kpn);
    sb.template issue<0>(buf1 + 2 * P8_PIECE, c.kt0 + 1, c.kt1, p.K, kpp, kpn);
    sa.template issue<0>(buf1 + 0 * P8_PIECE, c.kt0 + 1, c.kt1, p.K, kpp, kpn);
    sb.template issue<1>(buf1 + 3 * P8_PIECE, c.kt0 + 1, c.kt1, p.K, kpp, kpn);
    sa.template issue<1>(buf1 + 1 * P8_PIECE, c.kt0 + 1, c.kt1, p.K, kpp, kpn);
  };
  // vector-memory instructions the LDS-free epilogue issues per wave (stores only; vmcnt counts them like the DMA)
  // (bf16 rows: two 16-byte stores per 16-row group, 4 MH per wave; + a bf16 side output: 8 MH; + the one-byte factor codes of
  // GELU_SAVE_GRAD: one more store per group, 6 MH -- a count ABOVE the real one would let the wait below pass with primed
  // pieces still in flight)
  const int epi_vm = (p.split_k > 1 || p.out_f32 || (p.epilogue == SSAK_EPI_GELU && p.aux_out)) ? 8 * MH
                     : (p.epilogue == SSAK_EPI_GELU_SAVE_GRAD && p.aux_out)                      ? 6 * MH
                                                                                                 : 4 * MH;
  static_assert(EPI != P8_EPI_PLAIN_F32 || GROUPED, "the fp32 form is instantiated for the grouped weight gradients");
  const bool epi_early = !p.accumulate && p.epilogue != SSAK_EPI_MUL_GELU_GRAD && p.epilogue != SSAK_EPI_MUL_AUX && !p.colsum;  // epilogues that only store (a fixed count)

  // Persistent over tiles (the host launches one workgroup per CU).  The next tile's first two K tiles are put in
  // flight BEFORE the current tile's epilogue, and the epilogue's stores are left to drain under the next main loop:
  // vmcnt retires in issue order, so `vmcnt(stores of the epilogue)` at the top of the next tile means "the sixteen
  // priming DMAs have landed", and the first wait that must cover DMAs younger than the stores comes two K tiles later
  // (phase 4 of K tile 1).  Measured on K = 768 tiles (tools/probes/p8_probe.hip): every CU reaches its epilogue at the
  // same moment, the 32 MB burst of a round takes 2.7 us (bf16) to 8 us (GELU + saved pre-activation) to drain at HBM
  // speed, and the pipeline fill of a fresh workgroup costs another 2 us -- all of it used to be exposed.
  // Tile order.  Static (tile_ctr == null, the single-GPU default): workgroup b takes logical ids b, b + grid, ...
  // Dynamic (ssak_gemm_desc.dynamic_tiles; data-parallel runs): a workgroup draws every tile from a ticket counter, so that one which starts late -- its
  // CU held by another stream's kernel, e.g. the RCCL all-reduce of the previous layer's gradients -- simply takes fewer
  // tiles.  With the static stride such a workgroup ran its whole share after everyone else had finished
  // (tools/probes/hog_probe.hip: 32 busy CUs made these GEMMs 1.8x slower, not 1.14x).  There is one counter per XCD
  // (x = blockIdx.x & 7, ticket k -> logical id 8 k + x), which keeps xcd_remap's property that an XCD works through a
  // contiguous run of tiles (a single counter scattered the panels over all eight L2s: 16 % slower with nobody else on the
  // chip).  The ticket of the next tile is requested at the top of the current one (one lane, a returning atomic that
  // retires under the main loop, well away from the epilogue's store burst) and its value is only touched after the loop's
  // vmcnt(0), where it is published through LDS; only the very first ticket of a launch is waited for.  (Built with -amdgpu-atomic-optimizer-strategy=None: the optimizer's
  // readfirstlane broadcast would wait on the spot.)
  int* const tile_ctr = p.tile_ctr ? p.tile_ctr + (blockIdx.x & 7) : nullptr;
  int* const ticket_lds = reinterpret_cast<int*>(smem + P8_PIPE + 8 * 1024);  // written by one lane, read after the barrier
  int ticket = 0;  // lane 0 of wave 0: the ticket in flight (for the tile after the current one)
  int t_first = blockIdx.x;
  if (tile_ctr) {
    if (threadIdx.x == 0) *ticket_lds = atomicAdd(tile_ctr, 1);
    __syncthreads();
    t_first = 8 * __builtin_amdgcn_readfirstlane(*ticket_lds) + (int)(blockIdx.x & 7);
    __syncthreads();
  }
  bool primed = false;
  P8Tile cur_t = decode(min(t_first, ntiles - 1));
  for (int t = t_first; t < ntiles;) {
    const P8Tile c = cur_t;
    P8_STAMP(0);
    const bool was_primed = primed;
    if (!primed) {
      prime(c);
      wait_vmcnt<10>();  // BL(0), AT(0), BR(0) landed (this wave's share): the first load phase reads all three
    } else if (epi_vm == 8 * MH) {
      wait_vmcnt<8 * MH>();  // everything older than the previous epilogue's stores: the eight primed pieces
    } else if (epi_vm == 6 * MH) {
      wait_vmcnt<6 * MH>();
    } else {
      wait_vmcnt<4 * MH>();
    }
    P8_BARRIER();     // ... everyone's
    if (wr == 1) P8_BARRIER();  // wave row 1 runs one barrier behind wave row 0
    P8_STAMP(1);
    if (tile_ctr && threadIdx.x == 0) ticket = atomicAdd(tile_ctr, 1);  // next tile's ticket: retires under the main loop

    f32x4 acc[2 * MH][4];
#pragma unroll
    for (int i = 0; i < 2 * MH; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int kt0 = c.kt0, kt1 = c.kt1;
    bf16x8 fa[MH][2], fbl[2][2], fbr[2][2];
    for (int kt = kt0; kt < kt1; ++kt) {
      char* const cur = ((kt - kt0) & 1) ? buf1 : buf0;
      char* const nxt = ((kt - kt0) & 1) ? buf0 : buf1;
      const bool first = kt == kt0;
      const bool settled = was_primed && kt - kt0 < 2;
      // ---- LA
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fbl[j][kk] = frb.read(cur + 2 * P8_PIECE, j, kk);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fbr[j][kk] = frb.read(cur + 3 * P8_PIECE, j, kk);
      P8_FENCE();
#pragma unroll
      for (int i = 0; i < MH; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fa[i][kk] = fra.read(cur + 0 * P8_PIECE, i, kk);
      P8_FENCE();
      if (!first) sa.template issue<1>(nxt + 1 * P8_PIECE, kt + 1, kt1, p.K, kpp, kpn);  // AB(t+1)
      if (!settled) wait_vmcnt<8>();                                           // AB(t)
      __builtin_amdgcn_s_waitcnt(0xc07f);                                      // this wave's fragment reads are done
      P8_BARRIER();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < MH; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbl[j][kk], fa[i][kk], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbr[j][kk], fa[i][kk], acc[i][2 + j], 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
      P8_BARRIER();
      // ---- LB
#pragma unroll
      for (int i = 0; i < MH; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fa[i][kk] = fra.read(cur + 1 * P8_PIECE, i, kk);
      P8_FENCE();
      sa.template issue<0>(cur + 0 * P8_PIECE, kt + 2, kt1, p.K, kpp, kpn);  // AT(t+2)
      sb.template issue<0>(cur + 2 * P8_PIECE, kt + 2, kt1, p.K, kpp, kpn);  // BL(t+2)
      sb.template issue<1>(cur + 3 * P8_PIECE, kt + 2, kt1, p.K, kpp, kpn);  // BR(t+2)
      if (!(was_primed && first)) wait_vmcnt<8>();                 // AT(t+1), BL(t+1), BR(t+1)
      __builtin_amdgcn_s_waitcnt(0xc07f);
      P8_BARRIER();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < MH; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[MH + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbl[j][kk], fa[i][kk], acc[MH + i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[MH + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbr[j][kk], fa[i][kk], acc[MH + i][2 + j], 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
      P8_BARRIER();
    }
    if (wr == 0) P8_BARRIER();
    wait_vmcnt<0>();  // drain the trailing dummies before LDS is released
    P8_STAMP(2);
    if (tile_ctr && threadIdx.x == 0) *ticket_lds = ticket;

    __syncthreads();  // DMA drained in every wave, all fragment reads done: LDS is free
    const int t_next = tile_ctr ? 8 * __builtin_amdgcn_readfirstlane(*ticket_lds) + (int)(blockIdx.x & 7) : t + (int)gridDim.x;
    BiasRegs<4> bias_regs;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_lds + (16 * j + 4 * (lane >> 4)) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) bias_regs.v[j][r] = b4[r];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the slice is in registers before the next tile's DMA overwrites it
    P8_FENCE();
    // interior tile: every wave takes the LDS-free epilogue, so the next tile can start filling LDS right now
    GemmParams pe = p;  // the epilogue's view: in a grouped launch the output belongs to this tile's problem
    if (GROUPED) {
      pe.C = grp.pr[c.g].C;
      pe.ldc = grp.pr[c.g].ldc;
      pe.M = grp.pr[c.g].M;
      pe.N = grp.pr[c.g].N;
    }
    // "interior" = every wave can take the LDS-free epilogue (whole 256 columns; rows beyond M are masked per lane there)
    const bool interior = c.bn0 + 256 <= pe.N && epilogue_direct_ok(pe, c.bm0, c.bn0, 0, 0, BM, c.z1 * p.sc1 + c.z2 * p.sc2);
    primed = false;
    if (t_next < ntiles) {
      cur_t = decode(t_next);
      // early priming counts on the epilogue issuing exactly epi_vm stores per wave: only for tiles with all their rows
      if (interior && epi_early && c.bm0 + BM <= pe.M) {
        prime(cur_t);
        primed = true;
      }
    }
    if (interior) {
      gemm_epilogue_direct<2 * MH, EPI>(pe, acc, bias_regs, c.bm0, c.bn0, wr * 2 * SEGA, wc * 64, lane, c.z, c.z1, c.z2, c.split);
    } else {
      char* const lds_wave = smem + wave * 16384;
      gemm_epilogue<MH, 4>(pe, reinterpret_cast<f32x4(&)[MH][4]>(acc[0]), bias_regs, lds_wave, c.bm0, c.bn0, wr * 2 * SEGA, wc * 64, lane, c.z, c.z1, c.z2, c.split);
      gemm_epilogue<MH, 4>(pe, reinterpret_cast<f32x4(&)[MH][4]>(acc[MH]), bias_regs, lds_wave, c.bm0, c.bn0, wr * 2 * SEGA + SEGA, wc * 64, lane, c.z, c.z1, c.z2, c.split);
      __syncthreads();  // the transposition buffers are read out before the next tile's pieces overwrite them
    }
    P8_STAMP(3);
#ifdef P8_STAMPS
    __syncthreads();
    P8_STAMP(4);  // every wave has issued its epilogue
    wait_vmcnt<0>();
    __syncthreads();
    P8_STAMP(5);  // ... and its stores have drained
    p8_round = min(p8_round + 1, 3);
#endif
    t = t_next;
  }
  // the last workgroup out leaves the counters at zero for the next launch on this stream
  if (p.tile_ctr && threadIdx.x == 0) {
    // (agent-scope atomics only, no __threadfence(): a release fence here writes back the XCD's whole L2 -- the tiles just
    // stored -- once per workgroup, which cost 13 us per launch; the kernel boundary orders the reset before the next launch)
    if (atomicAdd(p.tile_ctr + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i) atomicExch(p.tile_ctr + i, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// "B-direct" form of the same kernel for products whose B operand is a WEIGHT (static between optimizer steps): B never
// enters LDS.  tools/probes/p8_loop.hip's ablations say why: the loop above is bound by LDS read bandwidth -- per 64-deep K
// tile a workgroup reads 192 KB of fragments (A: 16, B: 8 ds_read_b128 per wave) and takes 64 KB of LDS-DMA writes, 256 KB at
// 128 B/clk = the 2 048 cycles its 512 MFMAs take, and the first load phase of a K tile (16 reads per wave) alone exceeds the
// partner wave's 512 MFMA cycles; with the B fragment reads removed the same loop runs 13-26 % faster (8192^3: 1 194 -> 1 505
// TFLOP/s).  Here every wave loads its B fragments straight from global memory into registers, one K tile ahead, from a
// FRAGMENT-ORDERED copy of the weight (k_gemm_fragment_b: the 8 KB a wave column needs for one K tile are contiguous, one
// 1-KiB wave-instruction per fragment -- whole cache lines, the layout that made the direct positional convolution fast).  LDS
// then carries only A: four 32 KB stages (AT / AB pieces as above), 128 KB of fragment reads + 32 KB of DMA per K tile.
//   LA(t): read AT(t) | load B(t+1) groups 0, 1 -> the other register set | DMA AB(t+3) | vmcnt(8) | barrier | 8 MH MFMAs | barrier
//   LB(t): read AB(t) | load B(t+1) groups 2, 3                           | DMA AT(t+3) |          | barrier | 8 MH MFMAs | barrier
// (six vector-memory instructions per load phase: with all eight B loads in LA the four waves of a load phase queued 40 KiB
// behind the CU's one address unit -- longer than the partner's MFMA phase.)  One counted wait per K tile: everything older
// than the 8 youngest vector-memory instructions has landed = B(t) and every DMA issued before LB(t-1).  Hazards as above
// (waited in load phase w -> read in phase >= w + 1; a slot last read in load phase r is re-staged in load phase r + 1).
constexpr int P8BD_STAGE = 2 * P8_PIECE;  // AT + AB of one K tile

// B[n][k] (k-contiguous, ldb) or B[k][n] (b_km) -> frag[(cb * nkt + kt)][j][kk][lane][8]:  n = 64 cb + 16 j + (lane & 15),
// k = 64 kt + 32 kk + 8 (lane >> 4) + e; zeros beyond N / K; cb < nb64 (a multiple of 4: whole 256-column tiles).
// One workgroup per 8 KB block of one matrix of the batch: the 64 x 64 source tile goes through LDS (whole 128-byte lines
// in, 4 KB contiguous runs out; the K-major orientation is transposed on the way).
struct FragJob {
  const bf16* B;
  bf16* out;
  long ldb;
  int N, K, b_km, nkt, blk0;  // blk0: first workgroup of this matrix
};
constexpr int FRAG_JOBS_CAP = 80;
struct FragJobs {
  int n, total;
  FragJob j[FRAG_JOBS_CAP];
};
static_assert(sizeof(FragJobs) <= 4096, "kernel arguments are limited to 4 KiB");
__global__ __launch_bounds__(256) void gemm_fragment_b_kernel(const FragJobs jobs) {
  constexpr int PITCH = 72;  // elements: 144-byte rows keep 16-byte alignment and spread the banks
  __shared__ __attribute__((aligned(16))) bf16 tile[64 * PITCH];
  int ji = 0;
  for (int k = 1; k < jobs.n; ++k) ji = (int)blockIdx.x >= jobs.j[k].blk0 ? k : ji;
  const FragJob& J = jobs.j[ji];
  const int blk = (int)blockIdx.x - J.blk0;
  const int kt = blk % J.nkt, cb = blk / J.nkt;
  const int n0 = 64 * cb, k0 = 64 * kt;
  const bool aligned = ((uintptr_t)J.B & 15) == 0 && (J.ldb & 7) == 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (int)threadIdx.x / 8 + 32 * i, c8 = ((int)threadIdx.x & 7) * 8;  // tile row, first element of the 16-byte chunk
    // kc: row = n, chunk along k;  km: row = k, chunk along n
    const int gr = (J.b_km ? k0 : n0) + r, gc = (J.b_km ? n0 : k0) + c8;
    const int rows = J.b_km ? J.K : J.N, cols = J.b_km ? J.N : J.K;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
    if (gr < rows) {
      const bf16* src = J.B + (long)gr * J.ldb + gc;
      if (gc + 8 <= cols && aligned) {
        v = *reinterpret_cast<const bf16x8*>(src);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (gc + e < cols) v[e] = src[e];
      }
    }
    *reinterpret_cast<bf16x8*>(tile + r * PITCH + c8) = v;
  }
  __syncthreads();
  bf16* const dst = J.out + (long)blk * 4096;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (int)threadIdx.x + 256 * i;  // 16-byte chunk of the block
    const int lane = c & 63, kk = (c >> 6) & 1, jj = c >> 7;
    const int nl = 16 * jj + (lane & 15), kl = 32 * kk + 8 * (lane >> 4);
    bf16x8 v;
    if (!J.b_km) {
      v = *reinterpret_cast<const bf16x8*>(tile + nl * PITCH + kl);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[(kl + e) * PITCH + nl];
    }
    *reinterpret_cast<bf16x8*>(dst + c * 8) = v;
  }
}

template <int MH, int EPI>
__global__ __launch_bounds__(P8_THREADS) void gemm_p8bd_kernel(const GemmParams p) {
  constexpr int BM = 64 * MH, SEGA = 16 * MH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int per_z = p.tiles_m * p.tiles_n;
  const int ntiles = per_z * p.nz;
  const int nkt = (p.K + BK - 1) / BK;
  auto decode = [&](int t) {
    P8Tile c;
    const int id = xcd_remap(t, ntiles);
    c.g = 0;
    const int zs = id / per_z, rem = id % per_z;
    const int tm = rem / p.tiles_n, tn = rem % p.tiles_n;
    c.split = 0;
    c.z = zs;
    c.z1 = c.z / p.nb2;
    c.z2 = c.z % p.nb2;
    c.bm0 = tm * BM;
    c.bn0 = tn * 256;
    c.kt0 = 0;
    c.kt1 = nkt;
    return c;
  };
  const int kpp = p.kperm_p, kpn = p.kperm_n2;
  P8Stager<false, SEGA, 2 * SEGA, 4 * MH, 2> sa;
  P8Frag<false, MH> fra;
  fra.init(wr * SEGA, lane);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)p.ext_b, 0x00020000);
  const int b_voff = lane * 16;
  uint32_t b_col = 0;  // byte offset of this wave column's first K tile in the fragment-ordered B
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
  // fragments of column groups 2 * HALF, 2 * HALF + 1 (HALF = 2: all four) of the K tile visited at `step`
  auto load_b = [&](bf16x8(&dst)[4][2], int step, auto half_tag) {
    constexpr int HALF = decltype(half_tag)::value;
#if defined(P8BD_ABL) && (P8BD_ABL & 1)
    if (step >= 2) return;  // ablation: no B loads inside the main loop
#endif
    int kta = step < 2 * kpn ? (step >> 1) + ((step & 1) ? kpp : 0) : step - kpn;  // K tile of this step (P8Stager::issue's order)
    kta = min(kta, nkt - 1);                                                         // past the end: a valid tile nobody uses
    const uint32_t so = b_col + (uint32_t)kta * 8192u;
#pragma unroll
    for (int j = (HALF == 1 ? 2 : 0); j < (HALF == 0 ? 2 : 4); ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int f = j * 2 + kk;
        const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_voff + (f & 3) * 1024, so + (f >> 2) * 4096, 0);
        dst[j][kk] = __builtin_bit_cast(bf16x8, v);
      }
  };
  using Half0 = std::integral_constant<int, 0>;
  using Half1 = std::integral_constant<int, 1>;
  using Both = std::integral_constant<int, 2>;
  const __amdgpu_buffer_rsrc_t bias_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.bias, 0, p.bias ? (int)((((long)p.nb2 - 1) * p.bias_s2 + p.N) * 4) : 0, 0x00020000);
  char* const bias_lds = smem + P8_PIPE + wave * 1024;
  // (B loaded TWO K tiles ahead into a ring of three register sets -- to cover the HBM latency of weights that are read once per
  // train step -- was built for the 192-row form and is slower, warm or cold: 29 priming instructions and 96 fragment registers;
  // profiles/r03_gemm_bdirect_cold2.log.)
  bf16x8 fb0[4][2], fb1[4][2];
  // bias, AT(0), AB(0), B(0) -> fb0, AT(1), AB(1), AT(2), AB(2): 21 vector-memory instructions per wave
  auto prime = [&](const P8Tile& c) {
    {
      const long col = c.z2 * p.bias_s2 + c.bn0 + wc * 64 + 4 * lane;
      const uint32_t o = lane < 16 ? (uint32_t)(col * 4) : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(bias_rsrc, (lds_void*)bias_lds, 16, o, 0, 0, 0);
    }
    sa.init(p.A + c.z1 * p.sa1 + c.z2 * p.sa2, p.lda, c.bm0, p.M, 0, p.ext_a);
    b_col = (uint32_t)(((long)(c.bn0 >> 6) + wc) * nkt) * 8192u;
    sa.template issue<0>(smem + 0 * P8_PIECE, 0, nkt, p.K, kpp, kpn);
    sa.template issue<1>(smem + 1 * P8_PIECE, 0, nkt, p.K, kpp, kpn);
    load_b(fb0, 0, Both{});
    sa.template issue<0>(smem + P8BD_STAGE + 0 * P8_PIECE, 1, nkt, p.K, kpp, kpn);
    sa.template issue<1>(smem + P8BD_STAGE + 1 * P8_PIECE, 1, nkt, p.K, kpp, kpn);
    sa.template issue<0>(smem + 2 * P8BD_STAGE + 0 * P8_PIECE, 2, nkt, p.K, kpp, kpn);
    sa.template issue<1>(smem + 2 * P8BD_STAGE + 1 * P8_PIECE, 2, nkt, p.K, kpp, kpn);
  };
  const int epi_vm = (p.epilogue == SSAK_EPI_GELU && p.aux_out) ? 8 * MH : (p.epilogue == SSAK_EPI_GELU_SAVE_GRAD && p.aux_out) ? 6 * MH : 4 * MH;
  const bool epi_early = !p.accumulate && p.epilogue != SSAK_EPI_MUL_GELU_GRAD && p.epilogue != SSAK_EPI_MUL_AUX && !p.colsum;

  int* const tile_ctr = p.tile_ctr ? p.tile_ctr + (blockIdx.x & 7) : nullptr;
  int* const ticket_lds = reinterpret_cast<int*>(smem + P8_PIPE + 8 * 1024);
  int ticket = 0;
  int t_first = blockIdx.x;
  if (tile_ctr) {
    if (threadIdx.x == 0) *ticket_lds = atomicAdd(tile_ctr, 1);
    __syncthreads();
    t_first = 8 * __builtin_amdgcn_readfirstlane(*ticket_lds) + (int)(blockIdx.x & 7);
    __syncthreads();
  }
  bool primed = false;
  P8Tile cur_t = decode(min(t_first, ntiles - 1));
  for (int t = t_first; t < ntiles;) {
    const P8Tile c = cur_t;
    const bool was_primed = primed;
    if (!primed) {
      prime(c);
      wait_vmcnt<18>();  // AT(0) landed (this wave's share): AB(0), B(0) and the six pieces of tiles 1, 2 may still fly
    } else if (epi_vm == 8 * MH) {
      wait_vmcnt<8 * MH>();  // everything older than the previous epilogue's stores
    } else if (epi_vm == 6 * MH) {
      wait_vmcnt<6 * MH>();
    } else {
      wait_vmcnt<4 * MH>();
    }
    P8_BARRIER();
    if (wr == 1) P8_BARRIER();  // wave row 1 runs one barrier behind wave row 0
    if (tile_ctr && threadIdx.x == 0) ticket = atomicAdd(tile_ctr, 1);

    f32x4 acc[2 * MH][4];
#pragma unroll
    for (int i = 0; i < 2 * MH; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16x8 fa[MH][2];
    // one K tile: `cur` holds B(step), `nxt` receives B(step + 1)
    auto ktile = [&](bf16x8(&cur)[4][2], bf16x8(&nxt)[4][2], int step) {
      char* const st_cur = smem + (step & 3) * P8BD_STAGE;
      char* const st_new = smem + ((step + 3) & 3) * P8BD_STAGE;
      // ---- LA
#pragma unroll
      for (int i = 0; i < MH; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fa[i][kk] = fra.read(st_cur, i, kk);
      P8_FENCE();
      load_b(nxt, step + 1, Half0{});
      sa.template issue<1>(st_new + P8_PIECE, step + 3, nkt, p.K, kpp, kpn);  // AB(t+3)
      // B(t) and everything older: B(t)'s second half was loaded in LB(t-1), 2 + 6 instructions ago (K tile 0: in the priming
      // sequence, followed by AB(2)).  After early priming the previous tile's epilogue stores sit in between and stay in flight.
      if (step == 0 && was_primed) {
        if (epi_vm == 8 * MH) wait_vmcnt<8 + 8 * MH>();
        else wait_vmcnt<8 + 4 * MH>();
      } else {
#if defined(P8BD_ABL) && (P8BD_ABL & 1)
        wait_vmcnt<4>();
#else
        wait_vmcnt<8>();
#endif
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);  // this wave's fragment reads are done
      P8_BARRIER();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < MH; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[j][kk], fa[i][kk], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      P8_BARRIER();
      // ---- LB
#pragma unroll
      for (int i = 0; i < MH; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fa[i][kk] = fra.read(st_cur + P8_PIECE, i, kk);
      P8_FENCE();
      // (the loads first: LA(t+1)'s wait reaches back to them and must not force a DMA issued half a K tile before it)
      load_b(nxt, step + 1, Half1{});
      sa.template issue<0>(st_new, step + 3, nkt, p.K, kpp, kpn);  // AT(t+3)
      __builtin_amdgcn_s_waitcnt(0xc07f);
      P8_BARRIER();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < MH; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[MH + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[j][kk], fa[i][kk], acc[MH + i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      P8_BARRIER();
    };
    int step = 0;
    for (; step + 1 < nkt; step += 2) {
      ktile(fb0, fb1, step);
      ktile(fb1, fb0, step + 1);
    }
    if (step < nkt) ktile(fb0, fb1, step);
    if (wr == 0) P8_BARRIER();
    wait_vmcnt<0>();  // drain the trailing dummies before LDS is released
    // the last K tile's B loads (a clamped tile nobody multiplies) must stay in the instruction stream: the counted waits
    // above count them
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) asm volatile("" ::"v"(fb0[j][kk]), "v"(fb1[j][kk]));
    if (tile_ctr && threadIdx.x == 0) *ticket_lds = ticket;

    __syncthreads();
    const int t_next = tile_ctr ? 8 * __builtin_amdgcn_readfirstlane(*ticket_lds) + (int)(blockIdx.x & 7) : t + (int)gridDim.x;
    BiasRegs<4> bias_regs;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_lds + (16 * j + 4 * (lane >> 4)) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) bias_regs.v[j][r] = b4[r];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    P8_FENCE();
    const bool interior = c.bn0 + 256 <= p.N && epilogue_direct_ok(p, c.bm0, c.bn0, 0, 0, BM, c.z1 * p.sc1 + c.z2 * p.sc2);
    primed = false;
    if (t_next < ntiles) {
      cur_t = decode(t_next);
      if (interior && epi_early && c.bm0 + BM <= p.M) {
        prime(cur_t);
        primed = true;
      }
    }
    if (interior) {
      gemm_epilogue_direct<2 * MH, EPI>(p, acc, bias_regs, c.bm0, c.bn0, wr * 2 * SEGA, wc * 64, lane, c.z, c.z1, c.z2, 0);
    } else {
      char* const lds_wave = smem + wave * 16384;
      gemm_epilogue<MH, 4>(p, reinterpret_cast<f32x4(&)[MH][4]>(acc[0]), bias_regs, lds_wave, c.bm0, c.bn0, wr * 2 * SEGA, wc * 64, lane, c.z, c.z1, c.z2, 0);
      gemm_epilogue<MH, 4>(p, reinterpret_cast<f32x4(&)[MH][4]>(acc[MH]), bias_regs, lds_wave, c.bm0, c.bn0, wr * 2 * SEGA + SEGA, wc * 64, lane, c.z, c.z1, c.z2, 0);
      __syncthreads();
    }
    t = t_next;
  }
  if (p.tile_ctr && threadIdx.x == 0) {
    if (atomicAdd(p.tile_ctr + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i) atomicExch(p.tile_ctr + i, 0);
    }
  }
}

// Ticket counters (eight per-XCD counters + the count of finished workgroups): one 64-byte slot per (device, stream).  Launches on a stream run in order and every launch leaves its slot
// at zero, so a slot is reused without a reset; different streams never share one.
int p8_ticket_slot(hipStream_t st, int** out) {
  struct Key {
    int dev;
    hipStream_t st;
    bool operator<(const Key& o) const { return dev != o.dev ? dev < o.dev : st < o.st; }
  };
  static std::mutex mu;
  static std::map<Key, int*> slots;
  static std::map<int, std::pair<char*, int>> pools;  // device -> (zeroed pool, slots used)
  constexpr int POOL_SLOTS = 1024;
  int dev = 0;
  SSAK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto it = slots.find(Key{dev, st});
  if (it == slots.end()) {
    auto& pool = pools[dev];
    if (!pool.first) {
      SSAK_HIP(hipMalloc((void**)&pool.first, POOL_SLOTS * 64));
      SSAK_HIP(hipMemset(pool.first, 0, POOL_SLOTS * 64));
      pool.second = 0;
    }
    if (pool.second >= POOL_SLOTS) {  // more streams than slots: fall back to the static order on the extra ones
      *out = nullptr;
      return SSAK_OK;
    }
    it = slots.emplace(Key{dev, st}, reinterpret_cast<int*>(pool.first + 64 * pool.second++)).first;
  }
  *out = it->second;
  return SSAK_OK;
}

int p8_num_cu(int* out) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    SSAK_HIP(hipGetDevice(&dev));
    SSAK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  }
  *out = n_cu;
  return SSAK_OK;
}

template <int MH, bool A_KM, bool B_KM, int EPI = -1>
int launch_p8(const GemmParams& p, hipStream_t st) {
  auto kern = gemm_p8_kernel<MH, A_KM, B_KM, false, EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    SSAK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS));
    attr_done = true;
  }
  const long ntiles = (long)p.tiles_m * p.tiles_n * p.nz * p.split_k;
  int n_cu = 0;
  if (int rc = p8_num_cu(&n_cu)) return rc;
  P8GroupT<1> none;
  none.n = 0;
  none.total_tiles = 0;
  GemmParams q = p;
  q.tile_ctr = nullptr;
  if (ntiles > n_cu && n_cu % 8 == 0 && p.dynamic)
    if (int rc = p8_ticket_slot(st, &q.tile_ctr)) return rc;
  // one timing slot per (instantiation, N, K): an instantiation serves several products of the step (qkv, attention output and
  // feed-forward-down projections share one), and "algorithmic FLOPs per launch" only means something per product
  static std::mutex slot_mu;
  static std::map<std::pair<int, int>, int> slots;
  int slot;
  {
    std::lock_guard<std::mutex> lock(slot_mu);
    auto it = slots.find({p.N, p.K});
    if (it == slots.end()) {
      char nm[112];
      snprintf(nm, sizeof(nm), "gemm_p8_kernel<%d, %s, %s, false, %d> (N = %d, K = %d)", MH, A_KM ? "true" : "false", B_KM ? "true" : "false", EPI,
               p.N, p.K);
      it = slots.emplace(std::make_pair(p.N, p.K), ssak_prof_register(nm, SSAK_BOUND_MFMA)).first;
    }
    slot = it->second;
  }
  ProfScope prof_scope(slot, 2.0 * p.M * p.N * (double)p.K * p.nz, st);
  kern<<<dim3((unsigned)std::min<long>(ntiles, n_cu)), P8_THREADS, P8_LDS, st>>>(q, none);  // one persistent workgroup per CU
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <int MH, int EPI>
int launch_p8bd(const GemmParams& p, hipStream_t st) {
  auto kern = gemm_p8bd_kernel<MH, EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    SSAK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS));
    attr_done = true;
  }
  const long ntiles = (long)p.tiles_m * p.tiles_n * p.nz;
  int n_cu = 0;
  if (int rc = p8_num_cu(&n_cu)) return rc;
  GemmParams q = p;
  q.tile_ctr = nullptr;
  if (ntiles > n_cu && n_cu % 8 == 0 && p.dynamic)
    if (int rc = p8_ticket_slot(st, &q.tile_ctr)) return rc;
  static std::mutex slot_mu;
  static std::map<std::pair<int, int>, int> slots;  // one timing slot per (N, K), as launch_p8
  int slot;
  {
    std::lock_guard<std::mutex> lock(slot_mu);
    auto it = slots.find({p.N, p.K});
    if (it == slots.end()) {
      char nm[112];
      snprintf(nm, sizeof(nm), "gemm_p8bd_kernel<%d, %d> (N = %d, K = %d)", MH, EPI, p.N, p.K);
      it = slots.emplace(std::make_pair(p.N, p.K), ssak_prof_register(nm, SSAK_BOUND_MFMA)).first;
    }
    slot = it->second;
  }
  ProfScope prof_scope(slot, 2.0 * p.M * p.N * (double)p.K * p.nz, st);
  kern<<<dim3((unsigned)std::min<long>(ntiles, n_cu)), P8_THREADS, P8_LDS, st>>>(q);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <int MH>
int dispatch_p8(const GemmParams& p, int a_km, int b_km, hipStream_t st) {
  // the feed-forward pair: specialised instantiations for the layouts the encoder layers run them with (forward: both
  // operands K-contiguous; backward dX: the weight read K-major); ssak_gemm_bf16 keeps other layouts off this kernel
  if (p.epilogue == SSAK_EPI_GELU_SAVE_GRAD) {
    if (!a_km && !b_km) return launch_p8<MH, false, false, SSAK_EPI_GELU_SAVE_GRAD>(p, st);
    ssak_set_error("gemm_p8: GELU_SAVE_GRAD is built for K-contiguous operands");
    return SSAK_ERR_INVALID;
  }
  if (p.epilogue == SSAK_EPI_MUL_AUX) {
    if (!a_km && b_km) return launch_p8<MH, false, true, SSAK_EPI_MUL_AUX>(p, st);
    if (!a_km && !b_km) return launch_p8<MH, false, false, SSAK_EPI_MUL_AUX>(p, st);  // B = a transposed weight copy
    ssak_set_error("gemm_p8: MUL_AUX is built for a K-contiguous A");
    return SSAK_ERR_INVALID;
  }
  // lean forms of the common modes (gemm_common.h): no dropout, no column sums, no split-K slabs
  const bool no_extras = !p.drop_thresh && !p.colsum && p.split_k <= 1;
  if (MH >= 3 && no_extras && !a_km) {
    if (p.epilogue == SSAK_EPI_NONE && !p.out_f32 && !p.accumulate)
      return b_km ? launch_p8<MH, false, true, P8_EPI_PLAIN_BF16>(p, st) : launch_p8<MH, false, false, P8_EPI_PLAIN_BF16>(p, st);
    if (p.epilogue == SSAK_EPI_GELU && !p.aux_out && !p.out_f32 && !b_km) return launch_p8<MH, false, false, P8_EPI_GELU_ONLY>(p, st);
  }
  if (!a_km && !b_km) return launch_p8<MH, false, false>(p, st);
  if (!a_km && b_km) return launch_p8<MH, false, true>(p, st);
  if (a_km && b_km) return launch_p8<MH, true, true>(p, st);
  return launch_p8<MH, true, false>(p, st);
}

template <bool A_KM, bool B_KM, int EPI = -1>
int launch_p8_grouped(const GemmParams& p, const P8Group& grp, hipStream_t st) {
  auto kern = gemm_p8_kernel<4, A_KM, B_KM, true, EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    SSAK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS));
    attr_done = true;
  }
  int n_cu = 0;
  if (int rc = p8_num_cu(&n_cu)) return rc;
  GemmParams q = p;
  q.tile_ctr = nullptr;
  if (grp.total_tiles > n_cu && n_cu % 8 == 0 && p.dynamic)
    if (int rc = p8_ticket_slot(st, &q.tile_ctr)) return rc;
  static int slot = -1;
  if (slot < 0) {
    char nm[112];
    snprintf(nm, sizeof(nm), "gemm_p8_kernel<4, %s, %s, true, %d>", A_KM ? "true" : "false", B_KM ? "true" : "false", EPI);
    slot = ssak_prof_register(nm, SSAK_BOUND_MFMA);
  }
  double flops = 0.0;
  for (int i = 0; i < grp.n; ++i) flops += 2.0 * grp.pr[i].M * grp.pr[i].N * (double)p.K;
  ProfScope prof_scope(slot, flops, st);
  kern<<<dim3((unsigned)std::min(grp.total_tiles, n_cu)), P8_THREADS, P8_LDS, st>>>(q, grp);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

}  // namespace

int ssak_gemm_ticket_slot(hipStream_t st, int** out) { return p8_ticket_slot(st, out); }

int ssak_gemm_p8_launch(const void* params, int bm, int a_km, int b_km, hipStream_t st) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  if (bm == 256) return dispatch_p8<4>(p, a_km, b_km, st);
  if (bm == 192) return dispatch_p8<3>(p, a_km, b_km, st);
  return dispatch_p8<2>(p, a_km, b_km, st);
}

// B-direct form (B = the fragment-ordered copy made by k_gemm_fragment_b; A K-contiguous, bf16 out, no split-K).  The caller
// (gemm.hip) has checked the shape; tiles_m / tiles_n are counted for bm x 256 tiles.
template <int MH>
int dispatch_p8bd(const GemmParams& p, hipStream_t st) {
  const bool plain = p.epilogue == SSAK_EPI_NONE && !p.drop_thresh && !p.colsum && !p.out_f32 && !p.accumulate && p.split_k <= 1;
  if (plain) return launch_p8bd<MH, P8_EPI_PLAIN_BF16>(p, st);
  ssak_set_error("gemm_p8bd: only the plain bf16 epilogue (bias) is built for the fragment-ordered B form (epilogue %d)", p.epilogue);
  return SSAK_ERR_INVALID;
}
int ssak_gemm_p8bd_launch(const void* params, int bm, hipStream_t st) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  if (bm == 256) return dispatch_p8bd<4>(p, st);
  if (bm == 192) return dispatch_p8bd<3>(p, st);
  return dispatch_p8bd<2>(p, st);
}
size_t k_gemm_fragment_b_bytes(int N, int K) { return (size_t)ssak_cdiv(N, 256) * 4 * ssak_cdiv(K, BK) * 8192; }
int k_gemm_fragment_b_batched(int n, const void* const* B, const long* ldb, const int* N, const int* K, const int* b_km, void* const* out,
                              hipStream_t st) {
  for (int i0 = 0; i0 < n; i0 += FRAG_JOBS_CAP) {
    FragJobs jobs;
    jobs.n = std::min(n - i0, FRAG_JOBS_CAP);
    int blk = 0;
    for (int i = 0; i < jobs.n; ++i) {
      FragJob& J = jobs.j[i];
      SSAK_REQUIRE(B[i0 + i] && out[i0 + i] && N[i0 + i] > 0 && K[i0 + i] > 0, "gemm_fragment_b: bad matrix %d", i0 + i);
      SSAK_REQUIRE(((uintptr_t)out[i0 + i] & 15) == 0, "gemm_fragment_b: the copy must be 16-byte aligned");
      J.B = (const bf16*)B[i0 + i];
      J.out = (bf16*)out[i0 + i];
      J.ldb = ldb[i0 + i];
      J.N = N[i0 + i];
      J.K = K[i0 + i];
      J.b_km = b_km[i0 + i];
      J.nkt = ssak_cdiv(J.K, BK);
      J.blk0 = blk;
      blk += ssak_cdiv(J.N, 256) * 4 * J.nkt;
    }
    for (int i = jobs.n; i < FRAG_JOBS_CAP; ++i) jobs.j[i] = jobs.j[0];
    jobs.total = blk;
    gemm_fragment_b_kernel<<<blk, 256, 0, st>>>(jobs);
    SSAK_LAUNCH_CHECK();
  }
  return SSAK_OK;
}

// Grouped launch of n <= 48 problems sharing K, layouts, alpha and output type (256-row tiles).  `params` carries the shared
// fields (K, alpha, out_f32, accumulate, ...); A/B/C, M/N, leading dimensions and extents come per problem.
int ssak_gemm_p8_launch_grouped(const void* params, int n, const void* const* A, const void* const* B, void* const* C, const int* M,
                                const int* N, const long* lda, const long* ldb, const long* ldc, const uint32_t* ext_a,
                                const uint32_t* ext_b, int a_km, int b_km, hipStream_t st) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  P8Group grp;
  grp.n = n;
  int t0 = 0;
  for (int i = 0; i < n; ++i) {
    P8Problem& q = grp.pr[i];
    q.A = (const bf16*)A[i];
    q.B = (const bf16*)B[i];
    q.C = C[i];
    q.lda = lda[i];
    q.ldb = ldb[i];
    q.ldc = ldc[i];
    q.M = M[i];
    q.N = N[i];
    q.ext_a = ext_a[i];
    q.ext_b = ext_b[i];
    q.tile0 = t0;
    q.tiles_n = ssak_cdiv(N[i], 256);
    t0 += ssak_cdiv(M[i], 256) * q.tiles_n;
  }
  for (int i = n; i < P8_GROUP_CAP; ++i) grp.pr[i] = grp.pr[0];
  grp.total_tiles = t0;
  if (!a_km && !b_km) return launch_p8_grouped<false, false>(p, grp, st);
  if (!a_km && b_km) return launch_p8_grouped<false, true>(p, grp, st);
  if (a_km && b_km) {
    if (p.out_f32 && p.epilogue == SSAK_EPI_NONE) return launch_p8_grouped<true, true, P8_EPI_PLAIN_F32>(p, grp, st);  // weight gradients
    return launch_p8_grouped<true, true>(p, grp, st);
  }
  return launch_p8_grouped<true, false>(p, grp, st);
}
