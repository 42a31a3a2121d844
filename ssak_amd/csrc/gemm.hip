// Batched bf16 MFMA GEMM with fp32 accumulation: the one dense-contraction kernel behind every Linear,
// channels-last Conv1d (overlapping-row "Toeplitz" A operand), attention product and weight gradient of
// the Wav2Vec2 / Whisper encoder path.  gfx950 only.
//
//   C[z][m][n] = alpha * sum_k A(z; m,k) * B(z; n,k)   (+ bias[n], GELU, ...)
//
// Operand storage is chosen per operand: K-contiguous ([rows][K], an nn.Linear weight or a row-major
// activation) or K-major ([K][rows], what a transposed product -- dW = dY^T X, P^T dO -- reads).  Both are
// staged global -> VGPR -> LDS (16-byte chunks, zero-filled at every edge) and double-buffered with one
// workgroup barrier per 64-deep K step; K-contiguous tiles are XOR-swizzled and read with ds_read_b128,
// K-major tiles are row-padded and read with ds_read_b64_tr_b16 (the hardware transpose read), so no
// operand is ever transposed in HBM.  4 waves per workgroup, each owning a (BM/WM)x(BN/WN) sub-tile of
// 16x16x32 bf16 MFMAs.  The MFMA is issued with the operands swapped so each lane ends up with 4
// consecutive output columns (8-/16-byte stores).  Workgroup ids are remapped so that each XCD's L2 sees a
// contiguous run of tiles sharing the same A row panel.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include <cstdio>
#include "common.h"
#include "gemm_common.h"
#include "kernels.h"

namespace {


// ---- LDS tile geometry -------------------------------------------------------------------------
// K-contiguous tile: R rows x 128 B, 16-B chunk c of row r stored at chunk c ^ ((r >> 1) & 7)
//   (16 consecutive rows at one logical chunk land on 16 distinct 16-B slots of the 256-B bank row).
// K-major tile: 64 k-rows x (2R + 32) B; the 32-B pad makes 8 consecutive k-rows cover all 64 banks.
template <int R, bool KM>
struct Tile {
  static constexpr int STRIDE = KM ? (2 * R + 32) : 128;
  static constexpr int BYTES = KM ? BK * STRIDE : R * 128;
  static constexpr int NCHUNK = R * 8 / NTHREADS;  // 16-B chunks per thread per tile
};

__device__ __forceinline__ uint4 mask_chunk(uint4 v, int nvalid) {
  // keep the first nvalid (0..8) bf16 of a 16-B chunk
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = nvalid - 2 * i;
    w[i] = (r >= 2) ? w[i] : (r == 1 ? (w[i] & 0xffffu) : 0u);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

template <int N>
struct Regs {
  uint4 v[N];
};

// Per-thread staging state of one operand tile, set up once per workgroup: where each of this thread's
// 16-byte chunks comes from in global memory (for k-tile 0), where it goes in LDS, and how many of its 8
// elements are inside the matrix along the non-K axis.  Inside the K loop a chunk's source only advances by a
// constant, every load is issued unconditionally (out-of-range chunks read a safe in-range address and are
// zeroed later), and the zero-fill masking is applied when the registers are written to LDS -- one K step after
// the loads were issued, and only on edge tiles (workgroup-uniform branch), so the loads stay in flight together.
template <int R, bool KM>
struct Stager {
  static constexpr int N = Tile<R, KM>::NCHUNK;
  const char* base;     // workgroup-uniform operand base (batch offset folded in): loads use base + 32-bit offset
  uint32_t off[N];      // byte offset of the chunk in the CURRENT k-tile (advanced by kstep after every load)
  int lds_off[N];
  int kofs[N];          // KC: k offset of the chunk inside the k-tile (c*8); KM: k-row inside the k-tile
  int nrow[N];          // KC: 8 if the row is inside the matrix else 0; KM: valid elements along the row axis (0..8)
  uint32_t kstep;       // byte stride of one k-tile
  bool rows_full;       // every chunk of this tile is fully inside along the non-K axis

  __device__ __forceinline__ void init(const bf16* base_, long ld, int row0, int rows_total, int kt0) {
    base = reinterpret_cast<const char*>(base_);
    rows_full = true;
    kstep = (uint32_t)((KM ? (long)BK * ld : (long)BK) * 2);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int q = threadIdx.x + i * NTHREADS;
      if (!KM) {
        const int r = q >> 3, c = q & 7;
        const int gr = row0 + r;
        nrow[i] = gr < rows_total ? 8 : 0;
        kofs[i] = c * 8;
        off[i] = (uint32_t)(((long)(gr < rows_total ? gr : 0) * ld + c * 8) * 2);
        lds_off[i] = r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
      } else {
        constexpr int CPR = R / 8;
        const int kr = q / CPR, c = q % CPR;
        const int gr = row0 + c * 8;
        nrow[i] = min(max(rows_total - gr, 0), 8);
        kofs[i] = kr;
        off[i] = (uint32_t)(((long)kr * ld + (nrow[i] > 0 ? gr : 0)) * 2);
        lds_off[i] = kr * Tile<R, KM>::STRIDE + c * 16;
      }
      off[i] += (uint32_t)kt0 * kstep;
      rows_full = rows_full && (nrow[i] == 8);
    }
    rows_full = __syncthreads_and(rows_full);
  }
  // valid elements of chunk i in k-tile starting at k0 (0..8)
  __device__ __forceinline__ int nvalid(int i, int k0, int k_end) const {
    if (!KM) return nrow[i] ? min(max(k_end - (k0 + kofs[i]), 0), 8) : 0;
    return (k0 + kofs[i] < k_end) ? nrow[i] : 0;
  }
  // issue the loads of k-tile kt (all unconditional; chunks with no valid element read offset 0) and advance
  __device__ __forceinline__ void load(Regs<N>& regs, int kt, int k_end) {
    const int k0 = kt * BK;
    if (rows_full && (k0 + BK <= k_end)) {  // workgroup-uniform fast path
#pragma unroll
      for (int i = 0; i < N; ++i) regs.v[i] = *reinterpret_cast<const uint4*>(base + off[i]);
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i)
        regs.v[i] = *reinterpret_cast<const uint4*>(base + (nvalid(i, k0, k_end) > 0 ? off[i] : 0u));
    }
#pragma unroll
    for (int i = 0; i < N; ++i) off[i] += kstep;
  }
  __device__ __forceinline__ void store(const Regs<N>& regs, char* lds, int kt, int k_end) const {
    const int k0 = kt * BK;
    if (rows_full && (k0 + BK <= k_end)) {
#pragma unroll
      for (int i = 0; i < N; ++i) *reinterpret_cast<uint4*>(lds + lds_off[i]) = regs.v[i];
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) *reinterpret_cast<uint4*>(lds + lds_off[i]) = mask_chunk(regs.v[i], nvalid(i, k0, k_end));
    }
  }
};

// Fragment addressing.  Lane l of a wave gets k = 8*(l>>4)+j (j = 0..7) of row r0 + (l&15) of the 16-row group.
// Everything lane-dependent is folded into ONE per-thread byte offset computed before the K loop (frag_lane_off);
// the 16-row group index i, the 32-deep k half kk and the stage are compile-time / uniform immediates:
//   K-contiguous: off = lane_off ^ (kk << 6)  + i * 2048          (the XOR swizzle only sees (row>>1)&7, and
//                                                                  16*i rows leave it unchanged)
//   K-major     : off = lane_off + kk * 32 * STRIDE + i * 32      (+ 4 * STRIDE for the upper four k)
template <int R, bool KM>
__device__ __forceinline__ int frag_lane_off(int w0, int lane) {
  if (!KM) {
    const int r = w0 + (lane & 15);
    return r * 128 + (((lane >> 4) ^ ((r >> 1) & 7)) << 4);
  } else {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    return (8 * g + q) * Tile<R, KM>::STRIDE + (w0 + 4 * p) * 2;
  }
}
template <int R, bool KM>
__device__ __forceinline__ bf16x8 read_frag(const char* lds, int lane_off, int i, int kk) {
  if (!KM) {
    return *reinterpret_cast<const bf16x8*>(lds + ((lane_off ^ (kk << 6)) + i * 2048));
  } else {
    const int off = lane_off + kk * 32 * Tile<R, KM>::STRIDE + i * 32;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off + 4 * Tile<R, KM>::STRIDE));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  }
}


template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const GemmParams p) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 16, NI = TN / 16;
  using TA = Tile<BM, A_KM>;
  using TB = Tile<BN, B_KM>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = TA::BYTES + TB::BYTES;  // stage s: A at s*STAGE, B at s*STAGE + TA::BYTES

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm0 = (wave / WN) * TM, wn0 = (wave % WN) * TN;

  const int per_z = p.tiles_m * p.tiles_n;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int zs = id / per_z, rem = id % per_z;
  const int tm = rem / p.tiles_n, tn = rem % p.tiles_n;
  const int split = zs % p.split_k, z = zs / p.split_k;
  const int z1 = z / p.nb2, z2 = z % p.nb2;
  const bf16* Ab = p.A + z1 * p.sa1 + z2 * p.sa2;
  const bf16* Bb = p.B + z1 * p.sb1 + z2 * p.sb2;
  const int bm0 = tm * BM, bn0 = tn * BN;

  const int nkt = (p.K + BK - 1) / BK;
  const int kt0 = split * p.kt_per_split;
  const int kt1 = min(nkt, kt0 + p.kt_per_split);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  BiasRegs<NI> bias_regs;
  load_bias<NI>(p, bn0, wn0, lane, z2, bias_regs);

  const int a_lane_off = frag_lane_off<BM, A_KM>(wm0, lane);
  const int b_lane_off = frag_lane_off<BN, B_KM>(wn0, lane);
  Regs<TA::NCHUNK> ra;
  Regs<TB::NCHUNK> rb;
  Stager<BM, A_KM> sa;
  Stager<BN, B_KM> sb;
  sa.init(Ab, p.lda, bm0, p.M, kt0);
  sb.init(Bb, p.ldb, bn0, p.N, kt0);
  if (kt0 < kt1) {
    sa.load(ra, kt0, p.K);
    sb.load(rb, kt0, p.K);
    sa.store(ra, smem, kt0, p.K);
    sb.store(rb, smem + TA::BYTES, kt0, p.K);
  }
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    const bool more = kt + 1 < kt1;
    if (more) {
      sa.load(ra, kt + 1, p.K);
      sb.load(rb, kt + 1, p.K);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = read_frag<BM, A_KM>(smem + cur * STAGE, a_lane_off, i, kk);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = read_frag<BN, B_KM>(smem + cur * STAGE + TA::BYTES, b_lane_off, j, kk);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    if (more) {
      sa.store(ra, smem + (cur ^ 1) * STAGE, kt + 1, p.K);
      sb.store(rb, smem + (cur ^ 1) * STAGE + TA::BYTES, kt + 1, p.K);
    }
    __syncthreads();
  }
  // every wave is past its last fragment read (the barrier above): LDS becomes the epilogue's transposition buffer
  gemm_epilogue<MI, NI>(p, acc, bias_regs, smem + wave * (1024 * MI * NI), bm0, bn0, wm0, wn0, lane, z, z1, z2, split);
}

// =================================================================================================
// LDS-DMA variant: operand tiles go global -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR staging, no
// ds_write: on gfx950 a ds_write_b128 costs ~13 cycles of the SIMD->LDS path per wave-instruction, which made the
// register-staged kernel LDS-store-bound).  An LDS-DMA wave-instruction writes 1 KiB linearly (M0 base + lane*16),
// so the bank-conflict swizzles are applied on the SOURCE side: lane L fetches the global chunk whose swizzled
// home is slot L.  Out-of-range chunks are given an offset beyond the buffer descriptor's extent: the DMA then
// writes zeros (measured: tools/probes/glds_oob.hip), which is the zero fill the edges need.  Partial 16-byte
// chunks cannot be masked here, so this kernel is selected only when no chunk is partial or the caller
// guarantees that padding elements are zero in memory (desc.pads_are_zero).
//   K-contiguous tile: [R rows][128 B], chunk c of row r at slot c ^ ((r >> 1) & 7)               (as above)
//   K-major tile     : [64 k-rows][2R B], chunk ch of k-row kr at slot ch ^ f(kr); f spreads the 8 k-rows one
//                      transpose-read touches over the whole 256-B bank row (conflict-free ds_read_b64_tr_b16)
// Two stages, ONE barrier per K step: wait own DMA (vmcnt 0) -> barrier -> issue next tile's DMA -> MFMA.

// K-contiguous tile swizzle (XOR on the 16-byte chunk index of row r): 128-byte rows (BKT 64) / 64-byte rows (BKT 32)
template <int BKT>
__device__ __forceinline__ int kc_swz(int r) {
  return BKT == 64 ? ((r >> 1) & 7) : ((r >> 2) & 2);  // both conflict-free for ds_read_b128 (searched exhaustively)
}

template <int R, bool KM, int BKT>
struct DmaStager {
  static constexpr int NINST = R * BKT / 2048;  // 1-KiB wave-instructions per wave per tile (tile = R*BKT*2 bytes, 4 waves)
  static constexpr uint32_t OOB = 0x80000000u;
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t off[NINST];   // byte offset of this lane's chunk in the current k-tile (OOB if its row is outside)
  int kofs[NINST];       // KC: first k of the chunk inside the k-tile; KM: k-row inside the k-tile
  uint32_t kstep;
  int wave;

  __device__ __forceinline__ void init(const bf16* base, long ld, int row0, int rows_total, int K, int kt0,
                                       uint32_t extent_bytes) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)extent_bytes, 0x00020000);
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    kstep = (uint32_t)((KM ? (long)BKT * ld : (long)BKT) * 2);
#pragma unroll
    for (int j = 0; j < NINST; ++j) {
      const int S = (wave * NINST + j) * 64 + lane;
      if (!KM) {
        constexpr int CPK = BKT / 8;  // chunks per row
        const int r = S / CPK, pc = S % CPK;
        const int c = pc ^ kc_swz<BKT>(r);
        const int gr = row0 + r;
        kofs[j] = c * 8;
        off[j] = gr < rows_total ? (uint32_t)(((long)gr * ld + c * 8) * 2) : OOB;
      } else {
        constexpr int CPR = R / 8;
        const int kr = S / CPR, pc = S % CPR;
        const int ch = pc ^ km_swz<R>(kr);
        const int gr = row0 + ch * 8;
        kofs[j] = kr;
        off[j] = gr < rows_total ? (uint32_t)(((long)kr * ld + gr) * 2) : OOB;
      }
      if (off[j] != OOB) off[j] += (uint32_t)kt0 * kstep;
    }
  }
  __device__ __forceinline__ void issue(char* lds_tile, int kt, int K) {
    const int k0 = kt * BKT;
    const bool full = k0 + BKT <= K;  // uniform
#pragma unroll
    for (int j = 0; j < NINST; ++j) {
      uint32_t o = off[j];
      if (!full) o = (k0 + kofs[j] < K) ? o : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds_tile + (wave * NINST + j) * 1024), 16, o, 0, 0, 0);
      if (off[j] != OOB) off[j] += kstep;
    }
  }
};

// per-lane fragment offsets for the DMA tile images
template <int R, bool KM, int NF, int BKT>
struct DmaFrag {
  int off[KM ? NF : 1];
  __device__ __forceinline__ void init(int w0, int lane) {
    if (!KM) {
      const int r = w0 + (lane & 15);
      off[0] = r * (BKT * 2) + (((lane >> 4) ^ kc_swz<BKT>(r)) << 4);
    } else {
      const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
      const int kr = 8 * g + q;
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int ch = (w0 >> 3) + 2 * i + (p >> 1);
        off[i] = kr * (2 * R) + ((ch ^ km_swz<R>(kr)) << 4) + (p & 1) * 8;
      }
    }
  }
  __device__ __forceinline__ bf16x8 read(const char* lds, int i, int kk) const {
    if (!KM) {
      return *reinterpret_cast<const bf16x8*>(lds + ((off[0] ^ (kk << 6)) + i * (16 * BKT * 2)));
    } else {
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
      const char* a = lds + off[KM ? i : 0] + kk * 32 * (2 * R);
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * (2 * R)));
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      return __builtin_bit_cast(bf16x8, v);
    }
  }
};

// NJ (0 = all): column sub-tiles of a wave that are actually multiplied.  The grouped positional convolution has N = 48 per
// group: with 128 x 64 tiles as 2 x 2 waves a quarter of the MFMAs multiplied padding; as 4 x 1 waves with NJ = 3 every wave
// skips the fourth (all-padding) sub-tile -- its accumulators stay zero and the bounds-checked epilogue never stores them.
template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM, int BKT, int NJ = 0>
__global__ __launch_bounds__(NTHREADS) void gemm_dma_kernel(const GemmParams p) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int NJU = NJ ? NJ : NI;
  constexpr int ABYTES = BM * BKT * 2, BBYTES = BN * BKT * 2, STAGE = ABYTES + BBYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm0 = (wave / WN) * TM, wn0 = (wave % WN) * TN;
  const int per_z = p.tiles_m * p.tiles_n;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int zs = id / per_z, rem = id % per_z;
  const int tm = rem / p.tiles_n, tn = rem % p.tiles_n;
  const int split = zs % p.split_k, z = zs / p.split_k;
  const int z1 = z / p.nb2, z2 = z % p.nb2;
  const bf16* Ab = p.A + z1 * p.sa1 + z2 * p.sa2;
  const bf16* Bb = p.B + z1 * p.sb1 + z2 * p.sb2;
  const int bm0 = tm * BM, bn0 = tn * BN;
  const int nkt = (p.K + BKT - 1) / BKT;
  const int kps = p.kt_per_split * (BK / BKT);  // host sizes splits in 64-deep steps
  const int kt0 = split * kps;
  const int kt1 = min(nkt, kt0 + kps);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  BiasRegs<NI> bias_regs;
  load_bias<NI>(p, bn0, wn0, lane, z2, bias_regs);

  DmaStager<BM, A_KM, BKT> sa;
  DmaStager<BN, B_KM, BKT> sb;
  sa.init(Ab, p.lda, bm0, p.M, p.K, kt0, p.ext_a);
  sb.init(Bb, p.ldb, bn0, p.N, p.K, kt0, p.ext_b);
  DmaFrag<BM, A_KM, MI, BKT> fra;
  DmaFrag<BN, B_KM, NI, BKT> frb;
  fra.init(wm0, lane);
  frb.init(wn0, lane);

  if (kt0 < kt1) {
    sa.issue(smem, kt0, p.K);
    sb.issue(smem + ABYTES, kt0, p.K);
  }
  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    // own DMA of tile kt has landed (vmcnt 0), then everyone's has; also every wave is done reading stage cur^1
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0) only
    __syncthreads();
    if (kt + 1 < kt1) {
      sa.issue(smem + (cur ^ 1) * STAGE, kt + 1, p.K);
      sb.issue(smem + (cur ^ 1) * STAGE + ABYTES, kt + 1, p.K);
    }
    const char* la = smem + cur * STAGE;
    const char* lb = la + ABYTES;
#pragma unroll
    for (int kk = 0; kk < BKT / 32; ++kk) {
      bf16x8 fa[MI], fb[NJU];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = fra.read(la, i, kk);
#pragma unroll
      for (int j = 0; j < NJU; ++j) fb[j] = frb.read(lb, j, kk);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJU; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();  // all fragment reads done (no DMA is outstanding): LDS becomes the epilogue's transposition buffer
  gemm_epilogue<MI, NI>(p, acc, bias_regs, smem + wave * (1024 * MI * NI), bm0, bn0, wm0, wn0, lane, z, z1, z2, split);
}

// =================================================================================================
// 256x128 tile, 8 waves (4 x 2, 64x64 per wave), THREE LDS stages with two tiles of LDS-DMA in flight.
// The two-stage kernel above spends ~half of its wave cycles parked at the per-K-step barrier (rocprofv3 PMC:
// SQ_WAIT_ANY 48 % of SQ_WAVE_CYCLES, MFMA busy 29 %): one 32 KB tile in flight per workgroup cannot cover the
// L2/HBM latency of ~1-2 us with 0.25-0.5 us of MFMA work.  Here a K step is 1024 MFMA cycles per SIMD and the
// DMA of tiles kt+1 and kt+2 is in flight while tile kt is multiplied: counted `s_waitcnt vmcnt(N)` (N = this
// wave's DMA instructions per tile) + a raw s_barrier per K step, never vmcnt(0) inside the loop.
template <int R, bool KM, int NW>
struct DmaStagerW {
  static constexpr int NINST = R / (8 * NW);  // 1-KiB wave-instructions per wave per tile (tile = R*128 bytes)
  static constexpr uint32_t OOB = 0x80000000u;
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t off[NINST];
  int kofs[NINST];
  uint32_t kstep;
  int wave;

  __device__ __forceinline__ void init(const bf16* base, long ld, int row0, int rows_total, int kt0, uint32_t extent_bytes) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)extent_bytes, 0x00020000);
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    kstep = (uint32_t)((KM ? (long)BK * ld : (long)BK) * 2);
#pragma unroll
    for (int j = 0; j < NINST; ++j) {
      const int S = (wave * NINST + j) * 64 + lane;
      if (!KM) {
        const int r = S >> 3, pc = S & 7;
        const int c = pc ^ ((r >> 1) & 7);
        const int gr = row0 + r;
        kofs[j] = c * 8;
        off[j] = gr < rows_total ? (uint32_t)(((long)gr * ld + c * 8) * 2) : OOB;
      } else {
        constexpr int CPR = R / 8;
        const int kr = S / CPR, pc = S % CPR;
        const int ch = pc ^ km_swz<(R >= 128 ? 128 : 64)>(kr);
        const int gr = row0 + ch * 8;
        kofs[j] = kr;
        off[j] = gr < rows_total ? (uint32_t)(((long)kr * ld + gr) * 2) : OOB;
      }
      if (off[j] != OOB) off[j] += (uint32_t)kt0 * kstep;
    }
  }
  __device__ __forceinline__ void issue(char* lds_tile, int kt, int K) {
    const int k0 = kt * BK;
    const bool full = k0 + BK <= K;
#pragma unroll
    for (int j = 0; j < NINST; ++j) {
      uint32_t o = off[j];
      if (!full) o = (k0 + kofs[j] < K) ? o : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds_tile + (wave * NINST + j) * 1024), 16, o, 0, 0, 0);
      if (off[j] != OOB) off[j] += kstep;
    }
  }
  // issue NINST out-of-range loads (zeros): keeps the per-tile vmcnt bookkeeping uniform past the last tile
  __device__ __forceinline__ void issue_dummy(char* lds_tile) {
#pragma unroll
    for (int j = 0; j < NINST; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds_tile + (wave * NINST + j) * 1024), 16, OOB, 0, 0, 0);
  }
};

template <int R, bool KM, int NF>
struct DmaFragW {
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
        off[i] = kr * (2 * R) + ((ch ^ km_swz<(R >= 128 ? 128 : 64)>(kr)) << 4) + (p & 1) * 8;
      }
    }
  }
  __device__ __forceinline__ bf16x8 read(const char* lds, int i, int kk) const {
    if (!KM) {
      return *reinterpret_cast<const bf16x8*>(lds + ((off[0] ^ (kk << 6)) + i * 2048));
    } else {
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
      const char* a = lds + off[KM ? i : 0] + kk * 32 * (2 * R);
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * (2 * R)));
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      return __builtin_bit_cast(bf16x8, v);
    }
  }
};


template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM>
__global__ __launch_bounds__(64 * WM * WN) void gemm_dma3_kernel(const GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int ABYTES = BM * 128, BBYTES = BN * 128, STAGE = ABYTES + BBYTES;
  using SA = DmaStagerW<BM, A_KM, NW>;
  using SB = DmaStagerW<BN, B_KM, NW>;
  constexpr int NPT = SA::NINST + SB::NINST;  // this wave's DMA instructions per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm0 = (wave / WN) * TM, wn0 = (wave % WN) * TN;
  const int per_z = p.tiles_m * p.tiles_n;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int zs = id / per_z, rem = id % per_z;
  const int tm = rem / p.tiles_n, tn = rem % p.tiles_n;
  const int split = zs % p.split_k, z = zs / p.split_k;
  const int z1 = z / p.nb2, z2 = z % p.nb2;
  const bf16* Ab = p.A + z1 * p.sa1 + z2 * p.sa2;
  const bf16* Bb = p.B + z1 * p.sb1 + z2 * p.sb2;
  const int bm0 = tm * BM, bn0 = tn * BN;
  const int nkt = (p.K + BK - 1) / BK;
  const int kt0 = split * p.kt_per_split;
  const int kt1 = min(nkt, kt0 + p.kt_per_split);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  BiasRegs<NI> bias_regs;
  load_bias<NI>(p, bn0, wn0, lane, z2, bias_regs);

  SA sa;
  SB sb;
  sa.init(Ab, p.lda, bm0, p.M, kt0, p.ext_a);
  sb.init(Bb, p.ldb, bn0, p.N, kt0, p.ext_b);
  DmaFragW<BM, A_KM, MI> fra;
  DmaFragW<BN, B_KM, NI> frb;
  fra.init(wm0, lane);
  frb.init(wn0, lane);

  // prologue: two tiles in flight (past the end: zero-filling dummies keep the vmcnt arithmetic uniform)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (kt0 + s < kt1) {
      sa.issue(smem + s * STAGE, kt0 + s, p.K);
      sb.issue(smem + s * STAGE + ABYTES, kt0 + s, p.K);
    } else {
      sa.issue_dummy(smem + s * STAGE);
      sb.issue_dummy(smem + s * STAGE + ABYTES);
    }
  }
  int stage = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    // tile kt has landed once all but this wave's NPT youngest DMA instructions (tile kt+1) are done ...
    wait_vmcnt<NPT>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();  // ... in every wave; and every wave is done reading the stage refilled below
    __builtin_amdgcn_sched_barrier(0);
    {
      const int s2 = stage >= 1 ? stage - 1 : 2;  // (stage + 2) % 3: the stage read in the previous iteration
      if (kt + 2 < kt1) {
        sa.issue(smem + s2 * STAGE, kt + 2, p.K);
        sb.issue(smem + s2 * STAGE + ABYTES, kt + 2, p.K);
      } else {
        sa.issue_dummy(smem + s2 * STAGE);
        sb.issue_dummy(smem + s2 * STAGE + ABYTES);
      }
    }
    const char* la = smem + stage * STAGE;
    const char* lb = la + ABYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = fra.read(la, i, kk);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = frb.read(lb, j, kk);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
  wait_vmcnt<0>();  // drain the trailing dummy DMA ...
  __syncthreads();  // ... in every wave, and all fragment reads are done: LDS becomes the epilogue's transposition buffer
  gemm_epilogue<MI, NI>(p, acc, bias_regs, smem + wave * (1024 * MI * NI), bm0, bn0, wm0, wn0, lane, z, z1, z2, split);
}

// deterministic split-K combine: C = alpha * sum_s slab[s] (+ bias) (+ C)
__global__ void splitk_reduce_kernel(const GemmParams p) {
  const long per = (long)p.M * p.N;
  const long total = per * p.nz;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int z = (int)(e / per);
    const long r = e % per;
    const int m = (int)(r / p.N), n = (int)(r % p.N);
    float s = 0.f;
    for (int k = 0; k < p.split_k; ++k) s += p.slab[((long)k * p.nz + z) * per + r];
    s = s * p.alpha + (p.bias ? p.bias[(z % p.nb2) * p.bias_s2 + n] : 0.f);
    const long o = (long)(z / p.nb2) * p.sc1 + (long)(z % p.nb2) * p.sc2 + (long)m * p.ldc + n;
    if (p.out_f32) {
      float* dst = reinterpret_cast<float*>(p.C) + o;
      *dst = p.accumulate ? (*dst + s) : s;
    } else {
      reinterpret_cast<bf16*>(p.C)[o] = (bf16)s;
    }
  }
}

bool g_no_big_tile = false;  // development switch (SSAK_GEMM_NO_BIG=1): keep the 128x128 kernels
template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM, int NJ = 0>
int launch(const GemmParams& p, bool dma, hipStream_t st) {
  const size_t lds = dma ? 2 * (size_t)(BM + BN) * 128 : 2 * (size_t)(Tile<BM, A_KM>::BYTES + Tile<BN, B_KM>::BYTES);
  auto kern = dma ? gemm_dma_kernel<BM, BN, WM, WN, A_KM, B_KM, 64, NJ> : gemm_kernel<BM, BN, WM, WN, A_KM, B_KM>;
  if (lds > 64 * 1024) {
    static bool attr_done[2] = {false, false};  // per instantiation and kernel flavour
    if (!attr_done[dma]) {
      SSAK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_done[dma] = true;
    }
  }
  const long nblk = (long)p.tiles_m * p.tiles_n * p.nz * p.split_k;
  static int slots[2] = {-1, -1};
  if (slots[dma] < 0) {
    char nm[112];
    // (as rocprofv3 prints the instantiation: tests/test_profiles.py looks the slot names up in the kernel statistics)
    if (dma)
      snprintf(nm, sizeof(nm), "gemm_dma_kernel<%d, %d, %d, %d, %s, %s, 64, %d>", BM, BN, WM, WN, A_KM ? "true" : "false", B_KM ? "true" : "false", NJ);
    else
      snprintf(nm, sizeof(nm), "gemm_kernel<%d, %d, %d, %d, %s, %s>", BM, BN, WM, WN, A_KM ? "true" : "false", B_KM ? "true" : "false");
    slots[dma] = ssak_prof_register(nm, SSAK_BOUND_MFMA);
  }
  ProfScope prof_scope(slots[dma], 2.0 * p.M * p.N * (double)p.K * p.nz, st);
  kern<<<dim3((unsigned)nblk), NTHREADS, lds, st>>>(p);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <bool A_KM, bool B_KM>
int launch_big(const GemmParams& p, hipStream_t st) {
  constexpr size_t lds = 3 * (size_t)(256 + 128) * 128;  // 144 KiB
  auto kern = gemm_dma3_kernel<256, 128, 4, 2, A_KM, B_KM>;
  static bool attr_done = false;
  if (!attr_done) {
    SSAK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  const long nblk = (long)p.tiles_m * p.tiles_n * p.nz * p.split_k;
  static int slot = -1;
  if (slot < 0) {
    char nm[112];
    snprintf(nm, sizeof(nm), "gemm_dma3_kernel<256, 128, 4, 2, %s, %s>", A_KM ? "true" : "false", B_KM ? "true" : "false");
    slot = ssak_prof_register(nm, SSAK_BOUND_MFMA);
  }
  ProfScope prof_scope(slot, 2.0 * p.M * p.N * (double)p.K * p.nz, st);
  kern<<<dim3((unsigned)nblk), 512, lds, st>>>(p);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

int dispatch_big(const GemmParams& p, int a_km, int b_km, hipStream_t st) {
  if (!a_km && !b_km) return launch_big<false, false>(p, st);
  if (!a_km && b_km) return launch_big<false, true>(p, st);
  if (a_km && b_km) return launch_big<true, true>(p, st);
  return launch_big<true, false>(p, st);
}

template <int BM, int BN, int WM, int WN, int NJ = 0>
int dispatch_layout(const GemmParams& p, int a_km, int b_km, bool dma, hipStream_t st) {
  if (!a_km && !b_km) return launch<BM, BN, WM, WN, false, false, NJ>(p, dma, st);
  if (!a_km && b_km) return launch<BM, BN, WM, WN, false, true, NJ>(p, dma, st);
  if (a_km && b_km) return launch<BM, BN, WM, WN, true, true, NJ>(p, dma, st);
  return launch<BM, BN, WM, WN, true, false, NJ>(p, dma, st);
}


// out[n] += sum over slots of partial[slot][n], fixed order (the column sums the epilogues left per wave-tile row).
// Workgroup = 64 columns x 16 slot groups; the groups are combined in LDS in a fixed order.
__global__ __launch_bounds__(1024) void colsum_slots_kernel(const float* __restrict__ partial, int slots, int N, float* __restrict__ out) {
  __shared__ float red[16][65];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + cx;
  float s = 0.f;
  if (n < N)
    for (int k = ry; k < slots; k += 16) s += partial[(long)k * N + n];
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][cx];
    out[n] += t;
  }
}

// ---- kernel / tile / split-K selection ---------------------------------------------------------------------------
// A small cost model in microseconds (constants fitted to tools/bench_gemm.py and tools/probes/p8_probe.hip on the
// Wav2Vec2-base train-step shapes, MI355X): a workgroup costs a fixed prologue + epilogue plus a per-K-tile time, the
// launch runs ceil(workgroups / resident workgroups) rounds, a split-K launch adds its slab reduction.
//   phase-interleaved kernel (gemm_p8.hip): one 8-wave workgroup per CU, tile (256|192|128) x 256 -- the tile height is
//     what fills the last round (M = 15968: 192-row tiles give 84 x 3 = 252 tiles for N = 768);
//   128x128 kernel: two 4-wave workgroups per CU.
struct GemmPlan {
  bool p8;
  int bm;     // p8 tile height
  int split;
  double cost;
};
int g_env_p8 = -2, g_env_p8_bm = 0;
bool g_no_p4 = false;  // development switch SSAK_GEMM_NO_P4: keep everything on the 8-wave kernel
GemmPlan plan_gemm(const ssak_gemm_desc* d, bool dma, size_t workspace_bytes) {
  if (g_env_p8 == -2) {
    const char* v = SSAK_DEV_ENV("SSAK_GEMM_P8");  // development switches: 0 = never, 1 = whenever it applies
    g_env_p8 = v ? atoi(v) : -1;
    v = SSAK_DEV_ENV("SSAK_GEMM_P8_BM");
    g_env_p8_bm = v ? atoi(v) : 0;
    v = SSAK_DEV_ENV("SSAK_GEMM_NO_P4");
    g_no_p4 = v && v[0] == '1';
  }
  const int nkt = ssak_cdiv(d->K, BK);
  const long nz = (long)d->nb1 * d->nb2;
  const double out_mb = (double)nz * d->M * d->N * 4e-6;  // one fp32 slab, MB
  const bool wide = d->out_f32 != 0;
  int s_lo = d->split_k > 1 ? d->split_k : 1, s_hi = s_lo;
  if (d->split_k == 0) {  // auto: deterministic split-K sized by the model (plain epilogue only)
    s_lo = 1;
    s_hi = (d->epilogue == SSAK_EPI_NONE && !(d->drop_p > 0.f)) ? std::max(1, std::min(32, nkt / 4)) : 1;
  }
  // (the feed-forward epilogue pair exists on the persistent kernel only for the layouts the encoder uses: other layouts take
  // the 128 x 128 kernels, whose LDS-staged epilogue handles every mode)
  const bool p8_layout_ok = (d->epilogue != SSAK_EPI_GELU_SAVE_GRAD || (!d->a_kmajor && !d->b_kmajor)) &&
                            (d->epilogue != SSAK_EPI_MUL_AUX || !d->a_kmajor);
  const bool p8_ok = dma && d->M >= 256 && d->N >= 256 && g_env_p8 != 0 && p8_layout_ok;
  if (p8_ok && (d->plan_tile == 256 || d->plan_tile == 192 || d->plan_tile == 128))
    return GemmPlan{true, d->plan_tile, s_lo, 0.0};  // caller's choice
  GemmPlan best_def{false, 0, s_lo, 1e30}, best_p8{true, 256, s_lo, 1e30};
  for (int s = s_lo; s <= s_hi; ++s) {
    if (d->split_k == 0 && s > 1 && (size_t)s * nz * (size_t)d->M * d->N * sizeof(float) > workspace_bytes) break;
    const int kt = ssak_cdiv(nkt, s);
    const bool slab = s > 1;
    const double reduce = slab ? 2.0 + (s + 1) * out_mb / 3.5 : 0.0;
    {
      const long blocks = (long)ssak_cdiv(d->M, 128) * ssak_cdiv(d->N, d->N > 64 ? 128 : 64) * nz * s;
      const double c = (double)ssak_cdiv(blocks, 512) * (2.5 + ((slab || wide) ? 2.0 : 0.0) + kt * 1.0) + reduce;
      if (c < best_def.cost) best_def = GemmPlan{false, 0, s, c};
    }
    for (int bm = 256; p8_ok && bm >= 128; bm -= 64) {
      if (g_env_p8_bm && bm != g_env_p8_bm) continue;
      const long blocks = (long)ssak_cdiv(d->M, bm) * ssak_cdiv(d->N, 256) * nz * s;
      const double epi = ((slab || wide) ? 9.4 : 4.7) * bm / 256.0;
      const double c = (double)ssak_cdiv(blocks, 256) * (2.0 + epi + kt * (0.60 + 0.98 * bm / 256.0)) + reduce;
      if (c < best_p8.cost) best_p8 = GemmPlan{true, bm, s, c};
    }
  }
  if (p8_ok && best_p8.cost < 1e30 && (g_env_p8 == 1 || best_p8.cost < 0.97 * best_def.cost)) return best_p8;
  return best_def;
}

// The B-direct form of the persistent kernel (gemm_p8.hip) for this product and plan?  Measured on the train step's shapes
// (tools/probes/p8_loop.hip, profiles/r03_gemm_bdirect_probe*.log): 192-row tiles (the N = 768 products) gain 3-25 %, 256-row
// tiles gain for deep K and narrow N (K = 3072, N = 768: +15 %) and LOSE for wide outputs (N = 3072: -5..-9 %, the weight no
// longer fits the XCD's L2 next to the activations and every wave row fetches it again) and for the conv stack.
bool fragments_pay(const ssak_gemm_desc* d, const GemmPlan& plan) {
  if (!plan.p8 || plan.split != 1 || d->a_kmajor || d->out_f32 || d->accumulate || d->colsum || d->drop_p > 0.f) return false;
  if (d->sb1 || d->sb2 || d->lda < d->K) return false;  // one weight for every batch; no Toeplitz A (conv stack)
  if (d->epilogue != SSAK_EPI_NONE) return false;
  if ((size_t)ssak_cdiv(d->N, 256) * 4 * ssak_cdiv(d->K, BK) * 8192 >= (1ull << 31)) return false;
  if (plan.bm == 192) return d->N <= 2304;
  if (plan.bm == 256) return d->K >= 2048 && d->N <= 1024;
  return false;
}

}  // namespace

extern "C" int ssak_gemm_uses_fragments(const ssak_gemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
  const bool partial_a = d->a_kmajor ? (d->M & 7) : (d->K & 7);
  const bool partial_b = d->b_kmajor ? (d->N & 7) : (d->K & 7);
  const bool dma = d->pads_are_zero || !(partial_a || partial_b);
  return fragments_pay(d, plan_gemm(d, dma, 0)) ? 1 : 0;
}
extern "C" size_t ssak_gemm_fragment_b_bytes(int N, int K) { return N > 0 && K > 0 ? k_gemm_fragment_b_bytes(N, K) : 0; }
extern "C" int ssak_gemm_fragment_b_batched(int n, const void* const* B, const long* ldb, const int* N, const int* K, const int* b_kmajor,
                                            void* const* out, void* stream) {
  SSAK_REQUIRE(n >= 0 && (n == 0 || (B && ldb && N && K && b_kmajor && out)), "gemm_fragment_b: null pointer");
  if (n == 0) return SSAK_OK;
  return k_gemm_fragment_b_batched(n, B, ldb, N, K, b_kmajor, out, (hipStream_t)stream);
}
extern "C" int ssak_gemm_fragment_b(const void* B, long ldb, int N, int K, int b_kmajor, void* out, void* stream) {
  return ssak_gemm_fragment_b_batched(1, &B, &ldb, &N, &K, &b_kmajor, &out, stream);
}

extern "C" int ssak_gemm_bf16(const ssak_gemm_desc* d, const void* A, const void* B, void* C, const float* bias,
                              const void* aux_in, void* aux_out, void* workspace, size_t workspace_bytes,
                              void* stream) {
  SSAK_REQUIRE(d && A && B && C, "gemm: null pointer");
  SSAK_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  SSAK_REQUIRE((d->lda & 7) == 0 && (d->ldb & 7) == 0 && (d->ldc & 3) == 0, "gemm: lda/ldb must be multiples of 8, ldc of 4");
  SSAK_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)C & 15) == 0, "gemm: operands must be 16-byte aligned");
  SSAK_REQUIRE(((d->sa1 | d->sa2 | d->sb1 | d->sb2) & 7) == 0 && ((d->sc1 | d->sc2) & 3) == 0 &&
                   !(d->drop_p > 0.f && ((d->sc1 | d->sc2 | d->ldc) & 3)), "gemm: batch strides must keep 16-byte (A,B) / 8-byte (C) alignment");
  SSAK_REQUIRE(d->nb1 > 0 && d->nb2 > 0, "gemm: batch counts must be >= 1");
  SSAK_REQUIRE(d->epilogue >= 0 && d->epilogue <= SSAK_EPI_MUL_AUX, "gemm: bad epilogue %d", d->epilogue);
  SSAK_REQUIRE((d->epilogue != SSAK_EPI_MUL_GELU_GRAD && d->epilogue != SSAK_EPI_MUL_AUX) || aux_in, "gemm: MUL_GELU_GRAD / MUL_AUX need aux_in");
  SSAK_REQUIRE(!d->accumulate || d->out_f32, "gemm: accumulate needs fp32 output");
  SSAK_REQUIRE(d->split_k >= 0, "gemm: split_k must be >= 0 (0 = choose)");
  SSAK_REQUIRE(d->split_k <= 1 || d->epilogue == SSAK_EPI_NONE, "gemm: split_k supports the plain epilogue only");
  // the LDS-DMA kernels cannot mask partial 16-byte chunks: they need whole chunks or zero padding in memory
  const bool partial_a = d->a_kmajor ? (d->M & 7) : (d->K & 7);
  const bool partial_b = d->b_kmajor ? (d->N & 7) : (d->K & 7);
  const bool dma = d->pads_are_zero || !(partial_a || partial_b);
  const GemmPlan plan = plan_gemm(d, dma, workspace ? workspace_bytes : 0);
  const int split = plan.split;
  static const bool env_trace = SSAK_DEV_ENV("SSAK_GEMM_TRACE") != nullptr;  // development: one line per launch
  if (env_trace)
    fprintf(stderr, "gemm M=%d N=%d K=%d akm=%d bkm=%d nb=%dx%d epi=%d f32=%d acc=%d drop=%g dma=%d -> %s bm=%d split=%d cost=%.1f\n", d->M,
            d->N, d->K, d->a_kmajor, d->b_kmajor, d->nb1, d->nb2, d->epilogue, d->out_f32, d->accumulate, d->drop_p, (int)dma,
            plan.p8 ? "p8" : "tile128", plan.bm, plan.split, plan.cost);
  // column sums of C (desc.colsum): fused into the LDS-free epilogue of the 256-wide kernel when every tile takes it,
  // otherwise a separate pass over the stored bf16 C
  float* colsum_out = nullptr;
  bool colsum_fused = false;
  if (d->colsum) {
    SSAK_REQUIRE(aux_out && d->epilogue != SSAK_EPI_GELU && d->epilogue != SSAK_EPI_GELU_SAVE_GRAD && d->nb1 == 1 && d->nb2 == 1 && d->split_k <= 1 && !d->out_f32,
                 "gemm: colsum needs aux_out = float[N], a bf16 C, no batches / split_k / GELU side output");
    colsum_out = reinterpret_cast<float*>(aux_out);
    aux_out = nullptr;
    colsum_fused = plan.p8 && plan.split == 1 && (d->N % 256) == 0 && (d->ldc & 7) == 0 && workspace &&
                   workspace_bytes >= (size_t)ssak_cdiv(d->M, 64) * d->N * sizeof(float);
  }
  GemmParams p;
  p.A = (const bf16*)A;
  p.B = (const bf16*)B;
  p.C = C;
  p.bias = bias;
  p.aux_in = (const bf16*)aux_in;
  p.aux_out = (bf16*)aux_out;
  p.slab = (float*)workspace;
  p.M = d->M;
  p.N = d->N;
  p.K = d->K;
  p.lda = d->lda;
  p.ldb = d->ldb;
  p.ldc = d->ldc;
  p.nb2 = d->nb2;
  p.sa1 = d->sa1;
  p.sa2 = d->sa2;
  p.sb1 = d->sb1;
  p.sb2 = d->sb2;
  p.sc1 = d->sc1;
  p.sc2 = d->sc2;
  p.alpha = d->alpha;
  p.epilogue = d->epilogue;
  p.out_f32 = d->out_f32;
  p.accumulate = d->accumulate;
  p.dynamic = d->dynamic_tiles;
  p.split_k = split;
  p.nz = d->nb1 * d->nb2;
  // 16-bit dropout uniforms in the epilogue: threshold = round(p * 65536), scale from the realised keep probability
  p.drop_thresh = d->drop_p > 0.f ? (uint32_t)fminf(65535.f, roundf(d->drop_p * 65536.f)) : 0u;
  p.drop_scale = p.drop_thresh ? 1.f / (1.f - (float)p.drop_thresh / 65536.f) : 1.f;
  p.drop_stream = d->drop_stream;
  p.drop_seed = d->drop_seed;
  p.fq_a = p.fq_b = 0.f;
  if (d->epilogue == SSAK_EPI_MUL_AUX) {
    // the factor's mask is in its codes; drop_p only says which 1 / (1 - p) the forward product applied to its output
    p.fq_a = FQ_STEP * p.drop_scale;
    p.fq_b = -FQ_ZERO * FQ_STEP * p.drop_scale;
    p.drop_thresh = 0;
  }
  p.bias_s2 = d->bias_s2;
  p.colsum = nullptr;
  SSAK_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "gemm: drop_p must be in [0,1)");
  SSAK_REQUIRE(!(d->drop_p > 0.f && d->split_k > 1), "gemm: dropout epilogue is not available with split_k");
  SSAK_REQUIRE(!(d->drop_p > 0.f) || d->N <= DROP_TABLE_N, "gemm: the dropout epilogue is built for at most %d output columns", DROP_TABLE_N);
  const int nkt = ssak_cdiv(d->K, BK);
  p.kt_per_split = ssak_cdiv(nkt, split);
  if (split > 1)
    SSAK_REQUIRE(workspace && workspace_bytes >= (size_t)split * p.nz * (size_t)d->M * d->N * sizeof(float),
                 "gemm: split_k workspace too small");
  {
    // kernels address an operand slice with 32-bit byte offsets from its (batch-adjusted) base
    const double ext_a = d->a_kmajor ? ((double)(d->K - 1) * d->lda + d->M) : ((double)(d->M - 1) * d->lda + d->K);
    const double ext_b = d->b_kmajor ? ((double)(d->K - 1) * d->ldb + d->N) : ((double)(d->N - 1) * d->ldb + d->K);
    SSAK_REQUIRE(ext_a * 2 < 2.0e9 && ext_b * 2 < 2.0e9, "gemm: one batch slice of an operand must span < 2 GB");
    // round up to whole 16-byte chunks: a partial last chunk is still fetched (ld* and the allocation cover it)
    p.ext_a = (uint32_t)(((long)ext_a + 7) / 8 * 16);
    p.ext_b = (uint32_t)(((long)ext_b + 7) / 8 * 16);
  }
  hipStream_t st = (hipStream_t)stream;
  int rc;
  static const bool env_no_big = [] {
    const char* nb = SSAK_DEV_ENV("SSAK_GEMM_NO_BIG");
    return nb && nb[0] == '1';
  }();
  const long big_tiles = (long)ssak_cdiv(d->M, 256) * ssak_cdiv(d->N, 128) * p.nz * split;
  const int p8_bm = plan.bm;
  if (colsum_fused) p.colsum = (float*)workspace;
  p.kperm_p = p.kperm_n2 = 0;
  if (plan.p8) {
    p.tiles_m = ssak_cdiv(d->M, p8_bm);
    p.tiles_n = ssak_cdiv(d->N, 256);
    // Toeplitz A (conv as GEMM, rows overlap: lda < K): visit the K tiles so that the two reads of the same bytes -- tap t + s of
    // row i is tap t of row i + 1 -- are one K step apart instead of lda / BK steps (gemm_common.h: kperm_*)
    static const bool env_no_perm = [] {
      const char* e = SSAK_DEV_ENV("SSAK_GEMM_NO_KPERM");
      return e && e[0] == '1';
    }();
    if (!d->a_kmajor && split == 1 && d->K % BK == 0 && d->lda % BK == 0 && d->lda < d->K && !env_no_perm) {
      const int pp = (int)(d->lda / BK), n2 = nkt - pp;
      if (n2 > 0 && n2 <= pp) {
        p.kperm_p = pp;
        p.kperm_n2 = n2;
      }
    }
    if (d->b_fragments && fragments_pay(d, plan)) {
      SSAK_REQUIRE(((uintptr_t)d->b_fragments & 15) == 0, "gemm: b_fragments must be 16-byte aligned");
      p.B = (const bf16*)d->b_fragments;
      p.ext_b = (uint32_t)k_gemm_fragment_b_bytes(d->N, d->K);
      rc = ssak_gemm_p8bd_launch(&p, p8_bm, st);
    } else if (!g_no_p4 && ssak_gemm_p4_supports(&p, p8_bm, d->a_kmajor, d->b_kmajor)) {
      rc = ssak_gemm_p4_launch(&p, p8_bm, d->b_kmajor, st);
    } else {
      rc = ssak_gemm_p8_launch(&p, p8_bm, d->a_kmajor, d->b_kmajor, st);
    }
  } else if (d->N > 64 && dma && d->M >= 256 && !d->a_kmajor && !d->b_kmajor && big_tiles >= 2048 && !env_no_big && !g_no_big_tile) {
    p.tiles_m = ssak_cdiv(d->M, 256);
    p.tiles_n = ssak_cdiv(d->N, 128);
    rc = dispatch_big(p, d->a_kmajor, d->b_kmajor, st);
  } else if (d->N > 64) {
    p.tiles_m = ssak_cdiv(d->M, 128);
    p.tiles_n = ssak_cdiv(d->N, 128);
    rc = dispatch_layout<128, 128, 2, 2>(p, d->a_kmajor, d->b_kmajor, dma, st);
  } else {
    p.tiles_m = ssak_cdiv(d->M, 128);
    p.tiles_n = ssak_cdiv(d->N, 64);
    static const bool env_no_n48 = SSAK_DEV_ENV("SSAK_GEMM_NO_N48") != nullptr;  // development switches
    static const bool env_n48_128 = SSAK_DEV_ENV("SSAK_GEMM_N48_128") != nullptr;
    if (dma && d->N > 32 && d->N <= 48 && !env_no_n48 && !env_n48_128 && d->M >= 256 && !d->a_kmajor && !d->b_kmajor) {
      // K-contiguous operands: 256-row tiles -- twice the MFMA work per K step behind the same LDS-DMA round trip (this two-stage
      // kernel is latency-bound): 437 -> 378 us per step for the three forward / dX launches.  (K-major operands, the weight
      // gradient: 250 -> 325 us, so they stay on 128 rows.)
      // (512-row tiles, one workgroup per CU: 576 us)
      p.tiles_m = ssak_cdiv(d->M, 256);
      rc = launch<256, 64, 4, 1, false, false, 3>(p, dma, st);
    } else if (dma && d->N > 32 && d->N <= 48 && !env_no_n48)
      rc = dispatch_layout<128, 64, 4, 1, 3>(p, d->a_kmajor, d->b_kmajor, dma, st);  // N = 48 (grouped positional conv): no padding MFMAs
    else
      rc = dispatch_layout<128, 64, 2, 2>(p, d->a_kmajor, d->b_kmajor, dma, st);
  }
  if (rc != SSAK_OK) return rc;
  if (split > 1) {
    const long total = (long)p.nz * d->M * d->N;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    splitk_reduce_kernel<<<blocks, 256, 0, st>>>(p);
    SSAK_LAUNCH_CHECK();
  }
  if (colsum_out) {
    if (colsum_fused) {
      const int slots = 2 * ssak_cdiv(d->M, p8_bm);  // (tile row, wave row) pairs, wave tiles of p8_bm / 2 rows
      if (g_reduce_sink && g_reduce_sink->push((const float*)workspace, d->N, slots, d->N, colsum_out)) return SSAK_OK;
      colsum_slots_kernel<<<ssak_cdiv(d->N, 64), 1024, 0, st>>>((const float*)workspace, slots, d->N, colsum_out);
      SSAK_LAUNCH_CHECK();
    } else {
      // separate pass over the stored C; two-stage and fixed-order when the workspace can hold its partial rows
      // (min(64, ceil(M / 32)) * N floats), float atomics otherwise
      return k_colsum((const bf16*)C, d->ldc, d->M, d->N, colsum_out, st, (float*)workspace, workspace_bytes / sizeof(float));
    }
  }
  return SSAK_OK;
}

// Grouped launch: n <= 8 products of the same K, operand layouts, alpha and output type in ONE persistent launch of the
// phase-interleaved kernel (no bias / activation / dropout / split-K / batches).  This is the weight-gradient form of the
// train step: the four dW = dY^T X products of an encoder layer have 108 256x256 tiles between them and K = 16 k rows;
// launched one by one each needs a 7-way split-K with fp32 slabs and a reduction pass to fill the chip, two layers launched
// together are 216 tiles -- one round of workgroups, every accumulator written once, straight into the gradient buffer.
extern "C" int ssak_gemm_bf16_grouped(const ssak_gemm_desc* descs, int n, const void* const* A, const void* const* B, void* const* C,
                                      void* stream) {
  SSAK_REQUIRE(descs && A && B && C && n >= 1 && n <= 48, "gemm_grouped: need 1..48 problems");
  const ssak_gemm_desc& d0 = descs[0];
  int Ms[48], Ns[48];
  long lda[48], ldb[48], ldc[48];
  uint32_t ea[48], eb[48];
  for (int i = 0; i < n; ++i) {
    const ssak_gemm_desc& d = descs[i];
    SSAK_REQUIRE(A[i] && B[i] && C[i], "gemm_grouped: null operand");
    SSAK_REQUIRE(d.M > 0 && d.N > 0 && d.K == d0.K && d.K > 0, "gemm_grouped: all problems must share K");
    SSAK_REQUIRE(d.a_kmajor == d0.a_kmajor && d.b_kmajor == d0.b_kmajor && d.out_f32 == d0.out_f32 && d.alpha == d0.alpha &&
                     d.accumulate == d0.accumulate, "gemm_grouped: layouts, alpha and output type must match");
    SSAK_REQUIRE(d.nb1 == 1 && d.nb2 == 1 && d.epilogue == SSAK_EPI_NONE && !(d.drop_p > 0.f) && d.split_k <= 1,
                 "gemm_grouped: plain single products only");
    SSAK_REQUIRE((d.lda & 7) == 0 && (d.ldb & 7) == 0 && (d.ldc & 3) == 0, "gemm_grouped: lda/ldb must be multiples of 8, ldc of 4");
    SSAK_REQUIRE(((uintptr_t)A[i] & 15) == 0 && ((uintptr_t)B[i] & 15) == 0 && ((uintptr_t)C[i] & 15) == 0,
                 "gemm_grouped: operands must be 16-byte aligned");
    SSAK_REQUIRE(!d.accumulate || d.out_f32, "gemm_grouped: accumulate needs fp32 output");
    const bool partial_a = d.a_kmajor ? (d.M & 7) : (d.K & 7);
    const bool partial_b = d.b_kmajor ? (d.N & 7) : (d.K & 7);
    SSAK_REQUIRE(d.pads_are_zero || !(partial_a || partial_b), "gemm_grouped: operands need whole 16-byte chunks (or zero padding)");
    const double xa = d.a_kmajor ? ((double)(d.K - 1) * d.lda + d.M) : ((double)(d.M - 1) * d.lda + d.K);
    const double xb = d.b_kmajor ? ((double)(d.K - 1) * d.ldb + d.N) : ((double)(d.N - 1) * d.ldb + d.K);
    SSAK_REQUIRE(xa * 2 < 2.0e9 && xb * 2 < 2.0e9, "gemm_grouped: an operand must span < 2 GB");
    ea[i] = (uint32_t)(((long)xa + 7) / 8 * 16);
    eb[i] = (uint32_t)(((long)xb + 7) / 8 * 16);
    Ms[i] = d.M;
    Ns[i] = d.N;
    lda[i] = d.lda;
    ldb[i] = d.ldb;
    ldc[i] = d.ldc;
  }
  GemmParams p{};
  p.K = d0.K;
  p.M = d0.M;
  p.N = d0.N;
  p.nb2 = 1;
  p.nz = 1;
  p.alpha = d0.alpha;
  p.epilogue = SSAK_EPI_NONE;
  p.out_f32 = d0.out_f32;
  p.accumulate = d0.accumulate;
  p.dynamic = d0.dynamic_tiles;
  p.split_k = 1;
  p.drop_scale = 1.f;
  p.kt_per_split = ssak_cdiv(d0.K, BK);
  hipStream_t st = (hipStream_t)stream;
  // (the four-wave loop with both operands K-major -- transposing reads into pinned fragment registers, a padded two-rows-per-
  // block image -- was built and is bit-exact, but no faster than the eight-wave kernel here: 1 170 vs 1 206 TFLOP/s in situ.
  // These launches stream ~1 GB of activations per 250-tile round and both kernels sit at ~1.7 us per K tile on the bytes
  // a CU can keep in flight with two 64 KB stages, not on the matrix pipe: tools/probes/gemm_p4w_kmajor.hip.txt,
  // profiles/r04_gemm_p4w_kmajor_not_adopted.log)
  const int rc = ssak_gemm_p8_launch_grouped(&p, n, A, B, C, Ms, Ns, lda, ldb, ldc, ea, eb, d0.a_kmajor, d0.b_kmajor, st);
  return rc;
}

