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
//   * a K tile is four phases, one output quadrant (64x32 per wave, 16 MFMAs) each:
//       P1: read AT,BL  -> C(top,left)      DMA BR(t+1)   wait for BR(t)
//       P2: read BR     -> C(top,right)     DMA AB(t+1)   wait for AB(t)
//       P3: read AB     -> C(bottom,right)  DMA BL(t+2)
//       P4: (BL kept)   -> C(bottom,left)   DMA AT(t+2)   wait for BL(t+1), AT(t+1)
//     every wait is a counted `s_waitcnt vmcnt(8)`: four pieces (64 KB per CU) stay in flight across the raw
//     s_barriers and each piece has about four phases (>= 1000 MFMA cycles) to land; vmcnt never drains in the loop;
//   * the two wave rows run staggered by one barrier: while one does its 16 MFMAs the other issues its ds_reads and
//     DMA, so the matrix pipe of each SIMD alternates between its two resident waves instead of idling on LDS.
// Hazards (G0 = wave row 0, G1 = wave row 1, one barrier behind): a piece waited for in phase w (before the phase's
// first barrier) is read in phase >= w+1; a slot last read in phase r is re-staged in phase >= r+2.  Past the last K
// tile the same DMA instructions are issued with out-of-range offsets (the DMA then writes zeros into slots nobody
// reads any more), which keeps the vmcnt arithmetic uniform.
//
// LDS images per piece (as the 128x128 kernel): K-contiguous [128 rows][128 B], 16-B chunk c of row r at slot
// c ^ ((r >> 1) & 7) (ds_read_b128, conflict-free); K-major [64 k-rows][256 B], chunk ch of k-row kr at slot
// ch ^ km_swz(kr) (ds_read_b64_tr_b16).  The DMA writes linearly, so the swizzles are applied to the source address.
#include "common.h"
#include "gemm_common.h"
#include "kernels.h"

namespace {

constexpr int P8_THREADS = 512;
constexpr int P8_PIECE = 16384;          // bytes per piece
constexpr int P8_BUF = 4 * P8_PIECE;     // AT, AB, BL, BR
constexpr int P8_LDS = 2 * P8_BUF;       // 128 KiB

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
  __device__ __forceinline__ void issue(char* lds_piece, int kt, int kt_end, int K) {
    // k-rows / k-chunks of this tile that exist (uniform): all 64, a K tail, or none (dummy past the last tile)
    const int klim = kt < kt_end ? min(K - kt * BK, BK) : 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t o = kofs[j] < klim ? off[H][j] : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds_piece + (j * 8 + wave) * 1024), 16, o, 0, 0, 0);
      off[H][j] += kstep;  // an OOB offset stays >= 2^31 (total advance < 2 GB), i.e. out of range
    }
  }
};

// per-lane fragment offsets inside a piece; w0 = first piece row of this wave's panel, NF 16-row groups
template <bool KM, int NF>
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
    if (!KM) {
      return *reinterpret_cast<const bf16x8*>(piece + ((off[0] ^ (kk << 6)) + i * 2048));
    } else {
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
      const char* a = piece + off[KM ? i : 0] + kk * 32 * 256;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * 256));
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
    if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(p.slab)[blockIdx.x * 8 + (i)] = wall_clock64(); \
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
#ifdef P8_DMA_IN_MFMA
// variant: the phase's LDS-DMA is issued by the wave row that is multiplying, between the two 32-deep halves
#define P8_MFMA(I0, J0, FB, DMA)                                                                              \
  do {                                                                                                        \
    P8_BARRIER();                                                                                             \
    __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0) */                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                            \
    _Pragma("unroll") for (int i = 0; i < MH; ++i)                                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[(I0) + i][(J0) + j] =                               \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[j][0], fa[i][0], acc[(I0) + i][(J0) + j], 0, 0, 0);    \
    P8_FENCE();                                                                                               \
    DMA;                                                                                                      \
    P8_FENCE();                                                                                               \
    _Pragma("unroll") for (int i = 0; i < MH; ++i)                                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[(I0) + i][(J0) + j] =                               \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[j][1], fa[i][1], acc[(I0) + i][(J0) + j], 0, 0, 0);    \
    __builtin_amdgcn_s_setprio(0);                                                                            \
    P8_BARRIER();                                                                                             \
  } while (0)
#define P8_PRE(DMA)
#define P8_WAIT() wait_vmcnt<6>()
#else
#define P8_MFMA(I0, J0, FB, DMA)                                                                              \
  do {                                                                                                        \
    P8_BARRIER();                                                                                             \
    __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0) */                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                            \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int i = 0; i < MH; ++i)           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[(I0) + i][(J0) + j] =                               \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[j][kk], fa[i][kk], acc[(I0) + i][(J0) + j], 0, 0, 0);  \
    __builtin_amdgcn_s_setprio(0);                                                                            \
    P8_BARRIER();                                                                                             \
  } while (0)
#define P8_PRE(DMA) DMA
#define P8_WAIT() wait_vmcnt<8>()
#endif

// MH = 16-row groups per quadrant: the tile is (64 * MH) x 256, i.e. 256 / 192 / 128 rows -- picked by the host so
// that the tile count fills whole rounds of 256 workgroups (M = 15968: 192-row tiles give 84 x 3 = 252 tiles for N = 768).
template <int MH, bool A_KM, bool B_KM>
__global__ __launch_bounds__(P8_THREADS) void gemm_p8_kernel(const GemmParams p) {
  constexpr int BM = 64 * MH, SEGA = 16 * MH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  P8_STAMP(0);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int per_z = p.tiles_m * p.tiles_n;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int zs = id / per_z, rem = id % per_z;
  const int tm = rem / p.tiles_n, tn = rem % p.tiles_n;
  const int split = zs % p.split_k, z = zs / p.split_k;
  const int z1 = z / p.nb2, z2 = z % p.nb2;
  const bf16* Ab = p.A + z1 * p.sa1 + z2 * p.sa2;
  const bf16* Bb = p.B + z1 * p.sb1 + z2 * p.sb2;
  const int bm0 = tm * BM, bn0 = tn * 256;
  const int nkt = (p.K + BK - 1) / BK;
  const int kt0 = split * p.kt_per_split;
  const int kt1 = min(nkt, kt0 + p.kt_per_split);

  f32x4 acc[2 * MH][4];
#pragma unroll
  for (int i = 0; i < 2 * MH; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  P8Stager<A_KM, SEGA, 2 * SEGA, A_KM ? 16 : 4 * MH, 2> sa;
  P8Stager<B_KM, 32, 64, 16, 4> sb;
  sa.init(Ab, p.lda, bm0, p.M, kt0, p.ext_a);
  sb.init(Bb, p.ldb, bn0, p.N, kt0, p.ext_b);
  P8Frag<A_KM, MH> fra;
  P8Frag<B_KM, 2> frb;
  fra.init(wr * SEGA, lane);
  frb.init(wc * 32, lane);

  // piece slots of buffer b: AT = 0, AB = 1, BL = 2, BR = 3
  char* const buf0 = smem;
  char* const buf1 = smem + P8_BUF;
  // prologue: BL(0) AT(0) BR(0) AB(0) BL(1) AT(1)
  sb.template issue<0>(buf0 + 2 * P8_PIECE, kt0, kt1, p.K);
  sa.template issue<0>(buf0 + 0 * P8_PIECE, kt0, kt1, p.K);
  sb.template issue<1>(buf0 + 3 * P8_PIECE, kt0, kt1, p.K);
  sa.template issue<1>(buf0 + 1 * P8_PIECE, kt0, kt1, p.K);
  sb.template issue<0>(buf1 + 2 * P8_PIECE, kt0 + 1, kt1, p.K);
  sa.template issue<0>(buf1 + 0 * P8_PIECE, kt0 + 1, kt1, p.K);
  wait_vmcnt<8>();  // BL(0), AT(0) landed (this wave's share)
  P8_BARRIER();     // ... everyone's
  if (wr == 1) P8_BARRIER();  // wave row 1 runs one barrier behind wave row 0
  P8_STAMP(1);

  bf16x8 fa[MH][2], fbl[2][2], fbr[2][2];
  for (int kt = kt0; kt < kt1; ++kt) {
    char* const cur = ((kt - kt0) & 1) ? buf1 : buf0;
    char* const nxt = ((kt - kt0) & 1) ? buf0 : buf1;
    // ---- P1
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fbl[j][kk] = frb.read(cur + 2 * P8_PIECE, j, kk);
    P8_FENCE();
#pragma unroll
    for (int i = 0; i < MH; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fa[i][kk] = fra.read(cur + 0 * P8_PIECE, i, kk);
    P8_FENCE();
    P8_PRE(sb.template issue<1>(nxt + 3 * P8_PIECE, kt + 1, kt1, p.K));  // BR(t+1)
    P8_WAIT();                                                            // BR(t)
    P8_MFMA(0, 0, fbl, sb.template issue<1>(nxt + 3 * P8_PIECE, kt + 1, kt1, p.K));
    // ---- P2
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fbr[j][kk] = frb.read(cur + 3 * P8_PIECE, j, kk);
    P8_FENCE();
    P8_PRE(sa.template issue<1>(nxt + 1 * P8_PIECE, kt + 1, kt1, p.K));  // AB(t+1)
    P8_WAIT();                                                            // AB(t)
    P8_MFMA(0, 2, fbr, sa.template issue<1>(nxt + 1 * P8_PIECE, kt + 1, kt1, p.K));
    // ---- P3
#pragma unroll
    for (int i = 0; i < MH; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fa[i][kk] = fra.read(cur + 1 * P8_PIECE, i, kk);
    P8_FENCE();
    P8_PRE(sb.template issue<0>(cur + 2 * P8_PIECE, kt + 2, kt1, p.K));  // BL(t+2)
    P8_MFMA(MH, 2, fbr, sb.template issue<0>(cur + 2 * P8_PIECE, kt + 2, kt1, p.K));
    // ---- P4
    P8_PRE(sa.template issue<0>(cur + 0 * P8_PIECE, kt + 2, kt1, p.K));  // AT(t+2)
    P8_WAIT();                                                            // BL(t+1), AT(t+1)
    P8_MFMA(MH, 0, fbl, sa.template issue<0>(cur + 0 * P8_PIECE, kt + 2, kt1, p.K));
  }
  if (wr == 0) P8_BARRIER();
  wait_vmcnt<0>();  // drain the trailing dummies before LDS is released
  P8_STAMP(2);

  BiasRegs<4> bias_regs;
  load_bias<4>(p, bn0, wc * 64, lane, z2, bias_regs);
  __syncthreads();  // DMA drained in every wave, all fragment reads done: LDS becomes the transposition buffer
  char* const lds_wave = smem + wave * 16384;
  gemm_epilogue<MH, 4>(p, reinterpret_cast<f32x4(&)[MH][4]>(acc[0]), bias_regs, lds_wave, bm0, bn0, wr * 2 * SEGA, wc * 64, lane, z, z1, z2, split);
  gemm_epilogue<MH, 4>(p, reinterpret_cast<f32x4(&)[MH][4]>(acc[MH]), bias_regs, lds_wave, bm0, bn0, wr * 2 * SEGA + SEGA, wc * 64, lane, z, z1, z2, split);
#ifdef P8_STAMPS
  P8_STAMP(3);
  wait_vmcnt<0>();
  P8_STAMP(4);
  if (threadIdx.x == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(hw));
    unsigned hw2;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw2));
    reinterpret_cast<unsigned long long*>(p.slab)[blockIdx.x * 8 + 5] = ((unsigned long long)hw << 32) | hw2;
  }
#endif
}

template <int MH, bool A_KM, bool B_KM>
int launch_p8(const GemmParams& p, hipStream_t st) {
  auto kern = gemm_p8_kernel<MH, A_KM, B_KM>;
  static bool attr_done = false;
  if (!attr_done) {
    SSAK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS));
    attr_done = true;
  }
  const long nblk = (long)p.tiles_m * p.tiles_n * p.nz * p.split_k;
  kern<<<dim3((unsigned)nblk), P8_THREADS, P8_LDS, st>>>(p);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <int MH>
int dispatch_p8(const GemmParams& p, int a_km, int b_km, hipStream_t st) {
  if (!a_km && !b_km) return launch_p8<MH, false, false>(p, st);
  if (!a_km && b_km) return launch_p8<MH, false, true>(p, st);
  if (a_km && b_km) return launch_p8<MH, true, true>(p, st);
  return launch_p8<MH, true, false>(p, st);
}

}  // namespace

int ssak_gemm_p8_launch(const void* params, int bm, int a_km, int b_km, hipStream_t st) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  if (bm == 256) return dispatch_p8<4>(p, a_km, b_km, st);
  if (bm == 192) return dispatch_p8<3>(p, a_km, b_km, st);
  return dispatch_p8<2>(p, a_km, b_km, st);
}
