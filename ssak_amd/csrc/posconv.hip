// Grouped positional convolution (Wav2Vec2PositionalConvEmbedding: Conv1d(H, H, k = 128, pad = 64, groups = 16), transformers
// modeling_wav2vec2.py:326-368) as a DIRECT convolution on the matrix cores.  gfx950.
//
// As a Toeplitz GEMM (round 1-2: gemm_dma_kernel<256, 64, 4, 1, ..., NJ = 3>) every output row's A operand is the K = 128 taps x cg
// channels that follow it in the packed input, so consecutive rows share 127 / 128 of their bytes and the kernel streamed each
// input element 128 times through the CU's one global -> LDS path: 40 KB of LDS-DMA per 64-deep K step against 512 cycles of
// MFMAs, i.e. bound by the load path (0.30 of the matrix roof, 636 us per train step for forward + dX + dW).  Here a workgroup
// owns 512 output frames of one (utterance, group): the 512 + 127 input rows it needs are loaded ONCE into LDS (72 KB for cg =
// 48) and stay there; an A fragment of tap t is the fragment of tap 0 read 1 row further down -- an address, not a copy.  The
// weights of the group (cg x K cg, shared by every workgroup of the group: L2-resident) go straight from global memory into
// B-fragment registers, one tap step ahead, out of a fragment-ordered copy (1 KiB contiguous per wave-level load).  No barrier and no LDS write inside the tap loop.
//   4 waves x 128 frames, NJ = cg / 16 column groups: 8 x NJ accumulators per wave.
//   cg = 48 (base): a step = 2 taps = 96 k = 3 MFMA K-slices; 8-element chunk q = 4 kk + (lane >> 4) of the step is tap q / 6,
//     channels 8 (q % 6);  cg = 64 (XLSR-large): a step = 1 tap = 64 k = 2 slices, chunk q -> channels 8 q.
//   LDS row pitch 96 B (cg = 48) / 160 B (cg = 64): conflict-free ds_read_b128 fragments (see PcGeom::PITCH).
// Serves the forward (bias + GELU, pre-activation saved) and the input gradient (flipped taps prepared by k_posconv_prepare,
// plain store); the weight gradient (contraction over time) is posconv_wgrad_kernel further down.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int PC_TM = 512;     // output frames per workgroup
constexpr int PC_WROWS = 128;  // per wave

template <int CG>
struct PcGeom {
  static constexpr int TP = CG == 48 ? 2 : 1;   // taps per step
  static constexpr int KK = TP * CG / 32;       // MFMA K-slices per step
  static constexpr int NJ = CG / 16;            // 16-column groups
  static constexpr int CPR = CG / 8;            // 16-byte chunks per input row
  // LDS row pitch: the pitch for which a ds_read_b128 of an A fragment is conflict-free under the REAL service groups of the
  // instruction (MI355X_MICROARCH.md: lanes {0-3, 12-15, 20-27}, ... -- not 16 consecutive lanes): 96 B (no padding) for
  // cg = 48, 160 B for cg = 64; the "obvious" cg * 2 + 16 is 2-way conflicted in both cases (enumerated on the host)
  static constexpr int PITCH = CG == 48 ? 96 : 160;
};

struct PcParams {
  const bf16* x;       // packed input [G][rows_per_group][CG]
  const bf16* w;       // fragment-ordered weights (k_posconv_frag_weights): [G][K / TP][NJ][KK][64 lanes][8]
  const float* bias;   // [H] or null
  bf16* out;           // [B * F][H]
  bf16* pre;           // [B * F][H] pre-activation (forward) or null
  long rows_per_group; // packed rows per group
  long batch_rows;     // packed rows between utterances (F + K)
  int row0;            // first packed row of utterance 0's frame 0 window (0 forward, `shift` for the input gradient)
  int B, F, H, K;
  int gelu;
};

template <int CG>
__global__ __launch_bounds__(256) void posconv_direct_kernel(const PcParams p) {
  using G_ = PcGeom<CG>;
  constexpr int TP = G_::TP, KK = G_::KK, NJ = G_::NJ, CPR = G_::CPR, PITCH = G_::PITCH;
  extern __shared__ __attribute__((aligned(16))) char win[];  // [PC_TM + K - 1 (+ pad)][PITCH]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lc = lane & 15, g4 = lane >> 4;
  const int t0 = blockIdx.x * PC_TM, grp = blockIdx.y, b = blockIdx.z;
  const int nrows = PC_TM + p.K - 1;  // window rows
  // ---- the input window: packed rows [base + t0, base + t0 + nrows) of this group, loaded once
  {
    const long base = (long)grp * p.rows_per_group + (long)b * p.batch_rows + p.row0 + t0;
    const long limit = (long)(grp + 1) * p.rows_per_group;  // never read past the group's packed rows
    const bf16* src = p.x;
    for (int c = threadIdx.x; c < nrows * CPR; c += 256) {
      const int r = c / CPR, ch = c % CPR;
      const long row = base + r;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (row < limit) v = *reinterpret_cast<const uint4*>(src + row * CG + ch * 8);
      *reinterpret_cast<uint4*>(win + r * PITCH + ch * 16) = v;
    }
  }
  // per-lane offsets of the A fragments' chunks inside a step: [kk] -> (tap in step, channel chunk)
  int aoff[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const int q = 4 * kk + g4;
    aoff[kk] = (q / CPR) * PITCH + (q % CPR) * 16;
  }
  const char* arow = win + (wave * PC_WROWS + lc) * PITCH;  // row of A-fragment group 0, tap 0
  // B fragments come from a FRAGMENT-ORDERED copy of the group's weights (posconv_frag_kernel): [step][j][kk][lane][8], so a
  // wave-level load is 1 KiB contiguous.  Read straight from w[o][tap][c] the same bytes are 16 rows x 64 B -- sixteen
  // half-used cache lines per instruction -- and the loop was bound by the CU's address path: 334 us per step for forward + dX,
  // 217 without the weight loads, 226 with contiguous ones (profiles/r03_posconv_direct_ablations.log).
  const bf16* wfrag = p.w + (long)grp * CG * p.K * CG + lane * 8;
  f32x4 acc[8][NJ];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // Two register sets for the weight fragments, used alternately (the loop is unrolled by two steps): a set is loaded one
  // whole step before it is used.  (With one set copied into the other at the end of a step the compiler's wait-count
  // insertion, conservative across the back edge, made every step wait for the loads it had just issued.)
  bf16x8 b0[NJ][KK], b1[NJ][KK];
  auto load_b = [&](bf16x8 (&dst)[NJ][KK], int step) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
        dst[j][kk] = *reinterpret_cast<const bf16x8*>(wfrag + (((long)step * NJ + j) * KK + kk) * 512);
  };
  auto compute = [&](const bf16x8 (&bf)[NJ][KK], int step) {
    const char* a0 = arow + step * TP * PITCH;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      bf16x8 fa[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(a0 + i * 16 * PITCH + aoff[kk]);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][kk], fa[i], acc[i][j], 0, 0, 0);
    }
  };
  load_b(b0, 0);
  __syncthreads();  // the window is in LDS
  const int nsteps = p.K / TP;  // even (k_posconv_direct_supported)
  // (scheduling fences: left alone, the compiler sinks the loads of both sets to the end of the iteration, right in front of
  // their first use.  The last pair of steps is peeled: a conditional load inside the loop makes the wait counts of the
  // second half cover the loads it has just issued.)
  int s = 0;
  for (; s + 2 < nsteps; s += 2) {
    load_b(b1, s + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(b0, s);
    __builtin_amdgcn_sched_barrier(0);
    load_b(b0, s + 2);
    __builtin_amdgcn_sched_barrier(0);
    compute(b1, s + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  load_b(b1, s + 1);
  __builtin_amdgcn_sched_barrier(0);
  compute(b0, s);
  __builtin_amdgcn_sched_barrier(0);
  compute(b1, s + 1);
  // ---- epilogue: lane (lm = lc, lq = g4) holds C[frame 16 i + lm][channel 16 j + 4 lq + r]
  const int col0 = grp * CG + 4 * g4;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int t = t0 + wave * PC_WROWS + 16 * i + lc;
    if (t >= p.F) continue;
    const long o = ((long)b * p.F + t) * p.H + col0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + (p.bias ? p.bias[col0 + 16 * j + r] : 0.f);
      if (p.pre) {
        const bf16x4 q = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *reinterpret_cast<bf16x4*>(p.pre + o + 16 * j) = q;
      }
      if (p.gelu) {
        const f32x2 y0 = gelu2((f32x2){v[0], v[1]}), y1 = gelu2((f32x2){v[2], v[3]});
        v[0] = y0[0], v[1] = y0[1], v[2] = y1[0], v[3] = y1[1];
      }
      const bf16x4 q = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
      *reinterpret_cast<bf16x4*>(p.out + o + 16 * j) = q;
    }
  }
}

// wfrag[g][step][j][kk][lane = (lc, g4)][e] = w[g * CG + 16 j + lc][step * TP * CG + 32 kk + 8 g4 + e]   (16-byte chunks)
template <int CG>
__global__ void posconv_frag_kernel(const bf16* __restrict__ w, bf16* __restrict__ wfrag, int G, int K) {
  using G_ = PcGeom<CG>;
  const long nchunks = (long)G * CG * K * CG / 8;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(c & 63);
    long r = c >> 6;
    const int kk = (int)(r % G_::KK);
    r /= G_::KK;
    const int j = (int)(r % G_::NJ);
    r /= G_::NJ;
    const int nsteps = K / G_::TP;
    const int step = (int)(r % nsteps), g = (int)(r / nsteps);
    const int lc = lane & 15, g4 = lane >> 4;
    const bf16* src = w + ((long)g * CG + 16 * j + lc) * ((long)K * CG) + (long)step * G_::TP * CG + 32 * kk + 8 * g4;
    *reinterpret_cast<uint4*>(wfrag + c * 8) = *reinterpret_cast<const uint4*>(src);
  }
}

template <int CG>
int launch_posconv(const PcParams& p, hipStream_t st) {
  const size_t lds = (size_t)(PC_TM + p.K - 1 + 1) * PcGeom<CG>::PITCH;
  static bool attr_done = false;
  if (!attr_done) {
    SSAK_HIP(hipFuncSetAttribute((const void*)posconv_direct_kernel<CG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  posconv_direct_kernel<CG><<<dim3(ssak_cdiv(p.F, PC_TM), p.H / CG, p.B), 256, lds, st>>>(p);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

}  // namespace

// true when the direct kernel is built for this geometry (else the caller keeps the Toeplitz GEMM)
bool k_posconv_direct_supported(int H, int G, int K) {
  const int cg = G > 0 ? H / G : 0;
  if (G <= 0 || H % G || (cg != 48 && cg != 64)) return false;
  if (K < 4 || (K & (cg == 48 ? 3 : 1))) return false;  // an even number of steps (2 taps per step for cg = 48)
  return (size_t)(PC_TM + K) * (cg == 48 ? 96 : 160) <= 150 * 1024;
}

// w [H][K * cg] (k_posconv_prepare_t: forward or flipped taps) -> the fragment-ordered copy the direct kernel reads (same size)
int k_posconv_frag_weights(const bf16* w, bf16* wfrag, int H, int G, int K, hipStream_t st) {
  SSAK_REQUIRE(k_posconv_direct_supported(H, G, K), "posconv_frag_weights: H=%d G=%d K=%d not built", H, G, K);
  const long nchunks = (long)H * K * (H / G) / 8;
  const int grid = (int)std::min<long>(4096, (nchunks + 255) / 256);
  if (H / G == 48)
    posconv_frag_kernel<48><<<grid, 256, 0, st>>>(w, wfrag, G, K);
  else
    posconv_frag_kernel<64><<<grid, 256, 0, st>>>(w, wfrag, G, K);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

// x: packed [G][rows_per_group][cg] (k_posconv_pack_t), w: the FRAGMENT-ORDERED weights (k_posconv_frag_weights),
// out / pre: [B * F, H] bf16.  row0: 0 for the forward, `shift` (1 for even K) for the input gradient.
int k_posconv_direct(const bf16* x, long rows_per_group, int row0, const bf16* w, const float* bias, bf16* out, bf16* pre, int B, int F,
                     int H, int G, int K, int gelu, hipStream_t st) {
  SSAK_REQUIRE(k_posconv_direct_supported(H, G, K), "posconv_direct: H=%d G=%d K=%d not built", H, G, K);
  PcParams p{x, w, bias, out, pre, rows_per_group, (long)F + K, row0, B, F, H, K, gelu};
  ProfScope prof_scope(PROF_POSCONV_DIRECT, 2.0 * B * F * (double)H * K * (H / G), st);
  return H / G == 48 ? launch_posconv<48>(p, st) : launch_posconv<64>(p, st);
}

namespace {

// =====================================================================================================================
// Weight gradient of the grouped positional convolution as a direct contraction over time (round 3, second half):
//   dwf[g][tap * cg + c][n] = sum over packed rows r of x[g][r + tap][c] * dy[g][lead + r][n]
// As a Toeplitz GEMM (gemm_dma_kernel<128, 64, 4, 1, true, true, 64, 3>: M = K cg = 6 144 rows per group, N = cg, the
// contraction over B (F + K) = 20 k rows) the A operand of tap t + 1 is the A operand of tap t one row further down and every
// input row went through the global -> LDS path 128 times: 254 us per step at 0.30 of the matrix roof.  Here a workgroup owns
// TT = 16 (cg = 48) / 8 (cg = 64) taps of one group and a quarter of the rows; a stage of 128 rows (+ TT - 1) of x and dy is
// written to LDS once and the A fragment of every tap is a `ds_read_b64_tr_b16` of the SAME image at a row offset.  Both
// operands are contracted over their memory rows, so both are transposing reads; the image (128-byte rows, the 32-byte column
// block cb of row r at slot cb ^ ((r >> 1) & 1 | ((r >> 3) & 1) << 1)) is conflict-free for every row offset (enumerated on
// the host under the instruction's 2 x 32 lane groups).  4 waves x TW taps x cg channels x cg outputs: 36 / 32 accumulator
// tiles per wave; the four row quarters write fp32 partials that a second kernel adds in a fixed order.
constexpr int PW_RB = 128;    // contraction rows per stage
constexpr int PW_SPLIT = 4;   // row ranges (partials)
template <int CG>
struct PwGeom {
  static constexpr int CB = CG / 16;              // 16-column blocks of a row (= 32-byte slots)
  static constexpr int TW = CG == 48 ? 4 : 2;     // taps per wave
  static constexpr int TT = 4 * TW;               // taps per workgroup
  static constexpr int XR = PW_RB + TT - 1;       // x rows per stage
  static constexpr int CPR = CG / 8;              // 16-byte chunks per packed row
  static constexpr int NCH = (XR + PW_RB) * CPR;  // chunks per stage
  static constexpr int PER_T = (NCH + 255) / 256;
};
struct PwParams {
  const bf16* x;   // packed input [G][rows_per_group][CG]
  const bf16* dy;  // packed output gradient, same layout
  float* part;     // [PW_SPLIT][G][K * CG][CG]
  long rows_per_group;
  long rows;       // contraction rows: B (F + K)
  int lead, K, G;
};
__device__ __forceinline__ int pw_swz(int r) { return ((r >> 1) & 1) | (((r >> 3) & 1) << 1); }

template <int CG>
__global__ __launch_bounds__(256, 2) void posconv_wgrad_kernel(const PwParams p) {
  using G_ = PwGeom<CG>;
  constexpr int CB = G_::CB, TW = G_::TW, TT = G_::TT, CPR = G_::CPR;
  constexpr int XP = (G_::XR + 7) / 8, YP = PW_RB / 8;  // 1-KiB pieces (8 image rows) of the x and dy images
  constexpr int XROWS = 8 * XP;                          // image rows reserved for x (the dy image follows)
  constexpr int BUF = (XROWS + PW_RB) * 128;
  extern __shared__ __attribute__((aligned(16))) char img[];  // 2 stages x (x rows | dy rows), 128-byte rows
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tb = blockIdx.x, grp = blockIdx.y, split = blockIdx.z;
  const long per_split = ((p.rows + PW_RB - 1) / PW_RB + PW_SPLIT - 1) / PW_SPLIT * PW_RB;
  const long r_lo = (long)split * per_split, r_hi = min(p.rows, r_lo + per_split);
  const uint32_t ext = (uint32_t)(p.rows_per_group * CG * 2);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (long)grp * p.rows_per_group * CG), 0, (int)ext, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dy + (long)grp * p.rows_per_group * CG), 0, (int)ext, 0x00020000);
  typedef __attribute__((address_space(3))) void lds_void_;
  // LDS-DMA writes a piece lane-linearly (lane l -> byte 16 l of the piece = row l / 8, 16-byte slot l % 8), so the swizzle is
  // applied to the SOURCE: slot -> which chunk of the packed row it holds (none for the two padding slots of a 48-channel row).
  // The swizzle has period 16 rows: two source patterns, for even and odd pieces.
  uint32_t src_off[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int r = 8 * par + (lane >> 3), s16 = lane & 7;
    const int ch = 2 * ((s16 >> 1) ^ pw_swz(r)) + (s16 & 1);
    src_off[par] = ch < CPR ? (uint32_t)(((lane >> 3) * CG + ch * 8) * 2) : 0x80000000u;
  }
  // stage the rows of contraction block rb into buffer `buf`: pieces wave, wave + 4, ... (x pieces first, then dy)
  auto stage = [&](char* buf, long rb) {
    const long x0 = rb + (long)tb * TT, y0 = (long)p.lead + rb;
#pragma unroll
    for (int i = 0; i < (XP + YP + 3) / 4; ++i) {
      const int pc = wave + 4 * i;
      if (pc < XP + YP) {
        const bool is_x = pc < XP;
        const int ip = is_x ? pc : pc - XP;  // piece inside its image
        const long row0 = (is_x ? x0 : y0) + 8 * ip;
        const uint32_t so = src_off[ip & 1];
        // rows beyond the group's packed rows read as zeros through the descriptor's bounds check (32-bit offsets: < 4 MB here)
        const uint32_t o = so == 0x80000000u ? so : (uint32_t)(row0 * CG * 2) + so;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(is_x ? xr : yr, (lds_void_*)(buf + (is_x ? 0 : XROWS * 128) + ip * 1024), 16, o, 0, 0, 0);
      }
    }
  };
  // per-lane byte offsets of the transposing reads: row 8 g + q (+ 4 for the second read) of a 32-row step, columns 4 p .. 4 p + 3
  // of column block cb: row part + ((cb << 5) ^ swizzle part) -- the swizzle has period 16 in the row, so a step adds 4 096
  const int g4 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  typedef __attribute__((address_space(3))) char lds_char;
  const uint32_t img_a = (uint32_t)(uintptr_t)(lds_char*)img;
  uint32_t xrow[TW][2], xsw[TW][2], yrow[2], ysw[2];
#pragma unroll
  for (int hi = 0; hi < 2; ++hi) {
    const int kr = 8 * g4 + q4 + 4 * hi;
    yrow[hi] = img_a + (XROWS + kr) * 128 + 8 * p4;
    ysw[hi] = pw_swz(kr) << 5;
#pragma unroll
    for (int tw = 0; tw < TW; ++tw) {
      const int row = kr + wave * TW + tw;
      xrow[tw][hi] = img_a + row * 128 + 8 * p4;
      xsw[tw][hi] = pw_swz(row) << 5;
    }
  }
  f32x4 acc[TW * CB][CB];
#pragma unroll
  for (int i = 0; i < TW * CB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  struct Frag {
    s16x4 lo, hi;
  };
  // one fragment = two transposing reads (inline asm: untracked by the compiler; the caller counts lgkmcnt)
#define PW_TR(F, ROW, SW, CBI, OFF)                                                                               \
  do {                                                                                                            \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(F.lo) : "v"(ROW[0] + ((uint32_t)((CBI) << 5) ^ SW[0])), "n"(OFF) : "memory"); \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(F.hi) : "v"(ROW[1] + ((uint32_t)((CBI) << 5) ^ SW[1])), "n"(OFF) : "memory"); \
  } while (0)
  auto join = [](const Frag& f) { return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7)); };

  if (r_lo < r_hi) stage(img, r_lo);
  int cur = 0;
  for (long rb = r_lo; rb < r_hi; rb += PW_RB, cur ^= 1) {
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's pieces of the stage have landed
    __syncthreads();                      // ... everyone's; and every wave is done with the other buffer
    if (rb + PW_RB < r_hi) stage(img + (cur ^ 1) * BUF, rb + PW_RB);  // in flight under this stage's MFMAs
    // units (step, tw) of the stage, software-pipelined: the fragments of unit u + 1 are read while unit u multiplies
    Frag fy[2][CB], fx[2][CB];
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      PW_TR(fy[0][j], yrow, ysw, j, 0);
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      PW_TR(fx[0][cb], xrow[0], xsw[0], cb, 0);
    }
#pragma unroll
    for (int u = 0; u < 4 * TW; ++u) {
      const int step = u / TW, tw = u % TW;
      const int nu = u + 1, nstep = nu / TW, ntw = nu % TW;
      int issued = 0;
      if (nu < 4 * TW) {
        if (ntw == 0) {
#pragma unroll
          for (int j = 0; j < CB; ++j) {
            PW_TR(fy[nstep & 1][j], yrow, ysw, j, (nu / TW) * 4096);
          }
          issued += 2 * CB;
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          PW_TR(fx[nu & 1][cb], xrow[nu % TW], xsw[nu % TW], cb, (nu / TW) * 4096);
        }
        issued += 2 * CB;
      }
      // everything but the reads just issued has returned (LDS returns in order): lgkmcnt(issued)
      if (issued == 0) __builtin_amdgcn_s_waitcnt(0xc07f);
      else if (issued == 2 * CB) __builtin_amdgcn_s_waitcnt(0xc07f | ((2 * CB) << 8));
      else __builtin_amdgcn_s_waitcnt(0xc07f | ((4 * CB) << 8));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int j = 0; j < CB; ++j)
          acc[tw * CB + cb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join(fy[step & 1][j]), join(fx[u & 1][cb]), acc[tw * CB + cb][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // the read addresses follow the stage buffer
    const uint32_t d = cur ? (uint32_t)-BUF : (uint32_t)BUF;
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
      yrow[hi] += d;
#pragma unroll
      for (int tw = 0; tw < TW; ++tw) xrow[tw][hi] += d;
    }
  }
#undef PW_TR
  // ---- partial: lane (c = lane & 15, n = 4 (lane >> 4) + r) of block (tap, cb) x j
  float* out = p.part + (((long)split * p.G + grp) * p.K * CG) * CG;
#pragma unroll
  for (int tw = 0; tw < TW; ++tw)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const long row = (long)(tb * TT + wave * TW + tw) * CG + 16 * cb + (lane & 15);
#pragma unroll
      for (int j = 0; j < CB; ++j) *reinterpret_cast<f32x4*>(out + row * CG + 16 * j + 4 * g4) = acc[tw * CB + cb][j];
    }
}

__global__ void posconv_wgrad_sum_kernel(const float* __restrict__ part, float* __restrict__ dwf, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 s = reinterpret_cast<const f32x4*>(part)[i];
#pragma unroll
    for (int k = 1; k < PW_SPLIT; ++k) s += reinterpret_cast<const f32x4*>(part)[i + k * n4];
    reinterpret_cast<f32x4*>(dwf)[i] = s;
  }
}

}  // namespace

size_t k_posconv_wgrad_scratch_floats(int H, int G, int K) { return (size_t)PW_SPLIT * H * K * (H / G); }
// x, dy: packed [G][rows_per_group][cg] (k_posconv_pack_t); dwf [G][K * cg][cg] fp32 (what k_posconv_weight_bwd reads);
// scratch >= k_posconv_wgrad_scratch_floats floats.  rows = B (F + K) contraction rows, dy read `lead` rows further down.
int k_posconv_wgrad_direct(const bf16* x, const bf16* dy, long rows_per_group, long rows, int lead, float* dwf, float* scratch, int H, int G,
                           int K, hipStream_t st) {
  SSAK_REQUIRE(k_posconv_direct_supported(H, G, K), "posconv_wgrad_direct: H=%d G=%d K=%d not built", H, G, K);
  const int cg = H / G;
  PwParams p{x, dy, scratch, rows_per_group, rows, lead, K, G};
  ProfScope prof_scope(PROF_POSCONV_WGRAD, 2.0 * rows * (double)H * K * cg, st);
  if (cg == 48) {
    SSAK_REQUIRE(K % PwGeom<48>::TT == 0, "posconv_wgrad_direct: K must be a multiple of %d", PwGeom<48>::TT);
    constexpr size_t lds = 2 * (size_t)((PwGeom<48>::XR + 7) / 8 * 8 + PW_RB) * 128;
    static bool attr48 = false;
    if (!attr48) {
      SSAK_HIP(hipFuncSetAttribute((const void*)posconv_wgrad_kernel<48>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr48 = true;
    }
    posconv_wgrad_kernel<48><<<dim3(K / PwGeom<48>::TT, G, PW_SPLIT), 256, lds, st>>>(p);
  } else {
    SSAK_REQUIRE(K % PwGeom<64>::TT == 0, "posconv_wgrad_direct: K must be a multiple of %d", PwGeom<64>::TT);
    constexpr size_t lds = 2 * (size_t)((PwGeom<64>::XR + 7) / 8 * 8 + PW_RB) * 128;
    static bool attr64 = false;
    if (!attr64) {
      SSAK_HIP(hipFuncSetAttribute((const void*)posconv_wgrad_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr64 = true;
    }
    posconv_wgrad_kernel<64><<<dim3(K / PwGeom<64>::TT, G, PW_SPLIT), 256, lds, st>>>(p);
  }
  SSAK_LAUNCH_CHECK();
  const long n4 = (long)H * K * cg / 4;
  posconv_wgrad_sum_kernel<<<(int)std::min<long>(2048, (n4 + 255) / 256), 256, 0, st>>>(scratch, dwf, n4);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
