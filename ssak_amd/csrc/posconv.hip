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
// plain store); the weight gradient (contraction over time) stays a GEMM.
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
