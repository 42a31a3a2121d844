// Fused multi-head self-attention (head_dim 64), forward and backward.  gfx950.
//
// Stands behind Wav2Vec2Attention / eager_attention_forward (transformers modeling_wav2vec2.py:438-463,500-548) and
// WhisperAttention inside the encoder layers: softmax(Q K^T * d^-0.5 + key mask) -> dropout -> . V, per (utterance,
// head), and its autograd.  The unfused path (GEMM -> softmax kernel -> GEMM) moved every [frames x frames] score
// matrix through HBM four times per layer; here scores live only in MFMA accumulators:
//
//  forward : one workgroup = 128 queries of one (b, h); 4 waves x 32 queries.  K/V tiles of 64 keys stream through
//            LDS by LDS-DMA (double buffered, source-side swizzle).  S^T = K Q^T is computed with the KEY on the
//            accumulator row and the QUERY on the lane, so the online-softmax statistics (running max / sum) are
//            per-lane scalars and the probabilities are already the B operand of O^T = V^T P^T (the k order inside an
//            MFMA step is permuted identically on both operands).  Only O and the log-sum-exp per row are written.
//  backward: recompute P from Q, K and the saved log-sum-exp (flash-attention style), two kernels so that no
//            gradient needs cross-workgroup atomics: dQ (workgroup owns 128 queries, sweeps keys; same orientation
//            as the forward) and dK/dV (workgroup owns 128 keys, sweeps queries; S = Q K^T orientation so that P and
//            dS are the B operands of dV^T = dO^T P and dK^T = Q^T dS).
// Dropout bits (common.h): word(row, key) = rowkey(seed, stream, (b * nh + h) * F + q) * colmul(key) mod 2^32, keep iff
// word >= thresh16 << 16 -- one integer multiply, one compare and one select per score element; the forward and both
// backward kernels regenerate identical masks.  Key-padding masks (ragged batches) are applied as -inf before the
// softmax, as create_bidirectional_mask does.
#include <cstdio>
#include <cstdlib>
#include <utility>

#include "kernels.h"

namespace {

int attn_tile(int which);  // 16-row sub-tiles per wave: 0 forward (queries), 1 dQ kernel (queries), 2 dK/dV kernel (keys)

constexpr int HD = 64;    // head dim
constexpr int KT = 64;    // keys (or queries in dkv) per streamed tile
constexpr int TILE_BYTES = KT * HD * 2;  // 8 KiB
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

struct AttnParams {
  const bf16* qkv;     // [B*F, 3H]: q | k | v
  bf16* ctx;           // [B*F, H]    forward output
  float* lse;          // [B, nh, F]  log-sum-exp of the scaled, masked scores
  const int32_t* klens;
  const bf16* dctx;    // [B*F, H]    backward input
  float* delta;        // [B, nh, F]  rowsum(dO * O)
  bf16* dqkv;          // [B*F, 3H]   backward output
  float* bias_part;    // [B * blocks, 3H] or null: column sums of this workgroup's rows of dqkv (the q|k|v bias gradient's first stage)
  int B, F, nh, H, Fp;
  float scale;
  uint64_t seed;
  uint32_t stream, thresh16;
  float drop_scale;
};

__device__ __forceinline__ float shfl_xor_f(float v, int m) { return __shfl_xor(v, m, 64); }
// max over the four 16-lane groups of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48), the result in all of them: two VALU swaps
// (v_permlane16_swap / v_permlane32_swap of the value with a copy of itself leave {r0 r0 r2 r2 | r1 r1 r3 r3} and
// {lo lo | hi hi}) instead of two ds_bpermute round trips through the LDS crossbar in the softmax's dependent chain
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
__device__ __forceinline__ float max_over_lane_groups(float v) {
  const unsigned b = __builtin_bit_cast(unsigned, v);
  u32x2_t t = __builtin_amdgcn_permlane16_swap(b, b, false, false);
  float m;  // (asm: no canonicalising v_max x, x in front)
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(__builtin_bit_cast(float, t[0])), "v"(__builtin_bit_cast(float, t[1])));
  const unsigned c = __builtin_bit_cast(unsigned, m);
  t = __builtin_amdgcn_permlane32_swap(c, c, false, false);
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(__builtin_bit_cast(float, t[0])), "v"(__builtin_bit_cast(float, t[1])));
  return r;
}
// max of two without the canonicalising v_max x, x that fmaxf puts in front of values it cannot prove canonical (MFMA results,
// lane swaps): the scores are never NaN
__device__ __forceinline__ float max2_nc(float a, float b) {
  float d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// max of three without the canonicalising v_max x, x that fmaxf puts in front of every MFMA result in IEEE mode (the scores
// are never NaN): 8 instructions for 16 values instead of 31
__device__ __forceinline__ float max3_nc(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// a value the compiler cannot hoist across the branch it is used under (keeps the per-key comparisons of the one tile that
// crosses the key length out of every other tile's instruction stream)
__device__ __forceinline__ int opaque_s(int v) {
  asm volatile("" : "+s"(v));
  return v;
}

// Dropout bits.  Every (utterance, head, query) row has a key = hash(seed, stream, row index) | 1, computed once per row and
// kernel; every key index has a multiplier colmul(key) = hash(key) | 1 (a function of the key alone).  Kernels with the key on
// the accumulator rows (forward, dQ) keep the multipliers of all keys in LDS behind their tiles and read four consecutive ones
// per `ds_read_b128`; the dK / dV kernel has the key on the lane and keeps its multipliers in registers.  Per element: one
// `v_mul_lo_u32` (full rate on gfx950), one compare against thresh16 << 16, one select.  Rounds 2-4 spent 5.5 VALU per element
// on a word per key PAIR (add, xor-shift, multiply, xor-shift, two field tests); statistics of this form: common.h.
__device__ __forceinline__ uint32_t drop_rowseed(const AttnParams& p, int b, int h, int q) {
  return drop_rowkey(p.seed, p.stream, (uint64_t)((uint32_t)(b * p.nh + h) * (uint32_t)p.F + (uint32_t)min(q, p.F - 1)));
}
// the multipliers of keys [0, n) into LDS (n a multiple of 64; visible after the next workgroup barrier)
__device__ __forceinline__ void fill_colmul(uint32_t* cmt, int n) {
  for (int k = threadIdx.x; k < n; k += 256) cmt[k] = drop_colmul((uint32_t)k);
}

// LDS tile images (64 rows x 128 B each):
//   row-read image  : chunk c of row r at c ^ ((r >> 1) & 7)      -> ds_read_b128 fragments (row on the lane)
//   transpose image : chunk c of row r at c ^ (r & 6)             -> ds_read_b64_tr_b16 fragments (column on the lane)
//   dual image      : = the transpose image.  `ds_read_b128` is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19,
//                     28-31}, ... (MI355X_MICROARCH.md, LDS), not in runs of eight lanes, and under THOSE groups the transpose
//                     image is conflict-free for row reads as well (enumerated on the host: 0 extra cycles for both kinds of
//                     fragment; the rotated swizzle c ^ rotl3((r >> 1) & 7) this image used through round 2 was designed for
//                     contiguous groups and was two-way conflicted on every row read: SQ_LDS_BANK_CONFLICT = 26 % / 40 % of
//                     the LDS cycles of the dK-dV / dQ kernels, profiles/r03_pmc_sq.json).  One image per tensor instead of
//                     two: a wave issues an LDS-DMA instruction only every ~110 cycles, and the backward kernels staged 6 (dQ)
//                     and 8 (dK / dV) per wave and tile.
#ifdef ATT_OLD_DUAL  // A/B: the rotated swizzle of rounds 1-2
__device__ __forceinline__ int dual_swz(int r) {
  const int x = (r >> 1) & 7;
  return ((x << 1) & 7) | (x >> 2);
}
#else
__device__ __forceinline__ int dual_swz(int r) { return r & 6; }
#endif
__device__ __forceinline__ void dma_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, uint32_t col_byte, long ld_bytes,
                                         int row0, int nrows_total, int image /* 0 row-read, 1 transpose, 2 dual */, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int S = (wave * 2 + j) * 64 + lane;
    const int r = S >> 3, pc = S & 7;
    const int c = image == 2 ? (pc ^ dual_swz(r)) : image == 1 ? (pc ^ (r & 6)) : (pc ^ ((r >> 1) & 7));
    const int gr = row0 + r;
    const uint32_t off = gr < nrows_total ? (uint32_t)((long)gr * ld_bytes + col_byte + c * 16) : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(lds_tile + (wave * 2 + j) * 1024), 16, off, 0, 0, 0);
  }
}

// The same staging with the per-lane offsets computed ONCE and the tile's first row carried by the buffer descriptor: per tile the
// address work is scalar (base += row0 rows, records -= the same) -- dma_tile() spends an add, a compare and a select per instruction
// and tile on the VALU (12 per key tile in the forward / dQ kernels, 12 in dK / dV), in kernels whose time IS their VALU count.  Rows
// beyond the tensor fall outside the shrunken descriptor and come back as zeros, as before.
struct DmaOff {
  uint32_t o[2];
};
__device__ __forceinline__ DmaOff dma_offsets(uint32_t col_byte, long ld_bytes, int image, int wave, int lane) {
  DmaOff d;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int S = (wave * 2 + j) * 64 + lane;
    const int r = S >> 3, pc = S & 7;
    const int c = image == 2 ? (pc ^ dual_swz(r)) : image == 1 ? (pc ^ (r & 6)) : (pc ^ ((r >> 1) & 7));
    d.o[j] = (uint32_t)((long)r * ld_bytes + col_byte + c * 16);
  }
  return d;
}
// descriptor of rows [row0, nrows) of a [nrows, ld_bytes] tensor (row0 uniform: scalar arithmetic)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_rows(const void* base, long ld_bytes, int row0, int nrows) {
  const long skip = (long)row0 * ld_bytes;
  return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + skip), 0, (int)((long)max(nrows - row0, 0) * ld_bytes), 0x00020000);
}
__device__ __forceinline__ void dma_tile_at(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, const DmaOff& d, int wave) {
#pragma unroll
  for (int j = 0; j < 2; ++j)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(lds_tile + (wave * 2 + j) * 1024), 16, d.o[j], 0, 0, 0);
}

// row-read fragment: rows 16*sub + (lane&15), k = 32*kk + 8*(lane>>4) + j
__device__ __forceinline__ bf16x8 frag_rows(const char* tile, int sub, int kk, int lane) {
  const int r = 16 * sub + (lane & 15);
  const int c = kk * 4 + (lane >> 4);
  return *reinterpret_cast<const bf16x8*>(tile + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
}
// transpose fragment from the transpose image: A[row = column 16*ci + (lane&15) of the tile][k = tile rows]:
//   element j < 4: tile row rbase + 4*(lane>>4) + j, j >= 4: tile row rbase + 16 + 4*(lane>>4) + (j-4)
// (this is the k permutation of an accumulator tile used as the other operand)
__device__ __forceinline__ bf16x8 frag_cols_perm(const char* tile, int ci, int rbase, int lane) {
  const int g = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  const int r = rbase + 4 * g + q4;
  const int ch = 2 * ci + (p4 >> 1);
  const char* a = tile + r * 128 + ((ch ^ (r & 6)) << 4) + (p4 & 1) * 8;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(a + 16 * 128));
  s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// the same two fragments from a dual image
__device__ __forceinline__ bf16x8 frag_rows_d(const char* tile, int sub, int kk, int lane) {
  const int r = 16 * sub + (lane & 15);
  const int c = kk * 4 + (lane >> 4);
  return *reinterpret_cast<const bf16x8*>(tile + r * 128 + ((c ^ dual_swz(r)) << 4));
}
__device__ __forceinline__ bf16x8 frag_cols_perm_d(const char* tile, int ci, int rbase, int lane) {
  const int g = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  const int r = rbase + 4 * g + q4;
  const int ch = 2 * ci + (p4 >> 1);
  const char* a = tile + r * 128 + ((ch ^ dual_swz(r)) << 4) + (p4 & 1) * 8;  // row + 16: the same swizzle
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(a + 16 * 128));
  s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// (as vector conversions: element-by-element casts compiled to one v_cvt_pk_bf16_f32 PER ELEMENT plus a v_perm_b32 per pair --
// 48 instructions per 32 score elements where 16 do)
__device__ __forceinline__ bf16x8 pack_p(const f32x4& a, const f32x4& b) {
  const bf16x4 lo = __builtin_convertvector(a, bf16x4), hi = __builtin_convertvector(b, bf16x4);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8 load_row_frag(const bf16* base, long ld, int row, int nrows, int kk, int lane) {
  // global -> register fragment: row `row`, k = 32*kk + 8*(lane>>4) + j   (zero beyond nrows)
  bf16x8 z = {};
  if (row >= nrows) return z;
  return *reinterpret_cast<const bf16x8*>(base + (long)row * ld + 32 * kk + 8 * (lane >> 4));
}

// Workgroup -> (batch, head, block) with the blocks of one (batch, head) on ONE XCD.  The hardware deals consecutive
// workgroup ids round-robin over the 8 XCDs, each with its own L2: with the plain (block, head, batch) grid order the 4
// query blocks of a head landed on 4 different XCDs and each fetched that head's K and V from HBM again (rocprofv3
// FETCH_SIZE: 221 MB per forward launch for 98 MB of operands).  Here XCD x takes the heads bh = x (mod 8) and walks their
// blocks back to back, so the first block's fetches serve the others from L2.  Identity when batch * heads is not a
// multiple of 8.
struct AttnBlock {
  int b, h, blk;
};
__device__ __forceinline__ AttnBlock attn_block() {
  const int nblk = gridDim.x, nh = gridDim.y, nbh = gridDim.y * gridDim.z;
  AttnBlock r;
  if ((nbh & 7) != 0) {
    r.b = blockIdx.z, r.h = blockIdx.y, r.blk = blockIdx.x;
    return r;
  }
  const int L = blockIdx.x + nblk * (blockIdx.y + nh * blockIdx.z);
  const int xcd = L & 7, j = L >> 3;
  const int bh = (j / nblk) * 8 + xcd;
  r.blk = j % nblk;
  r.b = bh / nh;
  r.h = bh % nh;
  return r;
}

// Column sums of a workgroup's output rows, for the q|k|v bias gradient (the separate column-sum pass read all of dqkv again:
// 74 MB per layer).  `acc` is an output tile in the transposed orientation all three gradients use (row = d = 16 i + 4 g + r on
// the registers, output row on lane & 15), already rounded to bf16 like the stored values and zero for rows that are not stored.
// 16-lane DPP reductions, the four waves through LDS (the streamed tiles are dead: one barrier first), one float per d written to
// the workgroup's slot -- no atomics, fixed summation order.
template <int NS>
__device__ __forceinline__ void bias_partial(float (&cs)[NS][4][4], char* smem, float* dst /* slot row + column base */, int set_stride,
                                             int wave, int lane) {
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = cs[s][i][r];  // DPP steps at VALU rate (a __shfl_xor butterfly goes through ds_bpermute: +8 us on the dK / dV launch)
        v += dpp_f<0xB1, 0xf>(0.f, v);   // quad_perm [1,0,3,2]
        v += dpp_f<0x4E, 0xf>(0.f, v);   // quad_perm [2,3,0,1]
        v += dpp_f<0x141, 0xf>(0.f, v);  // row_half_mirror
        v += dpp_f<0x140, 0xf>(0.f, v);  // row_mirror: every lane holds its 16-lane row total
        cs[s][i][r] = v;
      }
  float* red = reinterpret_cast<float*>(smem);  // [set][wave][64 d]
  // raw barriers with an LDS-only wait: __syncthreads() is also a fence, i.e. `s_waitcnt vmcnt(0)` -- every wave sat out the
  // round trip of the output rows it had just stored (4 us per kernel: the bias sums cost 8 us per layer, measured standalone)
  lds_barrier();
  if ((lane & 15) == 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(red + (s * 4 + wave) * HD + 16 * i + 4 * (lane >> 4)) = (f32x4){cs[s][i][0], cs[s][i][1], cs[s][i][2], cs[s][i][3]};
  }
  lds_barrier();
  if (wave < NS) {  // wave s adds up set s
    const float* rs = red + wave * 4 * HD;
    dst[wave * set_stride + lane] = (rs[lane] + rs[HD + lane]) + (rs[2 * HD + lane] + rs[3 * HD + lane]);
  }
}
__device__ __forceinline__ float bf16_round(float v) { return (float)(bf16)v; }

// ================================================================================================ forward
typedef __attribute__((ext_vector_type(2))) float f32x2;

// DROP is a template parameter: as a run-time test it put a branch (and two register copies to merge its sides) around every
// key pair's dropout words -- sixteen per query sub-tile and key tile -- which also kept the exp / hash / MFMA streams apart.
// NQS = 16-query sub-tiles per wave (workgroup = 4 waves x 16 NQS queries).
template <bool DROP, int NQS>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x (K | V)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const AttnBlock ab = attn_block();
  const int b = ab.b, h = ab.h, q0 = ab.blk * (64 * NQS);
  const int F = p.F, H = p.H;
  const long ld = 3L * H;
  const bf16* base = p.qkv + (long)b * F * ld;
  const int kl = p.klens ? min(max(p.klens[b], 0), F) : F;
  const int nkt = (kl + KT - 1) / KT;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((long)F * ld * 2), 0x00020000);
  const uint32_t kcol = (uint32_t)((H + h * HD) * 2), vcol = (uint32_t)((2 * H + h * HD) * 2);

  // this lane's two query rows (one per 16-row sub-tile) and their Q fragments
  int qrow[NQS];
  bf16x8 qf[NQS][2];
#pragma unroll
  for (int qs = 0; qs < NQS; ++qs) {
    qrow[qs] = q0 + 16 * NQS * wave + 16 * qs + (lane & 15);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qf[qs][kk] = load_row_frag(base + h * HD, ld, qrow[qs], F, kk, lane);
  }
  float m_run[NQS], l_run[NQS];
#pragma unroll
  for (int qs = 0; qs < NQS; ++qs) {
    m_run[qs] = -INFINITY;
    l_run[qs] = 0.f;
  }
  const float c2 = p.scale * 1.4426950408889634f;  // softmax scale x log2(e)
  const uint32_t thi = p.thresh16 << 16;
  uint32_t rowseed[NQS];
#pragma unroll
  for (int qs = 0; qs < NQS; ++qs) rowseed[qs] = drop_rowseed(p, b, h, qrow[qs]);
  const uint32_t* cmt = reinterpret_cast<const uint32_t*>(smem + 2 * 2 * TILE_BYTES);  // [nkt * 64] key multipliers (DROP)
  if (DROP) fill_colmul(reinterpret_cast<uint32_t*>(smem + 2 * 2 * TILE_BYTES), nkt * KT);
  f32x4 oacc[NQS][4];
#pragma unroll
  for (int qs = 0; qs < NQS; ++qs)
#pragma unroll
    for (int i = 0; i < 4; ++i) oacc[qs][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const DmaOff koff = dma_offsets(kcol, ld * 2, 0, wave, lane), voff = dma_offsets(vcol, ld * 2, 1, wave, lane);
  if (nkt > 0) {
    dma_tile_at(rsrc, smem, koff, wave);
    dma_tile_at(rsrc, smem + TILE_BYTES, voff, wave);
  }
  const int g = lane >> 4;
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    __syncthreads();
    if (kt + 1 < nkt) {
      const __amdgpu_buffer_rsrc_t rn = rsrc_rows(base, ld * 2, (kt + 1) * KT, F);
      dma_tile_at(rn, smem + (cur ^ 1) * 2 * TILE_BYTES, koff, wave);
      dma_tile_at(rn, smem + (cur ^ 1) * 2 * TILE_BYTES + TILE_BYTES, voff, wave);
    }
    const char* kt_lds = smem + cur * 2 * TILE_BYTES;
    const char* vt_lds = kt_lds + TILE_BYTES;
    const int k0 = kt * KT;
    // ---- S^T = K Q^T : rows = keys (16*ks + 4g + r), column = this lane's query
    f32x4 s[NQS][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 ka = frag_rows(kt_lds, ks, 0, lane), kb = frag_rows(kt_lds, ks, 1, lane);
#pragma unroll
      for (int qs = 0; qs < NQS; ++qs) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[qs][0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb, qf[qs][1], a, 0, 0, 0);
        s[qs][ks] = a;
      }
    }
    // ---- online softmax per query (lane-local column), dropout, pack P^T as the B operand of the PV product.
    // Scores stay raw: the softmax scale and log2(e) are folded into one fma in front of v_exp_f32; keys are masked only
    // in the tile that crosses the key length; the dropout scale is applied once, to O, at the end.
    bf16x8 pb[NQS][2];
    const bool edge = k0 + KT > kl;  // uniform
    u32x4_t cm[4];  // multipliers of this lane's keys k0 + 16 ks + 4 g + r (shared by the query sub-tiles)
    if (DROP) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) cm[ks] = *reinterpret_cast<const u32x4_t*>(cmt + k0 + 16 * ks + 4 * g);
    }
#pragma unroll
    for (int qs = 0; qs < NQS; ++qs) {
      if (edge) {
        const int tl = opaque_s(kl) - k0 - 4 * g;  // keys 16 ks + r of this lane's column at or beyond it are padding
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * ks + r >= tl) s[qs][ks][r] = -INFINITY;
      }
      float mx = max3_nc(s[qs][0][0], s[qs][0][1], s[qs][0][2]);
      mx = max3_nc(mx, s[qs][0][3], s[qs][1][0]);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) {
        mx = max3_nc(mx, s[qs][ks][1], s[qs][ks][2]);
        if (ks < 3)
          mx = max3_nc(mx, s[qs][ks][3], s[qs][ks + 1][0]);
        else
          mx = max2_nc(mx, s[qs][ks][3]);
      }
      mx = max_over_lane_groups(mx);
      // key 0 is never masked (kl >= 1 whenever a tile is processed), so the running maximum is finite from the first tile on
      const float m_new = max2_nc(m_run[qs], mx);
      const float mc = m_new * c2;
      const float alpha = __builtin_amdgcn_exp2f(fmaf(m_run[qs], c2, -mc));  // first tile: exp2(-inf) = 0 on a zero accumulator
      f32x2 rs2 = {0.f, 0.f};  // packed fp32 (v_pk_fma_f32 / v_pk_add_f32): the pair's two exponent arguments and the row sum
      const f32x2 c2v = {c2, c2}, mcv = {-mc, -mc};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int r2 = 0; r2 < 4; r2 += 2) {
          const f32x2 a = (f32x2){s[qs][ks][r2], s[qs][ks][r2 + 1]} * c2v + mcv;
          f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
          rs2 += e;
          if (DROP) {
            e[0] = drop_keep(rowseed[qs], cm[ks][r2], thi) ? e[0] : 0.f;
            e[1] = drop_keep(rowseed[qs], cm[ks][r2 + 1], thi) ? e[1] : 0.f;
          }
          s[qs][ks][r2] = e[0];
          s[qs][ks][r2 + 1] = e[1];
        }
      }
      // (the row sum stays a per-lane partial over this lane's keys -- alpha is the same in the four lanes of a query -- and is
      // added up across them once, in the epilogue)
      l_run[qs] = l_run[qs] * alpha + (rs2[0] + rs2[1]);
      m_run[qs] = m_new;
      if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {  // the maximum moves in the first tiles only: skip 16 multiplies otherwise
#pragma unroll
        for (int i = 0; i < 4; ++i) oacc[qs][i] *= alpha;
      }
      pb[qs][0] = pack_p(s[qs][0], s[qs][1]);
      pb[qs][1] = pack_p(s[qs][2], s[qs][3]);
    }
    // ---- O^T += V^T P^T : rows = d (16*i + 4g + r), column = query; k = keys in the permuted order of pb
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8 va = frag_cols_perm(vt_lds, i, 32 * t2, lane);
#pragma unroll
        for (int qs = 0; qs < NQS; ++qs) oacc[qs][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb[qs][t2], oacc[qs][i], 0, 0, 0);
      }
  }
  // ---- epilogue: O = acc / l, log-sum-exp for the backward
#pragma unroll
  for (int qs = 0; qs < NQS; ++qs) {
    const int q = qrow[qs];
    float lsum = l_run[qs];
    lsum += shfl_xor_f(lsum, 16);
    lsum += shfl_xor_f(lsum, 32);
    if (q >= F) continue;
    const float inv = lsum > 0.f ? p.drop_scale / lsum : 0.f;
    if (g == 0 && p.lse)  // natural-log units of the SCALED scores, as the backward and the reference's logsumexp use them
      p.lse[((long)b * p.nh + h) * F + q] = lsum > 0.f ? m_run[qs] * p.scale + __logf(lsum) : -INFINITY;
    bf16* dst = p.ctx + ((long)b * F + q) * H + h * HD + 4 * g;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16x4 o = {(bf16)(oacc[qs][i][0] * inv), (bf16)(oacc[qs][i][1] * inv), (bf16)(oacc[qs][i][2] * inv),
                        (bf16)(oacc[qs][i][3] * inv)};
      *reinterpret_cast<bf16x4*>(dst + 16 * i) = o;
    }
  }
}

// ================================================================================================ backward
// delta[b,h,q] = sum_d dO[q,d] * O[q,d]   (one wave per row of [B*F, H], 64 lanes = 64 d of one head at a time)
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ dctx, const bf16* __restrict__ ctx,
                                                         float* __restrict__ delta, int B, int F, int nh, int H) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * F) return;
  const int b = (int)(row / F), q = (int)(row % F);
  for (int h = 0; h < nh; ++h) {
    const long o = row * H + h * HD + lane;
    float v = (float)dctx[o] * (float)ctx[o];
    v = wave_sum(v);
    if (lane == 0) delta[((long)b * nh + h) * F + q] = v;
  }
}

// dQ: workgroup = 128 queries of one (b, h); sweeps the keys.  Same orientation as the forward:
//   S^T[key][q], dP^T[key][q] = V dO^T, dS^T = P^T * (dP^T * mask/(1-p) - delta[q]); dQ^T[d][q] += K^T[d][key] dS^T[key][q].
template <bool DROP, int NQS>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x (K dual image | V rows)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const AttnBlock ab = attn_block();
  const int b = ab.b, h = ab.h, q0 = ab.blk * (64 * NQS);
  const int F = p.F, H = p.H;
  const long ld = 3L * H;
  const bf16* base = p.qkv + (long)b * F * ld;
  const int kl = p.klens ? min(max(p.klens[b], 0), F) : F;
  const int nkt = (kl + KT - 1) / KT;
  const uint32_t kcol = (uint32_t)((H + h * HD) * 2), vcol = (uint32_t)((2 * H + h * HD) * 2);
  const int g = lane >> 4;
  int qrow[NQS];
  bf16x8 qf[NQS][2], dof[NQS][2];
  float lse[NQS], dl[NQS], lse2[NQS];
  uint32_t rowbase[NQS];  // dropout row seeds
  const float c2 = p.scale * 1.4426950408889634f;
  const uint32_t thi = p.thresh16 << 16;
#pragma unroll
  for (int qs = 0; qs < NQS; ++qs) {
    qrow[qs] = q0 + 16 * NQS * wave + 16 * qs + (lane & 15);
    rowbase[qs] = drop_rowseed(p, b, h, qrow[qs]);
    const int qc = min(qrow[qs], F - 1);
    lse[qs] = p.lse[((long)b * p.nh + h) * F + qc];
    // exponent offset: lse in log2 units, minus log2 (scale / (1 - p)): P comes out multiplied by the softmax scale and the
    // dropout scale, and dS = P'' (keep ? dP : 0 - delta (1 - p)) needs neither multiply (delta is rescaled once per row)
    lse2[qs] = lse[qs] > -INFINITY ? fmaf(lse[qs], 1.4426950408889634f, -__log2f(p.scale * p.drop_scale)) : INFINITY;
    // delta[q] = sum_d dO[q,d] * O[q,d], computed here from the dO fragments the kernel holds anyway (it used to be a
    // separate pass over dO and O) and written out for the dK/dV kernel, which runs after this one
    float part = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      qf[qs][kk] = load_row_frag(base + h * HD, ld, qrow[qs], F, kk, lane);
      dof[qs][kk] = load_row_frag(p.dctx + (long)b * F * H + h * HD, H, qrow[qs], F, kk, lane);
      const bf16x8 of = load_row_frag(p.ctx + (long)b * F * H + h * HD, H, qrow[qs], F, kk, lane);
#pragma unroll
      for (int j = 0; j < 8; ++j) part = fmaf((float)dof[qs][kk][j], (float)of[j], part);
    }
    part += shfl_xor_f(part, 16);
    part += shfl_xor_f(part, 32);
    dl[qs] = part / p.drop_scale;
    if (g == 0 && qrow[qs] < F) p.delta[((long)b * p.nh + h) * F + qrow[qs]] = part;
  }
  f32x4 dq[NQS][4];
#pragma unroll
  for (int qs = 0; qs < NQS; ++qs)
#pragma unroll
    for (int i = 0; i < 4; ++i) dq[qs][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const uint32_t* cmt = reinterpret_cast<const uint32_t*>(smem + 2 * 2 * TILE_BYTES);  // [nkt * 64] key multipliers (DROP)
  if (DROP) fill_colmul(reinterpret_cast<uint32_t*>(smem + 2 * 2 * TILE_BYTES), nkt * KT);
  const DmaOff koff = dma_offsets(kcol, ld * 2, 2, wave, lane), voff = dma_offsets(vcol, ld * 2, 2, wave, lane);
  auto issue = [&](int kt, int stage) {
    char* s0 = smem + stage * 2 * TILE_BYTES;
    const __amdgpu_buffer_rsrc_t rn = rsrc_rows(base, ld * 2, kt * KT, F);
    dma_tile_at(rn, s0, koff, wave);
    dma_tile_at(rn, s0 + TILE_BYTES, voff, wave);
  };
  if (nkt > 0) issue(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (kt + 1 < nkt) issue(kt + 1, cur ^ 1);
    const char* k_rows = smem + cur * 2 * TILE_BYTES;  // one dual image of K serves the row and the transposing reads
    const char* k_tr = k_rows;
    const char* v_rows = k_rows + TILE_BYTES;
    const int k0 = kt * KT;
    const bool edge = k0 + KT > kl;  // uniform: only this tile needs the key-length mask
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      bf16x8 dsb[NQS];
      {
        f32x4 s[NQS][2], dp[NQS][2];  // [query sub-tile][key sub-tile of this half]
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          const int ks = 2 * t2 + kh;
          const bf16x8 ka = frag_rows_d(k_rows, ks, 0, lane), kb = frag_rows_d(k_rows, ks, 1, lane);
          const bf16x8 va = frag_rows_d(v_rows, ks, 0, lane), vb = frag_rows_d(v_rows, ks, 1, lane);
#pragma unroll
          for (int qs = 0; qs < NQS; ++qs) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[qs][0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb, qf[qs][1], a, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, dof[qs][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb, dof[qs][1], c, 0, 0, 0);
            s[qs][kh] = a;
            dp[qs][kh] = c;
          }
        }
        if (edge) {  // keys beyond the key length: -inf before the exponent (only the tile that crosses it pays)
          const int tl = opaque_s(kl) - k0 - 4 * g;
#pragma unroll
          for (int qs = 0; qs < NQS; ++qs)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (16 * (2 * t2 + kh) + r >= tl) s[qs][kh][r] = -INFINITY;
        }
        u32x4_t cm[2];  // multipliers of this lane's keys of this half (shared by the query sub-tiles)
        if (DROP) {
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) cm[kh] = *reinterpret_cast<const u32x4_t*>(cmt + k0 + 16 * (2 * t2 + kh) + 4 * g);
        }
#pragma unroll
        for (int qs = 0; qs < NQS; ++qs) {
          // P'' = exp2(s * c2 - lsc) = P * scale / (1 - p); dS = P'' (keep ? dP : 0 - delta (1 - p)).  Rows without any valid
          // key have lse = -inf -> lsc = +inf -> P'' = 0.  Packed fp32 over the key pair.
          const f32x2 c2v = {c2, c2}, lscv = {-lse2[qs], -lse2[qs]}, ndlv = {-dl[qs], -dl[qs]};
#pragma unroll
          for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int r2 = 0; r2 < 4; r2 += 2) {
              f32x2 d = {dp[qs][kh][r2], dp[qs][kh][r2 + 1]};
              if (DROP) {
                d[0] = drop_keep(rowbase[qs], cm[kh][r2], thi) ? d[0] : 0.f;
                d[1] = drop_keep(rowbase[qs], cm[kh][r2 + 1], thi) ? d[1] : 0.f;
              }
              const f32x2 a = (f32x2){s[qs][kh][r2], s[qs][kh][r2 + 1]} * c2v + lscv;
              const f32x2 pv = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
              // P'' (d - delta') as fma(P'', d, -P'' delta'): two packed instructions per key pair (the subtraction compiled to one
              // v_sub_f32 per element: the selected values do not sit in register pairs)
              const f32x2 ds2 = __builtin_elementwise_fma(pv, d, pv * ndlv);
              s[qs][kh][r2] = ds2[0];
              s[qs][kh][r2 + 1] = ds2[1];
            }
          dsb[qs] = pack_p(s[qs][0], s[qs][1]);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8 ka = frag_cols_perm_d(k_tr, i, 32 * t2, lane);
#pragma unroll
        for (int qs = 0; qs < NQS; ++qs) dq[qs][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, dsb[qs], dq[qs][i], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int qs = 0; qs < NQS; ++qs) {
    const int q = qrow[qs];
    if (q >= F) continue;
    bf16* dst = p.dqkv + ((long)b * F + q) * ld + h * HD + 4 * g;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16x4 o = {(bf16)dq[qs][i][0], (bf16)dq[qs][i][1], (bf16)dq[qs][i][2], (bf16)dq[qs][i][3]};
      *reinterpret_cast<bf16x4*>(dst + 16 * i) = o;
    }
  }
  if (p.bias_part) {  // (uniform)
    float cs[1][4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = 0.f;
#pragma unroll
        for (int qs = 0; qs < NQS; ++qs) v += qrow[qs] < F ? bf16_round(dq[qs][i][r]) : 0.f;
        cs[0][i][r] = v;
      }
    bias_partial<1>(cs, smem, p.bias_part + ((long)b * gridDim.x + ab.blk) * ld + h * HD, 0, wave, lane);
  }
}

// dK, dV: workgroup = 128 keys of one (b, h) (4 waves x 32 keys); sweeps the queries in tiles of 64.
//   S[q][key] = Q K^T (query on the accumulator row, key on the lane), dP[q][key] = dO V^T,
//   Pd = P * mask/(1-p), dS = P * (dP * mask/(1-p) - delta[q]) * scale,
//   dV^T[d][key] += dO^T[d][q] Pd[q][key],  dK^T[d][key] += Q^T[d][q] dS[q][key].
// MASK: key-padding lengths are given (ragged batch).  Without them no key needs masking HERE: the key is on the lane, a key
// beyond F only feeds its own dK / dV rows, which are not stored -- and the S accumulators start from the constant 0 instead of
// a copy of a per-lane vector (4 v_mov per S tile, 32 per query tile and lane: a tenth of the loop's VALU).
template <bool DROP, int NKS, bool MASK>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x (Q dual image | dO dual image), then lse|delta|seed per stage
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const AttnBlock ab = attn_block();
  const int b = ab.b, h = ab.h, key0 = ab.blk * (64 * NKS);
  const int F = p.F, H = p.H;
  const long ld = 3L * H;
  const bf16* base = p.qkv + (long)b * F * ld;
  const int kl = p.klens ? min(max(p.klens[b], 0), F) : F;
  const int nqt = (F + KT - 1) / KT;
  const uint32_t qcol = (uint32_t)(h * HD * 2);
  const int g = lane >> 4;
  float* stat = reinterpret_cast<float*>(smem + 2 * 2 * TILE_BYTES);  // [stage][lse * log2 e KT | delta KT | dropout row seed KT]
  int krow[NKS];
  bf16x8 kf[NKS][2], vf[NKS][2];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    krow[ks] = key0 + 16 * NKS * wave + 16 * ks + (lane & 15);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      kf[ks][kk] = load_row_frag(base + H + h * HD, ld, krow[ks], F, kk, lane);
      vf[ks][kk] = load_row_frag(base + 2 * H + h * HD, ld, krow[ks], F, kk, lane);
    }
  }
  const float c2 = p.scale * 1.4426950408889634f;
  // exponent offset: P'' = P * scale / (1 - p) -- Pd = keep ? P'' : 0 and dS = P'' (keep ? dP : 0 - delta (1 - p)) need no multiply
  // by the dropout scale (delta is rescaled once per row when it is staged), as in the dQ kernel
  const float log2scale = __log2f(p.scale * p.drop_scale);
  const float inv_drop_scale = 1.f / p.drop_scale;
  const uint32_t thi = p.thresh16 << 16;
  uint32_t cmk[NKS];  // dropout multiplier of this lane's key (common.h)
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) cmk[ks] = DROP ? drop_colmul((uint32_t)krow[ks]) : 1u;
  // keys beyond the key length (a per-lane constant) start S at -3e38: exp2(-huge) = 0, no select per element
  f32x4 sinit[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const float v = krow[ks] < kl ? 0.f : -3.0e38f;
    sinit[ks] = (f32x4){v, v, v, v};
  }
  f32x4 dk[NKS][4], dv[NKS][4];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dk[ks][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dv[ks][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  const DmaOff qoff = dma_offsets(qcol, ld * 2, 2, wave, lane), dooff = dma_offsets(qcol, (long)H * 2, 2, wave, lane);
  const bf16* const do_base = p.dctx + (long)b * F * H;
  auto issue = [&](int qt, int stage) {
    char* s0 = smem + stage * 2 * TILE_BYTES;
    dma_tile_at(rsrc_rows(base, ld * 2, qt * KT, F), s0, qoff, wave);
    dma_tile_at(rsrc_rows(do_base, (long)H * 2, qt * KT, F), s0 + TILE_BYTES, dooff, wave);
  };
  // Per-query statistics of a tile (log-sum-exp, delta, dropout row key; threads 0..63 hold one query each) travel TWO tiles ahead
  // in registers: loaded in front of a tile's LDS-DMA, written to LDS one iteration later.  Loaded behind the DMA and consumed at
  // once (rounds 2-4) they made wave 0 wait `vmcnt(0)` for the tile it had just requested -- a memory round trip at the top of
  // every iteration with nothing overlapped, and the other waves met it at the next barrier.
  float st_l = -INFINITY, st_d = 0.f;  // RAW loaded values: any arithmetic on them here would wait for the loads on the spot
  int st_q = 0;
  auto load_stat = [&](int qt) {
    if (threadIdx.x < KT) {
      st_q = qt * KT + threadIdx.x;
      const long o = ((long)b * p.nh + h) * F + min(st_q, F - 1);
      st_l = p.lse[o];
      st_d = p.delta[o];
    }
  };
  auto write_stat = [&](int stage) {
    if (threadIdx.x < KT) {
      const float ls = st_q < F ? st_l : -INFINITY;
      // rows beyond F or without a valid key: +inf here makes P = exp2(.. - inf) = 0 without a select per element; the softmax
      // scale rides in the exponent (P' = P * scale: dS needs no multiply, dV is rescaled once at the end)
      stat[stage * 3 * KT + threadIdx.x] = ls > -INFINITY ? fmaf(ls, 1.4426950408889634f, -log2scale) : INFINITY;
      stat[stage * 3 * KT + KT + threadIdx.x] = st_q < F ? st_d * inv_drop_scale : 0.f;
      reinterpret_cast<uint32_t*>(stat)[stage * 3 * KT + 2 * KT + threadIdx.x] = drop_rowseed(p, b, h, st_q);
    }
  };
  load_stat(0);
  write_stat(0);
  if (nqt > 1) load_stat(1);
  issue(0, 0);
  for (int qt = 0; qt < nqt; ++qt) {
    const int cur = qt & 1;
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (qt + 1 < nqt) {
      write_stat(cur ^ 1);                 // tile qt + 1's statistics (in registers since the previous iteration)
      if (qt + 2 < nqt) load_stat(qt + 2);  // ... and the next ones, ahead of the DMA in the memory queue
      issue(qt + 1, cur ^ 1);
    }
    const char* q_rows = smem + cur * 2 * TILE_BYTES;  // dual images: row reads and transposing reads from the same tile
    const char* q_tr = q_rows;
    const char* do_rows = q_rows + TILE_BYTES;
    const char* do_tr = do_rows;
    const float* lse_s = stat + cur * 3 * KT;
    const float* dl_s = lse_s + KT;
    const uint32_t* seed_s = reinterpret_cast<const uint32_t*>(lse_s + 2 * KT);
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      bf16x8 pdb[NKS], dsb[NKS];  // per key sub-tile, for the 32 queries of this half
      {
        f32x4 s[NKS][2], dp[NKS][2];  // [key sub-tile][query sub-tile of this half]: rows = queries, column = key
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) {
          const int qsb = 2 * t2 + qh;
          const bf16x8 qa = frag_rows_d(q_rows, qsb, 0, lane), qb = frag_rows_d(q_rows, qsb, 1, lane);
          const bf16x8 da = frag_rows_d(do_rows, qsb, 0, lane), db = frag_rows_d(do_rows, qsb, 1, lane);
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) {
            // keys beyond the key length (a per-lane constant) start S at -3e38: exp2(-huge) = 0, no select per element
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
            if (MASK) a = sinit[ks];
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[ks][0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qb, kf[ks][1], a, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[ks][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(db, vf[ks][1], c, 0, 0, 0);
            s[ks][qh] = a;
            dp[ks][qh] = c;
          }
        }
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) {
          // dropout keys of this lane's 4 query rows (staged with lse / delta): one 16-byte LDS read
          u32x4_t rk = {1u, 1u, 1u, 1u};
          if (DROP) rk = *reinterpret_cast<const u32x4_t*>(seed_s + 16 * (2 * t2 + qh) + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; r += 2) {  // packed fp32 over the row pair
            const int ql = 16 * (2 * t2 + qh) + 4 * g + r;  // query inside the tile
            const f32x2 c2v = {c2, c2}, lscv = {-lse_s[ql], -lse_s[ql + 1]}, ndlv = {-dl_s[ql], -dl_s[ql + 1]};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
              const f32x2 a = (f32x2){s[ks][qh][r], s[ks][qh][r + 1]} * c2v + lscv;
              const f32x2 pr = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};  // P * scale / (1 - p)
              const f32x2 dpv = {dp[ks][qh][r], dp[ks][qh][r + 1]};
              f32x2 pd = pr;
              if (DROP) {
#pragma unroll
                for (int e = 0; e < 2; ++e) pd[e] = drop_keep(rk[r + e], cmk[ks], thi) ? pd[e] : 0.f;
              }
              // dS = P'' (keep dP - delta') = Pd dP - P'' delta' (softmax scale included): ONE select per element (on P''), the mask
              // reaches dP through the product -- two selects and a packed subtraction before
              const f32x2 ds2 = __builtin_elementwise_fma(pd, dpv, pr * ndlv);
              s[ks][qh][r] = pd[0];                 // Pd * scale
              s[ks][qh][r + 1] = pd[1];
              dp[ks][qh][r] = ds2[0];
              dp[ks][qh][r + 1] = ds2[1];
            }
          }
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          pdb[ks] = pack_p(s[ks][0], s[ks][1]);
          dsb[ks] = pack_p(dp[ks][0], dp[ks][1]);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8 doa = frag_cols_perm_d(do_tr, i, 32 * t2, lane);
        const bf16x8 qa = frag_cols_perm_d(q_tr, i, 32 * t2, lane);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          dv[ks][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(doa, pdb[ks], dv[ks][i], 0, 0, 0);
          dk[ks][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, dsb[ks], dk[ks][i], 0, 0, 0);
        }
      }
    }
  }
  const float inv_scale = 1.f / p.scale;
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int key = krow[ks];
    if (key >= F) continue;
    bf16* dkd = p.dqkv + ((long)b * F + key) * ld + H + h * HD + 4 * g;
    bf16* dvd = dkd + H;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16x4 a = {(bf16)dk[ks][i][0], (bf16)dk[ks][i][1], (bf16)dk[ks][i][2], (bf16)dk[ks][i][3]};
      const bf16x4 c = {(bf16)(dv[ks][i][0] * inv_scale), (bf16)(dv[ks][i][1] * inv_scale), (bf16)(dv[ks][i][2] * inv_scale),
                        (bf16)(dv[ks][i][3] * inv_scale)};  // accumulated from P * scale
      *reinterpret_cast<bf16x4*>(dkd + 16 * i) = a;
      *reinterpret_cast<bf16x4*>(dvd + 16 * i) = c;
    }
  }
  if (p.bias_part) {  // (uniform)
    float cs[2][4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float vk = 0.f, vv = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          vk += krow[ks] < F ? bf16_round(dk[ks][i][r]) : 0.f;
          vv += krow[ks] < F ? bf16_round(dv[ks][i][r] * inv_scale) : 0.f;
        }
        cs[0][i][r] = vk;
        cs[1][i][r] = vv;
      }
    bias_partial<2>(cs, smem, p.bias_part + ((long)b * gridDim.x + ab.blk) * ld + H + h * HD, H, wave, lane);
  }
}

// (A fused single-pass backward -- one evaluation of P, the dropout words and dS for dQ, dK and dV -- lived here through round 3 as
// an opt-in mode.  It lost to the two-kernel form at the train step's shape in every variant measured (184-193 us against 149-166 us
// per launch at B = 32, F = 499: profiles/r03_bench_attn.log) and was removed in round 4.)

AttnParams make_params(const bf16* qkv, bf16* ctx, float* lse, const int32_t* klens, const bf16* dctx, float* delta, bf16* dqkv,
                       int B, int F, int nh, int H, const DropSpec& drop) {
  AttnParams p;
  p.qkv = qkv;
  p.ctx = ctx;
  p.lse = lse;
  p.klens = klens;
  p.dctx = dctx;
  p.delta = delta;
  p.dqkv = dqkv;
  p.bias_part = nullptr;
  p.B = B;
  p.F = F;
  p.nh = nh;
  p.H = H;
  p.Fp = (F + 7) & ~7;
  p.scale = 1.f / sqrtf((float)HD);
  p.seed = drop.seed;
  p.stream = drop.stream;
  p.thresh16 = drop.p > 0.f ? (uint32_t)fminf(65535.f, roundf(drop.p * 65536.f)) : 0u;
  p.drop_scale = p.thresh16 ? 1.f / (1.f - (float)p.thresh16 / 65536.f) : 1.f;
  return p;
}

// Per-wave tile of the three kernels (SSAK_ATTN_TILE="f,q,k", each 1 or 2; development switch).  The kernels are bound by
// how many waves per SIMD are in their VALU phase at once (tools/probes/valu_rate.hip: one wave issues a VALU instruction
// every ~6 cycles, the SIMD takes one every 3-4), so the smaller per-wave state of a 16-row tile can pay for its extra LDS reads.
int attn_tile(int which) {
  static int t[3] = {-1, -1, -1};
  if (t[0] < 0) {
    int v[3] = {2, 2, 2};
    if (const char* e = SSAK_DEV_ENV("SSAK_ATTN_TILE")) sscanf(e, "%d,%d,%d", &v[0], &v[1], &v[2]);
    for (int i = 0; i < 3; ++i) t[i] = v[i] == 1 ? 1 : 2;
  }
  return t[which];
}

}  // namespace

bool k_attention_supported(int H, int nh) { return nh > 0 && H / nh == HD && H % nh == 0; }

int k_attention_fwd(const bf16* qkv, bf16* ctx, float* lse, const int32_t* klens, int B, int F, int nh, int H,
                    const DropSpec& drop, hipStream_t st) {
  SSAK_REQUIRE(k_attention_supported(H, nh), "attention: fused kernels are built for head_dim 64 (got %d)", nh ? H / nh : 0);
  SSAK_REQUIRE((long)F * 3 * H * 2 < 2000000000L, "attention: one utterance of q|k|v must span < 2 GB");
  const AttnParams p = make_params(qkv, ctx, lse, klens, nullptr, nullptr, nullptr, B, F, nh, H, drop);
  ProfScope prof_scope(PROF_ATTN_FWD, 4.0 * B * nh * (double)F * F * HD, st);  // S = Q K^T and O = P V
  const int nqs = attn_tile(0);
  SSAK_REQUIRE(!p.thresh16 || F <= DROP_TABLE_N, "attention: dropout is built for at most %d frames", DROP_TABLE_N);
  const int cm_lds = p.thresh16 ? ssak_cdiv(F, KT) * KT * 4 : 0;  // the keys' dropout multipliers behind the tiles
  static bool fwd_attr_done = false;
  if (!fwd_attr_done) {
    SSAK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * TILE_BYTES + DROP_TABLE_N * 4));
    SSAK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * TILE_BYTES + DROP_TABLE_N * 4));
    fwd_attr_done = true;
  }
#define ATT_LAUNCH_FWD(D, N) attn_fwd_kernel<D, N><<<dim3(ssak_cdiv(F, 64 * N), nh, B), 256, 2 * 2 * TILE_BYTES + cm_lds, st>>>(p)
  if (p.thresh16) {
    if (nqs == 1) ATT_LAUNCH_FWD(true, 1); else ATT_LAUNCH_FWD(true, 2);
  } else {
    if (nqs == 1) ATT_LAUNCH_FWD(false, 1); else ATT_LAUNCH_FWD(false, 2);
  }
#undef ATT_LAUNCH_FWD
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

size_t k_attention_bwd_bias_floats(int B, int F, int H) {
  // the bias partials need the two kernels to cut the frames into the same blocks (they do unless the development switch says otherwise)
#ifdef SSAK_AB_NO_ATTN_BIAS  // A/B build: the separate column-sum pass of rounds 1-3
  return 0;
#endif
  return attn_tile(1) == attn_tile(2) ? (size_t)B * ssak_cdiv(F, 64 * attn_tile(1)) * 3 * H : 0;
}

int k_attention_bwd(const bf16* qkv, const bf16* ctx, const float* lse, const int32_t* klens, const bf16* dctx, float* delta,
                    bf16* dqkv, int B, int F, int nh, int H, const DropSpec& drop, int mode, hipStream_t st, float* bias_part,
                    float* bias_grad) {
  SSAK_REQUIRE(k_attention_supported(H, nh), "attention: fused kernels are built for head_dim 64 (got %d)", nh ? H / nh : 0);
  SSAK_REQUIRE(mode == SSAK_ATTN_BWD_DEFAULT || mode == SSAK_ATTN_BWD_TWO_KERNEL,
               "attention_bwd: mode %d (0 / 1 = two kernels; the single-pass form, 2, was removed in ABI 400)", mode);
  SSAK_REQUIRE(!bias_part == !bias_grad && (!bias_part || k_attention_bwd_bias_floats(B, F, H) > 0), "attention_bwd: bias partials need both pointers and equal block sizes");
  AttnParams p = make_params(qkv, const_cast<bf16*>(ctx), const_cast<float*>(lse), klens, dctx, delta, dqkv, B, F, nh, H, drop);
  p.bias_part = bias_part;
  ProfScope prof_scope(PROF_ATTN_BWD, 8.0 * B * nh * (double)F * F * HD, st);  // dV, dP, dQ, dK (the recomputed S is not algorithmic work)
  const int nqs = attn_tile(1), nks = attn_tile(2);
  SSAK_REQUIRE(!p.thresh16 || F <= DROP_TABLE_N, "attention: dropout is built for at most %d frames", DROP_TABLE_N);
  const int cm_lds = p.thresh16 ? ssak_cdiv(F, KT) * KT * 4 : 0;  // the keys' dropout multipliers behind the tiles
  static bool dq_attr_done = false;
  if (!dq_attr_done) {
    SSAK_HIP(hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * TILE_BYTES + DROP_TABLE_N * 4));
    SSAK_HIP(hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * TILE_BYTES + DROP_TABLE_N * 4));
    dq_attr_done = true;
  }
#define ATT_LAUNCH_DQ(D, N) attn_bwd_dq_kernel<D, N><<<dim3(ssak_cdiv(F, 64 * N), nh, B), 256, 2 * 2 * TILE_BYTES + cm_lds, st>>>(p)
  if (p.thresh16) {
    if (nqs == 1) ATT_LAUNCH_DQ(true, 1); else ATT_LAUNCH_DQ(true, 2);
  } else {
    if (nqs == 1) ATT_LAUNCH_DQ(false, 1); else ATT_LAUNCH_DQ(false, 2);
  }
#undef ATT_LAUNCH_DQ
  SSAK_LAUNCH_CHECK();
  constexpr int dkv_lds = 2 * 2 * TILE_BYTES + 2 * 3 * KT * 4;
  static bool attr_done = false;
  if (!attr_done) {
#define ATT_DKV_ATTR(D, N, K) SSAK_HIP(hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<D, N, K>, hipFuncAttributeMaxDynamicSharedMemorySize, dkv_lds))
    ATT_DKV_ATTR(true, 1, true); ATT_DKV_ATTR(false, 1, true); ATT_DKV_ATTR(true, 2, true); ATT_DKV_ATTR(false, 2, true);
    ATT_DKV_ATTR(true, 1, false); ATT_DKV_ATTR(false, 1, false); ATT_DKV_ATTR(true, 2, false); ATT_DKV_ATTR(false, 2, false);
#undef ATT_DKV_ATTR
    attr_done = true;
  }
#define ATT_LAUNCH_DKV(D, N)                                                                                        \
  do {                                                                                                              \
    if (klens) attn_bwd_dkv_kernel<D, N, true><<<dim3(ssak_cdiv(F, 64 * N), nh, B), 256, dkv_lds, st>>>(p);        \
    else attn_bwd_dkv_kernel<D, N, false><<<dim3(ssak_cdiv(F, 64 * N), nh, B), 256, dkv_lds, st>>>(p);             \
  } while (0)
  if (p.thresh16) {
    if (nks == 1) ATT_LAUNCH_DKV(true, 1); else ATT_LAUNCH_DKV(true, 2);
  } else {
    if (nks == 1) ATT_LAUNCH_DKV(false, 1); else ATT_LAUNCH_DKV(false, 2);
  }
#undef ATT_LAUNCH_DKV
  SSAK_LAUNCH_CHECK();
  if (bias_part) {
    // second stage: with the caller's queued reductions (ReduceSink) when there is one, else here
    const int slots = B * ssak_cdiv(F, 64 * nqs);
    if (!(g_reduce_sink && g_reduce_sink->push(bias_part, 3L * H, slots, 3 * H, bias_grad))) {
      const int rc = k_colsum_rows(bias_part, slots, 3 * H, bias_grad, st);
      if (rc != SSAK_OK) return rc;
    }
  }
  return SSAK_OK;
}

// exported for per-op parity tests
extern "C" int ssak_attention_fwd(const void* qkv, void* ctx, float* lse, const int32_t* klens, int B, int F, int nh, int H,
                                  float drop_p, uint64_t seed, uint32_t stream_id, void* stream) {
  SSAK_REQUIRE(qkv && ctx && lse, "attention_fwd: null pointer");
  DropSpec d;
  d.p = drop_p;
  d.seed = seed;
  d.stream = stream_id;
  return k_attention_fwd((const bf16*)qkv, (bf16*)ctx, lse, klens, B, F, nh, H, d, (hipStream_t)stream);
}

extern "C" int ssak_attention_bwd(const void* qkv, const void* ctx, const float* lse, const int32_t* klens, const void* dctx,
                                  float* delta, void* dqkv, int B, int F, int nh, int H, float drop_p, uint64_t seed,
                                  uint32_t stream_id, int mode, void* stream) {
  SSAK_REQUIRE(qkv && ctx && lse && dctx && delta && dqkv, "attention_bwd: null pointer");
  DropSpec d;
  d.p = drop_p;
  d.seed = seed;
  d.stream = stream_id;
  return k_attention_bwd((const bf16*)qkv, (const bf16*)ctx, lse, klens, (const bf16*)dctx, delta, (bf16*)dqkv, B, F, nh, H, d, mode,
                         (hipStream_t)stream);
}

// as ssak_attention_bwd, and bias_grad[3H] += the column sums of dqkv (the q|k|v projection bias gradient), taken inside the kernels
extern "C" size_t ssak_attention_bwd_bias_workspace_bytes(int B, int F, int H) {
  return (size_t)B * ssak_cdiv(F > 0 ? F : 1, 64) * 3 * (H > 0 ? H : 0) * sizeof(float);
}
extern "C" int ssak_attention_bwd_bias(const void* qkv, const void* ctx, const float* lse, const int32_t* klens, const void* dctx,
                                       float* delta, void* dqkv, float* bias_grad, int B, int F, int nh, int H, float drop_p,
                                       uint64_t seed, uint32_t stream_id, void* workspace, size_t workspace_bytes, void* stream) {
  SSAK_REQUIRE(qkv && ctx && lse && dctx && delta && dqkv && bias_grad && workspace, "attention_bwd_bias: null pointer");
  SSAK_REQUIRE(workspace_bytes >= ssak_attention_bwd_bias_workspace_bytes(B, F, H), "attention_bwd_bias: workspace too small");
  DropSpec d;
  d.p = drop_p;
  d.seed = seed;
  d.stream = stream_id;
  if (k_attention_bwd_bias_floats(B, F, H) == 0) {  // (development switch SSAK_ATTN_TILE with unequal blocks)
    const int rc = k_attention_bwd((const bf16*)qkv, (const bf16*)ctx, lse, klens, (const bf16*)dctx, delta, (bf16*)dqkv, B, F, nh, H, d,
                                   SSAK_ATTN_BWD_DEFAULT, (hipStream_t)stream);
    return rc != SSAK_OK ? rc : k_colsum((const bf16*)dqkv, 3L * H, B * F, 3 * H, bias_grad, (hipStream_t)stream);
  }
  return k_attention_bwd((const bf16*)qkv, (const bf16*)ctx, lse, klens, (const bf16*)dctx, delta, (bf16*)dqkv, B, F, nh, H, d,
                         SSAK_ATTN_BWD_DEFAULT, (hipStream_t)stream, (float*)workspace, bias_grad);
}
