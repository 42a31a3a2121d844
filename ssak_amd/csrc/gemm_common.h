// Shared device-side pieces of the bf16 MFMA GEMM kernels (gemm.hip, gemm_p8.hip): launch parameters, XCD-aware
// workgroup order, bias prefetch, the fused epilogue, LDS-DMA helpers.  gfx950 only.
#pragma once
#include "common.h"

namespace {

SSAK_DEFINE_DROP_TABLE

constexpr int BK = 64;
constexpr int NTHREADS = 256;

struct GemmParams {
  const bf16* A;
  const bf16* B;
  void* C;
  const float* bias;
  const bf16* aux_in;
  bf16* aux_out;
  float* slab;
  int M, N, K;
  long lda, ldb, ldc;
  int nb2;
  long sa1, sa2, sb1, sb2, sc1, sc2;
  float alpha;
  int epilogue, out_f32, accumulate, split_k;
  int tiles_m, tiles_n, nz, kt_per_split;
  uint32_t drop_thresh, drop_stream;
  float drop_scale;
  uint64_t drop_seed;
  long bias_s2;
  uint32_t ext_a, ext_b;  // bytes addressable from one batch slice of A / B (buffer descriptor extent)
  float* colsum;          // [wave-tile rows][N] column sums of the stored values (bias gradient of the producing Linear) or null
  float fq_a, fq_b;       // SSAK_EPI_MUL_AUX: factor = code * fq_a + fq_b (step * scale, -zero * step * scale)
  int dynamic;            // ssak_gemm_desc.dynamic_tiles: draw tiles from ticket counters (persistent kernels)
  int* tile_ctr;          // persistent kernels: [0] = tickets handed out past the first round, [1] = workgroups done (or null: static)
  // K-tile visiting order of the persistent kernel for a Toeplitz A (conv as GEMM: lda = stride * C < K = k * C, so K tile kt
  // of row i + 1 IS K tile kt + kperm_p of row i): the first 2 * kperm_n2 steps visit (c, c + kperm_p) pairs, so the second
  // read of the same bytes follows the first one K step later and hits the L2 instead of going back to the Infinity Cache /
  // HBM 16 steps later (0 / 0 = natural order).  Both operands follow the same order; only the summation order changes.
  int kperm_p, kperm_n2;
};

__device__ __forceinline__ int xcd_remap(int id, int n) {
  // contiguous run of logical ids per XCD (hardware deals consecutive workgroup ids round-robin over 8 XCDs);
  // bijective for any n.  Speed only.
  const int q = n >> 3, r = n & 7;
  const int x = id & 7, i = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// ---- 8-bit code of the feed-forward backward factor (SSAK_EPI_GELU_SAVE_GRAD writes it, SSAK_EPI_MUL_AUX reads it) ----------
// f = gelu'(x) * keep / (1 - p).  gelu' lies in [-0.129, 1.129]: a uniform grid of step 1.26 / 254 with its zero point ON the
// grid (code 26 <-> exactly 0, which is also the code of a dropped element) covers [-0.129, 1.136] with a rounding error of at
// most 0.0025 -- what bf16 does for |f| >= 0.6, about 2.5 x bf16's error on average (RMS 1.4e-3 against a factor of ~0.5; the
// dX product it scales is exact, so the gradient picks up ~0.3 % of uncorrelated noise next to the ~1.5 % of the bf16 engine).
// The dropout scale 1 / (1 - p) is applied at decode.  One byte per element: the forward product's second store stream and
// the backward product's factor read are 49 MB instead of 98 MB per layer at the train-step shape.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
constexpr float FQ_ZERO = 26.f, FQ_STEP = 1.26f / 254.f, FQ_INV = 254.f / 1.26f;
// four codes in one dword (byte k = element k); g = gelu'(x), already zeroed where dropped; v_cvt_pk_u8_f32 rounds to nearest
// even and saturates to [0, 255]
__device__ __forceinline__ uint32_t fq_pack4(float g0, float g1, float g2, float g3) {
  uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g0, FQ_INV, FQ_ZERO), 0u, 0u);
  w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g1, FQ_INV, FQ_ZERO), 1u, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g2, FQ_INV, FQ_ZERO), 2u, w);
  return __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g3, FQ_INV, FQ_ZERO), 3u, w);
}
// f = (code - 26) * step * scale as one fma per element: a = step * scale, b = -26 * step * scale
__device__ __forceinline__ void fq_unpack4(uint32_t w, float a, float b, float (&f)[4]) {
  f[0] = fmaf((float)(w & 0xffu), a, b);
  f[1] = fmaf((float)((w >> 8) & 0xffu), a, b);
  f[2] = fmaf((float)((w >> 16) & 0xffu), a, b);
  f[3] = fmaf((float)(w >> 24), a, b);
}
__device__ __forceinline__ uint8_t fq_pack1(float g) { return (uint8_t)(__builtin_amdgcn_cvt_pk_u8_f32(fmaf(g, FQ_INV, FQ_ZERO), 0u, 0u) & 0xffu); }
__device__ __forceinline__ float fq_unpack1(uint8_t c, float a, float b) { return fmaf((float)c, a, b); }
// bias for this lane's NI column groups, loaded BEFORE the K loop (vector loads; the round trip then overlaps the
// main loop instead of being exposed at the tail of every workgroup: measured 19 us of 133 on the FFN shape)
template <int NI>
struct BiasRegs {
  float v[NI][4];
};
template <int NI>
__device__ __forceinline__ void load_bias(const GemmParams& p, int bn0, int wn0, int lane, int z2, BiasRegs<NI>& br) {
  const int ln = (lane >> 4) * 4;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = bn0 + wn0 + 16 * j + ln;
#pragma unroll
    for (int r = 0; r < 4; ++r) br.v[j][r] = 0.f;
    if (p.bias && p.split_k == 1 && n < p.N) {
      const float* bp = p.bias + z2 * p.bias_s2 + n;
      if (n + 3 < p.N && ((z2 * p.bias_s2) & 3) == 0) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(bp);
        br.v[j][0] = t[0];
        br.v[j][1] = t[1];
        br.v[j][2] = t[2];
        br.v[j][3] = t[3];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.N) br.v[j][r] = bp[r];
      }
    }
  }
}

// Fused epilogue.  The MFMA leaves lane l with 4 consecutive columns of 16 DIFFERENT rows (row = .. + (l & 15)), so a
// direct store touches 16 rows x 32 B per wave-instruction: a quarter of each 128-B line.  Measured on the 256x256
// kernel (tools/probes/p8_probe.hip): 17-19 us to issue the stores of one tile, as long as a whole K = 768 main loop.
// So the accumulators take one round trip through the wave's private LDS region (fp32, alpha and bias applied on the
// way in, 16-B chunk c of row r at slot c ^ (r mod chunks-per-row): conflict-free both ways) and come back with 16
// (8 for 32-column wave tiles) consecutive lanes covering ONE whole row segment; activation, GELU', dropout, the side
// output and every load / store then work on full cache lines.  `lds_wave` = 1024 * MI * NI bytes private to the wave;
// the caller guarantees nobody still reads that memory (workgroup barrier after the K loop, LDS-DMA drained).
template <int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[MI][NI], const BiasRegs<NI>& br, char* lds_wave,
                                              int bm0, int bn0, int wm0, int wn0, int lane, int z, int z1, int z2, int split) {
  constexpr int CPR = 4 * NI;      // 16-byte chunks (4 fp32) per row of the wave tile
  constexpr int PITCH = 64 * NI;   // bytes per row
  constexpr int RPI = 64 / CPR;    // rows covered by one wave-instruction in the row domain
  constexpr int STEPS = 16 * MI / RPI;
  {
    const int lm = lane & 15, lq = lane >> 4;
    const bool raw = p.split_k > 1;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = 16 * i + lm;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int chunk = 4 * j + lq;
        f32x4 v = acc[i][j];
        if (!raw) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] * p.alpha + br.v[j][r];
        }
        *reinterpret_cast<f32x4*>(lds_wave + row * PITCH + ((chunk ^ (row & (CPR - 1))) << 4)) = v;
      }
    }
  }
  const long coff = z1 * p.sc1 + z2 * p.sc2;
  float* const S = p.split_k > 1 ? p.slab + ((long)split * p.nz + z) * (long)p.M * p.N : nullptr;
  // ---- interior wave tiles (all but the last tile row / column): 8 consecutive columns per lane, no bounds checks,
  // 16-byte bf16 / 2 x 16-byte fp32 accesses, the output offset advanced by a constant
  if (bm0 + wm0 + 16 * MI <= p.M && bn0 + wn0 + 16 * NI <= p.N && (p.N & 7) == 0 && (p.ldc & 7) == 0 && (coff & 7) == 0 &&
      (((uintptr_t)p.aux_in | (uintptr_t)p.aux_out) & 15) == 0) {
    constexpr int LPR = CPR / 2;    // lanes per row
    constexpr int RPI2 = 64 / LPR;  // rows per wave-instruction
    const int r0 = lane / LPR, cp = lane % LPR;
    const int n = bn0 + wn0 + 8 * cp;
    long o = S ? (long)(bm0 + wm0 + r0) * p.N + n : coff + (long)(bm0 + wm0 + r0) * p.ldc + n;
    const long ostep = (long)RPI2 * (S ? (long)p.N : p.ldc);
    // dropout (common.h): word = rowkey(seed, site, output row) * colmul(output column); this lane's 8 columns are fixed
    uint32_t cm[8];
    if (p.drop_thresh) {
      const uint4 t0 = *reinterpret_cast<const uint4*>(g_drop_colmul.v + n), t1 = *reinterpret_cast<const uint4*>(g_drop_colmul.v + n + 4);
      cm[0] = t0.x, cm[1] = t0.y, cm[2] = t0.z, cm[3] = t0.w, cm[4] = t1.x, cm[5] = t1.y, cm[6] = t1.z, cm[7] = t1.w;
    }
    const uint32_t thi = p.drop_thresh << 16;
#pragma unroll 2
    for (int row = r0; row < 16 * MI; row += RPI2, o += ostep) {
      const int c0 = (2 * cp) ^ (row & (CPR - 1));
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(lds_wave + row * PITCH + (c0 << 4));
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(lds_wave + row * PITCH + ((c0 ^ 1) << 4));
      float v[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
      if (S) {
        *reinterpret_cast<f32x4*>(S + o) = t0;
        *reinterpret_cast<f32x4*>(S + o + 4) = t1;
        continue;
      }
      float gd[8];
      const bool save_grad = p.epilogue == SSAK_EPI_GELU_SAVE_GRAD;
      if (p.epilogue == SSAK_EPI_GELU || save_grad) {
        if (!save_grad && p.aux_out) {
          bf16x8 q;
#pragma unroll
          for (int r = 0; r < 8; ++r) q[r] = (bf16)v[r];
          *reinterpret_cast<bf16x8*>(p.aux_out + o) = q;
        }
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
          const f32x2 x = {v[r], v[r + 1]};
          if (save_grad) {
            const f32x2 d = gelu_grad2(x);
            gd[r] = d[0];
            gd[r + 1] = d[1];
          }
          const f32x2 y = gelu2(x);
          v[r] = y[0];
          v[r + 1] = y[1];
        }
      } else if (p.epilogue == SSAK_EPI_MUL_AUX) {
        const u32x2 w2 = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint8_t*>(p.aux_in) + o);  // 8 one-byte codes
        float f0[4], f1[4];
        fq_unpack4(w2[0], p.fq_a, p.fq_b, f0);
        fq_unpack4(w2[1], p.fq_a, p.fq_b, f1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] *= f0[r];
          v[4 + r] *= f1[r];
        }
      } else if (p.epilogue == SSAK_EPI_MUL_GELU_GRAD) {
        const bf16x8 a8 = *reinterpret_cast<const bf16x8*>(p.aux_in + o);
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
          const f32x2 y = gelu_grad2((f32x2){(float)a8[r], (float)a8[r + 1]});
          v[r] *= y[0];
          v[r + 1] *= y[1];
        }
      }
      if (p.drop_thresh) {
        const uint32_t rk = drop_rowkey(p.drop_seed, p.drop_stream, (uint64_t)z * p.M + (bm0 + wm0 + row));
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          const bool keep = drop_keep(rk, cm[h], thi);
          v[h] = keep ? v[h] * p.drop_scale : 0.f;
          if (save_grad) gd[h] = keep ? gd[h] : 0.f;  // (the factor's code carries the mask; its 1 / (1 - p) is applied at decode)
        }
      }
      if (save_grad && p.aux_out)
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint8_t*>(p.aux_out) + o) =
            (u32x2){fq_pack4(gd[0], gd[1], gd[2], gd[3]), fq_pack4(gd[4], gd[5], gd[6], gd[7])};
      if (p.out_f32) {
        float* dst = reinterpret_cast<float*>(p.C) + o;
        if (p.accumulate) {
          const f32x4 c0v = *reinterpret_cast<const f32x4*>(dst);
          const f32x4 c1v = *reinterpret_cast<const f32x4*>(dst + 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] += c0v[r];
            v[4 + r] += c1v[r];
          }
        }
        *reinterpret_cast<f32x4*>(dst) = (f32x4){v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
      } else {
        bf16x8 q;
#pragma unroll
        for (int r = 0; r < 8; ++r) q[r] = (bf16)v[r];
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.C) + o) = q;
      }
    }
    return;
  }
  const int rr = lane / CPR, cl = lane % CPR;
  // a rolled loop on purpose: unrolled, the mode switches below were replicated STEPS times and the epilogue spent its
  // time fetching cold instructions (measured: 18-20 us per 256x256 tile, either store pattern)
#pragma unroll 1
  for (int s = 0; s < STEPS; ++s) {
    const int row = s * RPI + rr;
    const int chunk = cl ^ (row & (CPR - 1));
    const f32x4 t = *reinterpret_cast<const f32x4*>(lds_wave + row * PITCH + (cl << 4));
    const int m = bm0 + wm0 + row;
    const int n = bn0 + wn0 + 4 * chunk;
    if (m >= p.M || n >= p.N) continue;
    const bool full = n + 3 < p.N;
    if (S) {  // split-K partial sums: raw accumulators into this split's slab
      float* dst = S + (long)m * p.N + n;
      if (full && (p.N & 3) == 0) {
        *reinterpret_cast<f32x4*>(dst) = t;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.N) dst[r] = t[r];
      }
      continue;
    }
    const long o = coff + (long)m * p.ldc + n;
    float v[4] = {t[0], t[1], t[2], t[3]};
    float gd[4] = {0.f, 0.f, 0.f, 0.f};
    const bool save_grad = p.epilogue == SSAK_EPI_GELU_SAVE_GRAD;
    if (p.epilogue == SSAK_EPI_GELU || save_grad) {
      if (!save_grad && p.aux_out) {
        if (full) {
          bf16x4 q = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          *reinterpret_cast<bf16x4*>(p.aux_out + o) = q;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) p.aux_out[o + r] = (bf16)v[r];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (save_grad) gd[r] = gelu_grad_f(v[r]);
        v[r] = gelu_f(v[r]);
      }
    } else if (p.epilogue == SSAK_EPI_MUL_AUX) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < p.N) v[r] *= fq_unpack1(reinterpret_cast<const uint8_t*>(p.aux_in)[o + r], p.fq_a, p.fq_b);
    } else if (p.epilogue == SSAK_EPI_MUL_GELU_GRAD) {
      if (full) {
        const bf16x4 a4 = *reinterpret_cast<const bf16x4*>(p.aux_in + o);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= gelu_grad_f((float)a4[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.N) v[r] *= gelu_grad_f((float)p.aux_in[o + r]);
      }
    }
    if (p.drop_thresh) {  // (edge tiles: the column multipliers by their hash, no table bounds to mind)
      const uint32_t rk = drop_rowkey(p.drop_seed, p.drop_stream, (uint64_t)z * p.M + m);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool keep = drop_keep(rk, drop_colmul((uint32_t)(n + r)), p.drop_thresh << 16);
        v[r] = keep ? v[r] * p.drop_scale : 0.f;
        if (save_grad) gd[r] = keep ? gd[r] : 0.f;
      }
    }
    if (save_grad && p.aux_out) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < p.N) reinterpret_cast<uint8_t*>(p.aux_out)[o + r] = fq_pack1(gd[r]);
    }
    if (p.out_f32) {
      float* dst = reinterpret_cast<float*>(p.C) + o;
      if (p.accumulate) {
        if (full) {
          const f32x4 c = *reinterpret_cast<const f32x4*>(dst);
          *reinterpret_cast<f32x4*>(dst) = (f32x4){c[0] + v[0], c[1] + v[1], c[2] + v[2], c[3] + v[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) dst[r] += v[r];
        }
      } else if (full) {
        *reinterpret_cast<f32x4*>(dst) = (f32x4){v[0], v[1], v[2], v[3]};
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.N) dst[r] = v[r];
      }
    } else {
      bf16* dst = reinterpret_cast<bf16*>(p.C) + o;
      if (full) {
        bf16x4 q = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *reinterpret_cast<bf16x4*>(dst) = q;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.N) dst[r] = (bf16)v[r];
      }
    }
  }
}

// ---- LDS-free epilogue for interior 64-column wave tiles -----------------------------------------------------------
// Everything elementwise (alpha, bias, GELU, GELU', dropout) is done in the accumulator layout.  fp32 outputs (slabs,
// weight gradients) are stored from there directly: the 4 lanes of a row hold 64 contiguous bytes.  bf16 outputs would be
// 8 B per lane and a quarter line per row, so the packed pairs go through a 4x4 transposition between the wave's four
// 16-lane rows and the four column groups (v_permlane32_swap + v_permlane16_swap, 8 per 16 rows): lane (lm, lq) ends
// up with the 16 consecutive columns 16*lq.. of row lm = two 16-byte stores, the four lanes of a row cover one 128-B
// line.  No LDS traffic, no barrier, ~40 VALU + 2 stores per 16 rows on the plain path (the LDS round trip it replaces
// measured 4.6 us per 256x256 tile whatever the number of workgroups, tools/probes/p8_probe.hip).
__device__ __forceinline__ void xpose4(uint32_t& r0, uint32_t& r1, uint32_t& r2, uint32_t& r3) {
  u32x2 t = __builtin_amdgcn_permlane32_swap(r0, r2, false, false);
  r0 = t[0];
  r2 = t[1];
  t = __builtin_amdgcn_permlane32_swap(r1, r3, false, false);
  r1 = t[0];
  r3 = t[1];
  t = __builtin_amdgcn_permlane16_swap(r0, r1, false, false);
  r0 = t[0];
  r1 = t[1];
  t = __builtin_amdgcn_permlane16_swap(r2, r3, false, false);
  r2 = t[0];
  r3 = t[1];
}
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
// one 16-byte store of the LDS-free epilogue: NON-TEMPORAL.  A round of tiles writes 25-64 MB at once and nothing on this CU
// reads it back; leaving those lines in the XCD's L2 (a plain store does) pushes the operand panels out, which the next
// tiles then fetch again.  Same-box A/B of the train step (profiles/r03_ab_epilogue_store_policy.log, three alternations):
// plain 2 019 / 2 023, `nt` 2 031 / 2 033 utterances/s (+0.5 %: the N = 768 forward products 1 991 -> 1 958 us per step);
// write-through `sc1` stores 1 890 (-6.5 %: every store a fabric write of its own).  -DSSAK_EPI_STORE=0 / 1 build the plain /
// sc1 forms.
#ifndef SSAK_EPI_STORE
#define SSAK_EPI_STORE 2
#endif
__device__ __forceinline__ void epi_store16(void* dst, u32x4 v) {
#if SSAK_EPI_STORE == 1
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
#elif SSAK_EPI_STORE == 2
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst));
#else
  *reinterpret_cast<u32x4*>(dst) = v;
#endif
}
// store 16 rows x 64 columns of codes held as g[j][r] (accumulator layout) at dst + row lm, + 16 * lq: one 16-byte store per lane
__device__ __forceinline__ void store_fq_rows(uint8_t* dst_lane, const float (&g)[4][4], bool rowok) {
  uint32_t q[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) q[j] = fq_pack4(g[j][0], g[j][1], g[j][2], g[j][3]);
  xpose4(q[0], q[1], q[2], q[3]);
  if (rowok) epi_store16(dst_lane, (u32x4){q[0], q[1], q[2], q[3]});
}
// ... and the way back: f[j][r] in the accumulator layout from the 16 bytes of this lane's row segment
__device__ __forceinline__ void load_fq_rows(const uint8_t* src_lane, bool rowok, float a, float b, float (&f)[4][4]) {
  u32x4 w = rowok ? *reinterpret_cast<const u32x4*>(src_lane) : (u32x4){0u, 0u, 0u, 0u};
  uint32_t q[4] = {w[0], w[1], w[2], w[3]};
  xpose4(q[0], q[1], q[2], q[3]);
#pragma unroll
  for (int j = 0; j < 4; ++j) fq_unpack4(q[j], a, b, f[j]);
}
// store 16 rows x 64 columns of bf16 held as v[j][r] (accumulator layout) at dst + row lm, columns 16*lq..
__device__ __forceinline__ void store_bf16_rows(bf16* dst_lane /* + row lm, + 16*lq */, const float (&v)[4][4], bool rowok = true) {
  uint32_t lo[4], hi[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bf16x2 a = {(bf16)v[j][0], (bf16)v[j][1]};
    const bf16x2 b = {(bf16)v[j][2], (bf16)v[j][3]};
    lo[j] = __builtin_bit_cast(uint32_t, a);
    hi[j] = __builtin_bit_cast(uint32_t, b);
  }
  xpose4(lo[0], lo[1], lo[2], lo[3]);
  xpose4(hi[0], hi[1], hi[2], hi[3]);
  if (rowok) {  // (after the swaps: those need every lane)
    epi_store16(dst_lane, (u32x4){lo[0], hi[0], lo[1], hi[1]});
    epi_store16(dst_lane + 8, (u32x4){lo[2], hi[2], lo[3], hi[3]});
  }
}
// true when the wave tile [wm0, wm0 + rows) x [wn0, wn0 + 64) of this workgroup can take gemm_epilogue_direct
__device__ __forceinline__ bool epilogue_direct_ok(const GemmParams& p, int bm0, int bn0, int wm0, int wn0, int rows, long coff) {
  (void)rows;  // tile rows beyond M are masked per lane by the epilogue itself
  const bool fq = p.epilogue == SSAK_EPI_GELU_SAVE_GRAD || p.epilogue == SSAK_EPI_MUL_AUX;  // 16-byte row segments of one-byte codes
  return bn0 + wn0 + 64 <= p.N && (p.N & 7) == 0 && (p.ldc & 7) == 0 && (coff & 7) == 0 && (!fq || ((p.ldc | coff) & 15) == 0) &&
         (((uintptr_t)p.aux_in | (uintptr_t)p.aux_out | (uintptr_t)p.C) & 15) == 0;
}
// EPI: -1 = the general-purpose form (epilogue mode read from the parameters: NONE / GELU / MUL_GELU_GRAD); SSAK_EPI_GELU_SAVE_GRAD
// or SSAK_EPI_MUL_AUX = a form specialised for the feed-forward pair.  The two are separate instantiations because the modes
// are compiled into one body otherwise and the widest of them sets the register allocation of ALL (the saved-gradient forward
// keeps 16 more values live than the rest: compiled in, it pushed every 256-row instantiation of the persistent kernel from
// 36-60 to 80-144 bytes of scratch per lane and the 192-row ones from none to 12-48, and slowed every launch by 2-14 %).
// Further specialisations (internal codes, chosen by the launcher when the descriptor qualifies): the plain bf16 store
// (bias / alpha only -- the qkv, output and feed-forward-down projections and every plain dX product), the plain fp32 store
// (weight gradients, optionally accumulating) and GELU without a saved pre-activation (the frozen conv stack).
constexpr int P8_EPI_PLAIN_BF16 = 100, P8_EPI_PLAIN_F32 = 101, P8_EPI_GELU_ONLY = 102;
// SSAK_EPI_MUL_AUX: the 16 bytes of factor codes of every 16-row group of a wave tile, loaded ahead of the arithmetic (one load per
// group, all in flight together).  Inside the per-group loop each load's round trip (L2 / HBM: ~1-2 us) was exposed -- the groups
// are kept apart by scheduling fences -- and made the feed-forward dX product's epilogue ~9 us per tile.
template <int MI>
struct FqCodes {
  u32x4 w[MI];
};
template <int MI>
__device__ __forceinline__ void load_fq_codes(const GemmParams& p, FqCodes<MI>& c, int bm0, int bn0, int wm0, int wn0, int lane, int z1, int z2) {
  const int lm = lane & 15, lq = lane >> 4;
  const int rows_valid = p.M - (bm0 + wm0 + lm);
  const uint8_t* src = reinterpret_cast<const uint8_t*>(p.aux_in) + z1 * p.sc1 + z2 * p.sc2 + (long)(bm0 + wm0 + lm) * p.ldc + bn0 + wn0 + 16 * lq;
#pragma unroll
  for (int i = 0; i < MI; ++i) c.w[i] = 16 * i < rows_valid ? *reinterpret_cast<const u32x4*>(src + (long)16 * i * p.ldc) : (u32x4){0u, 0u, 0u, 0u};
}
// the 16 dropout column multipliers (common.h) of a lane's columns bn0 + wn0 + 16 j + 4 lq + r of a 64-column wave tile
struct DropCols {
  uint32_t v[4][4];
};
__device__ __forceinline__ void load_drop_cols(DropCols& d, int bn0, int wn0, int lane) {
  const uint32_t* cmp = g_drop_colmul.v + bn0 + wn0 + 4 * (lane >> 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint4 t = *reinterpret_cast<const uint4*>(cmp + 16 * j);
    d.v[j][0] = t.x, d.v[j][1] = t.y, d.v[j][2] = t.z, d.v[j][3] = t.w;
  }
}
template <int MI, int EPI = -1>
__device__ __forceinline__ void gemm_epilogue_direct(const GemmParams& p, f32x4 (&acc)[MI][4], const BiasRegs<4>& br, int bm0,
                                                     int bn0, int wm0, int wn0, int lane, int z, int z1, int z2, int split,
                                                     const FqCodes<MI>* codes = nullptr) {
  constexpr bool GENERAL = EPI < 0;
  const int lm = lane & 15, lq = lane >> 4;
  const int rows_valid = p.M - (bm0 + wm0 + lm);  // this lane's row 16 * i + lm exists iff 16 * i < rows_valid
  if (GENERAL && p.split_k > 1) {  // raw partial sums into this split's slab
    float* S = p.slab + ((long)split * p.nz + z) * (long)p.M * p.N + (long)(bm0 + wm0 + lm) * p.N + bn0 + wn0 + 4 * lq;
    const long step = 16L * p.N;
#pragma unroll
    for (int i = 0; i < MI; ++i, S += step)
      if (16 * i < rows_valid) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(S + 16 * j) = acc[i][j];
      }
    return;
  }
  long orow = z1 * p.sc1 + z2 * p.sc2 + (long)(bm0 + wm0 + lm) * p.ldc + bn0 + wn0;  // row lm of the 16-row group, column 0
  const long step = 16L * p.ldc;
  const bool plain = EPI == P8_EPI_PLAIN_BF16 || (GENERAL && p.epilogue == SSAK_EPI_NONE && !p.drop_thresh && !p.colsum && !p.out_f32);
  if (plain) {
#pragma unroll
    for (int i = 0; i < MI; ++i, orow += step) {
      float v[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[j][r] = acc[i][j][r] * p.alpha + br.v[j][r];
      store_bf16_rows(reinterpret_cast<bf16*>(p.C) + orow + 16 * lq, v, 16 * i < rows_valid);
      __builtin_amdgcn_sched_barrier(0);  // keep the 16-row groups apart: hoisted addresses cost registers
    }
    return;
  }
  if constexpr (EPI == P8_EPI_PLAIN_BF16) return;
  if constexpr (EPI == P8_EPI_PLAIN_F32) {
#pragma unroll
    for (int i = 0; i < MI; ++i, orow += step) {
      if (16 * i < rows_valid) {
        float* dst = reinterpret_cast<float*>(p.C) + orow + 4 * lq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 t = acc[i][j] * p.alpha;
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] += br.v[j][r];
          if (p.accumulate) t += *reinterpret_cast<const f32x4*>(dst + 16 * j);
          *reinterpret_cast<f32x4*>(dst + 16 * j) = t;
        }
      }
    }
    return;
  }
  constexpr bool WITH_COLSUM = GENERAL || EPI == SSAK_EPI_MUL_AUX;
  constexpr bool WITH_DROP = GENERAL || EPI == SSAK_EPI_GELU_SAVE_GRAD;
  // The feed-forward up-projection (GELU_SAVE_GRAD + dropout) is the kernel's widest epilogue, at the register limit.  Its
  // dropout needs the 16 column multipliers of this lane (common.h) for every 16-row group; they take the place of the 16 bias
  // values: alpha and bias are applied to ALL accumulators first (the same one fma per element), after which the bias
  // registers are dead.  (Read again per group instead, each group waited out an L1 round trip: +53 us per step, measured.)
  constexpr bool PRE_BIAS = EPI == SSAK_EPI_GELU_SAVE_GRAD;
  uint32_t cmv[PRE_BIAS ? 4 : 1][4];
  if constexpr (PRE_BIAS) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = acc[i][j][r] * p.alpha + br.v[j][r];
    if (p.drop_thresh) {
      DropCols dc;  // (loaded here, not ahead of the next tile's priming: 16 more values live across the priming spilled at
      load_drop_cols(dc, bn0, wn0, lane);  // 256-row tiles and cost the launch 5 %, profiles/r05_ab_dropout_hash.log)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) cmv[j][r] = dc.v[j][r];
    }
  }
  float cs[4][4];  // column sums over this lane's rows (p.colsum)
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) cs[j][r] = 0.f;
#pragma unroll
  for (int i = 0; i < MI; ++i, orow += step) {
    __builtin_amdgcn_sched_barrier(0);
    const bool rowok = 16 * i < rows_valid;
    const long oa = orow + 4 * lq;  // this lane's chunk of column group j: oa + 16 * j
    float v[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[j][r] = PRE_BIAS ? acc[i][j][r] : acc[i][j][r] * p.alpha + br.v[j][r];
    float gd[4][4];  // SSAK_EPI_GELU_SAVE_GRAD: gelu'(pre), masked and scaled like the output -- the backward's factor
    constexpr bool save_grad = EPI == SSAK_EPI_GELU_SAVE_GRAD;
    if ((GENERAL && p.epilogue == SSAK_EPI_GELU) || save_grad || EPI == P8_EPI_GELU_ONLY) {
      if (GENERAL && p.aux_out) store_bf16_rows(p.aux_out + orow + 16 * lq, v, rowok);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 x = {v[j][r], v[j][r + 1]};
          if (save_grad) {
            const f32x2 d = gelu_grad2(x);
            gd[j][r] = d[0];
            gd[j][r + 1] = d[1];
          }
          const f32x2 y = gelu2(x);
          v[j][r] = y[0];
          v[j][r + 1] = y[1];
        }
    } else if (EPI == SSAK_EPI_MUL_AUX) {
      float f[4][4];  // one 16-byte load of this lane's row segment of codes, transposed back to the accumulator layout
      if (codes) {
        uint32_t q[4] = {codes->w[i][0], codes->w[i][1], codes->w[i][2], codes->w[i][3]};
        xpose4(q[0], q[1], q[2], q[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) fq_unpack4(q[j], p.fq_a, p.fq_b, f[j]);
      } else {
        load_fq_rows(reinterpret_cast<const uint8_t*>(p.aux_in) + orow + 16 * lq, rowok, p.fq_a, p.fq_b, f);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[j][r] *= f[j][r];
    } else if (GENERAL && p.epilogue == SSAK_EPI_MUL_GELU_GRAD) {
      bf16x4 a[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = rowok ? *reinterpret_cast<const bf16x4*>(p.aux_in + oa + 16 * j) : (bf16x4){};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 y = gelu_grad2((f32x2){(float)a[j][r], (float)a[j][r + 1]});
          v[j][r] *= y[0];
          v[j][r + 1] *= y[1];
        }
    }
    if (WITH_DROP && p.drop_thresh) {
      // dropout (common.h): word = rowkey(seed, site, output row) * colmul(output column) -- one hash per 16-row group and lane,
      // the 16 column multipliers from the table (general form: read per group, the same 64 bytes per lane every time,
      // L1-resident; the feed-forward form holds them in the registers the bias had: PRE_BIAS above)
      const uint32_t rk = drop_rowkey(p.drop_seed, p.drop_stream, (uint64_t)z * p.M + (bm0 + wm0 + 16 * i + lm));
      const uint32_t thi = p.drop_thresh << 16;
      const uint32_t* cmp = g_drop_colmul.v + bn0 + wn0 + 4 * lq;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t cm[4];
        if constexpr (PRE_BIAS) {
#pragma unroll
          for (int r = 0; r < 4; ++r) cm[r] = cmv[j][r];
        } else {
          const uint4 t = *reinterpret_cast<const uint4*>(cmp + 16 * j);
          cm[0] = t.x, cm[1] = t.y, cm[2] = t.z, cm[3] = t.w;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool keep = drop_keep(rk, cm[r], thi);
          v[j][r] = keep ? v[j][r] * p.drop_scale : 0.f;
          if (save_grad) gd[j][r] = keep ? gd[j][r] : 0.f;  // (the factor's code carries the mask; its 1 / (1 - p) is applied at decode)
        }
      }
    }
    if (save_grad && p.aux_out) store_fq_rows(reinterpret_cast<uint8_t*>(p.aux_out) + orow + 16 * lq, gd, rowok);
    if (WITH_COLSUM && p.colsum) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[j][r] += rowok ? v[j][r] : 0.f;
    }
    if (GENERAL && p.out_f32) {
      float* dst = reinterpret_cast<float*>(p.C) + oa;
      if (rowok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 t = {v[j][0], v[j][1], v[j][2], v[j][3]};
          if (p.accumulate) t += *reinterpret_cast<const f32x4*>(dst + 16 * j);
          *reinterpret_cast<f32x4*>(dst + 16 * j) = t;
        }
      }
    } else {
      store_bf16_rows(reinterpret_cast<bf16*>(p.C) + orow + 16 * lq, v, rowok);
    }
  }
  if (WITH_COLSUM && p.colsum) {
    // sum over the 16 lanes of a row group (the wave tile's rows), fixed order; lane lm == 0 of each group writes the
    // 16 columns 16 * j + 4 * lq .. of its slot = (tile row, wave row); a small kernel adds the slots up afterwards
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = cs[j][r];
        s += dpp_f<0xB1, 0xf>(0.f, s);
        s += dpp_f<0x4E, 0xf>(0.f, s);
        s += dpp_f<0x141, 0xf>(0.f, s);
        s += dpp_f<0x140, 0xf>(0.f, s);
        cs[j][r] = s;
      }
    if (lm == 0) {
      float* dst = p.colsum + (long)((bm0 + wm0) / (16 * MI)) * p.N + bn0 + wn0 + 4 * lq;
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(dst + 16 * j) = (f32x4){cs[j][0], cs[j][1], cs[j][2], cs[j][3]};
    }
  }
}

typedef __attribute__((address_space(3))) void lds_void;

template <int R>
__device__ __forceinline__ int km_swz(int kr) {
  // XOR applied to the 16-B chunk index of k-row kr
  if (R == 128) return (((kr >> 3) & 1) << 3) | ((kr & 3) << 1);
  return ((((kr >> 3) & 1) << 1) | ((kr >> 1) & 1)) << 1;  // R == 64: two k-rows share a 256-B bank row
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  // s_waitcnt vmcnt(N) only (lgkmcnt / expcnt untouched); gfx9 encoding: vm[3:0] | exp 7<<4 | lgkm 15<<8 | vm[5:4]<<14
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

}  // namespace
