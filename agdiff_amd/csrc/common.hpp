// Device-side building blocks shared by all kernels (gfx950 / CDNA4 only).
//
// Register-resident MLP chains on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fmaf chain):
// a wave owns one tile of 32 edges (or nodes).  An activation vector of 32*KT features is kept
// as KT accumulator tiles `f32x16 x[KT]` in the MFMA C/D layout:
//     lane l  <->  edge  (l & 31),  half h = l >> 5
//     x[t][r] <->  feature 32*t + (r&3) + 8*(r>>2) + 4*h
// With weights pre-packed in the matching k-order ("pk", include/agdiff_hip.h) the accumulator
// of one layer IS the B operand of the next one: no LDS, no shuffles between layers.
//   std  orientation: D[feature][edge] = W . X      (A = weights, B = activations)
//   flip orientation: D[edge][feature] = X^T . W^T  (A = activations, B = weights)
// The flipped result has edges on rows (registers) and features on lanes, which is what the
// destination-segmented reduction and coalesced row gathers want.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "agdiff_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define AG_WAVE 64
#define AG_WG 256  // 4 waves per workgroup, each wave an independent tile

__device__ __forceinline__ int ag_lane() { return threadIdx.x & 63; }
__device__ __forceinline__ int ag_wave_in_wg() { return threadIdx.x >> 6; }

// Row (edge slot) held in register r by lane half h of a flipped tile; also the feature offset
// inside a std tile.
__device__ __forceinline__ constexpr int ag_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x4 ag_ld4(const float* __restrict__ p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void ag_st4(float* __restrict__ p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// ---------------------------------------------------------------------------------- math
// Per-element activations run 64..96 times per lane per layer, fully unrolled, so they are kept
// branch-free and short: v_exp_f32 / v_log_f32 / v_rcp_f32 based (<= ~1e-6 relative).
// raw v_exp_f32 / v_log_f32 (base 2, ~1 ulp, no denormal fix-up code): arguments are clamped by the
// callers so that neither overflows; results below 2^-126 flush to zero, which every caller tolerates
// (they are added to 1 or subtracted from 1).
__device__ __forceinline__ float ag_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float ag_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float ag_exp(float x) { return ag_exp2(fmaxf(x, -125.0f * 0.69314718f) * 1.44269504088896340736f); }
__device__ __forceinline__ float ag_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// torch F.gelu (erf form, edge.py:59,68,86): gelu(x) = x * Phi(x), Phi(x) = erfc(-x/sqrt2)/2.
// One range, branch-free: erfc(t)/2 = 2^(p(t)) for t = min(|x|/sqrt2, 4.1), p = -1 - log2(e) * g(t) with g a
// degree-8 fit of -ln erfc(t) weighted by erfc (absolute error of Phi <= 4e-8 in fp32, gelu within
// 4e-7 absolute / 1.2e-7 * |x|; numpy mirror checked against torch in tests/test_host_logic.py).
__device__ __forceinline__ float ag_gelu(float x) {
  const float t = fminf(fabsf(x) * 0.70710678118654752440f, 4.1f);
  float p = -4.535698463e-05f;
  p = fmaf(p, t, 4.454943992e-04f);
  p = fmaf(p, t, -1.489399001e-03f);
  p = fmaf(p, t, -7.746984484e-04f);
  p = fmaf(p, t, 2.825373970e-02f);
  p = fmaf(p, t, -1.484816372e-01f);
  p = fmaf(p, t, -9.184163809e-01f);
  p = fmaf(p, t, -1.627908587e+00f);
  p = fmaf(p, t, -1.0f);
  const float q = ag_exp2(p);                 // erfc(t) / 2
  return x * (x >= 0.0f ? 1.0f - q : q);
}
// schnet.py:71-80: softplus(beta*x) - log 2 with torch's threshold 20 (softplus(z) = z for z > 20).
// softplus(z) >= z and equals z to fp32 precision beyond ~17, so max(z, log(1 + e^z)) reproduces the
// threshold form within 1 ulp without a compare/select; the exponent is clamped so that 2^t stays finite.
__device__ __forceinline__ float ag_ssp(float beta, float x) {
  const float z = beta * x;
  const float t = fminf(z * 1.44269504088896340736f, 126.0f);
  const float l = ag_log2(1.0f + ag_exp2(t)) * 0.69314718055994530942f;
  return fmaxf(z, l) - 0.69314718055994530942f;
}
__device__ __forceinline__ float ag_sigmoid(float x) {
  return ag_rcp(1.0f + ag_exp2(fminf(-x * 1.44269504088896340736f, 126.0f)));
}
__device__ __forceinline__ float ag_relu(float x) { return x > 0.0f ? x : 0.0f; }
__device__ __forceinline__ float ag_lrelu(float x) { return x > 0.0f ? x : 0.2f * x; }

// ---------------------------------------------------------------------------------- tiles
// Fill std-orientation tiles from a natural-order vector (bias init): y[t][r] = v[32t + row(r,h)].
template <int MT>
__device__ __forceinline__ void ag_init_vec(f32x16 (&y)[MT], const float* __restrict__ v, int h) {
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 b = ag_ld4(v + 32 * t + 8 * rq + 4 * h);
#pragma unroll
      for (int q = 0; q < 4; ++q) y[t][4 * rq + q] = b[q];
    }
}

// Load a row-major row (stride given by caller through `row`) into std tiles T0..T0+NTL-1 of y.
template <int NTL, int T0, int MT>
__device__ __forceinline__ void ag_load_row(f32x16 (&y)[MT], const float* __restrict__ row, int h) {
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 b = ag_ld4(row + 32 * t + 8 * rq + 4 * h);
#pragma unroll
      for (int q = 0; q < 4; ++q) y[T0 + t][4 * rq + q] = b[q];
    }
}

template <int NTL, int T0, int MT>
__device__ __forceinline__ void ag_store_row(const f32x16 (&y)[MT], float* __restrict__ row, int h) {
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 b;
#pragma unroll
      for (int q = 0; q < 4; ++q) b[q] = y[T0 + t][4 * rq + q];
      ag_st4(row + 32 * t + 8 * rq + 4 * h, b);
    }
}

// Fragment-major edge-attr tiles: [tile][4][4][64][4] floats (512 B per edge, coalesced 1 KiB per
// wave instruction).
__device__ __forceinline__ size_t ag_frag_off(int64_t tile, int t, int rq, int lane) {
  return ((size_t)((tile * 4 + t) * 4 + rq) * 64 + lane) * 4;
}
// Same storage addressed by (edge e, feature f multiple of 4): used for gathers of single edges.
__device__ __forceinline__ size_t ag_frag_off_ef(int64_t e, int f) {
  int64_t tile = e >> 5;
  int j = (int)(e & 31), t = f >> 5, w = f & 31;
  int rq = w >> 3, hh = (w >> 2) & 1;
  return ag_frag_off(tile, t, rq, j + 32 * hh);
}

template <int T0, int MT>
__device__ __forceinline__ void ag_load_frag(f32x16 (&y)[MT], const float* __restrict__ frag, int64_t tile, int lane) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 b = ag_ld4(frag + ag_frag_off(tile, t, rq, lane));
#pragma unroll
      for (int q = 0; q < 4; ++q) y[T0 + t][4 * rq + q] = b[q];
    }
}

// one 32-feature slice t of a fragment-major tile into y[T0]
template <int T0, int MT>
__device__ __forceinline__ void ag_load_frag_tile(f32x16 (&y)[MT], const float* __restrict__ frag, int64_t tile, int t, int lane) {
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    f32x4 b = ag_ld4(frag + ag_frag_off(tile, t, rq, lane));
#pragma unroll
    for (int q = 0; q < 4; ++q) y[T0][4 * rq + q] = b[q];
  }
}

template <int MT>
__device__ __forceinline__ void ag_store_frag(const f32x16 (&y)[MT], float* __restrict__ frag, int64_t tile, int lane) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 b;
#pragma unroll
      for (int q = 0; q < 4; ++q) b[q] = y[t][4 * rq + q];
      ag_st4(frag + ag_frag_off(tile, t, rq, lane), b);
    }
}

// ---------------------------------------------------------------------------------- dense layers
// Two arithmetic modes share every kernel (template parameter MODE):
//   AG_F32: v_mfma_f32_32x32x2_f32, exact fp32 (k-ordered fmaf chain).  One weight block = 4 x 16 B per
//           lane = 16 MFMAs (1024 SIMD cycles).
//   AG_BF3: "split bf16": every fp32 operand is hi + lo with hi = bf16(x), lo = bf16(x - hi); a product is
//           hi.hi + lo.hi + hi.lo on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (~2^-16 relative per
//           product, the dropped term is lo.lo).  Same C/D layout, so the register-resident chaining is
//           unchanged; k-slot (s, h, j) of tile t carries feature 32t + 16s + 8(j>>2) + 4h + (j&3), i.e.
//           accumulator registers 8s..8s+7 in order.  One weight block = [s][hi,lo][64 lanes][8 bf16] =
//           again 4 x 16 B per lane, 6 MFMAs (192 SIMD cycles).
// Both run the same software pipeline: the loads of step s+PF are issued before the MFMAs of step s;
// __builtin_amdgcn_sched_barrier(0) between steps keeps hipcc from hoisting every weight load of the
// layer to its top (which costs > 256 VGPRs and spills).
enum { AG_F32 = 0, AG_BF3 = 1 };
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE> struct AgIn;                       // one 32-feature input tile in MFMA-operand form
template <> struct AgIn<AG_F32> { f32x16 v; };
template <> struct AgIn<AG_BF3> { bf16x8 hi[2], lo[2]; };

__device__ __forceinline__ void ag_cvt(const f32x16& a, AgIn<AG_F32>& o) { o.v = a; }
__device__ __forceinline__ void ag_cvt(const f32x16& a, AgIn<AG_BF3>& o) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = a[8 * s + j];
      const __bf16 hb = (__bf16)v;
      o.hi[s][j] = hb;
      o.lo[s][j] = (__bf16)(v - (float)hb);
    }
}
template <int MODE, int NT, int A0, int NA, int NO>
__device__ __forceinline__ void ag_cvt_tiles(const f32x16 (&a)[NA], AgIn<MODE> (&o)[NO]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) ag_cvt(a[A0 + t], o[t]);
}

__device__ __forceinline__ u32x4 ag_ldu(const u32x4* p) { return *p; }

// o[O0 + ot] += sum over k-tiles t of W-block(ot, t) x x[X0 + t]; FLIP selects the orientation (see top).
// Blocks are contiguous in iteration order: output-tile-outer ("pk") or k-tile-outer ("pkk", KOUTER).
// KPART = number of 8-feature groups used in the LAST k-tile (4 = all 32 features).
template <int MODE, bool FLIP, bool KOUTER, int KT, int OT, int X0, int O0, int KPART, int PF, int NX, int NO>
__device__ __forceinline__ void ag_dense_impl(const AgIn<MODE> (&x)[NX], f32x16 (&o)[NO], const void* wpk, int lane) {
  static_assert(X0 + KT <= NX && O0 + OT <= NO, "tile range");
  constexpr int S = OT * KT;
  constexpr int R = PF + 1;
  u32x4 w[R][4];
  const u32x4* wl = reinterpret_cast<const u32x4*>(wpk) + lane;
  auto tile_of = [](int s) { return KOUTER ? s / OT : s % KT; };
  auto units = [&](int s) {   // 16-byte units of block s that are actually used
    const bool last = tile_of(s) == KT - 1;
    return MODE == AG_F32 ? (last ? KPART : 4) : (last ? 2 * ((KPART + 1) / 2) : 4);
  };
#pragma unroll
  for (int s = 0; s < (PF < S ? PF : S); ++s)
#pragma unroll
    for (int u = 0; u < units(s); ++u) w[s % R][u] = ag_ldu(wl + (s * 4 + u) * 64);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int t = tile_of(s), ot = KOUTER ? s % OT : s / KT;
    if (s + PF < S) {
#pragma unroll
      for (int u = 0; u < units(s + PF); ++u) w[(s + PF) % R][u] = ag_ldu(wl + ((s + PF) * 4 + u) * 64);
    }
    if constexpr (MODE == AG_F32) {
#pragma unroll
      for (int rq = 0; rq < units(s); ++rq) {
        const f32x4 wf = __builtin_bit_cast(f32x4, w[s % R][rq]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (FLIP)
            o[O0 + ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[X0 + t].v[4 * rq + q], wf[q], o[O0 + ot], 0, 0, 0);
          else
            o[O0 + ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q], x[X0 + t].v[4 * rq + q], o[O0 + ot], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < units(s) / 2; ++ks) {
        const bf16x8 whi = __builtin_bit_cast(bf16x8, w[s % R][2 * ks]);
        const bf16x8 wlo = __builtin_bit_cast(bf16x8, w[s % R][2 * ks + 1]);
        const bf16x8 xhi = x[X0 + t].hi[ks], xlo = x[X0 + t].lo[ks];
        if (FLIP) {
          o[O0 + ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xhi, whi, o[O0 + ot], 0, 0, 0);
          o[O0 + ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xlo, whi, o[O0 + ot], 0, 0, 0);
          o[O0 + ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xhi, wlo, o[O0 + ot], 0, 0, 0);
        } else {
          o[O0 + ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, xhi, o[O0 + ot], 0, 0, 0);
          o[O0 + ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, xlo, o[O0 + ot], 0, 0, 0);
          o[O0 + ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, xhi, o[O0 + ot], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// one weight block (4 x 16 B per lane, already in registers) applied to one input tile
template <int MODE, bool FLIP>
__device__ __forceinline__ void ag_block_mma(f32x16& o, const AgIn<MODE>& x, const u32x4 (&w)[4]) {
  if constexpr (MODE == AG_F32) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const f32x4 wf = __builtin_bit_cast(f32x4, w[rq]);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        o = FLIP ? __builtin_amdgcn_mfma_f32_32x32x2f32(x.v[4 * rq + q], wf[q], o, 0, 0, 0)
                 : __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q], x.v[4 * rq + q], o, 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 whi = __builtin_bit_cast(bf16x8, w[2 * ks]), wlo = __builtin_bit_cast(bf16x8, w[2 * ks + 1]);
      if (FLIP) {
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.hi[ks], whi, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.lo[ks], whi, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.hi[ks], wlo, o, 0, 0, 0);
      } else {
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, x.hi[ks], o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, x.lo[ks], o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, x.hi[ks], o, 0, 0, 0);
      }
    }
  }
}

template <int MODE> struct AgPF { static constexpr int v = (MODE == AG_F32) ? 1 : 2; };

// y[M0 + mt] += W . x   (std orientation: features on registers), "pk" blocks [MT][KT]
template <int MODE, int KT, int MT, int X0, int M0, int KPART, int PF = AgPF<MODE>::v, int NX, int NY>
__device__ __forceinline__ void ag_dense_std(const AgIn<MODE> (&x)[NX], f32x16 (&y)[NY], const void* wpk, int lane) {
  ag_dense_impl<MODE, false, false, KT, MT, X0, M0, KPART, PF>(x, y, wpk, lane);
}
// same with "pkk" blocks [KT][MT]: a caller that streams its input in 32-feature slices passes KT = 1
// and the slice's block offset (slice t starts at block t*MT)
template <int MODE, int KT, int MT, int X0, int M0, int PF = AgPF<MODE>::v, int NX, int NY>
__device__ __forceinline__ void ag_dense_std_k(const AgIn<MODE> (&x)[NX], f32x16 (&y)[NY], const void* wpk, int lane) {
  ag_dense_impl<MODE, false, true, KT, MT, X0, M0, 4, PF>(x, y, wpk, lane);
}
// z[N0 + nt][row = edge] (lane = feature) += x^T . W^T   (flip orientation), "pk" blocks
template <int MODE, int KT, int NT, int X0, int N0, int PF = AgPF<MODE>::v, int NX, int NZ>
__device__ __forceinline__ void ag_dense_flip(const AgIn<MODE> (&x)[NX], f32x16 (&z)[NZ], const void* wpk, int lane) {
  ag_dense_impl<MODE, true, false, KT, NT, X0, N0, 4, PF>(x, z, wpk, lane);
}
// The same from LDS-resident weight blocks (address space known to the compiler: ds_read_b128).
template <int MODE, int KT, int NT, int X0, int N0, int NX, int NZ>
__device__ __forceinline__ void ag_dense_flip_lds(const AgIn<MODE> (&x)[NX], f32x16 (&z)[NZ],
                                                  const __attribute__((address_space(3))) u32x4* wl, int lane) {
  static_assert(X0 + KT <= NX && N0 + NT <= NZ, "tile range");
#pragma unroll
  for (int s = 0; s < NT * KT; ++s) {
    const int t = s % KT, ot = s / KT;
    u32x4 w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = wl[(s * 4 + u) * 64 + lane];
    if constexpr (MODE == AG_F32) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 wf = __builtin_bit_cast(f32x4, w[rq]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          z[N0 + ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[X0 + t].v[4 * rq + q], wf[q], z[N0 + ot], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 whi = __builtin_bit_cast(bf16x8, w[2 * ks]), wlo = __builtin_bit_cast(bf16x8, w[2 * ks + 1]);
        z[N0 + ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[X0 + t].hi[ks], whi, z[N0 + ot], 0, 0, 0);
        z[N0 + ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[X0 + t].lo[ks], whi, z[N0 + ot], 0, 0, 0);
        z[N0 + ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[X0 + t].hi[ks], wlo, z[N0 + ot], 0, 0, 0);
      }
    }
  }
}
// weight block b of a packed matrix (both modes: 4 KiB per block)
__device__ __forceinline__ const void* ag_wblock(const float* wpk, int b) { return wpk + (size_t)b * 1024; }

// z (flip orientation: rows = edges, lanes = features) += s (per edge, on its lane) (x) b (per feature, on
// its lane): the outer product as one extra k-step whose only non-zero k-slot is (lane half 0, element 0).
__device__ __forceinline__ void ag_rank1(f32x16& z, float s_edge, float b_feat, int h, AgIn<AG_F32>*) {
  z = __builtin_amdgcn_mfma_f32_32x32x2f32(h == 0 ? s_edge : 0.0f, h == 0 ? b_feat : 0.0f, z, 0, 0, 0);
}
__device__ __forceinline__ void ag_rank1(f32x16& z, float s_edge, float b_feat, int h, AgIn<AG_BF3>*) {
  const float sv = h == 0 ? s_edge : 0.0f, bv = h == 0 ? b_feat : 0.0f;
  const __bf16 sh = (__bf16)sv, bh = (__bf16)bv;
  const __bf16 sl = (__bf16)(sv - (float)sh), bl = (__bf16)(bv - (float)bh);
  bf16x8 ah = {}, al = {}, bhv = {}, blv = {};
  ah[0] = sh; al[0] = sl; bhv[0] = bh; blv[0] = bl;
  z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhv, z, 0, 0, 0);
  z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhv, z, 0, 0, 0);
  z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blv, z, 0, 0, 0);
}

// ---------------------------------------------------------------------------------- edge-attr storage
// e_attr / l_attr tiles are stored in the operand form of the mode that consumes them, 4 x 16 B per lane
// per 32-feature slice, unit index ((tile*4 + t)*4 + u)*64 + lane:
//   AG_F32: unit u = register group rq (the accumulator dumped as-is)
//   AG_BF3: unit u = 2*s + part (part 0 = hi, 1 = lo) of k-step s
__device__ __forceinline__ void ag_store_attr_slice(const AgIn<AG_F32>& x, float* frag, int64_t tile, int t, int lane) {
  u32x4* p = reinterpret_cast<u32x4*>(frag) + ((tile * 4 + t) * 4) * 64 + lane;
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    f32x4 b;
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = x.v[4 * rq + q];
    p[rq * 64] = __builtin_bit_cast(u32x4, b);
  }
}
__device__ __forceinline__ void ag_store_attr_slice(const AgIn<AG_BF3>& x, float* frag, int64_t tile, int t, int lane) {
  u32x4* p = reinterpret_cast<u32x4*>(frag) + ((tile * 4 + t) * 4) * 64 + lane;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    p[(2 * s) * 64] = __builtin_bit_cast(u32x4, x.hi[s]);
    p[(2 * s + 1) * 64] = __builtin_bit_cast(u32x4, x.lo[s]);
  }
}
__device__ __forceinline__ void ag_load_attr_slice(AgIn<AG_F32>& x, const float* frag, int64_t tile, int t, int lane) {
  const u32x4* p = reinterpret_cast<const u32x4*>(frag) + ((tile * 4 + t) * 4) * 64 + lane;
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    const f32x4 b = __builtin_bit_cast(f32x4, p[rq * 64]);
#pragma unroll
    for (int q = 0; q < 4; ++q) x.v[4 * rq + q] = b[q];
  }
}
__device__ __forceinline__ void ag_load_attr_slice(AgIn<AG_BF3>& x, const float* frag, int64_t tile, int t, int lane) {
  const u32x4* p = reinterpret_cast<const u32x4*>(frag) + ((tile * 4 + t) * 4) * 64 + lane;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    x.hi[s] = __builtin_bit_cast(bf16x8, p[(2 * s) * 64]);
    x.lo[s] = __builtin_bit_cast(bf16x8, p[(2 * s + 1) * 64]);
  }
}
// four consecutive features f..f+3 (f % 4 == 0) of edge e as fp32 (GIN message gather)
template <int MODE>
__device__ __forceinline__ f32x4 ag_attr_gather4(const float* frag, int64_t e, int f) {
  const int64_t tile = e >> 5;
  const int j = (int)(e & 31), t = f >> 5, w = f & 31;
  if constexpr (MODE == AG_F32) {
    const int rq = w >> 3, hh = (w >> 2) & 1;
    return ag_ld4(frag + (((tile * 4 + t) * 4 + rq) * 64 + j + 32 * hh) * 4);
  } else {
    const int s = w >> 4, jq = (w >> 3) & 1, hh = (w >> 2) & 1;
    const unsigned short* base = reinterpret_cast<const unsigned short*>(frag) +
                                 ((((tile * 4 + t) * 4 + 2 * s) * 64 + j + 32 * hh) * 8 + 4 * jq);
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    const u16x4 hi = *reinterpret_cast<const u16x4*>(base);
    const u16x4 lo = *reinterpret_cast<const u16x4*>(base + 64 * 8);
    f32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      r[q] = __uint_as_float((unsigned)hi[q] << 16) + __uint_as_float((unsigned)lo[q] << 16);
    return r;
  }
}

// dot product over the features of std tiles [0, MT) with a natural-order weight vector:
// returns sum_f w[f] * y[f] for the lane's edge (both halves hold the total).
template <int MT, int NY>
__device__ __forceinline__ float ag_dot_vec(const f32x16 (&y)[NY], const float* __restrict__ w, int h) {
  float s = 0.0f;
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 b = ag_ld4(w + 32 * t + 8 * rq + 4 * h);
#pragma unroll
      for (int q = 0; q < 4; ++q) s = fmaf(b[q], y[t][4 * rq + q], s);
    }
  s += __shfl_xor(s, 32);
  return s;
}

#define AG_FOR_TILE(y, MT, expr)                 \
  _Pragma("unroll") for (int _t = 0; _t < (MT); ++_t) \
  _Pragma("unroll") for (int _r = 0; _r < 16; ++_r) { float v = (y)[_t][_r]; (y)[_t][_r] = (expr); }

// Host-side launch check
#define AG_CHECK_LAUNCH()                                         \
  do {                                                            \
    if (hipGetLastError() != hipSuccess) return AGDIFF_ERR_LAUNCH; \
  } while (0)
