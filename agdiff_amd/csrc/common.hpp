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
// branch-free and short: v_exp_f32 / v_log_f32 / v_rcp_f32 based (<= ~1e-6 relative), and an erf
// that evaluates both polynomial ranges and selects (max error < 1 ulp; coefficients checked
// against scipy.special.erf in tests/test_host_logic.py through their numpy mirror).
__device__ __forceinline__ float ag_exp(float x) { return __expf(x); }
__device__ __forceinline__ float ag_log(float x) { return __logf(x); }
__device__ __forceinline__ float ag_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ float ag_erf(float a) {
  const float t = fabsf(a), s = a * a;
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  const float big = copysignf(1.0f - ag_exp(r), a);
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  const float small = fmaf(q, a, a);
  return t > 0.927734375f ? big : small;
}
__device__ __forceinline__ float ag_gelu(float x) {  // torch F.gelu (erf form), edge.py:59,68,86
  return 0.5f * x * (1.0f + ag_erf(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float ag_ssp(float beta, float x) {  // schnet.py:71-80, softplus threshold 20
  const float z = beta * x;
  const float sp = z > 20.0f ? z : ag_log(1.0f + ag_exp(z));
  return sp - 0.69314718055994530942f;
}
__device__ __forceinline__ float ag_sigmoid(float x) { return ag_rcp(1.0f + ag_exp(-x)); }
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
// Both orientations run the same software pipeline: one "step" = one (output tile, k-tile) pair =
// 4 x 16-byte weight loads per lane + 16 MFMAs (1024 SIMD cycles).  The loads of step s+1 are issued
// before the MFMAs of step s; __builtin_amdgcn_sched_barrier(0) between steps keeps hipcc from
// hoisting every weight load of the layer to its top (which costs > 256 VGPRs and spills).
template <bool FLIP, bool KOUTER, int KT, int OT, int X0, int O0, int RQL, int NX, int NO>
__device__ __forceinline__ void ag_dense_impl(const f32x16 (&x)[NX], f32x16 (&o)[NO], const float* __restrict__ wpk, int lane) {
  static_assert(X0 + KT <= NX && O0 + OT <= NO, "tile range");
  constexpr int S = OT * KT;
  f32x4 w[2][4];
  const float* wl = wpk + (size_t)lane * 4;
#pragma unroll
  for (int rq = 0; rq < ((KT == 1) ? RQL : 4); ++rq) w[0][rq] = ag_ld4(wl + (size_t)rq * 256);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int t = KOUTER ? s / OT : s % KT, ot = KOUTER ? s % OT : s / KT;
    if (s + 1 < S) {
      const int tn = KOUTER ? (s + 1) / OT : (s + 1) % KT;
#pragma unroll
      for (int rq = 0; rq < ((tn == KT - 1) ? RQL : 4); ++rq)
        w[(s + 1) & 1][rq] = ag_ld4(wl + (size_t)((s + 1) * 4 + rq) * 256);
    }
#pragma unroll
    for (int rq = 0; rq < ((t == KT - 1) ? RQL : 4); ++rq)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (FLIP)
          o[O0 + ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[X0 + t][4 * rq + q], w[s & 1][rq][q], o[O0 + ot], 0, 0, 0);
        else
          o[O0 + ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s & 1][rq][q], x[X0 + t][4 * rq + q], o[O0 + ot], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// y[M0 + mt] += W[mt-th row block] . x[X0 .. X0+KT)   (std orientation: features on registers).
// Weights packed output-tile-outer ("pk": [MT][KT] blocks).  RQL = number of 8-feature groups used
// in the last k-tile (4 = all 32 features).
template <int KT, int MT, int X0, int M0, int RQL, int NX, int NY>
__device__ __forceinline__ void ag_dense_std(const f32x16 (&x)[NX], f32x16 (&y)[NY], const float* __restrict__ wpk, int lane) {
  ag_dense_impl<false, false, KT, MT, X0, M0, RQL>(x, y, wpk, lane);
}
// Same with k-tile-outer packing ("pkk": [KT][MT] blocks): a caller that streams its input in
// 32-feature slices passes KT = 1 and the slice's block offset (slice t starts at t*MT*1024 floats).
template <int KT, int MT, int X0, int M0, int NX, int NY>
__device__ __forceinline__ void ag_dense_std_k(const f32x16 (&x)[NX], f32x16 (&y)[NY], const float* __restrict__ wpk, int lane) {
  ag_dense_impl<false, true, KT, MT, X0, M0, 4>(x, y, wpk, lane);
}

// z[N0 + nt][row = edge] (lane = feature) += x^T . W^T   (flip orientation), "pk" weights.
template <int KT, int NT, int X0, int N0, int NX, int NZ>
__device__ __forceinline__ void ag_dense_flip(const f32x16 (&x)[NX], f32x16 (&z)[NZ], const float* __restrict__ wpk, int lane) {
  ag_dense_impl<true, false, KT, NT, X0, N0, 4>(x, z, wpk, lane);
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
