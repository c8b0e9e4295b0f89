// Device-side building blocks shared by all kernels (gfx950 / CDNA4 only).
//
// Register-resident MLP chains on 16x16 MFMA tiles: a wave owns one tile of AG_TW = 16 edges (or
// nodes).  A 16-feature slice of an activation is one accumulator tile `f32x4` in the MFMA C/D layout
//     lane l  <->  edge (l & 15),  quarter q = l >> 4
//     y[r]    <->  feature 16*T + 4*q + r
// so a 128-feature activation is 32 VGPRs per lane and a whole filter / encoder chain fits in <= 128
// VGPRs: four waves per SIMD instead of the two a 32-edge tile allows, which is what lets one wave's
// MFMAs run beside another wave's VALU work and hides LDS / L2 latency.
// With weights pre-packed in the matching k-order ("pk", include/agdiff_hip.h) the accumulator of one
// layer IS the operand of the next one: no LDS, no shuffles between layers.
//   std  orientation: D[feature][edge] = W . X      (A = weights, B = activations)
//   flip orientation: D[edge][feature] = X^T . W^T  (A = activations, B = weights)
// The flipped result has edges on rows (registers x quarters) and features on lanes, which is what the
// destination-segmented reduction and coalesced row gathers want.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "agdiff_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;

#define AG_WAVE 64
#define AG_WG 256                 // 4 waves per workgroup for the non-persistent kernels
#define AG_TW AGDIFF_TILE         // 16 edges / nodes per tile

__device__ __forceinline__ int ag_lane() { return threadIdx.x & 63; }
__device__ __forceinline__ int ag_wave_in_wg() { return threadIdx.x >> 6; }

__device__ __forceinline__ f32x4 ag_ld4(const float* __restrict__ p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void ag_st4(float* __restrict__ p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// ---------------------------------------------------------------------------------- math
// Per-element activations run 32..48 times per lane per layer, fully unrolled, so they are kept
// branch-free and short: raw v_exp_f32 / v_log_f32 / v_rcp_f32 (base 2, ~1 ulp, no denormal fix-up
// code).  Arguments are clamped by the callers so that nothing overflows; results below 2^-126 flush
// to zero, which every caller tolerates (they are added to 1 or subtracted from 1).
__device__ __forceinline__ float ag_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float ag_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float ag_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// torch F.gelu (erf form, edge.py:59,68,86): gelu(x) = x * Phi(x), Phi(x) = erfc(-x/sqrt2)/2.
// One range, branch-free: erfc(t)/2 = 2^(p(t)) for t = min(|x|/sqrt2, 4.1), p(t) = -1 + t (c1 + c2 t + ... + c6 t^5)
// a degree-6 fit of log2(erfc(t)/2) weighted towards the absolute error of erfc (Phi within 1.2e-7, gelu within
// 4.3e-7 absolute / 1.6e-7 * |x| in fp32; numpy mirror checked against torch in tests/test_host_logic.py).
__device__ __forceinline__ float ag_gelu(float x) {
  const float t = fminf(fabsf(x) * 0.70710678118654752440f, 4.1f);
  float p = 1.420383199e-04f;
  p = fmaf(p, t, -3.664264106e-03f);
  p = fmaf(p, t, 3.089617305e-02f);
  p = fmaf(p, t, -1.496994283e-01f);
  p = fmaf(p, t, -9.181654693e-01f);
  p = fmaf(p, t, -1.627925070e+00f);
  p = fmaf(p, t, -1.0f);
  const float q = ag_exp2(p);                 // erfc(t) / 2
  // x * (x >= 0 ? 1 - q : q)  ==  max(x, 0) - |x| q   (two instructions instead of four)
  return fmaf(-fabsf(x), q, fmaxf(x, 0.0f));
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
// The same in base 2 for callers whose surrounding linear layers carry the constants (agdiff_conv_params_t):
// u = beta x log2(e)  ->  max(u, log2(1 + 2^u)) = (softplus(beta x)) / ln 2.
__device__ __forceinline__ float ag_ssp_base2(float u) {
  return fmaxf(u, ag_log2(1.0f + ag_exp2(fminf(u, 126.0f))));
}
__device__ __forceinline__ float ag_sigmoid(float x) {
  return ag_rcp(1.0f + ag_exp2(fminf(-x * 1.44269504088896340736f, 126.0f)));
}
__device__ __forceinline__ float ag_relu(float x) { return x > 0.0f ? x : 0.0f; }
__device__ __forceinline__ float ag_lrelu(float x) { return x > 0.0f ? x : 0.2f * x; }

// Sum of a value over the four quarters of the wave (lanes l, l^16, l^32, l^48), result in every lane:
// two VALU lane-swap instructions (gfx950 v_permlane16_swap / v_permlane32_swap) instead of two trips
// through the LDS crossbar (ds_bpermute).  Association: (q0 + q1) + (q2 + q3).
__device__ __forceinline__ float ag_quarter_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const unsigned w = __float_as_uint(s);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Reduce-scatter of four values over the quarters: returns, in the lanes of quarter j, the sum of v[j] over the
// four quarters (same lane & 15); three lane-swap instructions for four values instead of two per value.
// v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of the second,
// v_permlane32_swap the upper half of the first with the lower half of the second.  Association per value:
// (q0 + q1) + (q2 + q3), as ag_quarter_sum.
__device__ __forceinline__ float ag_quarter_reduce_scatter4(float v0, float v1, float v2, float v3) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v2), __float_as_uint(v3), false, false);
  const float s01 = __uint_as_float(a[0]) + __uint_as_float(a[1]);   // rows: v0(q0+q1) v1(q0+q1) v0(q2+q3) v1(q2+q3)
  const float s23 = __uint_as_float(b[0]) + __uint_as_float(b[1]);
  const auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(s01), __float_as_uint(s23), false, false);
  return __uint_as_float(c[0]) + __uint_as_float(c[1]);              // rows: v0 v1 v2 v3
}

// ---------------------------------------------------------------------------------- fp32 tiles
// Fill std-orientation tiles from a natural-order vector (bias init): y[T][r] = v[16T + 4q + r].
template <int NT, int NY>
__device__ __forceinline__ void ag_init_vec(f32x4 (&y)[NY], const float* __restrict__ v, int q) {
#pragma unroll
  for (int t = 0; t < NT; ++t) y[t] = ag_ld4(v + 16 * t + 4 * q);
}
// Row-major row (natural feature order) <-> std tiles T0 .. T0+NT-1.
template <int NT, int T0, int NY>
__device__ __forceinline__ void ag_load_row(f32x4 (&y)[NY], const float* __restrict__ row, int q) {
#pragma unroll
  for (int t = 0; t < NT; ++t) y[T0 + t] = ag_ld4(row + 16 * t + 4 * q);
}
template <int NT, int T0, int NY>
__device__ __forceinline__ void ag_store_row(const f32x4 (&y)[NY], float* __restrict__ row, int q) {
#pragma unroll
  for (int t = 0; t < NT; ++t) ag_st4(row + 16 * t + 4 * q, y[T0 + t]);
}
// sum_f w[f] * y[f] over the first NT tiles for the lane's edge (all four quarters hold the total)
template <int NT, int NY>
__device__ __forceinline__ float ag_dot_vec(const f32x4 (&y)[NY], const float* __restrict__ w, int q) {
  float s = 0.0f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f32x4 b = ag_ld4(w + 16 * t + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r) s = fmaf(b[r], y[t][r], s);
  }
  return ag_quarter_sum(s);
}
#define AG_FOR_TILE(y, NT, expr)                       \
  _Pragma("unroll") for (int _t = 0; _t < (NT); ++_t)  \
  _Pragma("unroll") for (int _r = 0; _r < 4; ++_r) { float v = (y)[_t][_r]; (y)[_t][_r] = (expr); }

// MultiLayerPerceptron's activation (models/common.py:62-66: getattr(F, config.mlp_act)) between the layers of the two heads;
// agdiff_head_params_t.act (AGDIFF_ACT_*).  `act` is uniform: one scalar branch per layer and tile.
__device__ __forceinline__ float ag_tanh(float x) {            // 1 - 2 / (1 + e^{2x}): absolute error ~1e-7
  return 1.0f - 2.0f * ag_rcp(1.0f + ag_exp2(fminf(x * 2.88539008177792681472f, 126.0f)));
}
__device__ __forceinline__ float ag_softplus(float x) {        // F.softplus(beta = 1, threshold = 20)
  return fmaxf(x, ag_log2(1.0f + ag_exp2(fminf(x * 1.44269504088896340736f, 126.0f))) * 0.69314718055994530942f);
}
template <int NT, int NY>
__device__ __forceinline__ void ag_head_act(f32x4 (&y)[NY], int act) {
  switch (act) {
    case AGDIFF_ACT_GELU: AG_FOR_TILE(y, NT, ag_gelu(v)); break;
    case AGDIFF_ACT_SILU: AG_FOR_TILE(y, NT, v * ag_sigmoid(v)); break;
    case AGDIFF_ACT_TANH: AG_FOR_TILE(y, NT, ag_tanh(v)); break;
    case AGDIFF_ACT_SIGMOID: AG_FOR_TILE(y, NT, ag_sigmoid(v)); break;
    case AGDIFF_ACT_SOFTPLUS: AG_FOR_TILE(y, NT, ag_softplus(v)); break;
    case AGDIFF_ACT_LEAKY_RELU: AG_FOR_TILE(y, NT, (v > 0.0f ? v : 0.01f * v)); break;
    case AGDIFF_ACT_ELU: AG_FOR_TILE(y, NT, (v > 0.0f ? v : ag_exp2(v * 1.44269504088896340736f) - 1.0f)); break;
    case AGDIFF_ACT_RELU6: AG_FOR_TILE(y, NT, fminf(fmaxf(v, 0.0f), 6.0f)); break;
    case AGDIFF_ACT_HARDTANH: AG_FOR_TILE(y, NT, fminf(fmaxf(v, -1.0f), 1.0f)); break;
    case AGDIFF_ACT_SELU:
      AG_FOR_TILE(y, NT, 1.0507009873554804934f * (v > 0.0f ? v : 1.6732632423543772848f * (ag_exp2(v * 1.44269504088896340736f) - 1.0f)));
      break;
    case AGDIFF_ACT_MISH: AG_FOR_TILE(y, NT, v * ag_tanh(ag_softplus(v))); break;
    case AGDIFF_ACT_HARDSWISH: AG_FOR_TILE(y, NT, v * fminf(fmaxf(v + 3.0f, 0.0f), 6.0f) * (1.0f / 6.0f)); break;
    case AGDIFF_ACT_HARDSIGMOID: AG_FOR_TILE(y, NT, fminf(fmaxf(v + 3.0f, 0.0f), 6.0f) * (1.0f / 6.0f)); break;
    case AGDIFF_ACT_SOFTSIGN: AG_FOR_TILE(y, NT, v * ag_rcp(1.0f + fabsf(v))); break;
    case AGDIFF_ACT_LOGSIGMOID: AG_FOR_TILE(y, NT, -ag_softplus(-v)); break;
    case AGDIFF_ACT_HARDSHRINK: AG_FOR_TILE(y, NT, fabsf(v) > 0.5f ? v : 0.0f); break;
    case AGDIFF_ACT_SOFTSHRINK: AG_FOR_TILE(y, NT, v > 0.5f ? v - 0.5f : v < -0.5f ? v + 0.5f : 0.0f); break;
    case AGDIFF_ACT_RRELU: AG_FOR_TILE(y, NT, v >= 0.0f ? v : v * ((1.0f / 8.0f + 1.0f / 3.0f) / 2.0f)); break;
    default: AG_FOR_TILE(y, NT, ag_relu(v)); break;
  }
}

// ---------------------------------------------------------------------------------- MFMA operands
// Two arithmetic modes share every kernel (template parameter MODE):
//   AG_F32: v_mfma_f32_16x16x4_f32, exact fp32 (k-ordered fmaf chain).
//   AG_BF3: "split bf16": every fp32 operand is hi + lo with hi = bf16(x), lo = bf16(x - hi); a product is
//           hi.hi + lo.hi + hi.lo on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (~2^-16 relative per
//           product; the dropped term is lo.lo).
// The unit of input is a "k-tile" of 32 features = two consecutive accumulator tiles (2m, 2m+1):
//   AG_F32: the two f32x4 as they are; MFMA (u, r) contracts feature 32m + 16u + 4q + r over the quarters q
//   AG_BF3: one bf16x8 (hi) + one (lo); element j <-> feature 32m + 16*(j>>2) + 4q + (j&3), i.e. the two
//           accumulator tiles' registers in order
// One weight block = 16 outputs x 32 inputs = 2 x 16 B per lane in both modes (2 KiB):
//   lane l holds W[16*ot + (l&15)][32*t + {4q..4q+3} U {16+4q..16+4q+3}], q = l>>4
//   AG_F32: unit u = the four fp32 of half u;  AG_BF3: unit 0 = the eight bf16 hi, unit 1 = the eight lo.
//   AG_H3:  "split fp16": the same three-pass scheme with hi = fp16(x), lo = fp16(x - hi) on v_mfma_f32_16x16x32_f16 (the
//           rate of the bf16 form): 11 + 11 mantissa bits per operand instead of 8 + 8, i.e. ~2^-20 per product (both parts truncated) while |x| stays
//           inside fp16's range (values beyond 65504 saturate, parts below 6e-8 are lost: an ABSOLUTE floor ~2^-24, harmless
//           next to O(1) operands).  Used for the local branch (GIN layers, local head, local edge_attr rows), whose outputs
//           carry the 64 -> 1 cancellation of the head and were the thin spot of the split-bf16 parity.
enum { AG_F32 = 0, AG_BF3 = 1, AG_H3 = 2 };

template <int MODE> struct AgIn;
template <> struct AgIn<AG_F32> { f32x4 v[2]; };
template <> struct AgIn<AG_BF3> { bf16x8 hi, lo; };
template <> struct AgIn<AG_H3> { f16x8 hi, lo; };
// the 16-bit MFMA of a split mode
__device__ __forceinline__ f32x4 ag_mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 ag_mfma16(const f16x8& a, const f16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void ag_cvt(const f32x4& a0, const f32x4& a1, AgIn<AG_F32>& o) { o.v[0] = a0; o.v[1] = a1; }
__device__ __forceinline__ void ag_cvt(const f32x4& a0, const f32x4& a1, AgIn<AG_BF3>& o) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = (j < 4) ? a0[j & 3] : a1[j & 3];
    const __bf16 hb = (__bf16)v;
    o.hi[j] = hb;
    o.lo[j] = (__bf16)(v - (float)hb);
  }
}
// split fp16: one v_cvt_pkrtz_f16_f32 per pair and part (round toward zero: hi never exceeds |x|, so lo keeps the sign and the
// remainder; a value beyond fp16's range saturates at 65504 instead of becoming infinite)
__device__ __forceinline__ void ag_cvt_pair(AgIn<AG_H3>& o, int j, float v0, float v1) {
  const auto hp = __builtin_amdgcn_cvt_pkrtz(v0, v1);
  // v - float(hi) as fma(v, one, -float(hi)) with a 1.0 the optimiser cannot see through: ONE v_fma_mix_f32 that reads the
  // fp16 half directly instead of v_cvt_f32_f16 + v_sub_f32 (the same single rounding: v * 1 is exact)
#ifdef AG_NO_FMA_MIX        // (A/B builds: the two-instruction form)
  const auto lp = __builtin_amdgcn_cvt_pkrtz(v0 - (float)hp[0], v1 - (float)hp[1]);
#else
  float one = 1.0f;
  asm("" : "+s"(one));
  const auto lp = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf(v0, one, -(float)hp[0]), __builtin_fmaf(v1, one, -(float)hp[1]));
#endif
  o.hi[j] = (_Float16)hp[0];
  o.hi[j + 1] = (_Float16)hp[1];
  o.lo[j] = (_Float16)lp[0];
  o.lo[j + 1] = (_Float16)lp[1];
}
__device__ __forceinline__ void ag_cvt(const f32x4& a0, const f32x4& a1, AgIn<AG_H3>& o) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) ag_cvt_pair(o, j, (j < 4) ? a0[j & 3] : a1[j & 3], (j < 4) ? a0[(j & 3) + 1] : a1[(j & 3) + 1]);
}
// Elements j, j + 1 (j even) of a k-tile from two fp32 values: the piecewise form of ag_cvt, for callers that
// spread the conversion between other work.
__device__ __forceinline__ void ag_cvt_pair(AgIn<AG_F32>& o, int j, float v0, float v1) {
  o.v[j >> 2][j & 3] = v0;
  o.v[j >> 2][(j & 3) + 1] = v1;
}
__device__ __forceinline__ void ag_cvt_pair(AgIn<AG_BF3>& o, int j, float v0, float v1) {
#ifdef AG_CVT_PAIR_SCALAR
  const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
  o.hi[j] = h0;
  o.hi[j + 1] = h1;
  o.lo[j] = (__bf16)(v0 - (float)h0);
  o.lo[j + 1] = (__bf16)(v1 - (float)h1);
#else
  // one v_cvt_pk_bf16_f32 per pair for hi and one for lo; the rounded values come back as floats by a shift / a mask of the
  // packed word (six instructions per pair: element-wise conversion compiled to eight)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 hp = __builtin_convertvector(f32x2{v0, v1}, bf16x2);
  const uint32_t hw = __builtin_bit_cast(uint32_t, hp);
  const float h0 = __uint_as_float(hw << 16), h1 = __uint_as_float(hw & 0xFFFF0000u);
  const bf16x2 lp = __builtin_convertvector(f32x2{v0 - h0, v1 - h1}, bf16x2);
  o.hi[j] = hp[0];
  o.hi[j + 1] = hp[1];
  o.lo[j] = lp[0];
  o.lo[j + 1] = lp[1];
#endif
}
// The mixed second operand of agdiff_params_t.poly_plan 1 (ag_poly_features): hi = the split's hi of (v0, v1) as above; lo =
// the hi part of (m0, m1) - own * (hi as float): with own = 1 and m = v the usual lo part, with own = 0 the hi part of m.
__device__ __forceinline__ void ag_cvt_pair_mixed(AgIn<AG_H3>& o, int j, float v0, float v1, float m0, float m1, float own) {
  const auto hp = __builtin_amdgcn_cvt_pkrtz(v0, v1);
  const auto lp = __builtin_amdgcn_cvt_pkrtz(fmaf(-own, (float)hp[0], m0), fmaf(-own, (float)hp[1], m1));
  o.hi[j] = (_Float16)hp[0];
  o.hi[j + 1] = (_Float16)hp[1];
  o.lo[j] = (_Float16)lp[0];
  o.lo[j + 1] = (_Float16)lp[1];
}
__device__ __forceinline__ void ag_cvt_pair_mixed(AgIn<AG_BF3>& o, int j, float v0, float v1, float m0, float m1, float own) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 hp = __builtin_convertvector(f32x2{v0, v1}, bf16x2);
  const uint32_t hw = __builtin_bit_cast(uint32_t, hp);
  const float h0 = __uint_as_float(hw << 16), h1 = __uint_as_float(hw & 0xFFFF0000u);
  const bf16x2 lp = __builtin_convertvector(f32x2{fmaf(-own, h0, m0), fmaf(-own, h1, m1)}, bf16x2);
  o.hi[j] = hp[0];
  o.hi[j + 1] = hp[1];
  o.lo[j] = lp[0];
  o.lo[j + 1] = lp[1];
}
// NK k-tiles from accumulator tiles A0, A0+1, ...
template <int MODE, int NK, int A0, int NA, int NO>
__device__ __forceinline__ void ag_cvt_tiles(const f32x4 (&a)[NA], AgIn<MODE> (&o)[NO]) {
  static_assert(A0 + 2 * NK <= NA && NK <= NO, "tile range");
#pragma unroll
  for (int t = 0; t < NK; ++t) ag_cvt(a[A0 + 2 * t], a[A0 + 2 * t + 1], o[t]);
}

// Split-fp16 operands saturate at 65504 (ag_cvt_pair).  Activations that depend on the state and that no workspace tensor shows (hidden
// layers of the node stage, the GIN layers, the heads) are tracked where they are converted: the largest magnitude a lane has seen, and
// one flag per node for the launch's caller (agdiff_ws_t.range_rows) when it reaches the edge of the range.  Nothing in the other modes.
template <int MODE, int N, int NA>
__device__ __forceinline__ float ag_absmax(const f32x4 (&a)[NA], float mx) {
  static_assert(N <= NA, "tile range");
  if constexpr (MODE == AG_H3) {
#pragma unroll
    for (int t = 0; t < N; ++t) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(a[t][0]), fabsf(a[t][1]))), fmaxf(fabsf(a[t][2]), fabsf(a[t][3])));
  }
  return mx;
}
// (`limit`: 65000 for a value that becomes an operand itself; 255 for a node state the pair heads multiply with another one, 60000
// for the aggregates and CFConv inputs -- the figures of the host's tensor watch, agdiff_amd/epsnet.py RANGE_LIMITS: checked where
// a kernel stores such a row, the flag is sticky between two polls where the tensor only shows its last state)
template <int MODE>
__device__ __forceinline__ void ag_report_range(float mx, int32_t* rows, int64_t row, bool live, float limit = 65000.0f) {
  if constexpr (MODE == AG_H3) {
    if (rows && live && mx >= limit) rows[row] = 1;
  }
}

// one weight block (2 x 16 B per lane, already in registers) applied to one k-tile
template <int MODE, bool FLIP>
__device__ __forceinline__ void ag_block_mma(f32x4& o, const AgIn<MODE>& x, const u32x4 (&w)[2]) {
  if constexpr (MODE == AG_F32) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const f32x4 wf = __builtin_bit_cast(f32x4, w[u]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        o = FLIP ? __builtin_amdgcn_mfma_f32_16x16x4f32(x.v[u][r], wf[r], o, 0, 0, 0)
                 : __builtin_amdgcn_mfma_f32_16x16x4f32(wf[r], x.v[u][r], o, 0, 0, 0);
    }
  } else {
    using V = decltype(x.hi);
    const V whi = __builtin_bit_cast(V, w[0]), wlo = __builtin_bit_cast(V, w[1]);
    if (FLIP) {
      o = ag_mfma16(x.hi, whi, o);
      o = ag_mfma16(x.lo, whi, o);
      o = ag_mfma16(x.hi, wlo, o);
    } else {
      o = ag_mfma16(whi, x.hi, o);
      o = ag_mfma16(whi, x.lo, o);
      o = ag_mfma16(wlo, x.hi, o);
    }
  }
}

// The same product one MFMA at a time (part p of AgParts<MODE>::n), for callers that interleave the parts of several
// independent accumulators: back-to-back MFMAs on ONE accumulator wait for each other's result.
template <int MODE> struct AgParts { static constexpr int n = (MODE == AG_F32) ? 8 : 3; };
template <int MODE, bool FLIP>
__device__ __forceinline__ void ag_block_mma_part(f32x4& o, const AgIn<MODE>& x, const u32x4 (&w)[2], int part) {
  if constexpr (MODE == AG_F32) {
    const int u = part >> 2, r = part & 3;
    const f32x4 wf = __builtin_bit_cast(f32x4, w[u]);
    o = FLIP ? __builtin_amdgcn_mfma_f32_16x16x4f32(x.v[u][r], wf[r], o, 0, 0, 0)
             : __builtin_amdgcn_mfma_f32_16x16x4f32(wf[r], x.v[u][r], o, 0, 0, 0);
  } else {
    using V = decltype(x.hi);
    const V wv = __builtin_bit_cast(V, w[part == 2 ? 1 : 0]);
    const V xv = (part == 1) ? x.lo : x.hi;
    o = FLIP ? ag_mfma16(xv, wv, o) : ag_mfma16(wv, xv, o);
  }
}

// agdiff_params_t.poly_plan 1 at poly_kt 1: the second of two passes -- the operand's `lo` member holds [lo of terms 0..15 |
// hi of terms 0..15] (ag_poly_features<.., true>) and unit 1 of the block [hi | lo] of their coefficients
template <int MODE, bool FLIP>
__device__ __forceinline__ void ag_block_mma_mixed(f32x4& o, const AgIn<MODE>& x, const u32x4 (&w)[2]) {
  static_assert(MODE != AG_F32, "split modes only");
  using V = decltype(x.hi);
  const V wv = __builtin_bit_cast(V, w[1]);
  o = FLIP ? ag_mfma16(x.lo, wv, o) : ag_mfma16(wv, x.lo, o);
}

// The first part of a product that starts from zero: the accumulator input is the literal 0 (an inline constant of the
// MFMA), not a register the compiler has to clear first.
template <int MODE, bool FLIP>
__device__ __forceinline__ f32x4 ag_block_mma_first(const AgIn<MODE>& x, const u32x4 (&w)[2]) {
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  if constexpr (MODE == AG_F32) {
    const f32x4 wf = __builtin_bit_cast(f32x4, w[0]);
    return FLIP ? __builtin_amdgcn_mfma_f32_16x16x4f32(x.v[0][0], wf[0], zero, 0, 0, 0)
                : __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0], x.v[0][0], zero, 0, 0, 0);
  } else {
    using V = decltype(x.hi);
    const V wv = __builtin_bit_cast(V, w[0]);
    return FLIP ? ag_mfma16(x.hi, wv, zero) : ag_mfma16(wv, x.hi, zero);
  }
}

// Weight blocks are consumed in storage order: output-tile-outer ("pk": [OT][KT]) or k-tile-outer
// ("pkk": [KT][OT]).  Global source: loads run PF blocks ahead of their MFMAs, with
// __builtin_amdgcn_sched_barrier(0) between blocks so that hipcc cannot hoist every load of a layer to its
// top (that costs hundreds of VGPRs).  LDS source: the compiler schedules the ds_reads itself.
//   o[O0 + ot] += W-block(ot, t) . x[X0 + t]
template <int MODE, bool FLIP, bool KOUTER, int KT, int OT, int X0, int O0, int PF, int NX, int NO>
__device__ __forceinline__ void ag_dense(const AgIn<MODE> (&x)[NX], f32x4 (&o)[NO], const void* wpk, int lane) {
  static_assert(X0 + KT <= NX && O0 + OT <= NO, "tile range");
  constexpr int S = OT * KT;
  constexpr int R = PF + 1;
  u32x4 w[R][2];
  const u32x4* wl = reinterpret_cast<const u32x4*>(wpk) + lane;
#pragma unroll
  for (int s = 0; s < (PF < S ? PF : S); ++s) {
    w[s % R][0] = wl[(s * 2) * 64];
    w[s % R][1] = wl[(s * 2 + 1) * 64];
  }
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int t = KOUTER ? s / OT : s % KT, ot = KOUTER ? s % OT : s / KT;
    if (s + PF < S) {
      w[(s + PF) % R][0] = wl[((s + PF) * 2) * 64];
      w[(s + PF) % R][1] = wl[((s + PF) * 2 + 1) * 64];
    }
    ag_block_mma<MODE, FLIP>(o[O0 + ot], x[X0 + t], w[s % R]);
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int MODE, bool FLIP, bool KOUTER, int KT, int OT, int X0, int O0, int NX, int NO>
__device__ __forceinline__ void ag_dense_lds(const AgIn<MODE> (&x)[NX], f32x4 (&o)[NO], const lds_u32x4* wl, int lane) {
  static_assert(X0 + KT <= NX && O0 + OT <= NO, "tile range");
#ifdef AG_DENSE_LDS_PLAIN
#pragma unroll
  for (int s = 0; s < OT * KT; ++s) {
    const int t = KOUTER ? s / OT : s % KT, ot = KOUTER ? s % OT : s / KT;
    u32x4 w[2];
    w[0] = wl[(s * 2) * 64 + lane];
    w[1] = wl[(s * 2 + 1) * 64 + lane];
    ag_block_mma<MODE, FLIP>(o[O0 + ot], x[X0 + t], w);
  }
#else
  // Output tiles in groups of G: the G blocks of a k-tile are read together and their MFMA passes interleaved over the G
  // accumulators (back-to-back MFMAs on one accumulator, each behind its own LDS read, made these layers a chain of
  // exposed latencies).  Per accumulator the order of the additions is unchanged (k-tile outer, pass inner).
  constexpr int G = (OT % 4 == 0) ? 4 : (OT % 2 == 0) ? 2 : 1;
#pragma unroll
  for (int og = 0; og < OT / G; ++og) {
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      u32x4 w[G][2];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int ot = og * G + g;
        const int s = KOUTER ? t * OT + ot : ot * KT + t;
#ifdef AG_DENSE_LDS_HALF      // (timing experiment: every second weight block read from LDS, the others reused -- wrong results)
        if (g & 1) {
          w[g][0] = w[g - 1][0];
          w[g][1] = w[g - 1][1];
          continue;
        }
#endif
        w[g][0] = wl[(s * 2) * 64 + lane];
        w[g][1] = wl[(s * 2 + 1) * 64 + lane];
      }
#pragma unroll
      for (int part = 0; part < AgParts<MODE>::n; ++part) {
#pragma unroll
        for (int g = 0; g < G; ++g) ag_block_mma_part<MODE, FLIP>(o[O0 + og * G + g], x[X0 + t], w[g], part);
      }
    }
  }
#endif
}
// Mixed source: unit 0 of every block (hi halves in AG_BF3, first k-half in AG_F32) from an LDS array that holds
// only those units (64 u32x4 per block), unit 1 streamed from the full packed matrix in global memory PF blocks
// ahead.  Halves the L2 traffic of a layer that does not fit in LDS next to the others.
template <int MODE, bool FLIP, bool KOUTER, int KT, int OT, int X0, int O0, int PF, int NX, int NO>
__device__ __forceinline__ void ag_dense_split(const AgIn<MODE> (&x)[NX], f32x4 (&o)[NO], const lds_u32x4* w0,
                                               const void* wpk, int lane, int lds_lane) {
  static_assert(X0 + KT <= NX && O0 + OT <= NO, "tile range");
  constexpr int S = OT * KT;
  constexpr int R = PF + 1;
  u32x4 w1[R];
  const u32x4* wl = reinterpret_cast<const u32x4*>(wpk) + 64 + lane;     // unit 1 of block 0
#pragma unroll
  for (int s = 0; s < (PF < S ? PF : S); ++s) w1[s % R] = wl[(s * 2) * 64];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int t = KOUTER ? s / OT : s % KT, ot = KOUTER ? s % OT : s / KT;
    if (s + PF < S) w1[(s + PF) % R] = wl[((s + PF) * 2) * 64];
    u32x4 w[2];
    w[0] = w0[s * 64 + lds_lane];
    w[1] = w1[s % R];
    ag_block_mma<MODE, FLIP>(o[O0 + ot], x[X0 + t], w);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// default prefetch depth for weights streamed from L2 (blocks of 48 / 256 MFMA cycles)
template <int MODE> struct AgPF { static constexpr int v = (MODE == AG_F32) ? 2 : 4; };
// weight block b of a packed matrix (both modes: 2 KiB = 512 floats per block)
__device__ __forceinline__ const void* ag_wblock(const float* wpk, int b) { return wpk + (size_t)b * 512; }


// ---------------------------------------------------------------------------------- edge-attr storage
// e_attr (128 features per edge) is stored in the operand form of the mode that consumes it: per k-tile t two
// 16-byte units per lane, loaded straight into MFMA operands, 512 B per edge.
// 16-byte unit index of (tile, k-tile t, unit u, edge column col = lane & 15, quarter q = lane >> 4):
// ((tile * 4 + t) * 2 + u) * 64 + col * 4 + q.  A tile-wise producer / consumer moves 1 KiB contiguous per wave
// instruction (all 64 slots of one (tile, t, u)); the four quarters of ONE edge are adjacent, so a writer that places
// single edges (canonical edge -> its own and its mirror's slot) writes whole 64-byte sectors.
__device__ __forceinline__ int64_t ag_attr_unit(int64_t tile, int t, int u, int lane) {
  return ((tile * 4 + t) * 2 + u) * 64 + (lane & 15) * 4 + (lane >> 4);
}
__device__ __forceinline__ void ag_store_attr(const AgIn<AG_F32>& x, float* frag, int64_t tile, int t, int lane) {
  u32x4* p = reinterpret_cast<u32x4*>(frag);
  p[ag_attr_unit(tile, t, 0, lane)] = __builtin_bit_cast(u32x4, x.v[0]);
  p[ag_attr_unit(tile, t, 1, lane)] = __builtin_bit_cast(u32x4, x.v[1]);
}
__device__ __forceinline__ void ag_store_attr(const AgIn<AG_BF3>& x, float* frag, int64_t tile, int t, int lane) {
  u32x4* p = reinterpret_cast<u32x4*>(frag);
  p[ag_attr_unit(tile, t, 0, lane)] = __builtin_bit_cast(u32x4, x.hi);
  p[ag_attr_unit(tile, t, 1, lane)] = __builtin_bit_cast(u32x4, x.lo);
}
__device__ __forceinline__ void ag_store_attr(const AgIn<AG_H3>& x, float* frag, int64_t tile, int t, int lane) {
  u32x4* p = reinterpret_cast<u32x4*>(frag);
  p[ag_attr_unit(tile, t, 0, lane)] = __builtin_bit_cast(u32x4, x.hi);
  p[ag_attr_unit(tile, t, 1, lane)] = __builtin_bit_cast(u32x4, x.lo);
}
__device__ __forceinline__ void ag_load_attr(AgIn<AG_H3>& x, const float* frag, int64_t tile, int t, int lane) {
  const u32x4* p = reinterpret_cast<const u32x4*>(frag);
  x.hi = __builtin_bit_cast(f16x8, p[ag_attr_unit(tile, t, 0, lane)]);
  x.lo = __builtin_bit_cast(f16x8, p[ag_attr_unit(tile, t, 1, lane)]);
}
__device__ __forceinline__ void ag_load_attr(AgIn<AG_F32>& x, const float* frag, int64_t tile, int t, int lane) {
  const u32x4* p = reinterpret_cast<const u32x4*>(frag);
  x.v[0] = __builtin_bit_cast(f32x4, p[ag_attr_unit(tile, t, 0, lane)]);
  x.v[1] = __builtin_bit_cast(f32x4, p[ag_attr_unit(tile, t, 1, lane)]);
}
__device__ __forceinline__ void ag_load_attr(AgIn<AG_BF3>& x, const float* frag, int64_t tile, int t, int lane) {
  const u32x4* p = reinterpret_cast<const u32x4*>(frag);
  x.hi = __builtin_bit_cast(bf16x8, p[ag_attr_unit(tile, t, 0, lane)]);
  x.lo = __builtin_bit_cast(bf16x8, p[ag_attr_unit(tile, t, 1, lane)]);
}
// DistanceWeightingNetwork (schnet.py:83-100) times the cutoff envelope (schnet.py:140-146).
// cos via v_cos_f32 on the half angle (0.5 (cos x + 1) = cos^2(x/2), argument <= 1/4 revolution inside the
// cutoff), sigmoid / gaussian via v_exp_f32: absolute error ~1e-6 on a factor in [0, 1].
// DistanceWeightingNetwork before its sigmoid is piecewise linear in d (agdiff_conv_params_t.dist_seg): binary search for
// the segment among the 32 sorted kinks (padded with +inf), then one FMA -- instead of 32 hidden units per edge and conv.
// SEG: a pointer to global memory or to LDS.
template <typename SEG>
__device__ __forceinline__ float cf_dist_weight(SEG seg, float d) {
  int s = 0;
#pragma unroll
  for (int step = 16; step >= 1; step >>= 1) s += (seg[s + step - 1] <= d) ? step : 0;
  s += (s == 31 && seg[31] <= d) ? 1 : 0;
  return ag_sigmoid(fmaf(seg[32 + s], d, seg[65 + s]));
}
// the cutoff envelope C(d) (schnet.py:140-146), the same for every CFConv
__device__ __forceinline__ float cf_envelope(float d, float cutoff, int smooth) {
  float C;
  if (smooth) {
    const float c = __builtin_amdgcn_cosf(d * (0.25f / cutoff));     // cos(pi d / (2 rc)), input in revolutions
    C = c * c;
  } else {
    const float t = d - cutoff;
    C = ag_exp2(-(t * t) / (2.0f * cutoff * cutoff) * 1.44269504088896340736f);
  }
  return (d <= cutoff && d >= 0.0f) ? C : 0.0f;
}

// ---------------------------------------------------------------------------------- filter polynomials
// Operand elements of the lane's edge for the NKT k-tiles of a d-polynomial (include/agdiff_hip.h: agdiff_params_t.poly_kt;
// host mirror: agdiff_amd/packing.py poly_features): element j of quarter q in k-tile t is
//   phi[8 (4 t + q) + j](x) = T_{8 (4 t + q)}(x) T_j(x),   x = 2 d / cutoff - 1 in [-1, 1].
// T_0..T_8 by the three-term recurrence, T_16 .. T_56 from the product rule 2 T_a T_b = T_{a+b} + T_{|a-b|}, T_64 .. T_120 (k-tiles
// 2, 3) by the recurrence in steps of eight: ~25 VALU
// instructions + the operand split per k-tile, instead of a 128-wide MLP chain per edge.
// `gmask` (1 or 0) multiplies every element: edges outside the group being evaluated contribute nothing.
// MIXED (agdiff_params_t.poly_plan 1 at one k-tile, split modes): o[0].lo is the operand of the SECOND of two passes instead
// of the lo parts: lanes of quarters 0, 1 keep the lo parts of their terms f = 8 q + j < 16, quarters 2, 3 hold the hi parts
// of terms 8 (q - 2) + j -- the K = 32 slots of one instruction then carry lo(phi_f) hi(c_f) + hi(phi_f) lo(c_f), f < 16.
template <int MODE, int NKT, bool MIXED = false>
__device__ __forceinline__ void ag_poly_features(float d, float two_over_rc, int q, AgIn<MODE> (&o)[NKT], float gmask = 1.0f) {
  static_assert(NKT >= 1 && NKT <= AGDIFF_POLY_MAX_KT, "poly_kt");
  static_assert(!MIXED || (NKT == 1 && MODE != AG_F32), "mixed operand: one k-tile, split modes");
  const float x = fminf(fmaxf(fmaf(d, two_over_rc, -1.0f), -1.0f), 1.0f);
  const float x2 = x + x;
  float T[9];
  T[0] = 1.0f;
  T[1] = x;
#pragma unroll
  for (int n = 2; n <= 8; ++n) T[n] = fmaf(x2, T[n - 1], -T[n - 2]);
  // (every candidate of the per-quarter selects below is computed first and made opaque: left to itself the optimiser sinks the
  // evaluation of T16 .. T56 into the arms of `q == k ? ... : ...` and emits divergent branches for them)
  float t8 = T[8];
  float g2 = fmaf(t8 + t8, t8, -1.0f);             // T16
  float g3 = fmaf(g2 + g2, t8, -t8);               // T24
  asm volatile("" : "+v"(t8), "+v"(g2), "+v"(g3));
  float G[NKT];
  {
    float sel = 1.0f;
    sel = (q == 1) ? t8 : sel;
    sel = (q == 2) ? g2 : sel;
    sel = (q == 3) ? g3 : sel;
    asm volatile("" : "+v"(sel));
    G[0] = gmask * sel;
  }
  if constexpr (NKT >= 2) {
    float g4 = fmaf(g2 + g2, g2, -1.0f);           // T32
    float g5 = fmaf(g4 + g4, t8, -g3);             // T40
    float g6 = fmaf(g3 + g3, g3, -1.0f);           // T48
    float g7 = fmaf(g6 + g6, t8, -g5);             // T56
    asm volatile("" : "+v"(g4), "+v"(g5), "+v"(g6), "+v"(g7));
    float sel = g4;
    sel = (q == 1) ? g5 : sel;
    sel = (q == 2) ? g6 : sel;
    sel = (q == 3) ? g7 : sel;
    asm volatile("" : "+v"(sel));
    G[1] = gmask * sel;
    if constexpr (NKT >= 3) {                      // T64 .. T120: T_{8 (k + 1)} = 2 T_8 T_{8 k} - T_{8 (k - 1)}
      const float t82 = t8 + t8;
      float lo2 = g6, lo1 = g7;
#pragma unroll
      for (int t = 2; t < NKT; ++t) {
        float h0 = fmaf(t82, lo1, -lo2);
        float h1 = fmaf(t82, h0, -lo1);
        float h2 = fmaf(t82, h1, -h0);
        float h3 = fmaf(t82, h2, -h1);
        asm volatile("" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3));
        float s_ = h0;
        s_ = (q == 1) ? h1 : s_;
        s_ = (q == 2) ? h2 : s_;
        s_ = (q == 3) ? h3 : s_;
        asm volatile("" : "+v"(s_));
        G[t] = gmask * s_;
        lo2 = h2, lo1 = h3;
      }
    }
  }
  if constexpr (MIXED) {
    float Gsel = (q & 1) ? t8 : 1.0f;                        // T_{8 (q & 1)}: the lane's own factor in quarters 0, 1
    asm volatile("" : "+v"(Gsel));
    const float Gm = gmask * Gsel;
    const float own = (q < 2) ? 1.0f : 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) ag_cvt_pair_mixed(o[0], j, G[0] * T[j], G[0] * T[j + 1], Gm * T[j], Gm * T[j + 1], own);
  } else {
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
      for (int j = 0; j < 8; j += 2) ag_cvt_pair(o[t], j, G[t] * T[j], G[t] * T[j + 1]);
    }
  }
}

// The same for the TWO scaled feature sets a CFConv tile needs (conv1's rows scaled by s1, conv2's by s2) at one k-tile under
// agdiff_params_t.poly_plan 1 (MIXED), sharing everything that does not depend on the scale: the recurrence, the lane's two
// Chebyshev factors and the selects.  The selects are written on values that are already computed (and opaque to the optimiser):
// left to itself it sinks T16 / T24 into the arms of `q == k ? ... : ...` and emits divergent branches -- ~45 of the ~125 VALU
// instructions the two calls of ag_poly_features cost per tile.  Bit-identical to two calls of ag_poly_features<MODE, 1, true>.
template <int MODE>
__device__ __forceinline__ void ag_poly_features2_mixed(float d, float two_over_rc, int q, AgIn<MODE>& o1, float s1, AgIn<MODE>& o2, float s2) {
  static_assert(MODE != AG_F32, "mixed operand: split modes");
  const float x = fminf(fmaxf(fmaf(d, two_over_rc, -1.0f), -1.0f), 1.0f);
  const float x2 = x + x;
  float T[9];
  T[0] = 1.0f;
  T[1] = x;
#pragma unroll
  for (int n = 2; n <= 8; ++n) T[n] = fmaf(x2, T[n - 1], -T[n - 2]);
  float t8 = T[8];
  float g2 = fmaf(t8 + t8, t8, -1.0f);             // T16
  float g3 = fmaf(g2 + g2, t8, -t8);               // T24
  asm volatile("" : "+v"(t8), "+v"(g2), "+v"(g3));
  float G = 1.0f;
  G = (q == 1) ? t8 : G;
  G = (q == 2) ? g2 : G;
  G = (q == 3) ? g3 : G;
  float Gm = (q & 1) ? t8 : 1.0f;                  // T_{8 (q & 1)}: the lane's own factor in quarters 0, 1
  asm volatile("" : "+v"(G), "+v"(Gm));
  const float own = (q < 2) ? 1.0f : 0.0f;
  const float a1 = G * s1, b1 = Gm * s1, a2 = G * s2, b2 = Gm * s2;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    ag_cvt_pair_mixed(o1, j, a1 * T[j], a1 * T[j + 1], b1 * T[j], b1 * T[j + 1], own);
    ag_cvt_pair_mixed(o2, j, a2 * T[j], a2 * T[j + 1], b2 * T[j], b2 * T[j + 1], own);
  }
}

// global -> LDS copy of n 16-byte units by the whole workgroup (weights that stay resident for a launch or a phase).
// The loads of U units per thread are issued back to back and stored afterwards: the plain loop `dst[i] = src[i]`
// compiles to load, s_waitcnt vmcnt(0), ds_write per iteration -- one exposed L2 round trip per 16 bytes and thread,
// 10..13 in a row for a 160-KiB fill (found in the ISA of every kernel that stages weights; the node stage and the GIN
// layer do it three times / once per 16-node tile).
template <int U = 8, typename F>
__device__ __forceinline__ void ag_copy_lds_map(lds_u32x4* dst, const u32x4* __restrict__ src, int n, F src_index) {
  const int stride = (int)blockDim.x;
  for (int base = (int)threadIdx.x; base < n; base += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int i = base + u * stride;
      i = i < n ? i : n - 1;                   // (clamped: every load is issued, stores are conditional)
      v[u] = src[src_index(i)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * stride;
      if (i < n) dst[i] = v[u];
    }
  }
}
template <int U = 8>
__device__ __forceinline__ void ag_copy_lds(lds_u32x4* dst, const u32x4* __restrict__ src, int n) {
  ag_copy_lds_map<U>(dst, src, n, [](int i) { return i; });
}

// Kernels that declare more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize, and HIP keeps
// function attributes PER DEVICE: `done` has one bit per device id, so a process that drives several GPUs sets the
// attribute on each of them (setting it twice from two threads is harmless, hence no lock).
#include <atomic>
template <typename... K>
static inline bool ag_allow_big_lds(std::atomic<uint64_t>& done, size_t smem, K... kernels) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return true;
  const bool ok = ((hipFuncSetAttribute((const void*)kernels, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess) && ...);
  if (ok) done.fetch_or(bit, std::memory_order_release);
  return ok;
}
// Host side: the variant a launcher chose, for agdiff_ws_t.variant_log (a host word the tests read), and a tuning field of
// agdiff_params_t with its library default (0 = default).
static inline void ag_log_variant(const agdiff_ws_t* ws, int64_t bits) {
  if (ws && ws->variant_log) *ws->variant_log |= bits;
}
static inline int64_t ag_tune(int64_t v, int64_t dflt) { return v != 0 ? v : dflt; }
// agdiff_sampler_front with the update's step read from a device table (front.hip; used by agdiff_step_graph_capture)
int ag_sampler_front_table(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, const agdiff_step_args_t* s,
                           const agdiff_step_args_t* step_table, const int32_t* step_index, int32_t mode, float cutoff, void* stream);
// Host-side launch check
#define AG_CHECK_LAUNCH()                                         \
  do {                                                            \
    if (hipGetLastError() != hipSuccess) return AGDIFF_ERR_LAUNCH; \
  } while (0)
