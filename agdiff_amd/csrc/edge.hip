// Per-edge kernels: MLPEdgeEncoder, fused CFConv (filter MLP + message + destination-segmented
// reduction), pair-feature heads.  One wave = one tile of 32 edges; activations stay in the MFMA
// accumulator layout between layers (common.hpp).
#include "common.hpp"
#include <type_traits>

namespace {

// ------------------------------------------------------------------------------ edge encoder
struct EncArgs {
  const float* fe_w;
  const float* fe_b;
  const float* t1;
  const float* w1_pk;
  const float* t3;
  const float* w23_pk;
  const float* w4_pk;
  const float* b4;
  const int32_t* n_dev;
  const float* e_len;
  const int32_t* e_type;
  float* out_frag;
  int64_t max_tiles;
};

// encoder/edge.py:84-103.  x0 = gelu(w*d+b); h1 = gelu(W1a x0 + T1[type]); h2 = gelu(W23 h1 + T3[type]);
// a = W4 h2 + b4.  (T1/T3: per-edge-type tables holding the bond_emb halves of the two 256->128
// layers; W23 = comb.0[:, :128] @ efm.2; the trailing attention factor is exactly 1.)
__global__ void __launch_bounds__(AG_WG, 2) k_edge_encoder(EncArgs a) {
  const int lane = ag_lane(), h = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + ag_wave_in_wg();
  const int E = *a.n_dev;
  if (tile >= a.max_tiles || tile * 32 >= E) return;
  const int64_t e = tile * 32 + (lane & 31);
  const bool valid = e < E;
  const float d = valid ? a.e_len[e] : 0.0f;
  const int ty = valid ? a.e_type[e] : 0;

  f32x16 x[4], y[4];
  ag_init_vec<4>(x, a.fe_w, h);
  ag_init_vec<4>(y, a.fe_b, h);
  AG_FOR_TILE(x, 4, ag_gelu(fmaf(v, d, y[_t][_r])));
  ag_init_vec<4>(y, a.t1 + (size_t)ty * 128, h);
  ag_dense_std<4, 4, 0, 0, 4>(x, y, a.w1_pk, lane);
  AG_FOR_TILE(y, 4, ag_gelu(v));
  ag_init_vec<4>(x, a.t3 + (size_t)ty * 128, h);
  ag_dense_std<4, 4, 0, 0, 4>(y, x, a.w23_pk, lane);
  AG_FOR_TILE(x, 4, ag_gelu(v));
  ag_init_vec<4>(y, a.b4, h);
  ag_dense_std<4, 4, 0, 0, 4>(x, y, a.w4_pk, lane);
  ag_store_frag<4>(y, a.out_frag, tile, lane);
}

// ------------------------------------------------------------------------------ fused CFConv
struct ConvArgs {
  agdiff_conv_params_t cp;
  const int32_t* n_dev;
  const int32_t* in_ptr;
  const int32_t* e_src;
  const int32_t* e_dst;
  const float* e_len;
  const float* e_attr;
  const float* xs;       // [N][192]
  float* agg;            // [N][192]
  float* agg_first;      // [chunks][192]
  int64_t max_chunks;
  float cutoff;
  int32_t smooth;
};

// DistanceWeightingNetwork (schnet.py:83-100) times the cutoff envelope (schnet.py:140-146).
__device__ __forceinline__ float cf_edge_scale(const float* __restrict__ dw, float d, float cutoff, int smooth) {
  float acc = dw[96];
#pragma unroll 8
  for (int k = 0; k < 32; ++k) acc = fmaf(dw[64 + k], ag_relu(fmaf(dw[k], d, dw[32 + k])), acc);
  const float lw = 1.0f / (1.0f + expf(-acc));
  float C;
  if (smooth) {
    C = 0.5f * (cosf(d * 3.14159265358979323846f / cutoff) + 1.0f);
    C = C * (d <= cutoff ? 1.0f : 0.0f);
  } else {
    const float t = d - cutoff;
    C = expf(-(t * t) / (2.0f * cutoff * cutoff));
  }
  C = C * (d <= cutoff ? 1.0f : 0.0f) * (d >= 0.0f ? 1.0f : 0.0f);
  return lw * C;
}

// encoder/schnet.py:136-162 for conv1 (F=128) and conv2 (F=64) of one InteractionBlock:
//   W_e = nn(edge_attr_e) * (lw(d_e) * C(d_e));  agg[dst] += x[src] * W_e   (aggr='add')
// Each wave walks AGDIFF_CHUNK_TILES consecutive destination-sorted tiles and keeps the running sum
// of the current target in registers; a target whose list started in an earlier chunk is written
// to agg_first[chunk] and added by the node stage (fixed order -> bitwise reproducible).
__global__ void __launch_bounds__(AG_WG, 2) k_cfconv_fused(ConvArgs a) {
  const int lane0 = ag_lane();
  const int64_t chunk = (int64_t)blockIdx.x * 4 + ag_wave_in_wg();
  const int E = *a.n_dev;
  const int64_t e_begin = chunk * (32 * AGDIFF_CHUNK_TILES);
  if (chunk >= a.max_chunks || e_begin >= E) return;
  const int h0 = lane0 >> 5, col0 = lane0 & 31;

  // running sums (6 channel tiles x 64 lanes) of target run_t, whose list is still open; kept in
  // LDS so that the channel-tile loops need not be unrolled
  __shared__ float carry_s[4][6 * 64];
  float* carry = carry_s[ag_wave_in_wg()];
  int run_t = -1;

  auto dest = [&](int t) -> float* {
    const int lo = a.in_ptr[t];
    return (lo >= e_begin) ? (a.agg + (size_t)t * 192) : (a.agg_first + (size_t)chunk * 192);
  };

  for (int tt = 0; tt < AGDIFF_CHUNK_TILES; ++tt) {
    const int64_t tile = chunk * AGDIFF_CHUNK_TILES + tt;
    const int64_t tbase = tile * 32;
    if (tbase >= E) break;
    // opaque copy of the lane id: keeps hipcc from hoisting every lane-derived weight / table address
    // out of the tile loop (they would stay live across the whole body and spill)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int h = lane >> 5, col = lane & 31;
    const int64_t e = tbase + col;
    const bool valid = e < E;
    const float d = valid ? a.e_len[e] : 0.0f;
    const int my_src = valid ? a.e_src[e] : 0;
    const float s1 = valid ? cf_edge_scale(a.cp.dist_w, d, a.cutoff, a.smooth) : 0.0f;
    const float s2 = valid ? cf_edge_scale(a.cp.dist_w + 97, d, a.cutoff, a.smooth) : 0.0f;

    const int64_t last = (tbase + 31 < E) ? tbase + 31 : (int64_t)E - 1;
    const int t0 = __builtin_amdgcn_readfirstlane(a.e_dst[tbase]);
    const int t1 = __builtin_amdgcn_readfirstlane(a.e_dst[last]);
    if (run_t >= 0 && run_t != t0) {   // previous tile ended exactly on a list boundary
      float* dp = dest(run_t);
      if (h == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) dp[32 * i + col] = carry[i * 64 + lane];
      }
      run_t = -1;
    }
    const bool cont = (run_t == t0);

    f32x16 hid[6];
    {
      // first filter layer, k-tile outer: only two 32-feature slices of edge_attr are live at a time
      f32x16 ea[2];
      ag_init_vec<6>(hid, a.cp.filt_b1, h);
      ag_load_frag_tile<0>(ea, a.e_attr, tile, 0, lane);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t + 1 < 4) {
          if (t & 1) ag_load_frag_tile<0>(ea, a.e_attr, tile, t + 1, lane);
          else ag_load_frag_tile<1>(ea, a.e_attr, tile, t + 1, lane);
        }
        if (t & 1) ag_dense_std_k<1, 6, 1, 0>(ea, hid, a.cp.filt_w1_pk + (size_t)t * 6 * 1024, lane);
        else ag_dense_std_k<1, 6, 0, 0>(ea, hid, a.cp.filt_w1_pk + (size_t)t * 6 * 1024, lane);
      }
    }
    // ssp, then fold the per-edge scale lw(d)*C(d) (a per-lane scalar here) into the hidden layer:
    // (s.H)^T W2 + s.b2 == s.(H^T W2 + b2)
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const float beta = (t < 4) ? a.cp.ssp_beta1 : a.cp.ssp_beta2;
      const float sc = (t < 4) ? s1 : s2;
#pragma unroll
      for (int r = 0; r < 16; ++r) hid[t][r] = ag_ssp(beta, hid[t][r]) * sc;
    }
    // gather row of every edge slot my half owns: slot (r,h) lives in lane ag_row(r,h)
    uint32_t xoff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) xoff[r] = (uint32_t)__shfl(my_src, ag_row(r, h)) * 192u + (uint32_t)col;

    // second filter layer per 32-channel tile, flipped (rows = edges, lanes = channels), then
    // message and destination-segmented reduction of that channel tile
    auto channel_tile = [&](f32x16 (&z)[1], int nt) {
      const float* xb = a.xs + 32 * nt;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[0][r] *= xb[xoff[r]];
      float newcarry = 0.0f;
      for (int t = t0; t <= t1; ++t) {
        const int lo = __builtin_amdgcn_readfirstlane(a.in_ptr[t]);
        const int hi = __builtin_amdgcn_readfirstlane(a.in_ptr[t + 1]);
        float p = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t er = tbase + ag_row(r, h);
          p += ((er >= lo) && (er < hi)) ? z[0][r] : 0.0f;
        }
        p += __shfl_xor(p, 32);
        if (t == t0 && cont) p = carry[nt * 64 + lane] + p;
        if (t < t1) {
          float* dp = dest(t);
          if (h == 0) dp[32 * nt + col] = p;
        } else {
          newcarry = p;
        }
      }
      carry[nt * 64 + lane] = newcarry;
    };
    // bias enters as one extra k-step: A = s_e on k-slot 0 (lane half 0), B = b2 on k-slot 0
#pragma unroll 1
    for (int nt = 0; nt < 4; ++nt) {
      f32x16 z[1];
      const float bb = (h == 0) ? a.cp.filt_b2[32 * nt + col] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[0][r] = 0.0f;
      z[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((h == 0) ? s1 : 0.0f, bb, z[0], 0, 0, 0);
      ag_dense_flip<4, 1, 0, 0>(hid, z, a.cp.filt_w2a_pk + (size_t)nt * 4 * 1024, lane);
      channel_tile(z, nt);
    }
#pragma unroll 1
    for (int nt = 4; nt < 6; ++nt) {
      f32x16 z[1];
      const float bb = (h == 0) ? a.cp.filt_b2[32 * nt + col] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[0][r] = 0.0f;
      z[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((h == 0) ? s2 : 0.0f, bb, z[0], 0, 0, 0);
      ag_dense_flip<2, 1, 4, 0>(hid, z, a.cp.filt_w2b_pk + (size_t)(nt - 4) * 2 * 1024, lane);
      channel_tile(z, nt);
    }
    run_t = t1;
  }
  if (run_t >= 0) {
    float* dp = dest(run_t);
    if (h0 == 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) dp[32 * i + col0] = carry[i * 64 + lane0];
    }
  }
}

// ------------------------------------------------------------------------------ pair head
struct HeadArgs {
  agdiff_head_params_t hp;
  const int32_t* n_dev;
  const int32_t* src;
  const int32_t* dst;
  const float* node_h;   // [N][128]
  const float* attr_frag;
  float* out;            // [E]
  int64_t max_tiles;
};

// assemble_atom_pair_feature (common.py:106-109) + MultiLayerPerceptron 256->128->64->1 (common.py:86-103)
__global__ void __launch_bounds__(AG_WG, 2) k_pair_head(HeadArgs a) {
  const int lane = ag_lane(), h = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + ag_wave_in_wg();
  const int E = *a.n_dev;
  if (tile >= a.max_tiles || tile * 32 >= E) return;
  const int64_t e = tile * 32 + (lane & 31);
  const bool valid = e < E;
  const int s = valid ? a.src[e] : 0, t = valid ? a.dst[e] : 0;

  // first layer streamed over eight 32-feature slices of [h_src * h_dst || edge_attr] (pkk weights)
  f32x16 y1[4];
  ag_init_vec<4>(y1, a.hp.b1, h);
  {
    const float* hs = a.node_h + (size_t)s * 128;
    const float* ht = a.node_h + (size_t)t * 128;
    f32x16 sl[2];
    auto load_slice = [&](auto which, int k) {
      constexpr int W = decltype(which)::value;
      if (k < 4) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          f32x4 u = ag_ld4(hs + 32 * k + 8 * rq + 4 * h), w = ag_ld4(ht + 32 * k + 8 * rq + 4 * h);
#pragma unroll
          for (int q = 0; q < 4; ++q) sl[W][4 * rq + q] = u[q] * w[q];
        }
      } else {
        ag_load_frag_tile<W>(sl, a.attr_frag, tile, k - 4, lane);
      }
    };
    load_slice(std::integral_constant<int, 0>{}, 0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k + 1 < 8) {
        if (k & 1) load_slice(std::integral_constant<int, 0>{}, k + 1);
        else load_slice(std::integral_constant<int, 1>{}, k + 1);
      }
      if (k & 1) ag_dense_std_k<1, 4, 1, 0>(sl, y1, a.hp.w1_pk + (size_t)k * 4 * 1024, lane);
      else ag_dense_std_k<1, 4, 0, 0>(sl, y1, a.hp.w1_pk + (size_t)k * 4 * 1024, lane);
    }
  }
  AG_FOR_TILE(y1, 4, ag_relu(v));
  f32x16 y2[2];
  ag_init_vec<2>(y2, a.hp.b2, h);
  ag_dense_std<4, 2, 0, 0, 4>(y1, y2, a.hp.w2_pk, lane);
  AG_FOR_TILE(y2, 2, ag_relu(v));
  const float o = ag_dot_vec<2>(y2, a.hp.w3, h) + a.hp.b3;
  if (valid && h == 0) a.out[e] = o;
}

// ------------------------------------------------------------------------------ stand-alone aggregate
// out[i][:] = sum_{e in in-list of i} x[src[e]][:] * W[e][:]   (PyG propagate, schnet.py:156,161-162).
// One wave per target node: F/4 lanes cover one edge row with 16-byte loads, so a wave streams
// 64/(F/4) edges per instruction; partial sums are combined with xor-shuffles.  HBM-bound on W.
template <int F>
__global__ void __launch_bounds__(AG_WG) k_cfconv_aggregate(const float* __restrict__ x, const float* __restrict__ W,
                                                            const int32_t* __restrict__ in_ptr,
                                                            const int32_t* __restrict__ src, int64_t n,
                                                            float* __restrict__ out) {
  constexpr int LPE = F / 4;        // lanes per edge row
  constexpr int EPI = 64 / LPE;     // edges per wave instruction
  const int lane = ag_lane();
  const int64_t node = (int64_t)blockIdx.x * 4 + ag_wave_in_wg();
  if (node >= n) return;
  const int lo = in_ptr[node], hi = in_ptr[node + 1];
  const int sub = lane / LPE, fl = (lane % LPE) * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int e = lo + sub; e < hi; e += EPI) {
    const f32x4 w = ag_ld4(W + (size_t)e * F + fl);
    const f32x4 xv = ag_ld4(x + (size_t)src[e] * F + fl);
    acc += w * xv;
  }
#pragma unroll
  for (int o = LPE; o < 64; o <<= 1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] += __shfl_xor(acc[q], o);
  }
  if (sub == 0) ag_st4(out + (size_t)node * F + fl, acc);
}

}  // namespace

extern "C" int agdiff_edge_encoder(const agdiff_params_t* p, const int32_t* n_edges_dev, int64_t max_tiles,
                                   const float* e_len, const int32_t* e_type, float* attr_frag, void* stream) {
  if (!p || !n_edges_dev || !e_len || !e_type || !attr_frag || max_tiles < 0) return AGDIFF_ERR_ARG;
  if (max_tiles == 0) return AGDIFF_OK;
  EncArgs a{p->ee_fe_w, p->ee_fe_b, p->ee_t1, p->ee_w1_pk, p->ee_t3, p->ee_w23_pk, p->ee_w4_pk, p->ee_b4,
            n_edges_dev, e_len, e_type, attr_frag, max_tiles};
  k_edge_encoder<<<dim3((unsigned)((max_tiles + 3) / 4)), dim3(AG_WG), 0, (hipStream_t)stream>>>(a);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}

extern "C" int agdiff_cfconv_fused(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, int32_t k,
                                   void* stream) {
  if (!p || !topo || !ws || k < 0 || k >= p->num_convs) return AGDIFF_ERR_ARG;
  if (topo->max_in_degree > AGDIFF_TILE * AGDIFF_CHUNK_TILES) return AGDIFF_ERR_LIMIT;
  const int64_t max_tiles = (topo->max_edges + 31) / 32;
  const int64_t max_chunks = (max_tiles + AGDIFF_CHUNK_TILES - 1) / AGDIFF_CHUNK_TILES;
  if (max_chunks == 0) return AGDIFF_OK;
  ConvArgs a;
  a.cp = p->conv[k];
  a.n_dev = ws->num_edges;
  a.in_ptr = ws->in_ptr;
  a.e_src = ws->e_src;
  a.e_dst = ws->e_dst;
  a.e_len = ws->e_len;
  a.e_attr = ws->e_attr;
  a.xs = ws->xs;
  a.agg = ws->agg;
  a.agg_first = ws->agg_first;
  a.max_chunks = max_chunks;
  a.cutoff = p->cutoff;
  a.smooth = p->smooth;
  k_cfconv_fused<<<dim3((unsigned)((max_chunks + 3) / 4)), dim3(AG_WG), 0, (hipStream_t)stream>>>(a);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}

extern "C" int agdiff_pair_head(const agdiff_head_params_t* hp, const int32_t* n_edges_dev, int64_t max_tiles,
                                const int32_t* src, const int32_t* dst, const float* node_h, const float* attr_frag,
                                float* out, void* stream) {
  if (!hp || !n_edges_dev || !src || !dst || !node_h || !attr_frag || !out || max_tiles < 0) return AGDIFF_ERR_ARG;
  if (max_tiles == 0) return AGDIFF_OK;
  HeadArgs a;
  a.hp = *hp;
  a.n_dev = n_edges_dev;
  a.src = src;
  a.dst = dst;
  a.node_h = node_h;
  a.attr_frag = attr_frag;
  a.out = out;
  a.max_tiles = max_tiles;
  k_pair_head<<<dim3((unsigned)((max_tiles + 3) / 4)), dim3(AG_WG), 0, (hipStream_t)stream>>>(a);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}

extern "C" int agdiff_cfconv_aggregate(const float* x, const float* W, const int32_t* in_ptr, const int32_t* src,
                                       int64_t num_nodes, int32_t F, float* out, void* stream) {
  if (!x || !W || !in_ptr || !src || !out || num_nodes < 0) return AGDIFF_ERR_ARG;
  if (F != 64 && F != 128) return AGDIFF_ERR_LIMIT;
  if (num_nodes == 0) return AGDIFF_OK;
  dim3 grid((unsigned)((num_nodes + 3) / 4));
  if (F == 128)
    k_cfconv_aggregate<128><<<grid, dim3(AG_WG), 0, (hipStream_t)stream>>>(x, W, in_ptr, src, num_nodes, out);
  else
    k_cfconv_aggregate<64><<<grid, dim3(AG_WG), 0, (hipStream_t)stream>>>(x, W, in_ptr, src, num_nodes, out);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}
