// Per-node kernels: SchNet node stages (everything of InteractionBlock / AdaptiveScaling that is not
// per-edge), GIN layers, and the per-molecule Langevin update.  One wave = one tile of 16 nodes in
// the MFMA accumulator layout (common.hpp): lane <-> node, registers <-> features.
#include "common.hpp"
#include <cstdlib>

namespace {

// ------------------------------------------------------------------------------ SchNet node stage
struct NodeStageArgs {
  agdiff_conv_params_t prev;   // block k-1 (finish)   -- valid when finish != 0
  agdiff_conv_params_t next;   // block k   (lin1)     -- valid when prep != 0
  const float* emb;            // [100][128]
  const int32_t* atom_type;
  const int32_t* in_ptr;
  const float* agg;
  const float* agg_first;
  float* h;
  const float* h_in;           // the block's input h for the residual (= h, or the cached stage-0 output for block 0)
  float* xs;
  int64_t n;
  int32_t finish;
  int32_t prep;
  int32_t chunk_edges;         // AG_TW * agdiff_conv_chunk_tiles(max_edges)
  // second CFConv pass of the block (split CFConv: agg = radius edges, agg2 = local edges), or in_ptr2 == null
  const int32_t* in_ptr2;
  const float* agg2;
  const float* agg_first2;
  int32_t chunk_edges2;
  int32_t* range_rows;         // agdiff_ws_t.range_rows (split-fp16: hidden activations at the edge of fp16's range)
};

// The aggregate of one node from one CFConv pass (edge.hip): agg[node] holds the part of the node's edge list that lies
// in the chunk where the list starts, every later chunk its part in agg_first[chunk]; they are added in chunk order.
struct AggSrc {
  const float* row;
  const float* first;
  int c_lo, c_hi;
  bool has;
};
__device__ __forceinline__ AggSrc ag_agg_src(const int32_t* in_ptr, const float* agg, const float* agg_first, int chunk_e,
                                             int64_t nd) {
  AggSrc r;
  if (!in_ptr) {            // agdiff_cfconv_node: one complete row per node, no chunks
    r.has = true;
    r.c_lo = r.c_hi = 0;
    r.row = agg + (size_t)nd * 192;
    r.first = agg_first;
    return r;
  }
  const int lo = in_ptr[nd], hi = in_ptr[nd + 1];
  r.has = hi > lo;
  r.c_lo = lo / chunk_e;
  r.c_hi = r.has ? (hi - 1) / chunk_e : r.c_lo;
  r.row = agg + (size_t)nd * 192;
  r.first = agg_first;
  return r;
}
// features 32 k + 4 q .. +3 (v0) and 32 k + 16 + 4 q .. +3 (v1) of the aggregate
__device__ __forceinline__ void ag_agg_slice(const AggSrc& r, int k, int q, f32x4& v0, f32x4& v1) {
  v0 = r.has ? ag_ld4(r.row + 32 * k + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  v1 = r.has ? ag_ld4(r.row + 32 * k + 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int c = r.c_lo + 1; c <= r.c_hi; ++c) {
    const float* fr = r.first + (size_t)c * 192;
    v0 += ag_ld4(fr + 32 * k + 4 * q);
    v1 += ag_ld4(fr + 32 * k + 16 + 4 * q);
  }
}

// Stage k of SchNetEncoder.forward (schnet.py:268-282):
//   finish: p1 = BN(lin2_1(agg1)), p2 = BN(lin2_2(agg2))         (schnet.py:157-158, BN folded)
//           x  = lin(ssp(cat[p1,p2]));  x *= sigmoid(att(x))      (schnet.py:206-214; ssp in base 2, its constants in lin2 / lin)
//           h += x * sigmoid(fc2(relu(fc1(x))))                   (schnet.py:230-234, 280)
//   prep:   xs = LeakyReLU(BN(lin1(h))) for conv1 | conv2         (schnet.py:153-155)
//   stage 0 (finish == 0): h = embedding[z]                       (schnet.py:271)
//
// One wave = one 16-node tile; a workgroup of W waves (W = 4..16, chosen by the launcher so that the grid is about
// one workgroup per CU) shares every weight matrix through LDS: the 188 weight blocks (376 KiB) of a stage are
// copied global -> LDS once per WORKGROUP in three steps instead of being streamed from L2 once per WAVE
// (1.1 GB of L2 reads per launch at 2,928 tiles, which is what bounded the first version of this kernel).
//   step A: [0,48) lin2a | lin2b     [48,80) lin blocks 0..31 (output tiles 0..3)
//   step B: [0,32) lin blocks 32..63 [32,48) gate1
//   step C: [0,4) scale1  [4,12) scale2  [12,60) lin1 of the next block
// Waves without a tile (tail of the grid) still take part in the copies and barriers.
// Small batches (fewer tiles than keep every CU busy that way) use LDSW = false: 4-wave workgroups, every wave
// streams the weights from L2 through a register ring (no copies, no barriers).
#define AG_NODE_LDS_BLOCKS 80
#define AG_NODE_DENSE(KOUTER, KT, OT, X0, O0, x, o, gsrc, lblock)                                   \
  do {                                                                                              \
    if constexpr (LDSW) ag_dense_lds<MODE, false, KOUTER, KT, OT, X0, O0>(x, o, L + (lblock) * 128, lane); \
    else ag_dense<MODE, false, KOUTER, KT, OT, X0, O0, PF>(x, o, gsrc, lane);                       \
  } while (0)
template <int MODE, bool LDSW>
__global__ void __launch_bounds__(LDSW ? 1024 : 256, LDSW ? 1 : 2) k_schnet_node_stage(NodeStageArgs a) {
  extern __shared__ u32x4 ag_node_smem[];
  lds_u32x4* L = (lds_u32x4*)ag_node_smem;
  constexpr int PF = (MODE == AG_F32) ? 3 : 10;   // LDSW = false, few waves per SIMD: hide the L2 latency in registers
  auto stage_in = [&](int dst_block, const float* src, int nblocks) {
    if constexpr (LDSW) {
      const u32x4* g = reinterpret_cast<const u32x4*>(src);
      ag_copy_lds<4>(L + dst_block * 128, g, nblocks * 128);     // (4 deep: tile state is live across the later phases)
    }
  };
  auto sync = [&]() {
    if constexpr (LDSW) __syncthreads();
  };
  const int lane = ag_lane(), q = lane >> 4;
  const int64_t tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const bool active = tile * AG_TW < a.n;
  const int64_t node = tile * AG_TW + (lane & 15);
  const bool valid = active && node < a.n;
  const int64_t nd = valid ? node : 0;

  f32x4 hv[8];
  if (!a.finish) {
    if (a.prep) stage_in(12, a.next.lin1_pk, 48);
    if (active) ag_load_row<8, 0>(hv, a.emb + (size_t)a.atom_type[nd] * 128, q);
    sync();
  } else {
    stage_in(0, a.prev.lin2a_pk, 32);
    stage_in(32, a.prev.lin2b_pk, 16);
    stage_in(48, a.prev.lin_pk, 32);
    sync();
    f32x4 xc[8];
    AgIn<MODE> ub[8];
    if (active) {
      f32x4 u[16];
      {
        // aggregates of the node: one CFConv pass, or the sum of the radius and the local pass (split CFConv)
        const AggSrc src1 = ag_agg_src(a.in_ptr, a.agg, a.agg_first, a.chunk_edges, nd);
        const bool two = a.in_ptr2 != nullptr;
        const AggSrc src2 = two ? ag_agg_src(a.in_ptr2, a.agg2, a.agg_first2, a.chunk_edges2, nd) : src1;
        ag_init_vec<16>(u, a.prev.lin2_b, q);
        AgIn<MODE> g[2];
        auto load_slice = [&](AgIn<MODE>& dst, int k) {
          f32x4 v0, v1;
          ag_agg_slice(src1, k, q, v0, v1);
          if (two) {
            f32x4 w0, w1;
            ag_agg_slice(src2, k, q, w0, w1);
            v0 += w0;
            v1 += w1;
          }
          ag_cvt(v0, v1, dst);
        };
        load_slice(g[0], 0);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          if (k + 1 < 6) load_slice(g[(k + 1) & 1], k + 1);
          // conv1.lin2: k-tiles 0..3 -> u[0..7]; conv2.lin2: k-tiles 4..5 -> u[8..15]   (pkk blocks)
          if (k < 4) {
            if (k & 1) AG_NODE_DENSE(true, 1, 8, 1, 0, g, u, ag_wblock(a.prev.lin2a_pk, k * 8), k * 8);
            else AG_NODE_DENSE(true, 1, 8, 0, 0, g, u, ag_wblock(a.prev.lin2a_pk, k * 8), k * 8);
          } else {
            if (k & 1) AG_NODE_DENSE(true, 1, 8, 1, 8, g, u, ag_wblock(a.prev.lin2b_pk, (k - 4) * 8), 32 + (k - 4) * 8);
            else AG_NODE_DENSE(true, 1, 8, 0, 8, g, u, ag_wblock(a.prev.lin2b_pk, (k - 4) * 8), 32 + (k - 4) * 8);
          }
        }
      }
      // InteractionBlock.act in base 2: act.beta log2(e) rides in lin2, ln 2 and the -ln 2 shift in lin (packing.py)
      AG_FOR_TILE(u, 16, ag_ssp_base2(v));
      ag_report_range<MODE>(ag_absmax<MODE, 16>(u, 0.0f), a.range_rows, nd, valid);      // (hidden activations as operands: common.hpp)
      ag_cvt_tiles<MODE, 8, 0>(u, ub);
      ag_init_vec<8>(xc, a.prev.lin_b, q);
      AG_NODE_DENSE(false, 8, 4, 0, 0, ub, xc, a.prev.lin_pk, 48);                             // output tiles 0..3
    }
    sync();
    stage_in(0, a.prev.lin_pk + (size_t)32 * 512, 32);
    stage_in(32, a.prev.gate1_pk, 16);
    sync();
    if (active) {
      AG_NODE_DENSE(false, 8, 4, 0, 4, ub, xc, ag_wblock(a.prev.lin_pk, 32), 0);               // output tiles 4..7
      f32x4 g1[4];
      ag_init_vec<4>(g1, a.prev.gate1_b, q);
      {
        AgIn<MODE> xb[4];
        ag_report_range<MODE>(ag_absmax<MODE, 8>(xc, 0.0f), a.range_rows, nd, valid);        // (the gate below only shrinks it)
        ag_cvt_tiles<MODE, 4, 0>(xc, xb);
        AG_NODE_DENSE(false, 4, 4, 0, 0, xb, g1, a.prev.gate1_pk, 32);
      }
      AG_FOR_TILE(g1, 4, ag_relu(v));
      const float gate = ag_sigmoid(ag_dot_vec<4>(g1, a.prev.gate2_w, q) + a.prev.gate2_b);
      AG_FOR_TILE(xc, 8, v * gate);
    }
    sync();
    stage_in(0, a.prev.scale1_pk, 4);
    stage_in(4, a.prev.scale2_pk, 8);
    if (a.prep) stage_in(12, a.next.lin1_pk, 48);
    sync();
    if (active) {
      // AdaptiveScaling: 128 -> 8 (one 16-row output tile, rows 8..15 zero) -> relu -> 8 -> 128 (one k-tile,
      // input features 8..31 zero)
      f32x4 s1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      f32x4 s2[8];
      {
        AgIn<MODE> xb[4];
        ag_cvt_tiles<MODE, 4, 0>(xc, xb);
        AG_NODE_DENSE(false, 4, 1, 0, 0, xb, s1, a.prev.scale1_pk, 0);
      }
      AG_FOR_TILE(s1, 1, ag_relu(v));
#pragma unroll
      for (int t = 0; t < 8; ++t) s2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        AgIn<MODE> sb[1];
        ag_report_range<MODE>(ag_absmax<MODE, 2>(s1, 0.0f), a.range_rows, nd, valid);
        ag_cvt_tiles<MODE, 1, 0>(s1, sb);
        AG_NODE_DENSE(false, 1, 8, 0, 0, sb, s2, a.prev.scale2_pk, 4);
      }
      ag_load_row<8, 0>(hv, a.h_in + (size_t)nd * 128, q);
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[t][r] = hv[t][r] + xc[t][r] * ag_sigmoid(s2[t][r]);
    }
  }
  if (!active) return;                     // no barrier below this point
  ag_report_range<MODE>(ag_absmax<MODE, 8>(hv, 0.0f), a.range_rows, nd, valid, 255.0f);       // (the node state: the heads multiply two of them)
  if (valid) ag_store_row<8, 0>(hv, a.h + (size_t)node * 128, q);
  if (a.prep) {
    f32x4 xo[12];
    ag_init_vec<12>(xo, a.next.lin1_b, q);
    {
      AgIn<MODE> hb[4];
      ag_cvt_tiles<MODE, 4, 0>(hv, hb);
      AG_NODE_DENSE(false, 4, 12, 0, 0, hb, xo, a.next.lin1_pk, 12);
    }
    AG_FOR_TILE(xo, 12, ag_lrelu(v));
    ag_report_range<MODE>(ag_absmax<MODE, 12>(xo, 0.0f), a.range_rows, nd, valid, 60000.0f);
    if (valid) ag_store_row<12, 0>(xo, a.xs + (size_t)node * 192, q);
  }
}

// ------------------------------------------------------------------------------ SchNet node stage, small batches
// The same stage for batches with so few node tiles that one wave per tile leaves most of the chip idle and every wave
// spends its time waiting for its own 188 weight blocks from L2 (376 KiB per tile, ~25 us per launch at 270 tiles):
// here a WORKGROUP of four waves owns one 16-node tile, wave w computes a quarter of every layer's output tiles
// (so it streams 50 blocks instead of 188, through one register ring that runs ahead across the layer boundaries), and
// the waves hand the activations to each other through LDS in MFMA operand form between the layers.
//   step list of wave w (s = ring position; FINISH part 38 steps, PREP part 12):
//   [0,8) lin2 conv1: out tiles 2w,2w+1 | [8,12) lin2 conv2 | [12,28) lin: out tiles 2w,2w+1 | [28,32) gate1: out tile w
//   [32,36) scale fc.0 (all waves alike) | [36,38) scale fc.2: out tiles 2w,2w+1 | [38,50) next lin1: out tiles 3w..3w+2
template <int MODE>
__device__ __forceinline__ void ag_xch_put(lds_u32x4* x, int kt, const AgIn<MODE>& v, int lane) {
  if constexpr (MODE == AG_F32) {
    x[(kt * 2) * 64 + lane] = __builtin_bit_cast(u32x4, v.v[0]);
    x[(kt * 2 + 1) * 64 + lane] = __builtin_bit_cast(u32x4, v.v[1]);
  } else {
    x[(kt * 2) * 64 + lane] = __builtin_bit_cast(u32x4, v.hi);
    x[(kt * 2 + 1) * 64 + lane] = __builtin_bit_cast(u32x4, v.lo);
  }
}
template <int MODE>
__device__ __forceinline__ void ag_xch_get(const lds_u32x4* x, int kt, AgIn<MODE>& v, int lane) {
  if constexpr (MODE == AG_F32) {
    v.v[0] = __builtin_bit_cast(f32x4, x[(kt * 2) * 64 + lane]);
    v.v[1] = __builtin_bit_cast(f32x4, x[(kt * 2 + 1) * 64 + lane]);
  } else {
    v.hi = __builtin_bit_cast(decltype(v.hi), x[(kt * 2) * 64 + lane]);
    v.lo = __builtin_bit_cast(decltype(v.lo), x[(kt * 2 + 1) * 64 + lane]);
  }
}

template <int MODE, bool FINISH, bool PREP>
__global__ void __launch_bounds__(256, 2) k_schnet_node_stage_split(NodeStageArgs a) {
  __shared__ u32x4 ag_split_xch[20 * 128];         // 20 k-tiles of 2 KiB: U 0..7 | X 8..11 | gated X 12..15 | H 16..19
  __shared__ float ag_split_red[4][16];
  lds_u32x4* xch = (lds_u32x4*)ag_split_xch;
  const int lane = ag_lane(), q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(ag_wave_in_wg());
  const int64_t tile = blockIdx.x;
  const int64_t node = tile * AG_TW + (lane & 15);
  const bool valid = node < a.n;
  const int64_t nd = valid ? node : 0;
  constexpr int NF = FINISH ? 38 : 0, NS = NF + (PREP ? 12 : 0);
  constexpr int PF = 12, R = PF + 1;      // 12 blocks (24 KiB) in flight per wave: the stage is a chain of L2 round trips

  auto block_of = [&](int s) -> const float* {
    if (FINISH && s < 8) return a.prev.lin2a_pk + (size_t)((s >> 1) * 8 + 2 * w + (s & 1)) * 512;
    if (FINISH && s < 12) return a.prev.lin2b_pk + (size_t)(((s - 8) >> 1) * 8 + 2 * w + (s & 1)) * 512;
    if (FINISH && s < 28) return a.prev.lin_pk + (size_t)((2 * w + ((s - 12) >> 3)) * 8 + ((s - 12) & 7)) * 512;
    if (FINISH && s < 32) return a.prev.gate1_pk + (size_t)(w * 4 + (s - 28)) * 512;
    if (FINISH && s < 36) return a.prev.scale1_pk + (size_t)(s - 32) * 512;
    if (FINISH && s < 38) return a.prev.scale2_pk + (size_t)(2 * w + (s - 36)) * 512;
    return a.next.lin1_pk + (size_t)((3 * w + ((s - NF) >> 2)) * 4 + ((s - NF) & 3)) * 512;
  };
  u32x4 ring[R][2];
  auto request = [&](int s) {
    const u32x4* p = reinterpret_cast<const u32x4*>(block_of(s)) + lane;
    ring[s % R][0] = p[0];
    ring[s % R][1] = p[64];
  };
#pragma unroll
  for (int s = 0; s < (PF < NS ? PF : NS); ++s) request(s);

  f32x4 u4[4], xc2[2], g1[1], s1[2], s2[2], hv2[2], xo3[3];
  AgIn<MODE> gk[6], ub[8], xb[4], sb[1], hb[4];
  if (FINISH) {
    // aggregates of the tile's nodes (every wave needs the whole 192-wide row), as k_schnet_node_stage
    const AggSrc src1 = ag_agg_src(a.in_ptr, a.agg, a.agg_first, a.chunk_edges, nd);
    const bool two = a.in_ptr2 != nullptr;
    const AggSrc src2 = two ? ag_agg_src(a.in_ptr2, a.agg2, a.agg_first2, a.chunk_edges2, nd) : src1;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      f32x4 v0, v1;
      ag_agg_slice(src1, k, q, v0, v1);
      if (two) {
        f32x4 w0, w1;
        ag_agg_slice(src2, k, q, w0, w1);
        v0 += w0;
        v1 += w1;
      }
      ag_cvt(v0, v1, gk[k]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      u4[j] = ag_ld4(a.prev.lin2_b + 16 * (2 * w + j) + 4 * q);
      u4[2 + j] = ag_ld4(a.prev.lin2_b + 128 + 16 * (2 * w + j) + 4 * q);
      xc2[j] = ag_ld4(a.prev.lin_b + 16 * (2 * w + j) + 4 * q);
      s2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    g1[0] = ag_ld4(a.prev.gate1_b + 16 * w + 4 * q);
    s1[0] = s1[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) hv2[j] = ag_ld4(a.h_in + (size_t)nd * 128 + 16 * (2 * w + j) + 4 * q);   // used at the end
  } else {
    // stage 0: h = embedding[z]; wave w writes its two tiles of the row, every wave keeps the whole row as operands
    f32x4 hrow[8];
    ag_load_row<8, 0>(hrow, a.emb + (size_t)a.atom_type[nd] * 128, q);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // tiles 2w, 2w+1 of the row (runtime w: select with a short chain)
      f32x4 t = hrow[j];
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) t = (w == ww) ? hrow[2 * ww + j] : t;
      if (valid) ag_st4(a.h + (size_t)node * 128 + 16 * (2 * w + j) + 4 * q, t);
    }
    ag_cvt_tiles<MODE, 4, 0>(hrow, hb);
  }
  if (PREP) {
#pragma unroll
    for (int j = 0; j < 3; ++j) xo3[j] = ag_ld4(a.next.lin1_b + 16 * (3 * w + j) + 4 * q);
  }

  // one ring position: request the block PF steps ahead, then use this step's block (every loop below is short and
  // fully unrolled, so ring / operand / accumulator indices are compile-time constants)
#define AG_SPLIT_STEP(s, acc, x)                        \
  do {                                                  \
    if ((s) + PF < NS) request((s) + PF);               \
    ag_block_mma<MODE, false>(acc, x, ring[(s) % R]);   \
    __builtin_amdgcn_sched_barrier(0);                  \
  } while (0)
  if constexpr (FINISH) {
#pragma unroll
    for (int i = 0; i < 8; ++i) AG_SPLIT_STEP(i, u4[i & 1], gk[i >> 1]);
#pragma unroll
    for (int i = 0; i < 4; ++i) AG_SPLIT_STEP(8 + i, u4[2 + (i & 1)], gk[4 + (i >> 1)]);
    {
      AG_FOR_TILE(u4, 4, ag_ssp_base2(v));
      ag_report_range<MODE>(ag_absmax<MODE, 4>(u4, 0.0f), a.range_rows, nd, valid);
      AgIn<MODE> k0, k1;
      ag_cvt(u4[0], u4[1], k0);
      ag_cvt(u4[2], u4[3], k1);
      ag_xch_put<MODE>(xch, w, k0, lane);
      ag_xch_put<MODE>(xch, 4 + w, k1, lane);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 8; ++t) ag_xch_get<MODE>(xch, t, ub[t], lane);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) AG_SPLIT_STEP(12 + i, xc2[i >> 3], ub[i & 7]);
    {
      AgIn<MODE> k0;
      ag_report_range<MODE>(ag_absmax<MODE, 2>(xc2, 0.0f), a.range_rows, nd, valid);
      ag_cvt(xc2[0], xc2[1], k0);
      ag_xch_put<MODE>(xch, 8 + w, k0, lane);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 4; ++t) ag_xch_get<MODE>(xch, 8 + t, xb[t], lane);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) AG_SPLIT_STEP(28 + i, g1[0], xb[i]);
    {
      AG_FOR_TILE(g1, 1, ag_relu(v));
      const f32x4 gw = ag_ld4(a.prev.gate2_w + 16 * w + 4 * q);
      float part = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) part = fmaf(gw[r], g1[0][r], part);
      part = ag_quarter_sum(part);
      if (q == 0) ag_split_red[w][lane & 15] = part;
      __syncthreads();
      const int c = lane & 15;
      const float tot = ((ag_split_red[0][c] + ag_split_red[1][c]) + ag_split_red[2][c]) + ag_split_red[3][c];
      const float gate = ag_sigmoid(tot + a.prev.gate2_b);
      AG_FOR_TILE(xc2, 2, v * gate);
      AgIn<MODE> k0;
      ag_cvt(xc2[0], xc2[1], k0);
      ag_xch_put<MODE>(xch, 12 + w, k0, lane);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 4; ++t) ag_xch_get<MODE>(xch, 12 + t, xb[t], lane);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) AG_SPLIT_STEP(32 + i, s1[0], xb[i]);
    AG_FOR_TILE(s1, 1, ag_relu(v));
    ag_report_range<MODE>(ag_absmax<MODE, 2>(s1, 0.0f), a.range_rows, nd, valid);
    ag_cvt(s1[0], s1[1], sb[0]);
#pragma unroll
    for (int i = 0; i < 2; ++i) AG_SPLIT_STEP(36 + i, s2[i], sb[0]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) hv2[j][r] = hv2[j][r] + xc2[j][r] * ag_sigmoid(s2[j][r]);
      if (valid) ag_st4(a.h + (size_t)node * 128 + 16 * (2 * w + j) + 4 * q, hv2[j]);
    }
    ag_report_range<MODE>(ag_absmax<MODE, 2>(hv2, 0.0f), a.range_rows, nd, valid, 255.0f);
    if constexpr (PREP) {
      AgIn<MODE> k0;
      ag_cvt(hv2[0], hv2[1], k0);
      ag_xch_put<MODE>(xch, 16 + w, k0, lane);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 4; ++t) ag_xch_get<MODE>(xch, 16 + t, hb[t], lane);
    }
  }
  if constexpr (PREP) {
#pragma unroll
    for (int i = 0; i < 12; ++i) AG_SPLIT_STEP(NF + i, xo3[i >> 2], hb[i & 3]);
  }
#undef AG_SPLIT_STEP
  if (PREP) {
    AG_FOR_TILE(xo3, 3, ag_lrelu(v));
    ag_report_range<MODE>(ag_absmax<MODE, 3>(xo3, 0.0f), a.range_rows, nd, valid, 60000.0f);
    if (valid) {
#pragma unroll
      for (int j = 0; j < 3; ++j) ag_st4(a.xs + (size_t)node * 192 + 16 * (3 * w + j) + 4 * q, xo3[j]);
    }
  }
}

// ------------------------------------------------------------------------------ GIN layer
struct GinArgs {
  agdiff_gin_params_t gp;
  const float* emb;            // non-null on layer 0: input = node_emb[z]
  const int32_t* atom_type;
  const int32_t* loc_in_ptr;
  const int32_t* in_src;       // [L] by in-slot: source node
  const int32_t* in_row;       // [L] by in-slot: row of l_attr_rows
  const float* l_attr_rows;    // fp32 local edge attrs, row-major
  const float* h_in;
  float* h_out;
  int64_t n;
  int32_t* range_rows;         // agdiff_ws_t.range_rows
};

// GINEConv message sum (gin.py:57-63): m_i = sum_{e: dst = i} relu(h_src(e) + edge_attr_e) + (1 + eps) h_i, written to
// the layer's OUTPUT buffer (k_gin_layer replaces it by the layer's result).  Its own launch because it is bound by
// memory latency: inside the MLP kernel (16-node tiles, ~110 VGPRs, one 12..16-wave workgroup per CU next to 128 KiB of
// weights) a wave had no load in flight 80 % of the time and the gather was 55 of the layer's 74 us.  Here a half-wave
// owns a node, the lanes lie along the 512-byte rows (16 B each), four messages are in flight per node, and 8 waves fit
// a SIMD.  Messages are added in list order (sources ascending), as before.
__global__ void __launch_bounds__(256) k_gin_gather(GinArgs a) {
  const int lane = ag_lane(), half = lane >> 5, j = lane & 31;
  // Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8): give each XCD a contiguous range of nodes, so that a
  // molecule's rows -- every attribute row is read by both of its end points, every h row by all neighbours -- are fetched
  // into ONE L2
  const int64_t wg = (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;
  const int64_t node = (wg * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + half;
  const bool valid = node < a.n;
  const int nd = valid ? (int)node : 0;
  const int lo = a.loc_in_ptr[nd], hi = valid ? a.loc_in_ptr[nd + 1] : lo;
  const int deg2 = max(__builtin_amdgcn_readlane(hi - lo, 0), __builtin_amdgcn_readlane(hi - lo, 32));
  const float* hbase = a.emb ? a.emb : a.h_in;
  const f32x4 hself = ag_ld4((a.emb ? a.emb + (size_t)a.atom_type[nd] * 128 : a.h_in + (size_t)nd * 128) + 4 * j);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 4;           // messages in flight per node (8: slower, the extra slots of short lists cost more than they hide)
  for (int base = 0; base < deg2; base += 32) {
    // lane j of a half holds the indices of its node's message base + j (-1: past the list)
    const int slot = lo + base + j;
    const bool have = slot < hi;
    int my_src = have ? a.in_src[slot] : -1;
    const int my_row = have ? a.in_row[slot] : 0;
    if (a.emb && have) my_src = a.atom_type[my_src];        // layer 0: the source's row of the embedding table
    const int cnt = min(32, deg2 - base);
    for (int k = 0; k < cnt; k += U) {
      f32x4 hv[U], ev[U];
      bool on[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int s = __shfl(my_src, 32 * half + ((k + u) & 31)), r = __shfl(my_row, 32 * half + ((k + u) & 31));
        on[u] = s >= 0;
        hv[u] = ag_ld4(hbase + (size_t)(on[u] ? s : 0) * 128 + 4 * j);
        ev[u] = ag_ld4(a.l_attr_rows + (size_t)r * 128 + 4 * j);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += on[u] ? ag_relu(hv[u][r] + ev[u][r]) : 0.0f;
      }
    }
  }
  acc = acc + a.gp.one_plus_eps * hself;
  if (valid) *reinterpret_cast<f32x4*>(a.h_out + (size_t)node * 128 + 4 * j) = acc;
}

// GINEConv MLP + BN + relu + residual (gin.py:57-63, 131-138) on the message sums k_gin_gather left in h_out:
// u = MLP(m_i); u = BN(u) (folded); relu except last layer; h = u + h.
// Like the SchNet node stage, a workgroup of W one-tile waves shares the layer's two weight matrices (64 blocks,
// 128 KiB) through LDS.
template <int MODE, bool LDSW>
__global__ void __launch_bounds__(LDSW ? 1024 : 256, LDSW ? 1 : 2) k_gin_layer(GinArgs a) {
  extern __shared__ u32x4 ag_gin_smem[];
  lds_u32x4* L = (lds_u32x4*)ag_gin_smem;
  constexpr int PF = (MODE == AG_F32) ? 3 : 10;
  if constexpr (LDSW) {
    const u32x4* g1 = reinterpret_cast<const u32x4*>(a.gp.w1_pk);
    const u32x4* g2 = reinterpret_cast<const u32x4*>(a.gp.w2_pk);
    ag_copy_lds(L, g1, 32 * 128);
    ag_copy_lds(L + 32 * 128, g2, 32 * 128);
  }
  const int lane = ag_lane(), q = lane >> 4;
  const int64_t tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const bool active = tile * AG_TW < a.n;
  const int64_t node = tile * AG_TW + (lane & 15);
  const bool valid = active && node < a.n;
  const int64_t nd = valid ? node : 0;
  const float* hin_self = a.emb ? a.emb + (size_t)a.atom_type[nd] * 128 : a.h_in + (size_t)nd * 128;

  f32x4 m[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) m[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 hself[8];
  if (active) {
    ag_load_row<8, 0>(m, a.h_out + (size_t)nd * 128, q);
    ag_load_row<8, 0>(hself, hin_self, q);
  }
  if constexpr (LDSW) __syncthreads();
  if (!active) return;
  f32x4 y1[8];
  ag_init_vec<8>(y1, a.gp.b1, q);
  {
    AgIn<MODE> mb[4];
    ag_cvt_tiles<MODE, 4, 0>(m, mb);
    AG_NODE_DENSE(false, 4, 8, 0, 0, mb, y1, a.gp.w1_pk, 0);
  }
  AG_FOR_TILE(y1, 8, ag_relu(v));
  ag_init_vec<8>(m, a.gp.b2, q);
  {
    AgIn<MODE> yb[4];
    ag_report_range<MODE>(ag_absmax<MODE, 8>(y1, 0.0f), a.range_rows, nd, valid);
    ag_cvt_tiles<MODE, 4, 0>(y1, yb);
    AG_NODE_DENSE(false, 4, 8, 0, 0, yb, m, a.gp.w2_pk, 32);
  }
  if (a.gp.relu_out) { AG_FOR_TILE(m, 8, ag_relu(v)); }
#pragma unroll
  for (int t = 0; t < 8; ++t) m[t] += hself[t];
  ag_report_range<MODE>(ag_absmax<MODE, 8>(m, 0.0f), a.range_rows, nd, valid, 255.0f);
  if (valid) ag_store_row<8, 0>(m, a.h_out + (size_t)node * 128, q);
}

// The same layer with PERSISTENT workgroups (the launcher: more 16-wave rounds than CUs): the two matrices are staged once per CU
// and every wave walks its tiles without another barrier -- 768 one-round workgroups staged the 128 KiB three times per CU on
// 196 k atoms (288 -> 218 us in-step).  Its own kernel: with the loop around the one-round body the small-batch and exact-fp32
// instantiations compiled to slower code (alanine dipeptide +28 %, exact fp32 on 47 k atoms +17 % per step).
template <int MODE>
__global__ void __launch_bounds__(1024, 1) k_gin_layer_persistent(GinArgs a) {
  constexpr bool LDSW = true;
  extern __shared__ u32x4 ag_gin_smem[];
  lds_u32x4* L = (lds_u32x4*)ag_gin_smem;
  constexpr int PF = (MODE == AG_F32) ? 3 : 10;
  if constexpr (LDSW) {
    const u32x4* g1 = reinterpret_cast<const u32x4*>(a.gp.w1_pk);
    const u32x4* g2 = reinterpret_cast<const u32x4*>(a.gp.w2_pk);
    ag_copy_lds(L, g1, 32 * 128);
    ag_copy_lds(L + 32 * 128, g2, 32 * 128);
    __syncthreads();
  }
  const int lane = ag_lane(), q = lane >> 4;
  const int64_t tiles = (a.n + AG_TW - 1) / AG_TW;
  // LDSW: PERSISTENT workgroups (at most one per CU: the launcher) -- the 128 KiB of weights are staged once and every wave
  // walks its tiles without another barrier (768 one-round workgroups staged them three times per CU on 196 k atoms)
  const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < tiles; tile += stride) {
    const int64_t node = tile * AG_TW + (lane & 15);
    const bool valid = node < a.n;
    const int64_t nd = valid ? node : 0;
    const float* hin_self = a.emb ? a.emb + (size_t)a.atom_type[nd] * 128 : a.h_in + (size_t)nd * 128;
    f32x4 m[8], hself[8];
    ag_load_row<8, 0>(m, a.h_out + (size_t)nd * 128, q);
    ag_load_row<8, 0>(hself, hin_self, q);
    f32x4 y1[8];
    ag_init_vec<8>(y1, a.gp.b1, q);
    {
      AgIn<MODE> mb[4];
      ag_cvt_tiles<MODE, 4, 0>(m, mb);
      AG_NODE_DENSE(false, 4, 8, 0, 0, mb, y1, a.gp.w1_pk, 0);
    }
    AG_FOR_TILE(y1, 8, ag_relu(v));
    ag_init_vec<8>(m, a.gp.b2, q);
    {
      AgIn<MODE> yb[4];
      ag_report_range<MODE>(ag_absmax<MODE, 8>(y1, 0.0f), a.range_rows, nd, valid);
      ag_cvt_tiles<MODE, 4, 0>(y1, yb);
      AG_NODE_DENSE(false, 4, 8, 0, 0, yb, m, a.gp.w2_pk, 32);
    }
    if (a.gp.relu_out) { AG_FOR_TILE(m, 8, ag_relu(v)); }
#pragma unroll
    for (int t = 0; t < 8; ++t) m[t] += hself[t];
    ag_report_range<MODE>(ag_absmax<MODE, 8>(m, 0.0f), a.range_rows, nd, valid, 255.0f);
    if (valid) ag_store_row<8, 0>(m, a.h_out + (size_t)node * 128, q);
  }
}

// ------------------------------------------------------------------------------ Langevin update
struct UpdateArgs {
  agdiff_step_args_t s;
  const int32_t* graph_ptr;
  // local edges
  const int32_t* loc_src;
  const int32_t* loc_dst;
  const int32_t* loc_out_ptr;
  const int32_t* loc_in_ptr;
  const int32_t* loc_in_eid;
  const float* l_len;
  const float* l_inv;
  // dynamic edges
  const int32_t* in_ptr;
  const int32_t* out_ptr;
  const int32_t* ref2dst;
  const int32_t* e_src;
  const int32_t* e_dst;
  const int32_t* e_type;
  const float* e_len;
  const float* e_inv;
  int32_t* nan_flag;
  int32_t parts;         // lanes per atom (power of two, 1..16)
};

__device__ __forceinline__ void clip3(float& x, float& y, float& z, float limit) {  // dualenc.py:586-589
  const float nrm = sqrtf(x * x + y * y + z * z);
  if (nrm > limit) {
    const float d = limit / nrm;
    x *= d; y *= d; z *= d;
  }
}

// One workgroup per molecule: eq_transform of the local and (optionally) global edge scores
// (geometry.py:9-17), clip_norm, the Langevin move (dualenc.py:526-538), NaN flag, center_pos,
// clamp, trajectory copy (dualenc.py:539-545).  P = a.parts lanes share one atom's edge lists (strided) and
// combine their partial forces with xor-shuffles (P is a power of two <= 16, so the lanes of an atom sit in one wave).
__global__ void __launch_bounds__(1024) k_langevin_update(UpdateArgs a) {
  const int g = blockIdx.x;
  const int g0 = a.graph_ptr[g], n = a.graph_ptr[g + 1] - g0;
  __shared__ float red[3][16];
  __shared__ int nanw[16];
  float sx = 0.f, sy = 0.f, sz = 0.f;
  int bad = 0;
  // A graph in which a NaN has appeared is QUARANTINED from then on (sticky flag, cleared by the host when a job starts):
  // its positions are replaced by a finite placeholder geometry and no longer updated, so that no NaN ever enters another
  // forward -- graphs are independent by construction, but a 0 x NaN in a masked sum of a kernel whose 16-edge tile spans
  // two graphs would not be.  The reference stops the whole job at this step (dualenc.py:539-541); here the host raises at
  // its next poll of nan_flag[0], or re-samples the flagged molecules (agdiff_amd/driver.py).
  const int was_bad = a.nan_flag[1 + g];
  const int P = a.parts, part = threadIdx.x & (P - 1);
  const int per_pass = blockDim.x / P;
  // pass 1: new (uncentred) positions into scratch, per-thread partial sums for the centroid
  for (int base = 0; base < n; base += per_pass) {
    const int li = base + (int)(threadIdx.x / P);
    const bool on = li < n;
    const int i = g0 + (on ? li : 0);
    const float px = a.s.pos_in[3 * i], py = a.s.pos_in[3 * i + 1], pz = a.s.pos_in[3 * i + 2];
    float lx = 0.f, ly = 0.f, lz = 0.f;
    if (on) {
      // (as the global loops below: four edges at a time, the loads of each dependency level issued together)
      constexpr int UL = 4;
      const int lo1 = a.loc_out_ptr[i + 1];
      for (int e0 = a.loc_out_ptr[i] + part; e0 < lo1; e0 += UL * P) {            // row == i: + dd_dr * score
        int j[UL];
        float ln[UL], sc[UL];
        bool ok[UL];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
          ok[u] = e0 + u * P < lo1;
          const int e = ok[u] ? e0 + u * P : 0;
          j[u] = ok[u] ? a.loc_dst[e] : i;
          ln[u] = ok[u] ? a.l_len[e] : 1.0f;
          sc[u] = ok[u] ? a.l_inv[e] : 0.0f;
        }
        float qx[UL], qy[UL], qz[UL];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
          qx[u] = a.s.pos_in[3 * j[u]]; qy[u] = a.s.pos_in[3 * j[u] + 1]; qz[u] = a.s.pos_in[3 * j[u] + 2];
        }
#pragma unroll
        for (int u = 0; u < UL; ++u) {
          if (!ok[u]) continue;
          const float w = 1.0f / ln[u];
          lx += (w * (px - qx[u])) * sc[u];
          ly += (w * (py - qy[u])) * sc[u];
          lz += (w * (pz - qz[u])) * sc[u];
        }
      }
      const int li1 = a.loc_in_ptr[i + 1];
      for (int k0 = a.loc_in_ptr[i] + part; k0 < li1; k0 += UL * P) {              // col == i: - dd_dr * score
        int e[UL], j[UL];
        float ln[UL], sc[UL];
        bool ok[UL];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
          ok[u] = k0 + u * P < li1;
          e[u] = ok[u] ? a.loc_in_eid[k0 + u * P] : 0;
        }
#pragma unroll
        for (int u = 0; u < UL; ++u) {
          j[u] = ok[u] ? a.loc_src[e[u]] : i;
          ln[u] = ok[u] ? a.l_len[e[u]] : 1.0f;
          sc[u] = ok[u] ? a.l_inv[e[u]] : 0.0f;
        }
        float qx[UL], qy[UL], qz[UL];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
          qx[u] = a.s.pos_in[3 * j[u]]; qy[u] = a.s.pos_in[3 * j[u] + 1]; qz[u] = a.s.pos_in[3 * j[u] + 2];
        }
#pragma unroll
        for (int u = 0; u < UL; ++u) {
          if (!ok[u]) continue;
          const float w = 1.0f / ln[u];
          lx -= (w * (qx[u] - px)) * sc[u];
          ly -= (w * (qy[u] - py)) * sc[u];
          lz -= (w * (qz[u] - pz)) * sc[u];
        }
      }
    }
    for (int o = P >> 1; o > 0; o >>= 1) { lx += __shfl_xor(lx, o); ly += __shfl_xor(ly, o); lz += __shfl_xor(lz, o); }
    if (a.s.clip_local >= 0.0f) clip3(lx, ly, lz, a.s.clip_local);
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (a.s.use_global) {
      if (on) {
        // Both loops are chains of dependent loads (index -> edge record -> neighbour position): four iterations are
        // taken at a time with the loads of each level issued together, so that a lane waits for three round trips per
        // four edges instead of per edge.  The order of the additions is that of the plain loop.
        constexpr int U = 4;
        const int q1 = a.out_ptr[i + 1];
        for (int q0 = a.out_ptr[i] + part; q0 < q1; q0 += U * P) {
          int e[U], ty[U], j[U];
          float ln[U], sc[U];
          bool ok[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            ok[u] = q0 + u * P < q1;
            e[u] = ok[u] ? a.ref2dst[q0 + u * P] : 0;
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            ty[u] = ok[u] ? a.e_type[e[u]] : 1;
            j[u] = ok[u] ? a.e_dst[e[u]] : i;
            ln[u] = ok[u] ? a.e_len[e[u]] : 1.0f;
            sc[u] = ok[u] ? a.e_inv[e[u]] : 0.0f;
          }
          float qx[U], qy[U], qz[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            qx[u] = a.s.pos_in[3 * j[u]]; qy[u] = a.s.pos_in[3 * j[u] + 1]; qz[u] = a.s.pos_in[3 * j[u] + 2];
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (!ok[u] || ty[u] != 0) continue;      // edge_inv_global * (1 - local_edge_mask), dualenc.py:516-518
            const float w = 1.0f / ln[u];
            gx += (w * (px - qx[u])) * sc[u];
            gy += (w * (py - qy[u])) * sc[u];
            gz += (w * (pz - qz[u])) * sc[u];
          }
        }
        const int e1 = a.in_ptr[i + 1];
        for (int e0 = a.in_ptr[i] + part; e0 < e1; e0 += U * P) {
          int ty[U], j[U];
          float ln[U], sc[U];
          bool ok[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            ok[u] = e0 + u * P < e1;
            const int e = ok[u] ? e0 + u * P : 0;
            ty[u] = ok[u] ? a.e_type[e] : 1;
            j[u] = ok[u] ? a.e_src[e] : i;
            ln[u] = ok[u] ? a.e_len[e] : 1.0f;
            sc[u] = ok[u] ? a.e_inv[e] : 0.0f;
          }
          float qx[U], qy[U], qz[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            qx[u] = a.s.pos_in[3 * j[u]]; qy[u] = a.s.pos_in[3 * j[u] + 1]; qz[u] = a.s.pos_in[3 * j[u] + 2];
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (!ok[u] || ty[u] != 0) continue;
            const float w = 1.0f / ln[u];
            gx -= (w * (qx[u] - px)) * sc[u];
            gy -= (w * (qy[u] - py)) * sc[u];
            gz -= (w * (qz[u] - pz)) * sc[u];
          }
        }
      }
      for (int o = P >> 1; o > 0; o >>= 1) { gx += __shfl_xor(gx, o); gy += __shfl_xor(gy, o); gz += __shfl_xor(gz, o); }
      clip3(gx, gy, gz, a.s.clip);
    }
    if (on && part == 0) {
      const float ex = lx + gx * a.s.w_global, ey = ly + gy * a.s.w_global, ez = lz + gz * a.s.w_global;
      const float nx = (px + (a.s.step_size * ex) / a.s.sigma) + a.s.noise[3 * i] * a.s.noise_scale;
      const float ny = (py + (a.s.step_size * ey) / a.s.sigma) + a.s.noise[3 * i + 1] * a.s.noise_scale;
      const float nz = (pz + (a.s.step_size * ez) / a.s.sigma) + a.s.noise[3 * i + 2] * a.s.noise_scale;
      bad |= (nx != nx) | (ny != ny) | (nz != nz);
      sx += nx; sy += ny; sz += nz;
      a.s.scratch[3 * i] = nx; a.s.scratch[3 * i + 1] = ny; a.s.scratch[3 * i + 2] = nz;
    }
  }
  __syncthreads();     // scratch[] of this molecule is complete (pass 2 reads other threads' rows)
  // centroid
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o);
    bad |= __shfl_xor(bad, o);
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wv] = sx; red[1][wv] = sy; red[2][wv] = sz; nanw[wv] = bad; }
  __syncthreads();
  const int nw = blockDim.x >> 6;
  float cx = 0.f, cy = 0.f, cz = 0.f;
  int anybad = 0;
  for (int w = 0; w < nw; ++w) { cx += red[0][w]; cy += red[1][w]; cz += red[2][w]; anybad |= nanw[w]; }
  const float inv_n = 1.0f / (float)(n > 0 ? n : 1);
  cx *= inv_n; cy *= inv_n; cz *= inv_n;
  if (anybad && threadIdx.x == 0) {
    a.nan_flag[0] = 1;
    a.nan_flag[1 + g] = 1;       // per graph: the driver re-samples only the molecules that diverged (test.py:143-181)
  }
  const bool frozen = was_bad || anybad;
  for (int li = threadIdx.x; li < n; li += blockDim.x) {
    const int i = g0 + li;
    float x = a.s.scratch[3 * i] - cx, y = a.s.scratch[3 * i + 1] - cy, z = a.s.scratch[3 * i + 2] - cz;
    if (a.s.clip_pos >= 0.0f) {
      x = fminf(fmaxf(x, -a.s.clip_pos), a.s.clip_pos);
      y = fminf(fmaxf(y, -a.s.clip_pos), a.s.clip_pos);
      z = fminf(fmaxf(z, -a.s.clip_pos), a.s.clip_pos);
    }
    float tx = x, ty = y, tz = z;
    if (frozen) {          // placeholder: a centred straight chain, 1.5 apart (finite, no two atoms at one place)
      x = ((float)li - 0.5f * (float)(n - 1)) * 1.5f;
      y = z = 0.0f;
      tx = ty = tz = __uint_as_float(0x7FC00000u);       // the trajectory shows the graph as lost
    }
    a.s.pos_out[3 * i] = x; a.s.pos_out[3 * i + 1] = y; a.s.pos_out[3 * i + 2] = z;
    if (a.s.traj_out) { a.s.traj_out[3 * i] = tx; a.s.traj_out[3 * i + 1] = ty; a.s.traj_out[3 * i + 2] = tz; }
  }
}

// ------------------------------------------------------------------------------ diffusion loss (forward only)
// get_loss_diffusion, dualenc.py:284-395, as scripts/train.py:160-170 evaluates it under no_grad.
struct PerturbArgs {
  const int32_t* graph_ptr;
  const float* pos;
  const float* noise;
  const float* alpha_graph;
  float* out;
};

// pos_perturbed = pos + pos_noise * sqrt(1 - a) / sqrt(a), a = alphas[time_step[graph]]  (dualenc.py:308-312)
__global__ void __launch_bounds__(256) k_perturb_positions(PerturbArgs a) {
  const int g = blockIdx.x;
  const int g0 = a.graph_ptr[g], n = a.graph_ptr[g + 1] - g0;
  const float al = a.alpha_graph[g];
  const float s1 = sqrtf(1.0f - al), s2 = sqrtf(al);
  for (int k = threadIdx.x; k < 3 * n; k += blockDim.x) {
    const int idx = 3 * g0 + k;
    a.out[idx] = a.pos[idx] + (a.noise[idx] * s1) / s2;
  }
}

struct LossArgs {
  const int32_t* graph_ptr;
  const float* pos_gt;
  const float* pos;          // perturbed positions the score network was evaluated on
  const float* alpha_graph;
  float* loss;               // [3][N]: total, global, local
  int64_t n_total;
  float cutoff;
  const int32_t* loc_src;
  const int32_t* loc_dst;
  const int32_t* loc_out_ptr;
  const int32_t* loc_in_ptr;
  const int32_t* loc_in_eid;
  const float* l_len;
  const float* l_inv;
  const int32_t* in_ptr;
  const int32_t* out_ptr;
  const int32_t* ref2dst;
  const int32_t* e_src;
  const int32_t* e_dst;
  const int32_t* e_type;
  const float* e_len;
  const float* e_inv;
};

__device__ __forceinline__ float ag_dist3(const float* __restrict__ p, int i, int j) {   // geometry.py:5-6
  const float dx = p[3 * i] - p[3 * j], dy = p[3 * i + 1] - p[3 * j + 1], dz = p[3 * i + 2] - p[3 * j + 2];
  return sqrtf(dx * dx + dy * dy + dz * dz);
}

// One workgroup per molecule, one thread per atom: the four eq_transforms (target / prediction x global / local,
// geometry.py:9-17) gathered over the atom's out- and in-edges, then the squared differences
// (dualenc.py:337-385).  d_target = (d_gt - d_perturbed) / sqrt(1 - a) * sqrt(a) per edge.
__global__ void __launch_bounds__(256) k_diffusion_loss(LossArgs a) {
  const int g = blockIdx.x;
  const int g0 = a.graph_ptr[g], n = a.graph_ptr[g + 1] - g0;
  const float al = a.alpha_graph[g];
  const float s1 = sqrtf(1.0f - al), s2 = sqrtf(al);
  for (int li = threadIdx.x; li < n; li += blockDim.x) {
    const int i = g0 + li;
    const float px = a.pos[3 * i], py = a.pos[3 * i + 1], pz = a.pos[3 * i + 2];
    float t[3] = {0.f, 0.f, 0.f}, p[3] = {0.f, 0.f, 0.f};      // target / predicted, accumulated over both roles
    float ti[3] = {0.f, 0.f, 0.f}, pi[3] = {0.f, 0.f, 0.f};
    // ---- local edges (type > 0)
    for (int e = a.loc_out_ptr[i]; e < a.loc_out_ptr[i + 1]; ++e) {
      const int j = a.loc_dst[e];
      const float dp = a.l_len[e], w = 1.0f / dp;
      const float dt = ((ag_dist3(a.pos_gt, i, j) - dp) / s1) * s2, sc = a.l_inv[e];
      const float ux = w * (px - a.pos[3 * j]), uy = w * (py - a.pos[3 * j + 1]), uz = w * (pz - a.pos[3 * j + 2]);
      t[0] += ux * dt; t[1] += uy * dt; t[2] += uz * dt;
      p[0] += ux * sc; p[1] += uy * sc; p[2] += uz * sc;
    }
    for (int k = a.loc_in_ptr[i]; k < a.loc_in_ptr[i + 1]; ++k) {
      const int e = a.loc_in_eid[k];
      const int j = a.loc_src[e];
      const float dp = a.l_len[e], w = 1.0f / dp;
      const float dt = ((ag_dist3(a.pos_gt, j, i) - dp) / s1) * s2, sc = a.l_inv[e];
      const float ux = w * (a.pos[3 * j] - px), uy = w * (a.pos[3 * j + 1] - py), uz = w * (a.pos[3 * j + 2] - pz);
      ti[0] -= ux * dt; ti[1] -= uy * dt; ti[2] -= uz * dt;
      pi[0] -= ux * sc; pi[1] -= uy * sc; pi[2] -= uz * sc;
    }
    float ll = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = (p[c] + pi[c]) - (t[c] + ti[c]);
      ll += d * d;
      t[c] = p[c] = ti[c] = pi[c] = 0.f;
    }
    ll *= 5.0f;
    // ---- global: edges with global_mask = (d_perturbed <= cutoff | local) & ~local  (dualenc.py:346-356)
    for (int q = a.out_ptr[i]; q < a.out_ptr[i + 1]; ++q) {
      const int e = a.ref2dst[q];
      const float dp = a.e_len[e];
      if (a.e_type[e] != 0 || !(dp <= a.cutoff)) continue;
      const int j = a.e_dst[e];
      const float w = 1.0f / dp;
      const float dt = ((ag_dist3(a.pos_gt, i, j) - dp) / s1) * s2, sc = a.e_inv[e];
      const float ux = w * (px - a.pos[3 * j]), uy = w * (py - a.pos[3 * j + 1]), uz = w * (pz - a.pos[3 * j + 2]);
      t[0] += ux * dt; t[1] += uy * dt; t[2] += uz * dt;
      p[0] += ux * sc; p[1] += uy * sc; p[2] += uz * sc;
    }
    for (int e = a.in_ptr[i]; e < a.in_ptr[i + 1]; ++e) {
      const float dp = a.e_len[e];
      if (a.e_type[e] != 0 || !(dp <= a.cutoff)) continue;
      const int j = a.e_src[e];
      const float w = 1.0f / dp;
      const float dt = ((ag_dist3(a.pos_gt, j, i) - dp) / s1) * s2, sc = a.e_inv[e];
      const float ux = w * (a.pos[3 * j] - px), uy = w * (a.pos[3 * j + 1] - py), uz = w * (a.pos[3 * j + 2] - pz);
      ti[0] -= ux * dt; ti[1] -= uy * dt; ti[2] -= uz * dt;
      pi[0] -= ux * sc; pi[1] -= uy * sc; pi[2] -= uz * sc;
    }
    float lg = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = (p[c] + pi[c]) - (t[c] + ti[c]);
      lg += d * d;
    }
    lg *= 2.0f;
    a.loss[i] = lg + ll;
    a.loss[a.n_total + i] = lg;
    a.loss[2 * a.n_total + i] = ll;
  }
}

}  // namespace

// Waves (= 16-node tiles) per workgroup for the LDS-sharing node kernels: about one workgroup per CU.
// Below agdiff_params_t.tune_node_ldsw_min_tiles the 4-wave streaming variants are faster (measured crossover between 275
// and 2,928 tiles: 24 vs 43 us and 75 vs 56 us per node stage).
static int64_t ag_node_ldsw_min_tiles(const agdiff_params_t* p) { return ag_tune(p->tune_node_ldsw_min_tiles, 1536); }
// Up to this many node tiles the stage runs with four waves per tile (k_schnet_node_stage_split); measured (1 molecule x
// 25 / 100 / 200 conformers): 70 vs 130, 118 vs 138, 186 vs 150 us per 7 stages
static int64_t ag_node_split_max_tiles(const agdiff_params_t* p) { return ag_tune(p->tune_node_split_max_tiles, 320); }
static int ag_node_waves_per_wg(const agdiff_params_t* p, int64_t tiles) {
  if (tiles < ag_node_ldsw_min_tiles(p)) return 4;
  int64_t w = (tiles + 255) / 256;
  if (w > 16) w = 16;
  return (int)w;
}

extern "C" int agdiff_schnet_node_stage(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                                        int32_t k, void* stream) {
  return agdiff_schnet_node_stage_split(p, topo, ws, k, 0, stream);
}

extern "C" int agdiff_schnet_node_stage_split(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                                              int32_t k, int32_t split, void* stream) {
  if (!p || !topo || !ws || k < 0 || k > p->num_convs || p->num_convs > AGDIFF_MAX_CONVS) return AGDIFF_ERR_ARG;
  if ((split & 8) && (!(split & 1) || !ws->agg_loc || !ws->agg_first_loc || !topo->lp_ptr)) return AGDIFF_ERR_ARG;
  if (topo->num_nodes <= 0) return AGDIFF_OK;
  NodeStageArgs a;
  a.finish = k > 0;
  a.prep = k < p->num_convs;
  a.prev = p->conv[k > 0 ? k - 1 : 0];
  a.next = p->conv[k < p->num_convs ? k : 0];
  a.emb = p->schnet_emb;
  a.atom_type = topo->atom_type;
  a.in_ptr = ws->in_ptr;
  a.agg = ws->agg;
  a.agg_first = ws->agg_first;
  a.h = ws->h;
  a.h_in = ws->h;
  a.xs = ws->xs;
  a.n = topo->num_nodes;
  a.chunk_edges = AG_TW * agdiff_conv_chunk_tiles(topo->max_edges);
  a.in_ptr2 = nullptr;
  a.agg2 = nullptr;
  a.agg_first2 = nullptr;
  a.chunk_edges2 = 1;
  a.range_rows = ws->range_rows;
  // Stage 0 (embedding + block 0's lin1) does not depend on the positions: split bit 2 makes stage 0 write its outputs to
  // the cache ws->h0 / ws->xs0 instead of ws->h / ws->xs, bit 1 makes stage 1 take block 0's input h from that cache.
  if ((split & 4) && k == 0 && ws->h0 && ws->xs0) {
    a.h = ws->h0;
    a.xs = ws->xs0;
  }
  if ((split & 2) && k == 1 && ws->h0) a.h_in = ws->h0;
  if (split & 1) {     // block k-1 ran as agdiff_cfconv_node: ws->agg holds one complete row per node ...
    a.in_ptr = nullptr;
    if ((split & 8) && topo->num_local > 0) {     // ... plus agdiff_cfconv_local's ws->agg_loc (local edges through the filter MLPs)
      a.in_ptr2 = topo->lp_ptr;
      a.agg2 = ws->agg_loc;
      a.agg_first2 = ws->agg_first_loc;
      a.chunk_edges2 = AG_TW * agdiff_conv_chunk_tiles(topo->num_local_padded);
    }
  }
  const int64_t tiles = (a.n + AG_TW - 1) / AG_TW;
  hipStream_t st = (hipStream_t)stream;
  if (tiles <= ag_node_split_max_tiles(p)) {       // small batch: four waves per tile (k_schnet_node_stage_split)
    ag_log_variant(ws, AGDIFF_VAR_NODE_SPLIT4);
    const dim3 grid((unsigned)tiles), block(256);
#define AG_LAUNCH_SPLIT(M)                                                                  \
    do {                                                                                    \
      if (a.finish && a.prep) k_schnet_node_stage_split<M, true, true><<<grid, block, 0, st>>>(a);        \
      else if (a.finish) k_schnet_node_stage_split<M, true, false><<<grid, block, 0, st>>>(a);            \
      else k_schnet_node_stage_split<M, false, true><<<grid, block, 0, st>>>(a);                          \
    } while (0)
    if (p->precision == AG_H3) AG_LAUNCH_SPLIT(AG_H3);
    else if (p->precision == AG_BF3) AG_LAUNCH_SPLIT(AG_BF3);
    else AG_LAUNCH_SPLIT(AG_F32);
#undef AG_LAUNCH_SPLIT
    AG_CHECK_LAUNCH();
    return AGDIFF_OK;
  }
  const int waves = ag_node_waves_per_wg(p, tiles);
  const bool ldsw = tiles >= ag_node_ldsw_min_tiles(p);
  ag_log_variant(ws, ldsw ? AGDIFF_VAR_NODE_LDSW : AGDIFF_VAR_NODE_STREAM);
  const size_t smem = ldsw ? (size_t)AG_NODE_LDS_BLOCKS * 2048 : 0;
  static std::atomic<uint64_t> attr_done{0};
  if (!ag_allow_big_lds(attr_done, (size_t)AG_NODE_LDS_BLOCKS * 2048, k_schnet_node_stage<AG_BF3, true>,
                        k_schnet_node_stage<AG_F32, true>, k_schnet_node_stage<AG_H3, true>))
    return AGDIFF_ERR_LAUNCH;
  const dim3 grid((unsigned)((tiles + waves - 1) / waves)), block(64 * waves);
  if (p->precision == AG_H3) {
    if (ldsw) k_schnet_node_stage<AG_H3, true><<<grid, block, smem, st>>>(a);
    else k_schnet_node_stage<AG_H3, false><<<grid, block, 0, st>>>(a);
  } else if (p->precision == AG_BF3) {
    if (ldsw) k_schnet_node_stage<AG_BF3, true><<<grid, block, smem, st>>>(a);
    else k_schnet_node_stage<AG_BF3, false><<<grid, block, 0, st>>>(a);
  } else {
    if (ldsw) k_schnet_node_stage<AG_F32, true><<<grid, block, smem, st>>>(a);
    else k_schnet_node_stage<AG_F32, false><<<grid, block, 0, st>>>(a);
  }
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}

extern "C" int agdiff_gin_encoder(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                                  int32_t rows_per_canonical_edge, void* stream) {
  if (!p || !topo || !ws || p->num_convs_local <= 0 || p->num_convs_local > AGDIFF_MAX_CONVS_LOCAL ||
      (topo->num_local > 0 && (!topo->loc_in_src || !topo->loc_in_row)))
    return AGDIFF_ERR_ARG;
  if (topo->num_nodes <= 0) return AGDIFF_OK;
  const int64_t tiles = (topo->num_nodes + AG_TW - 1) / AG_TW;
  const int waves = ag_node_waves_per_wg(p, tiles);
  const bool ldsw = tiles >= ag_node_ldsw_min_tiles(p);
  if (ldsw) ag_log_variant(ws, AGDIFF_VAR_GIN_LDSW);
  const size_t smem = ldsw ? (size_t)64 * 2048 : 0;
  static std::atomic<uint64_t> attr_done{0};
  if (!ag_allow_big_lds(attr_done, (size_t)64 * 2048, k_gin_layer<AG_BF3, true>, k_gin_layer<AG_F32, true>, k_gin_layer<AG_H3, true>,
                        k_gin_layer_persistent<AG_BF3>, k_gin_layer_persistent<AG_H3>))
    return AGDIFF_ERR_LAUNCH;
  int64_t wgs = (tiles + waves - 1) / waves;
  const bool persistent = ldsw && wgs > 256 && p->precision_local != AG_F32;   // (one workgroup per CU walks its tiles, weights resident)
  if (persistent) wgs = 256;
  const dim3 grid((unsigned)wgs), block(64 * waves);
  // ping-pong so that the final layer lands in ws->hl
  float* bufs[2] = {ws->hl, ws->hl2};
  int cur = (p->num_convs_local & 1) ? 0 : 1;   // layer 0 writes bufs[cur]; the last write must hit bufs[0]
  const float* in = nullptr;
  for (int k = 0; k < p->num_convs_local; ++k) {
    GinArgs a;
    a.gp = p->gin[k];
    a.emb = (k == 0) ? p->gin_emb : nullptr;
    a.atom_type = topo->atom_type;
    a.loc_in_ptr = topo->loc_in_ptr;
    a.in_src = topo->loc_in_src;
    a.in_row = rows_per_canonical_edge ? topo->loc_in_row : topo->loc_in_eid;
    a.l_attr_rows = ws->l_attr_rows;
    a.h_in = in;
    a.h_out = bufs[cur];
    a.n = topo->num_nodes;
    a.range_rows = ws->range_rows;
    hipStream_t st = (hipStream_t)stream;
    k_gin_gather<<<dim3((unsigned)((((topo->num_nodes + 7) / 8) + 7) / 8 * 8)), dim3(256), 0, st>>>(a);   // grid: multiple of 8 (XCD ranges)
    AG_CHECK_LAUNCH();
    if (p->precision_local == AG_H3) {
      if (persistent) k_gin_layer_persistent<AG_H3><<<grid, block, smem, st>>>(a);
      else if (ldsw) k_gin_layer<AG_H3, true><<<grid, block, smem, st>>>(a);
      else k_gin_layer<AG_H3, false><<<grid, block, 0, st>>>(a);
    } else if (p->precision_local == AG_BF3) {
      if (persistent) k_gin_layer_persistent<AG_BF3><<<grid, block, smem, st>>>(a);
      else if (ldsw) k_gin_layer<AG_BF3, true><<<grid, block, smem, st>>>(a);
      else k_gin_layer<AG_BF3, false><<<grid, block, 0, st>>>(a);
    } else {
      if (ldsw) k_gin_layer<AG_F32, true><<<grid, block, smem, st>>>(a);
      else k_gin_layer<AG_F32, false><<<grid, block, 0, st>>>(a);
    }
    AG_CHECK_LAUNCH();
    in = bufs[cur];
    cur ^= 1;
  }
  return AGDIFF_OK;
}

extern "C" int agdiff_langevin_update(const agdiff_topo_t* topo, const agdiff_ws_t* ws, const agdiff_step_args_t* s,
                                      void* stream) {
  if (!topo || !ws || !s || !s->pos_in || !s->pos_out || !s->noise || !s->scratch || !ws->nan_flag) return AGDIFF_ERR_ARG;
  if (topo->num_graphs <= 0) return AGDIFF_OK;
  UpdateArgs a;
  a.s = *s;
  a.graph_ptr = topo->graph_ptr;
  a.loc_src = topo->loc_src;
  a.loc_dst = topo->loc_dst;
  a.loc_out_ptr = topo->loc_out_ptr;
  a.loc_in_ptr = topo->loc_in_ptr;
  a.loc_in_eid = topo->loc_in_eid;
  a.l_len = ws->l_len;
  a.l_inv = ws->l_inv;
  a.in_ptr = ws->in_ptr;
  a.out_ptr = ws->out_ptr;
  a.ref2dst = ws->ref2dst;
  a.e_src = ws->e_src;
  a.e_dst = ws->e_dst;
  a.e_type = ws->e_type;
  a.e_len = ws->e_len;
  a.e_inv = ws->e_inv_global;
  a.nan_flag = ws->nan_flag;
  // lanes per atom: 16 when a workgroup of up to 1024 threads offers that many for the largest molecule of the batch (the
  // edge loops are chains of dependent loads: 16 lanes per atom instead of 4 for 46-atom molecules took the launch from
  // 80 to 4x us), else as many as 1024 threads offer
  const int bd_max = topo->num_graphs >= 512 ? 512 : 1024;      // many molecules: keep more workgroups resident
  int bd = 256, parts = 16;
  while (bd < bd_max && (int64_t)parts * topo->max_atoms_per_graph > bd) bd <<= 1;
  while (parts > 1 && (int64_t)parts * topo->max_atoms_per_graph > bd) parts >>= 1;
  a.parts = parts;
  k_langevin_update<<<dim3((unsigned)topo->num_graphs), dim3(bd), 0, (hipStream_t)stream>>>(a);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}

extern "C" int agdiff_perturb_positions(const agdiff_topo_t* topo, const float* pos, const float* noise,
                                        const float* alpha_graph, float* pos_out, void* stream) {
  if (!topo || !pos || !noise || !alpha_graph || !pos_out) return AGDIFF_ERR_ARG;
  if (topo->num_graphs <= 0) return AGDIFF_OK;
  PerturbArgs a{topo->graph_ptr, pos, noise, alpha_graph, pos_out};
  k_perturb_positions<<<dim3((unsigned)topo->num_graphs), dim3(256), 0, (hipStream_t)stream>>>(a);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}

extern "C" int agdiff_diffusion_loss(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                                     const float* pos_gt, const float* pos_perturbed, const float* alpha_graph,
                                     float* loss, void* stream) {
  if (!p || !topo || !ws || !pos_gt || !pos_perturbed || !alpha_graph || !loss) return AGDIFF_ERR_ARG;
  if (topo->num_graphs <= 0) return AGDIFF_OK;
  LossArgs a;
  a.graph_ptr = topo->graph_ptr;
  a.pos_gt = pos_gt;
  a.pos = pos_perturbed;
  a.alpha_graph = alpha_graph;
  a.loss = loss;
  a.n_total = topo->num_nodes;
  a.cutoff = p->cutoff;
  a.loc_src = topo->loc_src;
  a.loc_dst = topo->loc_dst;
  a.loc_out_ptr = topo->loc_out_ptr;
  a.loc_in_ptr = topo->loc_in_ptr;
  a.loc_in_eid = topo->loc_in_eid;
  a.l_len = ws->l_len;
  a.l_inv = ws->l_inv;
  a.in_ptr = ws->in_ptr;
  a.out_ptr = ws->out_ptr;
  a.ref2dst = ws->ref2dst;
  a.e_src = ws->e_src;
  a.e_dst = ws->e_dst;
  a.e_type = ws->e_type;
  a.e_len = ws->e_len;
  a.e_inv = ws->e_inv_global;
  int bd = (int)(((topo->max_atoms_per_graph + 63) / 64) * 64);
  if (bd > 256) bd = 256;
  if (bd < 64) bd = 64;
  k_diffusion_loss<<<dim3((unsigned)topo->num_graphs), dim3(bd), 0, (hipStream_t)stream>>>(a);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}
