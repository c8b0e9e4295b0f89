// Per-edge kernels: MLPEdgeEncoder, per-edge CFConv scales, fused CFConv (filter MLP + message +
// destination-segmented reduction), pair-feature heads, and the stand-alone aggregate.
// One wave = one tile of 16 edges; activations stay in the MFMA accumulator layout between layers
// (common.hpp).  Every MFMA kernel is instantiated for both arithmetic modes (AG_F32 / AG_BF3).
#include "common.hpp"
#include <cstdlib>
#include <type_traits>

namespace {

// A per-lane LDS base pointer the compiler must keep in a register: ds_read takes a 16-bit immediate offset, so a weight
// block more than 64 KiB above the wave's only address register costs one v_add_u32 per read (64 per tile in the fused
// CFConv's second layer alone).  One opaque base per 64-KiB window of resident weights removes them.
__device__ __forceinline__ const lds_u32x4* ag_lds_base(const lds_u32x4* p, int lane) {
  const lds_u32x4* b = p + lane;
  asm volatile("" : "+v"(b));
  return b;
}

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
  float* out_frag;       // optional: operand-form tiles
  float* out_rows;       // optional: fp32 rows, row e or row row_index[e] (< 0: none)
  const int32_t* row_index;
  const int32_t* pos_index;   // optional (with mir_index): edge e's results go to positions pos_index[e] and, if
  const int32_t* mir_index;   // >= 0, mir_index[e] of out_frag / row_index instead of position e
  int64_t max_tiles;
  const int32_t* tile_flags;  // optional: [0] = number of tiles to do (0: the launch returns at once), [1 + tile] = 16-bit mask of
                              // the tile's rows to evaluate and store (0: skip the tile)
};

// Where one edge's results go (shared by the encoder kernels).
struct AgEdgeOut {
  int64_t p, pm;      // output positions (pm < 0: none)
};
__device__ __forceinline__ AgEdgeOut ag_edge_out(const int32_t* pos_index, const int32_t* mir_index, int64_t e, bool valid) {
  if (!pos_index) return AgEdgeOut{valid ? e : (int64_t)-1, -1};
  return AgEdgeOut{valid ? (int64_t)pos_index[e] : (int64_t)-1, valid ? (int64_t)mir_index[e] : (int64_t)-1};
}
template <int MODE>
__device__ __forceinline__ void ag_emit_edge_attr(const f32x4 (&y)[8], const AgEdgeOut& o, float* out_frag, float* out_rows,
                                                  const int32_t* row_index, bool scattered, int64_t tile, int lane) {
  const int q = lane >> 4;
  if (out_rows) {
    if (o.p >= 0) {
      const int64_t row = row_index ? (int64_t)row_index[o.p] : o.p;
      if (row >= 0) ag_store_row<8, 0>(y, out_rows + (size_t)row * 128, q);
    }
    if (o.pm >= 0) {       // (through row_index a mirror pair may share one row: written once)
      const int64_t row = row_index ? (int64_t)row_index[o.pm] : o.pm;
      const int64_t rowp = (row_index && o.p >= 0) ? (int64_t)row_index[o.p] : -1;
      if (row >= 0 && row != rowp) ag_store_row<8, 0>(y, out_rows + (size_t)row * 128, q);
    }
  }
  if (out_frag) {
    AgIn<MODE> x[4];
    ag_cvt_tiles<MODE, 4, 0>(y, x);
    if (!scattered) {
#pragma unroll
      for (int t = 0; t < 4; ++t) ag_store_attr(x[t], out_frag, tile, t, lane);
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (o.p >= 0) ag_store_attr(x[t], out_frag, o.p >> 4, t, q * 16 + (int)(o.p & 15));
        if (o.pm >= 0) ag_store_attr(x[t], out_frag, o.pm >> 4, t, q * 16 + (int)(o.pm & 15));
      }
    }
  }
}

// encoder/edge.py:84-103.  x0 = gelu(w*d+b); h1 = gelu(W1a x0 + T1[type]); h2 = gelu(W23 h1 + T3[type]);
// a = W4 h2 + b4.  (T1/T3: per-edge-type tables holding the bond_emb halves of the two 256->128
// layers; W23 = comb.0[:, :128] @ efm.2; the trailing attention factor is exactly 1.)
// The result is stored in the operand form of the consuming mode (common.hpp: edge-attr storage).
#define AG_PERSIST_WAVES 16
// Persistent launch, one 16-wave workgroup per CU: w1 and w23 (64 KiB each) and the first unit of every w4
// block (32 KiB) stay in LDS for the whole launch; only w4's second units (32 KiB per tile) stream from L2.
// (Streaming all three matrices made the kernel L2-bandwidth-bound: 192 KiB per 16-edge tile.)
template <int MODE>
__global__ void __launch_bounds__(64 * AG_PERSIST_WAVES, 4) k_edge_encoder(EncArgs a) {
  extern __shared__ u32x4 ag_enc_smem[];
  lds_u32x4* lw1 = (lds_u32x4*)ag_enc_smem;
  lds_u32x4* lw23 = lw1 + 32 * 128;
  lds_u32x4* lw4 = lw23 + 32 * 128;           // 32 blocks x 64 (unit 0 only)
  if (a.tile_flags && a.tile_flags[0] == 0) return;      // nothing flagged this step (uniform: before the fill)
  {
    const u32x4* g1 = reinterpret_cast<const u32x4*>(a.w1_pk);
    const u32x4* g2 = reinterpret_cast<const u32x4*>(a.w23_pk);
    const u32x4* g4 = reinterpret_cast<const u32x4*>(a.w4_pk);
    ag_copy_lds(lw1, g1, 32 * 128);
    ag_copy_lds(lw23, g2, 32 * 128);
    ag_copy_lds_map(lw4, g4, 32 * 64, [](int i) { return (i >> 6) * 128 + (i & 63); });
  }
  __syncthreads();
  const int lane0 = ag_lane();
  const int E = *a.n_dev;
  const int64_t stride = (int64_t)gridDim.x * AG_PERSIST_WAVES;
  for (int64_t tile = (int64_t)blockIdx.x * AG_PERSIST_WAVES + (threadIdx.x >> 6); tile < a.max_tiles; tile += stride) {
    if (tile * AG_TW >= E) break;
    const int row_mask = a.tile_flags ? a.tile_flags[1 + tile] : 0xFFFF;
    if (row_mask == 0) continue;
    int lane = lane0;
    asm volatile("" : "+v"(lane));      // keep lane-derived addresses out of the loop-invariant set
    const int q = lane >> 4;
    const int64_t e = tile * AG_TW + (lane & 15);
    // (flagged mode: only the flagged rows are stored -- what an edge's row holds depends on that edge alone, never on
    // which other edges share its tile)
    const bool valid = e < E && ((row_mask >> (lane & 15)) & 1);
    const float d = valid ? a.e_len[e] : 0.0f;
    const int ty = valid ? a.e_type[e] : 0;

    f32x4 y[8];
    AgIn<MODE> x[4];
    {
      f32x4 w[8];
      ag_init_vec<8>(w, a.fe_w, q);
      ag_init_vec<8>(y, a.fe_b, q);
      AG_FOR_TILE(y, 8, ag_gelu(fmaf(w[_t][_r], d, v)));
    }
    ag_cvt_tiles<MODE, 4, 0>(y, x);
    ag_init_vec<8>(y, a.t1 + (size_t)ty * 128, q);
    ag_dense_lds<MODE, false, false, 4, 8, 0, 0>(x, y, ag_lds_base(lw1, lane), 0);
    AG_FOR_TILE(y, 8, ag_gelu(v));
    ag_cvt_tiles<MODE, 4, 0>(y, x);
    ag_init_vec<8>(y, a.t3 + (size_t)ty * 128, q);
    ag_dense_lds<MODE, false, false, 4, 8, 0, 0>(x, y, ag_lds_base(lw23, lane), 0);
    AG_FOR_TILE(y, 8, ag_gelu(v));
    ag_cvt_tiles<MODE, 4, 0>(y, x);
    ag_init_vec<8>(y, a.b4, q);
    ag_dense_split<MODE, false, false, 4, 8, 0, 0, 6>(x, y, ag_lds_base(lw4, lane), a.w4_pk, lane, 0);
    ag_emit_edge_attr<MODE>(y, ag_edge_out(a.pos_index, a.mir_index, e, valid), a.out_frag, a.out_rows, a.row_index,
                            a.pos_index != nullptr, tile, lane);
  }
}

// ------------------------------------------------------------------------------ gaussian edge encoder
// encoder/edge.py:17-42 + GaussianSmearing (schnet.py:18-27): a[e] = [exp(coeff (d - offset_k)^2), k < 64 |
// bond_emb[type] (64)].  No GEMM: one 16-edge tile per wave, written straight in operand form.
struct GaussArgs {
  const float* offset;
  const float* emb;
  const int32_t* n_dev;
  const float* e_len;
  const int32_t* e_type;
  float* out_frag;
  float* out_rows;
  const int32_t* row_index;
  const int32_t* pos_index;
  const int32_t* mir_index;
  int64_t max_tiles;
  float coeff_log2e;
};

template <int MODE>
__global__ void __launch_bounds__(256) k_edge_gaussian(GaussArgs a) {
  const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int E = *a.n_dev;
  if (tile >= a.max_tiles || tile * AG_TW >= E) return;
  const int lane = ag_lane();
  const int q = lane >> 4;
  const int64_t e = tile * AG_TW + (lane & 15);
  const bool valid = e < E;
  const float d = valid ? a.e_len[e] : 0.0f;
  const int ty = valid ? a.e_type[e] : 0;
  f32x4 y[8];
  ag_load_row<4, 0>(y, a.offset, q);
  AG_FOR_TILE(y, 4, ag_exp2(a.coeff_log2e * ((d - v) * (d - v))));
  ag_load_row<4, 4>(y, a.emb + (size_t)ty * 64, q);
  ag_emit_edge_attr<MODE>(y, ag_edge_out(a.pos_index, a.mir_index, e, valid), a.out_frag, a.out_rows, a.row_index,
                          a.pos_index != nullptr, tile, lane);
}

// ------------------------------------------------------------------------------ per-edge conv scales
struct ScaleArgs {
  const float* dw[2 * AGDIFF_MAX_CONVS];
  const int32_t* n_dev;
  const float* e_len;
  const int32_t* pos_index;    // optional (with mir_index): entry e is the canonical edge of positions pos_index[e] and,
  const int32_t* mir_index;    // when >= 0, mir_index[e] (a mirror pair has one length, hence one scale)
  const int32_t* e_type;       // optional (with type_slot): entries whose type has a polynomial slot get scale 0 -- the MLP
  const int32_t* type_slot;    // pass of a mixed batch must not count them a second time
  float* out;
  int64_t epad;
  int32_t n;
  float cutoff;
  int32_t smooth;
};

// lw(d) * C(d) for the 2*num_convs CFConvs (schnet.py:138-149), once per step instead of once per block launch: one
// thread per edge (or mirror pair) evaluates ALL the convs -- the length (and the pair's two positions) are read once, the
// envelope is evaluated once, and in per-edge mode every conv's store is coalesced across the workgroup (one thread
// per (edge, conv) with scattered per-pair stores was bound by exactly those stores).  The segment tables of all convs
// (n x 100 floats) sit in LDS.
__global__ void __launch_bounds__(256) k_edge_scales(ScaleArgs a) {
  __shared__ float seg[2 * AGDIFF_MAX_CONVS * 100];
  for (int i = threadIdx.x; i < a.n * 100; i += blockDim.x) seg[i] = a.dw[i / 100][i % 100];
  __syncthreads();
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= *a.n_dev) return;
  const float d = a.e_len[e];
  float C = cf_envelope(d, a.cutoff, a.smooth);
  if (a.e_type && a.type_slot[a.e_type[e]] >= 0) C = 0.0f;
  const int64_t p0 = a.pos_index ? (int64_t)a.pos_index[e] : e;
  const int64_t p1 = a.pos_index ? (int64_t)a.mir_index[e] : -1;
  for (int c = 0; c < a.n; ++c) {
    const float s = cf_dist_weight(seg + c * 100, d) * C;
    float* out = a.out + (size_t)c * a.epad;
    out[p0] = s;
    if (p1 >= 0) out[p1] = s;
  }
}

// ------------------------------------------------------------------------------ fused CFConv
struct ConvArgs {
  agdiff_conv_params_t cp;
  const int32_t* n_dev;
  const int32_t* in_ptr;
  const int32_t* e_src;
  const int32_t* e_dst;
  const float* scale1;   // [E] lw(d)*C(d) of conv1 of this block
  const float* scale2;   // [E] ... conv2
  const float* e_attr;
  const float* xs;       // [N][192]
  float* agg;            // [N][192]
  float* agg_first;      // [chunks][192]
  int64_t max_chunks;
  int32_t chunk_tiles;   // tiles per chunk (agdiff_conv_chunk_tiles)
  int32_t ablate;        // timing experiments only (AGDIFF_ABLATE env): bit0 skip layer 1, bit1 skip ssp,
                         // bit2 skip layer 2, bit3 skip x gather, bit4 skip reduction, bit6 skip the next tile's e_attr
                         // loads, bit7 skip the first layer's LDS weight reads
};

#ifdef AG_CONV_STAMPS
// Diagnostic build only (make EXTRA=-DAG_CONV_STAMPS): per-phase wave cycles of k_cfconv_fused, summed over waves.
__device__ unsigned long long ag_conv_stamp_acc[8];
#define AG_STAMP(var)                                  \
  do {                                                 \
    __builtin_amdgcn_sched_barrier(0);                 \
    var = __builtin_readcyclecounter();                \
    __builtin_amdgcn_sched_barrier(0);                 \
  } while (0)
#else
#define AG_STAMP(var) do { } while (0)
#endif

// Timing experiments (AGDIFF_ABLATE env) exist only in the diagnostic build (make EXTRA=-DAG_CONV_ABLATE): in the
// product build the tile body has no uniform branches around its phases, which also gives the scheduler one region.
#ifdef AG_CONV_ABLATE
#define AG_ABL(bit) (a.ablate & (bit))
#else
#define AG_ABL(bit) false
#endif

#ifndef AG_CONV_WAVES
#define AG_CONV_WAVES 8     // waves per workgroup (= per CU): 2 per SIMD
#endif
#define AG_CONV_LDS_BLOCKS 80   // resident 2-KiB weight blocks: filt_w1 (48: both convs' first layer) | filt_w2a (32)
#define AG_CONV_NCH 12          // 16-channel tiles of the 192 filter channels (conv1: 0..7, conv2: 8..11)

// encoder/schnet.py:136-162 for conv1 (F=128) and conv2 (F=64) of one InteractionBlock:
//   W_e = nn(edge_attr_e) * (lw(d_e) * C(d_e));  agg[dst] += x[src] * W_e   (aggr='add')
// Persistent launch, one 8-wave workgroup per CU (2 waves per SIMD, ~240 VGPRs: the tile body is a software
// pipeline).  Each wave walks chunk_tiles (1..8, fewer for small batches) consecutive
// destination-sorted 16-edge tiles per chunk and keeps the running sum of the open target in registers; a target
// whose list started in an earlier chunk is written to agg_first[chunk] and added by the node stage (fixed
// order -> bitwise reproducible, no atomics).
template <int MODE>
__global__ void __launch_bounds__(64 * AG_CONV_WAVES, AG_CONV_WAVES / 4) k_cfconv_fused(ConvArgs a) {
  // LDS: all 160 KiB hold filter weights for the whole launch -- the fused first layer of both convs (96 KiB)
  // and conv1's second layer (64 KiB).  Only conv2's second layer (8 blocks = 16 KiB per tile) is streamed
  // from L2, each pair of blocks requested one channel tile ahead of its use.
  extern __shared__ u32x4 ag_conv_smem[];
  lds_u32x4* w1 = (lds_u32x4*)ag_conv_smem;
  lds_u32x4* w2a = w1 + 48 * 128;
  {
    const u32x4* g1 = reinterpret_cast<const u32x4*>(a.cp.filt_w1_pk);
    const u32x4* ga = reinterpret_cast<const u32x4*>(a.cp.filt_w2a_pk);
    ag_copy_lds(w1, g1, 48 * 128);
    ag_copy_lds(w2a, ga, 32 * 128);
  }
  __syncthreads();
  const int lane0 = ag_lane();
  const int wave = threadIdx.x >> 6;
  const int E = *a.n_dev;
  const int64_t cstride = (int64_t)gridDim.x * AG_CONV_WAVES;
  // Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8); give each XCD a contiguous range of the
  // destination-sorted edge list per round, so that the x rows and list bounds of a molecule stay in ONE L2.
  const int wg = (gridDim.x % 8 == 0) ? (int)((blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8) : (int)blockIdx.x;
  [[maybe_unused]] unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c0 = 0, c1 = 0;

  // edge attributes of the wave's NEXT tile are requested as soon as the current tile's first layer has consumed
  // its own (they land during the rest of the tile)
  AgIn<MODE> ea[4];
  // first-layer bias of the next pair of output tiles (see the first-layer loop)
  f32x4 nb0 = ag_ld4(a.cp.filt_b1 + 4 * (lane0 >> 4)), nb1 = ag_ld4(a.cp.filt_b1 + 16 + 4 * (lane0 >> 4));
  // ... and so are its per-edge scalars (source, the two conv scales) and its first / last target
  int pf_src = 0, pf_t0 = 0, pf_t1 = 0;
  float pf_s1 = 0.0f, pf_s2 = 0.0f;
  auto prefetch_meta = [&](int64_t tl, int ln) {
    const int64_t tb = tl * AG_TW, e = tb + (ln & 15);
    const bool valid = e < E;
    pf_src = valid ? a.e_src[e] : 0;
    pf_s1 = valid ? a.scale1[e] : 0.0f;
    pf_s2 = valid ? a.scale2[e] : 0.0f;
    const int64_t last = (tb + AG_TW - 1 < E) ? tb + AG_TW - 1 : (int64_t)E - 1;
    pf_t0 = a.e_dst[tb];
    pf_t1 = a.e_dst[last];
  };
  {
    const int64_t first = ((int64_t)wg * AG_CONV_WAVES + wave) * a.chunk_tiles;
    if (first * AG_TW < E) {
#pragma unroll
      for (int t = 0; t < 4; ++t) ag_load_attr(ea[t], a.e_attr, first, t, lane0);
      prefetch_meta(first, lane0);
    }
  }
  for (int64_t chunk = (int64_t)wg * AG_CONV_WAVES + wave; chunk < a.max_chunks; chunk += cstride) {
    const int64_t e_begin = chunk * (AG_TW * a.chunk_tiles);
    if (e_begin >= E) break;
    int run_t = -1;
    // running sums of target run_t, whose list is still open: 192 channels spread over the wave, entry g of
    // lane (col, q) = channel 16 (4 g + q) + col; zero while no list is open
    float carry[AG_CONV_NCH / 4];
#pragma unroll
    for (int i = 0; i < AG_CONV_NCH / 4; ++i) carry[i] = 0.0f;

    auto dest = [&](int t) -> float* {
      const int lo = a.in_ptr[t];
      return (lo >= e_begin) ? (a.agg + (size_t)t * 192) : (a.agg_first + (size_t)chunk * 192);
    };

    for (int tt = 0; tt < a.chunk_tiles; ++tt) {
      const int64_t tile = chunk * a.chunk_tiles + tt;
      const int64_t tbase = tile * AG_TW;
      if (tbase >= E) break;
      AG_STAMP(c0);
      // opaque copy of the lane id: keeps hipcc from hoisting every lane-derived weight / table address
      // out of the tile loop (they would stay live across the whole body and spill)
      int lane = lane0;
      asm volatile("" : "+v"(lane));
      const int q = lane >> 4, col = lane & 15;
      const int my_src = pf_src;
      const float s1 = pf_s1, s2 = pf_s2;
      const int t0 = __builtin_amdgcn_readfirstlane(pf_t0);
      const int t1 = __builtin_amdgcn_readfirstlane(pf_t1);
      if (run_t >= 0 && run_t != t0) {   // previous tile ended exactly on a list boundary
        float* dp = dest(run_t);
#pragma unroll
        for (int i = 0; i < AG_CONV_NCH / 4; ++i) {
          dp[16 * (4 * i + q) + col] = carry[i];
          carry[i] = 0.0f;
        }
        run_t = -1;
      }
      AG_STAMP(c1); st[0] += c1 - c0; c0 = c1;       // meta loads, carry flush

      // per edge slot of my quarter (slot 4q + r lives in lane 4q + r): gather row of x, and the scales
      // lw(d)*C(d) of the two convs, which multiply the message: (H^T W2 + b2) . s . x[src]
      uint32_t xoff[4];
      f32x4 sr;                          // scale of the conv being processed, per edge slot
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xoff[r] = ((uint32_t)__shfl(my_src, 4 * q + r) * 192u + (uint32_t)col) * 4u;     // byte offset into xs
        sr[r] = __shfl(s1, 4 * q + r);
      }
      // list boundaries of the targets present in this tile, one per lane (in_ptr[t0 + lane]); the
      // reduction loops read them with readlane instead of dependent global loads
      const int ntg = t1 - t0 + 1;
      const int ipl = a.in_ptr[t0 + (lane <= ntg ? lane : ntg)];
      auto bound = [&](int i) -> int {   // in_ptr[t0 + i], i <= ntg
        return (i < 64) ? __builtin_amdgcn_readlane(ipl, i) : __builtin_amdgcn_readfirstlane(a.in_ptr[t0 + i]);
      };
      auto dest_lo = [&](int t, int lo) -> float* {
        return (lo >= e_begin) ? (a.agg + (size_t)t * 192) : (a.agg_first + (size_t)chunk * 192);
      };

      // second filter layer per 16-channel tile, flipped (rows = edges, lanes = channels), then message and
      // destination-segmented reduction of that channel tile.  The x[src] rows of the next channel tile are
      // fetched before the current tile's reduction.
      // x[src] rows (and the second-layer bias) of channel tile nt are requested AG_X_AHEAD channel tiles before their
      // use through a small register ring: one channel tile (12 MFMAs) does not cover an L2 round trip under load
#ifndef AG_X_AHEAD
#define AG_X_AHEAD 2
#endif
      constexpr int XA = AG_X_AHEAD, XR = XA + 1;
      f32x4 xring[XR];
      float bring[XR];
      auto fetch_x = [&](int nt) {
        if (AG_ABL(8)) return;
        // base (SGPR pair) + 32-bit byte offset (VGPR) + immediate 64 nt: one instruction per gathered value
        const char* xb = reinterpret_cast<const char*>(a.xs);
#pragma unroll
        for (int r = 0; r < 4; ++r) xring[nt % XR][r] = *reinterpret_cast<const float*>(xb + (size_t)xoff[r] + 64 * nt);
        bring[nt % XR] = a.cp.filt_b2[16 * nt + col];
      };
#pragma unroll
      for (int nt = 0; nt < XA; ++nt) fetch_x(nt);      // requested here, before the first layer: land while its MFMAs run
      // the wave's next tile: edge attributes (HBM) + per-edge scalars.  Loads return in issue order (vmcnt), so every
      // x gather issued after this request waits for it too.  Issuing it later (inside the channel-tile loop, at tile 3 /
      // 6 / 9) was measured: 0.64 / 0.49 / 0.49 ms per launch against 0.476 here.
      auto prefetch_next = [&]() {
        int64_t nxt = tile + 1;
        if (tt + 1 >= a.chunk_tiles) nxt = (chunk + cstride) * a.chunk_tiles;
        if (nxt * AG_TW < E && nxt < a.max_chunks * a.chunk_tiles) {
          if (!(AG_ABL(64))) {
#pragma unroll
            for (int t = 0; t < 4; ++t) ag_load_attr(ea[t], a.e_attr, nxt, t, lane);
          }
          prefetch_meta(nxt, lane);
        }
      };
      AgIn<MODE> hidb[6];
      {
        // First filter layer of both convs (128 -> 192), all 48 weight blocks from LDS (block (t, ot) at
        // t * 12 + ot), two output tiles (= one k-tile of the second layer) at a time, so that the softplus
        // and the operand split of one pair can issue between the MFMAs of the next.
        // weight blocks are read from LDS one step (two blocks) ahead of the MFMAs that use them
#ifndef AG_L1_LDS_AHEAD
#define AG_L1_LDS_AHEAD 1       // steps (of two blocks) the LDS weight reads run ahead of their MFMAs
#endif
        constexpr int WA = AG_L1_LDS_AHEAD, WR = WA + 1;
        u32x4 wq[WR][2][2];
        const lds_u32x4* w1_lo = ag_lds_base(w1, lane);
        const lds_u32x4* w1_hi = ag_lds_base(w1 + 32 * 128, lane);
        auto fetch_w = [&](u32x4 (&dst)[2][2], int step) {     // step = m * 4 + t
          const int m = step >> 2, t = step & 3;
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int bi = t * AG_CONV_NCH + 2 * m + b;        // blocks 0..31 from the first 64-KiB window, 32..47 from the second
            const lds_u32x4* wb = (bi < 32) ? w1_lo + bi * 128 : w1_hi + (bi - 32) * 128;
            dst[b][0] = wb[0];
            dst[b][1] = wb[64];
          }
        };
#pragma unroll
        for (int s0 = 0; s0 < WA; ++s0) fetch_w(wq[s0 % WR], s0);
        // Software pipeline, fenced per step: the six MFMAs of pair m's step t run beside a quarter of pair
        // m-1's softplus + operand split (two of its eight values per lane).
        f32x4 hp0 = {0.f, 0.f, 0.f, 0.f}, hp1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m <= AG_CONV_NCH / 2; ++m) {
          f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
          if (m < AG_CONV_NCH / 2) {
            // The accumulators start from the bias, which the first MFMA of the pair needs at once: it is requested a
            // whole pair ahead (nb0 / nb1; the last pair requests pair 0's for the wave's next tile).  Loaded where it is
            // used, every pair began with an exposed L2 round trip -- six per tile.
            h0 = nb0;
            h1 = nb1;
            const int mn = (m + 1) % (AG_CONV_NCH / 2);
            nb0 = ag_ld4(a.cp.filt_b1 + 32 * mn + 4 * q);
            nb1 = ag_ld4(a.cp.filt_b1 + 32 * mn + 16 + 4 * q);
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int step = m * 4 + t;
            if (m < AG_CONV_NCH / 2) {
              if (step + WA < 4 * (AG_CONV_NCH / 2) && !(AG_ABL(128))) fetch_w(wq[(step + WA) % WR], step + WA);
              if (!(AG_ABL(1))) {
                ag_block_mma<MODE, false>(h0, ea[t], wq[step % WR][0]);
                ag_block_mma<MODE, false>(h1, ea[t], wq[step % WR][1]);
              }
            }
            if (m > 0) {
              float v0 = (t < 2) ? hp0[2 * t] : hp1[2 * t - 4], v1 = (t < 2) ? hp0[2 * t + 1] : hp1[2 * t - 3];
              if (!(AG_ABL(2))) { v0 = ag_ssp_base2(v0); v1 = ag_ssp_base2(v1); }
              asm volatile("" : "+v"(v0), "+v"(v1));      // keep the softplus here (the IR sinks it to its use otherwise)
              ag_cvt_pair(hidb[m - 1], 2 * t, v0, v1);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          hp0 = h0; hp1 = h1;
        }
        prefetch_next();
      }
      AG_STAMP(c1); st[1] += c1 - c0; c0 = c1;       // layer 1 + softplus + split (one pipeline)
      // Row masks of the first two targets of the tile, built once per tile: almost every 16-edge tile holds
      // the in-lists of one or two targets (in-degree >= 8), so the per-channel-tile reduction is 8 FMAs and two
      // quarter sums; tiles with more targets take the general loop.
      f32x4 m0, m1;
      {
        const int b0 = bound(0), b1 = bound(1), b2 = bound(ntg >= 2 ? 2 : 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int er = (int)tbase + 4 * q + r;
          m0[r] = (er >= b0 && er < b1) ? 1.0f : 0.0f;
          m1[r] = (ntg >= 2 && er >= b1 && er < b2) ? 1.0f : 0.0f;
        }
      }
      float* const dp0 = dest_lo(t0, bound(0));
      const bool fast = ntg <= 2 && !(AG_ABL(32));
      const bool two = ntg == 2;
      // Fast reduction (one or two targets in the tile), free of branches so that it can be issued in the shadow
      // of the next channel tile's MFMAs: both masked sums are always formed; the first target's running sum is
      // always stored (when its list goes on, a later tile or the flush repeats the store with the final value);
      // the open list's sum stays in carry[].  carry[] is zero whenever no list is open, so adding it needs no
      // condition.  The sums over the four quarters are taken four channel tiles at a time (reduce-scatter).
      // General reduction (three or more targets in the tile): one masked sum per target, replicated over the
      // quarters; quarter nt & 3 keeps / stores it (the fast path's distribution).
      auto reduce_general = [&](f32x4 z, int nt, float& cr) {
        const bool mine = q == (nt & 3);
        float newcarry = 0.0f;
        for (int i = 0; i < ntg; ++i) {
          const int lo = bound(i), hi = bound(i + 1);
          float p = 0.0f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int er = (int)tbase + 4 * q + r;
            p += ((er >= lo) && (er < hi)) ? z[r] : 0.0f;
          }
          p = ag_quarter_sum(p);
          if (i == 0) p = cr + p;
          if (i < ntg - 1) {
            float* dp = dest_lo(t0 + i, lo);
            if (mine) dp[16 * nt + col] = p;
          } else {
            newcarry = p;
          }
        }
        cr = mine ? newcarry : cr;
      };
      // conv2's second-layer blocks (pk [4][2]: one pair per channel tile) stream from L2; the pair of the
      // next channel tile is requested right after the current pair's MFMAs
      const u32x4* gl = reinterpret_cast<const u32x4*>(a.cp.filt_w2b_pk) + lane;
      u32x4 g[2][2];
      auto fetch_g = [&](int pair) {   // blocks 2*pair, 2*pair+1
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          g[b][0] = gl[((2 * pair + b) * 2) * 64];
          g[b][1] = gl[((2 * pair + b) * 2 + 1) * 64];
        }
      };
      AG_STAMP(c1); st[2] += c1 - c0; c0 = c1;       // bounds, masks
      // second-layer MFMAs of channel tile nt (flipped: rows = edges, lanes = channels), raw accumulators
      const lds_u32x4* w2a_l = ag_lds_base(w2a, lane);
      // conv1's 32 second-layer blocks are read from LDS AG_L2_LDS_AHEAD blocks ahead of their MFMAs through a small
      // register ring (left to the compiler, every block's two reads were issued and waited for on the spot: 32
      // exposed LDS round trips per tile)
#ifndef AG_L2_LDS_AHEAD
#define AG_L2_LDS_AHEAD 1
#endif
      constexpr int LA = AG_L2_LDS_AHEAD, LR = LA + 1;
      u32x4 w2q[LR][2];
      auto fetch_w2 = [&](int b) {             // b = nt * 4 + k
        w2q[b % LR][0] = w2a_l[(b * 2) * 64];
        w2q[b % LR][1] = w2a_l[(b * 2 + 1) * 64];
      };
#pragma unroll
      for (int b = 0; b < LA; ++b) fetch_w2(b);
      auto dense2 = [&](int nt) -> f32x4 {
        f32x4 z[1] = {{0.f, 0.f, 0.f, 0.f}};
        if (nt == 7) fetch_g(0);
        if (!(AG_ABL(4))) {
          if (nt < 8) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int b = nt * 4 + k;
              if (b + LA < 32) fetch_w2(b + LA);
              ag_block_mma<MODE, true>(z[0], hidb[k], w2q[b % LR]);
            }
          } else {
            ag_block_mma<MODE, true>(z[0], hidb[4], g[0]);
            ag_block_mma<MODE, true>(z[0], hidb[5], g[1]);
            if (nt + 1 < AG_CONV_NCH) fetch_g(nt + 1 - 8);
          }
        }
        return z[0];
      };
      // message factors of channel tile nt: bias (per channel = per lane) and scale . x[src] per edge row
      // message factors of channel tile nt: bias (per channel = per lane) and x[src] per edge row (times the
      // conv's per-edge scale unless the caller folds that into its row masks)
      auto factors = [&](int nt, float& bb, f32x4& m, bool with_scale) {
        if (nt == 8) {
#pragma unroll
          for (int r = 0; r < 4; ++r) sr[r] = __shfl(s2, 4 * q + r);
        }
        bb = bring[nt % XR];
        m = with_scale ? sr * xring[nt % XR] : xring[nt % XR];
        if (nt + XA < AG_CONV_NCH) fetch_x(nt + XA);
      };
      // software pipeline, fenced per channel tile: the MFMAs of tile nt beside message + reduction of tile nt-1.
      // TWO = the tile holds two targets; with one, the second masked sum, its quarter sums and the store are
      // not needed at all (the sum stays in carry[] until the list ends).
      auto run_fast = [&](auto TWO) {
        constexpr bool kTwo = decltype(TWO)::value;
        f32x4 zp = {0.f, 0.f, 0.f, 0.f}, mp = {0.f, 0.f, 0.f, 0.f};
        float bp = 0.0f;
        float p0[4], p1[4] = {0.f, 0.f, 0.f, 0.f};
        f32x4 w0 = m0 * sr, w1 = m1 * sr;          // conv1's scale folded into the row masks
#pragma unroll
        for (int nt = 0; nt <= AG_CONV_NCH; ++nt) {
          f32x4 z = {0.f, 0.f, 0.f, 0.f}, m = {0.f, 0.f, 0.f, 0.f};
          float bb = 0.0f;
          if (nt < AG_CONV_NCH) {
            z = dense2(nt);
            factors(nt, bb, m, false);
          }
          if (nt == 9) { w0 = m0 * sr; w1 = m1 * sr; }   // tile 8 (reduced in this step) starts conv2: sr is its scale now
          if (nt > 0) {
            const int j = (nt - 1) & 3, g4 = (nt - 1) >> 2;
            f32x4 t;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = (zp[r] + bp) * mp[r];
            if (AG_ABL(16)) {
              p0[j] = t[0];
            } else {
              p0[j] = t[0] * w0[0];
#pragma unroll
              for (int r = 1; r < 4; ++r) p0[j] = fmaf(t[r], w0[r], p0[j]);
              if constexpr (kTwo) {
                p1[j] = t[0] * w1[0];
#pragma unroll
                for (int r = 1; r < 4; ++r) p1[j] = fmaf(t[r], w1[r], p1[j]);
              }
            }
            if (j == 3) {      // quarter q ends up with the sums of channel tile 4 g4 + q
              const float r0 = carry[g4] + ag_quarter_reduce_scatter4(p0[0], p0[1], p0[2], p0[3]);
              if constexpr (kTwo) {
                dp0[16 * (4 * g4 + q) + col] = r0;
                carry[g4] = ag_quarter_reduce_scatter4(p1[0], p1[1], p1[2], p1[3]);
              } else {
                carry[g4] = r0;
              }
            }
          }
          // (no fence here: experiment)
          zp = z; mp = m; bp = bb;
        }
      };
      if (fast) {
        if (two) run_fast(std::true_type{});
        else run_fast(std::false_type{});
      } else {
#pragma unroll
        for (int nt = 0; nt < AG_CONV_NCH; ++nt) {
          f32x4 z = dense2(nt), m;
          float bb;
          factors(nt, bb, m, true);
#pragma unroll
          for (int r = 0; r < 4; ++r) z[r] = (z[r] + bb) * m[r];
          reduce_general(z, nt, carry[nt >> 2]);
        }
      }
      AG_STAMP(c1); st[3] += c1 - c0; c0 = c1;       // layer 2 + message + reduction (one pipeline)
      run_t = t1;
    }
    if (run_t >= 0) {
      float* dp = dest(run_t);
#pragma unroll
      for (int i = 0; i < AG_CONV_NCH / 4; ++i) dp[16 * (4 * i + (lane0 >> 4)) + (lane0 & 15)] = carry[i];
    }
  }  // chunk loop
#ifdef AG_CONV_STAMPS
  if (lane0 == 0) {
    for (int i = 0; i < 4; ++i) atomicAdd(&ag_conv_stamp_acc[i], st[i]);
    atomicAdd(&ag_conv_stamp_acc[7], 1ull);
  }
#endif
}

// ------------------------------------------------------------------------------ pair head
struct HeadArgs {
  agdiff_head_params_t hp;
  const int32_t* n_dev;
  const int32_t* src;
  const int32_t* dst;
  const float* node_h;   // [N][128]
  const float* attr_frag;   // operand-form edge_attr tiles, or
  const float* attr_rows;   // fp32 rows [E][128] (exactly one of the two)
  const int32_t* pos_index; // optional (with mir_index): edge e is position pos_index[e] of attr_frag / out, and its
  const int32_t* mir_index; // result is also written to position mir_index[e] when that is >= 0
  float* out;            // [E]
  int64_t max_tiles;
};

// assemble_atom_pair_feature (common.py:106-109) + MultiLayerPerceptron 256->128->64->1 (common.py:86-103)
// Persistent launch, one 16-wave workgroup per CU; both weight matrices (128 + 32 KiB) stay in LDS.
template <int MODE>
__global__ void __launch_bounds__(64 * AG_PERSIST_WAVES, 4) k_pair_head(HeadArgs a) {
  extern __shared__ u32x4 ag_head_smem[];
  lds_u32x4* lw1 = (lds_u32x4*)ag_head_smem;      // pkk [8][8] = 64 blocks
  lds_u32x4* lw2 = lw1 + 64 * 128;                // pk  [4][4] = 16 blocks
  {
    const u32x4* g1 = reinterpret_cast<const u32x4*>(a.hp.w1_pk);
    const u32x4* g2 = reinterpret_cast<const u32x4*>(a.hp.w2_pk);
    ag_copy_lds(lw1, g1, 64 * 128);
    ag_copy_lds(lw2, g2, 16 * 128);
  }
  __syncthreads();
  const int lane0 = ag_lane();
  const int E = *a.n_dev;
  const int64_t stride = (int64_t)gridDim.x * AG_PERSIST_WAVES;
  // (workgroups are dealt round-robin to the 8 XCDs: a contiguous tile range per XCD and round keeps a molecule's h rows,
  // which all of its edges gather, in ONE L2)
  const int64_t wg = (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;
  for (int64_t tile = wg * AG_PERSIST_WAVES + (threadIdx.x >> 6); tile < a.max_tiles; tile += stride) {
    if (tile * AG_TW >= E) break;
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int q = lane >> 4;
    const int64_t e = tile * AG_TW + (lane & 15);
    const bool valid = e < E;
    const int s = valid ? a.src[e] : 0, t = valid ? a.dst[e] : 0;
    const int64_t pe = a.pos_index ? (valid ? (int64_t)a.pos_index[e] : 0) : e;     // where the edge's attrs / result live
    const int64_t pm = (a.pos_index && valid) ? (int64_t)a.mir_index[e] : -1;

    // first layer over eight 32-feature k-tiles of [h_src * h_dst || edge_attr] (pkk weights)
    f32x4 y1[8];
    ag_init_vec<8>(y1, a.hp.b1, q);
    {
      const float* hs = a.node_h + (size_t)s * 128;
      const float* ht = a.node_h + (size_t)t * 128;
      AgIn<MODE> sl[2];
      auto load_slice = [&](AgIn<MODE>& dst, int k) {
        if (k < 4) {
          const f32x4 p0 = ag_ld4(hs + 32 * k + 4 * q) * ag_ld4(ht + 32 * k + 4 * q);
          const f32x4 p1 = ag_ld4(hs + 32 * k + 16 + 4 * q) * ag_ld4(ht + 32 * k + 16 + 4 * q);
          ag_cvt(p0, p1, dst);
        } else if (a.attr_frag && a.pos_index) {
          ag_load_attr(dst, a.attr_frag, pe >> 4, k - 4, q * 16 + (int)(pe & 15));
        } else if (a.attr_frag) {
          ag_load_attr(dst, a.attr_frag, tile, k - 4, lane);
        } else {
          const float* ar = a.attr_rows + (size_t)(valid ? e : 0) * 128 + 32 * (k - 4) + 4 * q;
          ag_cvt(ag_ld4(ar), ag_ld4(ar + 16), dst);
        }
      };
      load_slice(sl[0], 0);
      const lds_u32x4* lw1_lo = ag_lds_base(lw1, lane);
      const lds_u32x4* lw1_hi = ag_lds_base(lw1 + 32 * 128, lane);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k + 1 < 8) load_slice(sl[(k + 1) & 1], k + 1);
        const lds_u32x4* wk = (k < 4) ? lw1_lo + (k * 8) * 128 : lw1_hi + ((k - 4) * 8) * 128;
        if (k & 1) ag_dense_lds<MODE, false, true, 1, 8, 1, 0>(sl, y1, wk, 0);
        else ag_dense_lds<MODE, false, true, 1, 8, 0, 0>(sl, y1, wk, 0);
      }
    }
    ag_head_act<8>(y1, a.hp.act);
    f32x4 y2[4];
    ag_init_vec<4>(y2, a.hp.b2, q);
    {
      AgIn<MODE> y1b[4];
      ag_report_range<MODE>(ag_absmax<MODE, 8>(y1, 0.0f), a.hp.range_rows, s, valid);      // (the hidden layer as an operand: common.hpp)
      ag_cvt_tiles<MODE, 4, 0>(y1, y1b);
      ag_dense_lds<MODE, false, false, 4, 4, 0, 0>(y1b, y2, ag_lds_base(lw2, lane), 0);
    }
    ag_head_act<4>(y2, a.hp.act);
    const float o = ag_dot_vec<4>(y2, a.hp.w3, q) + a.hp.b3;
    if (valid && q == 0) {
      a.out[pe] = o;
      if (pm >= 0) a.out[pm] = o;
    }
  }
}

// lw(d) * C(d) of the 2*num_convs CFConvs for the radius rows (ws->rad_*: AGDIFF_RAD_STRIDE rows per target), and the pad
// rows that complete a target's last 16-row tile: src = the target itself, length 0, every scale 0.
struct RadScaleArgs {
  const float* dw[2 * AGDIFF_MAX_CONVS];
  const int32_t* rad_cnt;
  int32_t* rad_src;
  float* rad_len;
  float* out;
  int64_t rpad;      // N * AGDIFF_RAD_STRIDE
  int32_t n;
  float cutoff;
  int32_t smooth;
};
__global__ void __launch_bounds__(256) k_rad_scales(RadScaleArgs a) {
  __shared__ float seg[2 * AGDIFF_MAX_CONVS * 100];
  for (int i = threadIdx.x; i < a.n * 100; i += blockDim.x) seg[i] = a.dw[i / 100][i % 100];
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= a.rpad) return;
  const int i = (int)(row / AGDIFF_RAD_STRIDE), k = (int)(row - (int64_t)i * AGDIFF_RAD_STRIDE);
  const int cnt = a.rad_cnt[i];
  if (k >= ((cnt + AG_TW - 1) / AG_TW) * AG_TW) return;
  if (k >= cnt) {
    a.rad_src[row] = i;
    a.rad_len[row] = 0.0f;
    for (int c = 0; c < a.n; ++c) a.out[(size_t)c * a.rpad + row] = 0.0f;
    return;
  }
  const float d = a.rad_len[row];
  const float C = cf_envelope(d, a.cutoff, a.smooth);
  for (int c = 0; c < a.n; ++c) a.out[(size_t)c * a.rpad + row] = cf_dist_weight(seg + c * 100, d) * C;
}

// ------------------------------------------------------------------------------ local edge_attr rows by polynomial
__global__ void k_zero_word(int32_t* p) {
  if (threadIdx.x == 0) *p = 0;
}

struct AttrPolyArgs {
  const float* poly_pk;       // [num_slots] x pk [8][1]
  const int32_t* type_slot;
  const int32_t* n_dev;
  const float* e_len;
  const int32_t* e_type;
  float* out_rows;            // [n][128]
  int32_t* flags;             // [1 + tiles]
  int64_t max_tiles;
  int32_t num_slots;
  int32_t far_slots;          // far sets behind the num_slots near ones: edge_attr on (cutoff, far_hi]
  const int32_t* type_far;    // [100] edge type -> its far set (index into poly_pk) or -1; null without far sets
  float cutoff;
  float two_over_rc;
  float far_hi;
  float two_over_far;         // 2 / (far_hi - cutoff)
};

// edge_attr rows (fp32, natural feature order) of the canonical local edges from the per-type polynomials: every row whose
// length lies in [0, cutoff] -- or in (cutoff, far_hi] when its type has a far set -- and whose type has a slot is evaluated
// here (features once, one masked MFMA round per coefficient set present in the tile); the other rows are flagged for the
// encoder MLP (agdiff_local_edge_rows).
#ifndef AG_ATTRP_WAVES
#define AG_ATTRP_WAVES 16        // (one 16-wave workgroup per CU: the kernel is bound by its row stores -- 0.055 ms without them, 0.135 with, at 8
#endif                           // waves; twice the tiles in flight drain them 6 % faster: 0.127 ms for 417 MB on the 196 k-atom batch)
#define AG_ATTRP_MAX_SLOTS 9         // 9 x 16 KiB of coefficients in LDS
template <int MODE>
__global__ void __launch_bounds__(64 * AG_ATTRP_WAVES, AG_ATTRP_WAVES / 4) k_edge_attr_poly(AttrPolyArgs a) {
  extern __shared__ u32x4 ag_attrp_smem[];
  lds_u32x4* wl = (lds_u32x4*)ag_attrp_smem;
  __shared__ int wg_flagged;
  if (threadIdx.x == 0) wg_flagged = 0;
  ag_copy_lds(wl, reinterpret_cast<const u32x4*>(a.poly_pk), (a.num_slots + a.far_slots) * 8 * 128);
  __syncthreads();
  const int lane0 = ag_lane();
  const int E = *a.n_dev;
  const int64_t stride = (int64_t)gridDim.x * AG_ATTRP_WAVES;
  int my_flagged = 0;
  for (int64_t tile = (int64_t)blockIdx.x * AG_ATTRP_WAVES + (threadIdx.x >> 6); tile < a.max_tiles; tile += stride) {
    if (tile * AG_TW >= E) break;
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int q = lane >> 4;
    const int64_t e = tile * AG_TW + (lane & 15);
    const bool valid = e < E;
    const float d = valid ? a.e_len[e] : 0.0f;
    const int ety = valid ? a.e_type[e] : 0;
    const int tslot = valid ? a.type_slot[ety] : -1;
    const int fset = (valid && a.far_slots > 0) ? a.type_far[ety] : -1;
    // (NaN lengths by their bit pattern: this file is built with -fno-honor-nans, under which `!(d >= 0 && d <= rc)` may be
    // rewritten into comparisons that a NaN passes)
    const bool is_nan = (__float_as_uint(d) & 0x7FFFFFFFu) > 0x7F800000u;
    // beyond the cutoff: the type's far set (encoder/edge.py:84-103 on (cutoff, far_hi]), where it has one
    const bool far = d > a.cutoff && d <= a.far_hi && tslot >= 0 && fset >= 0;
    const int slot = far ? fset : tslot;
    const bool hard = valid && (is_nan || d < 0.0f || (d > a.cutoff && !far) || tslot < 0);
    // rows the polynomials do not cover (longer than the cutoff, a type without a slot) are left to the encoder MLP: the
    // tile's flag word is the mask of those rows, everything else is stored here -- per edge, whatever its tile-mates are
    const int hard_rows = (int)(__ballot(hard) & 0xFFFFull);
    if (lane == 0) a.flags[1 + tile] = hard_rows;
    if (hard_rows) ++my_flagged;
    if (hard_rows == (int)(__ballot(valid) & 0xFFFFull)) continue;        // nothing for the polynomials in this tile
    AgIn<MODE> phall[1], ph[1];
    ag_poly_features<MODE, 1>(far ? d - a.cutoff : d, far ? a.two_over_far : a.two_over_rc, q, phall);
    f32x4 y[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) y[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const lds_u32x4* wl_l = ag_lds_base(wl, lane);
    uint64_t todo = __ballot(slot >= 0 && !hard) & 0xFFFFull;
    while (todo) {
      const int g = __builtin_amdgcn_readlane(slot, (int)__builtin_ctzll(todo));
      const bool in = slot == g && !hard;
      todo &= ~__ballot(in);
      const u32x4 zero = {0u, 0u, 0u, 0u};
      if constexpr (MODE == AG_F32) {
        ph[0].v[0] = in ? phall[0].v[0] : f32x4{0.f, 0.f, 0.f, 0.f};
        ph[0].v[1] = in ? phall[0].v[1] : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        ph[0].hi = __builtin_bit_cast(decltype(ph[0].hi), in ? __builtin_bit_cast(u32x4, phall[0].hi) : zero);
        ph[0].lo = __builtin_bit_cast(decltype(ph[0].lo), in ? __builtin_bit_cast(u32x4, phall[0].lo) : zero);
      }
      const lds_u32x4* wg_ = wl_l + (size_t)g * (8 * 128);
      u32x4 w[8][2];
#pragma unroll
      for (int ot = 0; ot < 8; ++ot) {
        w[ot][0] = wg_[(ot * 2) * 64];
        w[ot][1] = wg_[(ot * 2 + 1) * 64];
      }
#pragma unroll
      for (int part = 0; part < AgParts<MODE>::n; ++part) {
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) ag_block_mma_part<MODE, false>(y[ot], ph[0], w[ot], part);
      }
    }
    if (valid && !hard) ag_store_row<8, 0>(y, a.out_rows + (size_t)e * 128, q);
  }
  // flagged tiles of the launch: one atomic per workgroup (a count of integers: the order does not matter)
  if (lane0 == 0 && my_flagged) atomicAdd(&wg_flagged, my_flagged);
  __syncthreads();
  if (threadIdx.x == 0 && wg_flagged) atomicAdd(&a.flags[0], wg_flagged);
}

// ------------------------------------------------------------------------------ pair head, edge_attr half by polynomial
struct HeadPolyArgs {
  agdiff_head_params_t hp;
  const int32_t* n_dev;
  const int32_t* src;
  const int32_t* dst;
  const float* len;
  const float* node_h;
  const int32_t* pos_index;   // optional (with mir_index): results go to out[pos_index[e]] and, when >= 0, out[mir_index[e]]
  const int32_t* mir_index;
  float* out;
  int64_t max_tiles;
  float two_over_rc;
  const int32_t* seg_tile_live;   // optional: the list is cut into 16-aligned segments, one per molecule (agdiff_sampler_front);
                                  // [max_tiles] live entries (0..16) of every tile; n_dev is unused then
};

// k_pair_head for edges whose edge_attr is MLPEdgeEncoder(d, type 0): the edge_attr half of the first layer,
// layers.0.weight[:, 128:] @ edge_attr, is the polynomial hp.attr_poly_pk of d (NKT k-tiles instead of four, and no
// edge_attr to read).  Entries of another type get a finite but meaningless result: the denoising loop multiplies them
// by 1 - local_edge_mask = 0 (dualenc.py:516-518) and the update kernel never reads them.
template <int MODE, int NKT>
__global__ void __launch_bounds__(64 * AG_PERSIST_WAVES, 4) k_pair_head_poly(HeadPolyArgs a) {
  extern __shared__ u32x4 ag_headp_smem[];
  lds_u32x4* lw1 = (lds_u32x4*)ag_headp_smem;     // k-tiles 0..3 of pkk [8][8] (h_src * h_dst half) = 32 blocks
  lds_u32x4* lwp = lw1 + 32 * 128;                // pkk [NKT][8]
  lds_u32x4* lw2 = lwp + 8 * NKT * 128;           // pk  [4][4] = 16 blocks
  {
    const u32x4* g1 = reinterpret_cast<const u32x4*>(a.hp.w1_pk);
    const u32x4* gp = reinterpret_cast<const u32x4*>(a.hp.attr_poly_pk);
    const u32x4* g2 = reinterpret_cast<const u32x4*>(a.hp.w2_pk);
    ag_copy_lds(lw1, g1, 32 * 128);
    ag_copy_lds(lwp, gp, 8 * NKT * 128);
    ag_copy_lds(lw2, g2, 16 * 128);
  }
  __syncthreads();
  const int lane0 = ag_lane();
  const int E = a.seg_tile_live ? 0 : *a.n_dev;
  const int64_t stride = (int64_t)gridDim.x * AG_PERSIST_WAVES;
  const int64_t wg = (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;
  for (int64_t tile = wg * AG_PERSIST_WAVES + (threadIdx.x >> 6); tile < a.max_tiles; tile += stride) {
    int64_t live_end;                    // entries of the list at or beyond this index are not live
    if (a.seg_tile_live) {
      const int live = a.seg_tile_live[tile];
      if (live == 0) continue;
      live_end = tile * AG_TW + live;
    } else {
      live_end = E;
      if (tile * AG_TW >= E) break;
    }
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int q = lane >> 4;
    const int64_t e = tile * AG_TW + (lane & 15);
    const bool valid = e < live_end;
    const int s = valid ? a.src[e] : 0, t = valid ? a.dst[e] : 0;
    const float d = valid ? a.len[e] : 0.0f;
    const int64_t pe = a.pos_index ? (valid ? (int64_t)a.pos_index[e] : 0) : e;
    const int64_t pm = (a.pos_index && valid) ? (int64_t)a.mir_index[e] : -1;

    f32x4 y1[8];
    ag_init_vec<8>(y1, a.hp.b1, q);
    {
      const float* hs = a.node_h + (size_t)s * 128;
      const float* ht = a.node_h + (size_t)t * 128;
      AgIn<MODE> sl[2];
      auto load_slice = [&](AgIn<MODE>& dst, int k) {
        const f32x4 p0 = ag_ld4(hs + 32 * k + 4 * q) * ag_ld4(ht + 32 * k + 4 * q);
        const f32x4 p1 = ag_ld4(hs + 32 * k + 16 + 4 * q) * ag_ld4(ht + 32 * k + 16 + 4 * q);
        ag_cvt(p0, p1, dst);
      };
      load_slice(sl[0], 0);
      const lds_u32x4* lw1_l = ag_lds_base(lw1, lane);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k + 1 < 4) load_slice(sl[(k + 1) & 1], k + 1);
        if (k & 1) ag_dense_lds<MODE, false, true, 1, 8, 1, 0>(sl, y1, lw1_l + (k * 8) * 128, 0);
        else ag_dense_lds<MODE, false, true, 1, 8, 0, 0>(sl, y1, lw1_l + (k * 8) * 128, 0);
      }
      AgIn<MODE> ph[NKT];
      ag_poly_features<MODE, NKT>(d, a.two_over_rc, q, ph);
      ag_dense_lds<MODE, false, true, NKT, 8, 0, 0>(ph, y1, ag_lds_base(lwp, lane), 0);
    }
    ag_head_act<8>(y1, a.hp.act);
    f32x4 y2[4];
    ag_init_vec<4>(y2, a.hp.b2, q);
    {
      AgIn<MODE> y1b[4];
      ag_report_range<MODE>(ag_absmax<MODE, 8>(y1, 0.0f), a.hp.range_rows, s, valid);      // (the hidden layer as an operand: common.hpp)
      ag_cvt_tiles<MODE, 4, 0>(y1, y1b);
      ag_dense_lds<MODE, false, false, 4, 4, 0, 0>(y1b, y2, ag_lds_base(lw2, lane), 0);
    }
    ag_head_act<4>(y2, a.hp.act);
    const float o = ag_dot_vec<4>(y2, a.hp.w3, q) + a.hp.b3;
    if (valid && q == 0) {
      a.out[pe] = o;
      if (pm >= 0) a.out[pm] = o;
    }
  }
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

#ifdef AG_CONV_STAMPS
extern "C" int agdiff_debug_conv_stamps(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ag_conv_stamp_acc), sizeof(ag_conv_stamp_acc)) != hipSuccess) return AGDIFF_ERR_LAUNCH;
  if (reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(ag_conv_stamp_acc), z, sizeof(z)) != hipSuccess) return AGDIFF_ERR_LAUNCH;
  }
  return AGDIFF_OK;
}
#endif

// (set by agdiff_local_edge_rows around its call of agdiff_edge_encoder: the launch then only does the flagged tiles)
static thread_local const int32_t* g_enc_tile_flags = nullptr;

extern "C" int agdiff_edge_encoder(const agdiff_params_t* p, const int32_t* n_edges_dev, int64_t max_tiles,
                                   const float* e_len, const int32_t* e_type, float* attr_frag, float* attr_rows,
                                   const int32_t* row_index, const int32_t* pos_index, const int32_t* mir_index,
                                   void* stream) {
  if (!p || !n_edges_dev || !e_len || !e_type || (!attr_frag && !attr_rows) || max_tiles < 0 || (!pos_index != !mir_index))
    return AGDIFF_ERR_ARG;
  if (max_tiles == 0) return AGDIFF_OK;
  if (p->edge_encoder == 1) {
    if (!p->ge_offset || !p->ge_emb) return AGDIFF_ERR_ARG;
    GaussArgs g{p->ge_offset, p->ge_emb, n_edges_dev, e_len, e_type, attr_frag, attr_rows, row_index, pos_index, mir_index,
                max_tiles, p->ge_coeff * 1.44269504088896340736f};
    const dim3 grid((unsigned)((max_tiles + 3) / 4));
    if (p->precision == AG_H3)
      k_edge_gaussian<AG_H3><<<grid, dim3(256), 0, (hipStream_t)stream>>>(g);
    else if (p->precision == AG_BF3)
      k_edge_gaussian<AG_BF3><<<grid, dim3(256), 0, (hipStream_t)stream>>>(g);
    else
      k_edge_gaussian<AG_F32><<<grid, dim3(256), 0, (hipStream_t)stream>>>(g);
    AG_CHECK_LAUNCH();
    return AGDIFF_OK;
  }
  if (p->edge_encoder != 0) return AGDIFF_ERR_ARG;
  EncArgs a{p->ee_fe_w, p->ee_fe_b, p->ee_t1, p->ee_w1_pk, p->ee_t3, p->ee_w23_pk, p->ee_w4_pk, p->ee_b4,
            n_edges_dev, e_len, e_type, attr_frag, attr_rows, row_index, pos_index, mir_index, max_tiles, g_enc_tile_flags};
  g_enc_tile_flags = nullptr;
  int64_t wgs = (max_tiles + AG_PERSIST_WAVES - 1) / AG_PERSIST_WAVES;
  if (wgs > 256) wgs = 256;
  const size_t smem = (size_t)80 * 2048;     // w1 (32 blocks) + w23 (32) + unit 0 of w4's 32 blocks
  static std::atomic<uint64_t> attr_done{0};
  if (!ag_allow_big_lds(attr_done, smem, k_edge_encoder<AG_BF3>, k_edge_encoder<AG_F32>, k_edge_encoder<AG_H3>)) return AGDIFF_ERR_LAUNCH;
  if (p->precision == AG_H3)
    k_edge_encoder<AG_H3><<<dim3((unsigned)wgs), dim3(64 * AG_PERSIST_WAVES), smem, (hipStream_t)stream>>>(a);
  else if (p->precision == AG_BF3)
    k_edge_encoder<AG_BF3><<<dim3((unsigned)wgs), dim3(64 * AG_PERSIST_WAVES), smem, (hipStream_t)stream>>>(a);
  else
    k_edge_encoder<AG_F32><<<dim3((unsigned)wgs), dim3(64 * AG_PERSIST_WAVES), smem, (hipStream_t)stream>>>(a);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}

namespace {
int launch_edge_scales(const agdiff_params_t* p, const int32_t* n_dev, int64_t max_n, const float* e_len,
                       const int32_t* pos_index, const int32_t* mir_index, float* out, int64_t epad, void* stream,
                       const int32_t* zero_slotted_types = nullptr) {
  if (max_n == 0) return AGDIFF_OK;
  ScaleArgs a;
  a.e_type = zero_slotted_types;
  a.type_slot = zero_slotted_types ? p->poly_type_slot : nullptr;
  for (int k = 0; k < p->num_convs; ++k) {
    a.dw[2 * k] = p->conv[k].dist_seg;
    a.dw[2 * k + 1] = p->conv[k].dist_seg + 100;
  }
  a.n_dev = n_dev;
  a.e_len = e_len;
  a.pos_index = pos_index;
  a.mir_index = mir_index;
  a.out = out;
  a.epad = epad;
  a.n = 2 * p->num_convs;
  a.cutoff = p->cutoff;
  a.smooth = p->smooth;
  k_edge_scales<<<dim3((unsigned)((max_n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}
}  // namespace

extern "C" int agdiff_edge_scales(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                                  int32_t per_canonical_edge, void* stream) {
  if (!p || !topo || !ws || !ws->e_scale || !ws->e_len || !ws->num_edges || p->num_convs > AGDIFF_MAX_CONVS)
    return AGDIFF_ERR_ARG;
  if (per_canonical_edge && (!ws->num_canon || !ws->c_len || !ws->c_pos || !ws->c_mir)) return AGDIFF_ERR_ARG;
  const int64_t epad = ((topo->max_edges + AG_TW - 1) / AG_TW) * AG_TW;
  return launch_edge_scales(p, per_canonical_edge ? ws->num_canon : ws->num_edges, topo->max_edges,
                            per_canonical_edge ? ws->c_len : ws->e_len, per_canonical_edge ? ws->c_pos : nullptr,
                            per_canonical_edge ? ws->c_mir : nullptr, ws->e_scale, epad, stream);
}

extern "C" int agdiff_edge_scales_split(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                                        int32_t which, void* stream) {
  if (!p || !topo || !ws || p->num_convs > AGDIFF_MAX_CONVS) return AGDIFF_ERR_ARG;
  if (which == 0) {        // radius rows by target (+ the pad rows of every target's last tile)
    if (!ws->r_scale || !ws->rad_len || !ws->rad_src || !ws->rad_cnt) return AGDIFF_ERR_ARG;
    if (topo->num_nodes <= 0) return AGDIFF_OK;
    RadScaleArgs a;
    for (int k = 0; k < p->num_convs; ++k) {
      a.dw[2 * k] = p->conv[k].dist_seg;
      a.dw[2 * k + 1] = p->conv[k].dist_seg + 100;
    }
    a.rad_cnt = ws->rad_cnt;
    a.rad_src = ws->rad_src;
    a.rad_len = ws->rad_len;
    a.out = ws->r_scale;
    a.rpad = topo->num_nodes * (int64_t)AGDIFF_RAD_STRIDE;
    a.n = 2 * p->num_convs;
    a.cutoff = p->cutoff;
    a.smooth = p->smooth;
    k_rad_scales<<<dim3((unsigned)((a.rpad + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
    AG_CHECK_LAUNCH();
    return AGDIFF_OK;
  }
  // local edges: one evaluation per canonical local edge (a mirror pair has one length), written to both of its positions
  if (!ws->lc_len || !ws->num_local_canon) return AGDIFF_ERR_ARG;
  if (which == 1) {        // ... in the padded local list (agdiff_cfconv_local)
    if (!ws->l_scale || !topo->lc_ppos || !topo->lc_pmir) return AGDIFF_ERR_ARG;
    return launch_edge_scales(p, ws->num_local_canon, topo->num_local_canon, ws->lc_len, topo->lc_ppos, topo->lc_pmir,
                              ws->l_scale, ((topo->num_local_padded + AG_TW - 1) / AG_TW) * AG_TW, stream,
                              agdiff_local_poly_enabled(p, topo, ws) == 2 ? topo->lc_type : nullptr);
  }
  // ... in the quad tiles (agdiff_cfconv_node)
  if (!ws->lt_scale || !topo->lc_tpos || !topo->lc_tmir) return AGDIFF_ERR_ARG;
  return launch_edge_scales(p, ws->num_local_canon, topo->num_local_canon, ws->lc_len, topo->lc_tpos, topo->lc_tmir,
                            ws->lt_scale, topo->num_local_tiles * AG_TW, stream);
}

namespace {
// k_cfconv_fused over a destination-sorted edge list of `max_e` slots (live count *n_dev)
int launch_cfconv_fused(const agdiff_params_t* p, int32_t k, int64_t max_e, const int32_t* n_dev, const int32_t* in_ptr,
                        const int32_t* e_src, const int32_t* e_dst, const float* scales, const float* e_attr, const float* xs,
                        float* agg, float* agg_first, void* stream) {
  const int64_t max_tiles = (max_e + AG_TW - 1) / AG_TW;
  const int chunk_tiles = agdiff_conv_chunk_tiles(max_e);
  const int64_t max_chunks = (max_tiles + chunk_tiles - 1) / chunk_tiles;
  if (max_chunks == 0) return AGDIFF_OK;
  ConvArgs a;
  a.cp = p->conv[k];
  a.n_dev = n_dev;
  a.in_ptr = in_ptr;
  a.e_src = e_src;
  a.e_dst = e_dst;
  {
    const size_t epad = (size_t)max_tiles * AG_TW;
    a.scale1 = scales + (size_t)(2 * k) * epad;
    a.scale2 = scales + (size_t)(2 * k + 1) * epad;
  }
  a.e_attr = e_attr;
  a.xs = xs;
  a.agg = agg;
  a.agg_first = agg_first;
  a.max_chunks = max_chunks;
  a.chunk_tiles = chunk_tiles;
  a.ablate = 0;
#ifdef AG_CONV_ABLATE
  {
    static int abl = -1;
    if (abl < 0) {
      const char* e = getenv("AGDIFF_ABLATE");
      abl = e ? atoi(e) : 0;
    }
    a.ablate = abl;
  }
#endif
  // persistent launch: one 8-wave workgroup per CU keeps 160 KiB of filter weights in LDS
  int64_t wgs = (max_chunks + AG_CONV_WAVES - 1) / AG_CONV_WAVES;
  if (wgs > 256) wgs = 256;
  const size_t smem = (size_t)AG_CONV_LDS_BLOCKS * 2048;
  static std::atomic<uint64_t> attr_done{0};
  if (!ag_allow_big_lds(attr_done, smem, k_cfconv_fused<AG_BF3>, k_cfconv_fused<AG_F32>, k_cfconv_fused<AG_H3>)) return AGDIFF_ERR_LAUNCH;
  if (p->precision == AG_H3)
    k_cfconv_fused<AG_H3><<<dim3((unsigned)wgs), dim3(64 * AG_CONV_WAVES), smem, (hipStream_t)stream>>>(a);
  else if (p->precision == AG_BF3)
    k_cfconv_fused<AG_BF3><<<dim3((unsigned)wgs), dim3(64 * AG_CONV_WAVES), smem, (hipStream_t)stream>>>(a);
  else
    k_cfconv_fused<AG_F32><<<dim3((unsigned)wgs), dim3(64 * AG_CONV_WAVES), smem, (hipStream_t)stream>>>(a);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}

}  // namespace

extern "C" int agdiff_cfconv_fused(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, int32_t k,
                                   void* stream) {
  if (!p || !topo || !ws || k < 0 || k >= p->num_convs) return AGDIFF_ERR_ARG;
  ag_log_variant(ws, AGDIFF_VAR_CFCONV_FUSED);
  return launch_cfconv_fused(p, k, topo->max_edges, ws->num_edges, ws->in_ptr, ws->e_src, ws->e_dst, ws->e_scale, ws->e_attr,
                             ws->xs, ws->agg, ws->agg_first, stream);
}

// Local edges by per-type filter polynomials?  (all of: slots built by the host, the quad-tile inputs present)
extern "C" int agdiff_local_poly_enabled(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws) {
  const bool on = p && topo && ws && !p->tune_local_poly_off && p->poly_kt >= 1 && p->poly_kt <= AGDIFF_POLY_MAX_KT &&
                  p->poly_num_slots > 0 && p->poly_num_slots <= AGDIFF_POLY_MAX_SLOTS && p->poly_type_slot && topo->lt_ptr &&
                  topo->lt_src && topo->lt_type && ws->lt_len && ws->lt_scale;
  if (!on) return 0;
  const uint64_t miss0 = (uint64_t)topo->local_type_mask[0] & ~(uint64_t)p->poly_slot_mask[0];
  const uint64_t miss1 = (uint64_t)topo->local_type_mask[1] & ~(uint64_t)p->poly_slot_mask[1];
  if (!(miss0 | miss1)) return 1;                          // every local type of the batch has a slot
  const uint64_t have = ((uint64_t)topo->local_type_mask[0] & (uint64_t)p->poly_slot_mask[0]) |
                        ((uint64_t)topo->local_type_mask[1] & (uint64_t)p->poly_slot_mask[1]);
  return have ? 2 : 0;                                     // some do (mixed) / none does
}

// The local edges through the filter MLPs (no per-type polynomials): k_cfconv_fused over the padded local list
extern "C" int agdiff_cfconv_local(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, int32_t k,
                                   void* stream) {
  if (!p || !topo || !ws || k < 0 || k >= p->num_convs) return AGDIFF_ERR_ARG;
  if (topo->num_local == 0) return AGDIFF_OK;
  if (!topo->lp_ptr || !topo->lp_src || !topo->lp_dst || !ws->l_scale || !ws->l_attr_frag || !ws->agg_loc ||
      !ws->agg_first_loc || !ws->num_local_padded)
    return AGDIFF_ERR_ARG;
  ag_log_variant(ws, AGDIFF_VAR_CFCONV_LOCAL_MLP);
  return launch_cfconv_fused(p, k, topo->num_local_padded, ws->num_local_padded, topo->lp_ptr, topo->lp_src, topo->lp_dst,
                             ws->l_scale, ws->l_attr_frag, ws->xs, ws->agg_loc, ws->agg_first_loc, stream);
}

namespace {
int launch_pair_head_poly(const agdiff_params_t* p, const int32_t* n_edges_dev, int64_t max_tiles,
                          const int32_t* src, const int32_t* dst, const float* len, const float* node_h,
                          const int32_t* pos_index, const int32_t* mir_index, float* out, const int32_t* seg_tile_live,
                          int32_t* range_rows, void* stream);
}

extern "C" int agdiff_pair_head_poly(const agdiff_params_t* p, const int32_t* n_edges_dev, int64_t max_tiles,
                                     const int32_t* src, const int32_t* dst, const float* len, const float* node_h,
                                     const int32_t* pos_index, const int32_t* mir_index, float* out, void* stream) {
  if (!n_edges_dev) return AGDIFF_ERR_ARG;
  return launch_pair_head_poly(p, n_edges_dev, max_tiles, src, dst, len, node_h, pos_index, mir_index, out, nullptr, nullptr, stream);
}

extern "C" int agdiff_pair_head_poly_rows(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                                          int32_t parity, void* stream) {
  if (!p || !topo || !ws || !ws->canon_counter || !ws->inv_r) return AGDIFF_ERR_ARG;
  const int64_t tiles = (topo->max_edges - topo->num_local + AG_TW - 1) / AG_TW + 1;
  return launch_pair_head_poly(p, ws->canon_counter + (parity & 1), tiles, ws->c_src, ws->c_dst, ws->c_len, ws->h, ws->c_pos,
                               ws->c_mir, ws->inv_r, nullptr, ws->range_rows, stream);
}

namespace {
int launch_pair_head_poly(const agdiff_params_t* p, const int32_t* n_edges_dev, int64_t max_tiles,
                          const int32_t* src, const int32_t* dst, const float* len, const float* node_h,
                          const int32_t* pos_index, const int32_t* mir_index, float* out, const int32_t* seg_tile_live,
                          int32_t* range_rows, void* stream) {
  if (!p || !src || !dst || !len || !node_h || !out || max_tiles < 0 || (!pos_index != !mir_index))
    return AGDIFF_ERR_ARG;
  if (p->poly_kt < 1 || p->poly_kt > AGDIFF_POLY_MAX_KT || !p->head_global.attr_poly_pk) return AGDIFF_ERR_ARG;
  if (max_tiles == 0) return AGDIFF_OK;
  HeadPolyArgs a;
  a.hp = p->head_global;
  if (range_rows) a.hp.range_rows = range_rows;
  a.seg_tile_live = seg_tile_live;
  a.n_dev = n_edges_dev;
  a.src = src;
  a.dst = dst;
  a.len = len;
  a.node_h = node_h;
  a.pos_index = pos_index;
  a.mir_index = mir_index;
  a.out = out;
  a.max_tiles = max_tiles;
  a.two_over_rc = 2.0f / p->cutoff;
  int64_t wgs = (max_tiles + AG_PERSIST_WAVES - 1) / AG_PERSIST_WAVES;
  if (wgs > 256) wgs = 256;
  const size_t smem = (size_t)(48 + 8 * p->poly_kt) * 2048;
  static std::atomic<uint64_t> attr_done{0};
#define AG_HEADP_KERNELS(M) k_pair_head_poly<M, 1>, k_pair_head_poly<M, 2>, k_pair_head_poly<M, 3>, k_pair_head_poly<M, 4>
  if (!ag_allow_big_lds(attr_done, (size_t)(48 + 8 * AGDIFF_POLY_MAX_KT) * 2048, AG_HEADP_KERNELS(AG_BF3), AG_HEADP_KERNELS(AG_F32),
                        AG_HEADP_KERNELS(AG_H3)))
    return AGDIFF_ERR_LAUNCH;
#undef AG_HEADP_KERNELS
  const dim3 grid((unsigned)wgs), block(64 * AG_PERSIST_WAVES);
  hipStream_t st = (hipStream_t)stream;
  auto by_terms = [&](auto MODE_) {
    constexpr int MODE = decltype(MODE_)::value;
    switch (p->poly_kt) {
      case 1: k_pair_head_poly<MODE, 1><<<grid, block, smem, st>>>(a); break;
      case 2: k_pair_head_poly<MODE, 2><<<grid, block, smem, st>>>(a); break;
      case 3: k_pair_head_poly<MODE, 3><<<grid, block, smem, st>>>(a); break;
      default: k_pair_head_poly<MODE, 4><<<grid, block, smem, st>>>(a); break;
    }
  };
  if (p->precision == AG_H3) by_terms(std::integral_constant<int, AG_H3>{});
  else if (p->precision == AG_BF3) by_terms(std::integral_constant<int, AG_BF3>{});
  else by_terms(std::integral_constant<int, AG_F32>{});
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}
}  // namespace

extern "C" int agdiff_local_edge_rows(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, void* stream) {
  if (!p || !topo || !ws) return AGDIFF_ERR_ARG;
  if (topo->num_local == 0) return AGDIFF_OK;
  if (!ws->l_attr_rows || !ws->lc_len || !ws->num_local_canon || !topo->lc_type || topo->num_local_canon <= 0) return AGDIFF_ERR_ARG;
  const int64_t ctiles = (topo->num_local_canon + AG_TW - 1) / AG_TW;
  // (all slots' 16-KiB sets sit in LDS: up to AG_ATTRP_MAX_SLOTS local edge types, else the encoder MLP for every tile)
  const bool poly = !p->tune_attr_poly_off && p->edge_encoder == 0 && p->poly_kt == 1 && p->poly_num_slots > 0 &&
                    p->poly_num_slots <= AG_ATTRP_MAX_SLOTS && p->attr_poly_typed_pk && p->poly_type_slot && ws->enc_flags;
  if (p->attr_poly_far_slots < 0 ||
      (p->attr_poly_far_slots > 0 && (p->poly_num_slots + p->attr_poly_far_slots > AG_ATTRP_MAX_SLOTS || !p->attr_poly_far_set ||
                                      !(p->attr_poly_far_hi > p->cutoff))))
    return AGDIFF_ERR_ARG;
  if (!poly)
    return agdiff_edge_encoder(p, ws->num_local_canon, ctiles, ws->lc_len, topo->lc_type, nullptr, ws->l_attr_rows, nullptr,
                               nullptr, nullptr, stream);
  hipStream_t st = (hipStream_t)stream;
  k_zero_word<<<1, 64, 0, st>>>(ws->enc_flags);       // (hipMemsetAsync of these four bytes ran a 17-us fill kernel)
  AG_CHECK_LAUNCH();
  AttrPolyArgs a;
  a.poly_pk = p->attr_poly_typed_pk;
  a.type_slot = p->poly_type_slot;
  a.n_dev = ws->num_local_canon;
  a.e_len = ws->lc_len;
  a.e_type = topo->lc_type;
  a.out_rows = ws->l_attr_rows;
  a.flags = ws->enc_flags;
  a.max_tiles = ctiles;
  a.num_slots = p->poly_num_slots;
  a.far_slots = poly ? p->attr_poly_far_slots : 0;
  a.type_far = p->attr_poly_far_set;
  a.cutoff = p->cutoff;
  a.two_over_rc = 2.0f / p->cutoff;
  a.far_hi = a.far_slots > 0 ? p->attr_poly_far_hi : p->cutoff;
  a.two_over_far = a.far_slots > 0 ? 2.0f / (p->attr_poly_far_hi - p->cutoff) : 0.0f;
  int64_t wgs = (ctiles + AG_ATTRP_WAVES - 1) / AG_ATTRP_WAVES;
  if (wgs > (AG_ATTRP_WAVES >= 16 ? 256 : 512)) wgs = AG_ATTRP_WAVES >= 16 ? 256 : 512;
  const size_t smem = (size_t)(p->poly_num_slots + a.far_slots) * 8 * 2048;
  static std::atomic<uint64_t> attr_done{0};
  if (!ag_allow_big_lds(attr_done, (size_t)AG_ATTRP_MAX_SLOTS * 8 * 2048, k_edge_attr_poly<AG_BF3>, k_edge_attr_poly<AG_F32>,
                        k_edge_attr_poly<AG_H3>))
    return AGDIFF_ERR_LAUNCH;
  // (the local branch's arithmetic mode: attr_poly_typed_pk is packed in it)
  if (p->precision_local == AG_H3)
    k_edge_attr_poly<AG_H3><<<dim3((unsigned)wgs), dim3(64 * AG_ATTRP_WAVES), smem, st>>>(a);
  else if (p->precision_local == AG_BF3)
    k_edge_attr_poly<AG_BF3><<<dim3((unsigned)wgs), dim3(64 * AG_ATTRP_WAVES), smem, st>>>(a);
  else
    k_edge_attr_poly<AG_F32><<<dim3((unsigned)wgs), dim3(64 * AG_ATTRP_WAVES), smem, st>>>(a);
  AG_CHECK_LAUNCH();
  ag_log_variant(ws, AGDIFF_VAR_ATTR_POLY);
  // the flagged tiles (if any) through the MLP: the launch returns before staging its weights when the count is 0
  g_enc_tile_flags = ws->enc_flags;
  const int rc = agdiff_edge_encoder(p, ws->num_local_canon, ctiles, ws->lc_len, topo->lc_type, nullptr, ws->l_attr_rows,
                                     nullptr, nullptr, nullptr, stream);
  g_enc_tile_flags = nullptr;        // (also when the call returned before it consumed the pointer)
  return rc;
}

extern "C" int agdiff_pair_head(const agdiff_head_params_t* hp, const int32_t* n_edges_dev, int64_t max_tiles,
                                const int32_t* src, const int32_t* dst, const float* node_h, const float* attr_frag,
                                const float* attr_rows, const int32_t* pos_index, const int32_t* mir_index, float* out,
                                void* stream) {
  if (!hp || !n_edges_dev || !src || !dst || !node_h || (!attr_frag == !attr_rows) || !out || max_tiles < 0 ||
      (!pos_index != !mir_index))
    return AGDIFF_ERR_ARG;
  if (max_tiles == 0) return AGDIFF_OK;
  HeadArgs a;
  a.hp = *hp;
  a.n_dev = n_edges_dev;
  a.src = src;
  a.dst = dst;
  a.node_h = node_h;
  a.attr_frag = attr_frag;
  a.attr_rows = attr_rows;
  a.pos_index = pos_index;
  a.mir_index = mir_index;
  a.out = out;
  a.max_tiles = max_tiles;
  int64_t wgs = (max_tiles + AG_PERSIST_WAVES - 1) / AG_PERSIST_WAVES;
  if (wgs > 256) wgs = 256;
  const size_t smem = (size_t)80 * 2048;     // w1 (64 blocks) + w2 (16 blocks)
  static std::atomic<uint64_t> attr_done{0};
  if (!ag_allow_big_lds(attr_done, smem, k_pair_head<AG_BF3>, k_pair_head<AG_F32>, k_pair_head<AG_H3>)) return AGDIFF_ERR_LAUNCH;
  // (operand-form edge_attr tiles must have been written in the head's mode: the global encoder writes them in p->precision,
  // which is what head_global carries; head_local reads fp32 rows)
  if (hp->precision == AG_H3)
    k_pair_head<AG_H3><<<dim3((unsigned)wgs), dim3(64 * AG_PERSIST_WAVES), smem, (hipStream_t)stream>>>(a);
  else if (hp->precision == AG_BF3)
    k_pair_head<AG_BF3><<<dim3((unsigned)wgs), dim3(64 * AG_PERSIST_WAVES), smem, (hipStream_t)stream>>>(a);
  else
    k_pair_head<AG_F32><<<dim3((unsigned)wgs), dim3(64 * AG_PERSIST_WAVES), smem, (hipStream_t)stream>>>(a);
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
