// The two CFConvs of an InteractionBlock with their filters from d-polynomials, one wave per quad of targets
// (include/agdiff_hip.h: agdiff_cfconv_node).  Its own translation unit: built with -fno-slp-vectorize -- the SLP
// vectoriser turns the per-row accumulation FMAs into v_pk_fma_f32 fed by register shuffles (358 v_mov per kernel), and
// packed fp32 next to MFMAs costs issue time instead of saving it (MI355X_MICROARCH.md, per-instruction cycle constants).
#include "common.hpp"
#include <type_traits>

#define AG_CONV_NCH 12          // 16-channel tiles of the 192 filter channels (conv1: 0..7, conv2: 8..11)

#ifdef AG_QUAD_STAMPS      // (debug build: start / end clock of every workgroup of the last k_cfconv_quad launch -- tools/quad_stamps.py)
__device__ unsigned long long ag_quad_stamp[2 * 256];
extern "C" int agdiff_debug_quad_stamps(unsigned long long* out) {      // (reads the stamps and clears them for the next launch)
  static unsigned long long zero[2 * 256];
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ag_quad_stamp), sizeof(ag_quad_stamp)) != hipSuccess) return AGDIFF_ERR_LAUNCH;
  return hipMemcpyToSymbol(HIP_SYMBOL(ag_quad_stamp), zero, sizeof(zero)) == hipSuccess ? AGDIFF_OK : AGDIFF_ERR_LAUNCH;
}
#endif

namespace {

template <int I, int N, typename F>
__device__ __forceinline__ void ag_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ag_static_for<I + 1, N>(f);
  }
}

// ------------------------------------------------------------------------------ CFConv by filter polynomials, per target
struct NodeConvArgs {
  const float* poly_rad;      // pk [12][NKT]: filter polynomials of conv1 (channel tiles 0..7) and conv2 (8..11), bias included
  const float* poly_typed;    // [num_slots] x pk [12][NKT]: the same per local edge type, or null
  const int32_t* type_slot;   // [100] edge type -> set
  int32_t num_slots;          // 0: no local tiles in this launch
  int32_t lds_slots;          // typed sets 0..lds_slots-1 are copied to LDS, the others are read from L2
  // radius rows, AGDIFF_RAD_STRIDE per target (agdiff_ws_t.rad_*)
  const int32_t* rad_cnt;
  const int32_t* rad_src;
  const float* rad_len;
  const float* r_scale1;      // lw(d)*C(d) of conv1 / conv2 of this block by radius row
  const float* r_scale2;
  // local quad tiles (agdiff_topo_t.lt_*, agdiff_ws_t.lt_*)
  const int32_t* quad_tgt;    // [4 Q]: the four targets of a quad (-1: none)
  const int32_t* lt_ptr;
  const int32_t* lt_src;
  const int32_t* lt_type;
  const float* lt_len;
  const float* l_scale1;
  const float* l_scale2;
  const float* xs;            // [N][192]
  float* agg;                 // [N][192]
  int32_t n;                  // N
  const int32_t* wg_ptr;      // topo->quad_wg_ptr (k_cfconv_quad on a grid of 256) or null
  int32_t num_quads;          // Q
  int32_t qshift;             // a local tile's rows of lane quarter q belong to the quad's (q >> qshift)-th target (GT = 4 >> qshift)
  float two_over_rc;
  float unscale;              // conv[k].filt_poly_unscale: the coefficient sets carry its inverse
};

// encoder/schnet.py:136-162 for conv1 and conv2 of one InteractionBlock, filters from d-polynomials:
//   W_e = nn(MLPEdgeEncoder(d_e, type_e)) = P_type(d_e);   agg[dst] += x[src] * W_e * (lw(d_e) C(d_e)).
// One wave owns a QUAD of targets (topo->quad_tgt: four atoms of one molecule whose local in-lists need like tiles) and
// walks, in this order, the quad's local tiles (static, topo->lt_*: 16 rows of ONE edge type, rows 4 k .. 4 k + 3 = in-edges
// of the quad's k-th target) and the radius tiles of its first, second, third and fourth target (every 16-row tile of the
// radius rows belongs to one target, ws->rad_*).  EVERY tile runs the same body: the K = 32 NKT polynomial features of each
// row, SCALED by the row's lw C (one set per conv: the per-edge scale rides through the MFMAs), times the coefficient
// blocks of the tile's set (flipped product: rows = edges, lanes = channels), then x[src] gathered per (row, channel) and
//   sum[channel tile] += sum_r z[r] x[r]  -- four FMAs per channel tile, no masks, no list bounds, no carries between waves:
// a lane's four rows (4 q + r) always belong to one target.  Radius tiles add into acc (all four quarters = the current
// target), local tiles into accL (quarter k = the quad's k-th target).  When a target's radius tiles are done, the sums
// over the wave's quarters are taken once (reduce-scatter over the quarters, three lane swaps per four channel tiles) --
// with accL entering from quarter k only (as the start value of that quarter's sums) -- and the target's row of agg is written once, complete (zeros for a target
// without edges): no agg_first, no second aggregate for the node stage to add, no atomics, fixed order => bitwise
// reproducible.  Coefficient sets: the radius edges' one and the first lds_slots typed ones in LDS, rarer ones from L2.
// Everything a tile needs from memory (sources, lengths, the two scales, the type; then the first x groups) is requested
// during the wave's previous tile, the features are evaluated there too; x groups are double-buffered inside a tile.
// Lengths beyond the cutoff are clamped into the fitted range: their CFConv scale is exactly 0 (schnet.py:140-146).
// A local tile whose type has no polynomial (mixed batches: agdiff_local_poly_enabled = 2) runs with scale 0: its edges
// go through agdiff_cfconv_local.
// Channel tiles per x / MFMA group and waves per workgroup (= per CU) of an instantiation.  At one k-tile, groups of TWO channel
// tiles need 135 VGPRs (one-pass plan; groups of three: 156): under the 128-VGPR cap of FOUR waves per SIMD that leaves 5..7
// spilled registers (reloaded once per local tile, none in the radius loop) and takes 15 % less time than three waves per SIMD
// at groups of three (six launches on 196 k atoms: 3.53 -> 3.00 ms; groups of two at three waves: 3.83; groups of three at four
// waves, 27 spills: 3.32).  Two k-tiles hold twice the coefficient registers and stay at 168 VGPRs, three waves per SIMD: groups
// of two spill 28 registers at 16 waves (1.14 against 1.01 ms per launch); ONE channel tile per group fits with 5 spills under the
// one-pass plan and wins 4 % on a 36-molecule batch but LOSES 6 % on the default job's batches (0.976 against 0.920 ms).
// Small launches (fewer than two quads per wave of a full grid: tune_cfconv_four_min_quads) keep the 12-wave shape: more
// workgroups for the same quads (23 k atoms: even; 4 k atoms: 12 waves 3 % ahead).  -DAG_NODE_GRP / -DAG_NODECONV_WAVES force one
// shape on every instantiation (A/B builds).
template <int NKT, bool FOUR, int PLAN = 0>
struct NodeConvShape {
#ifdef AG_NODE_GRP
  static constexpr int GRP = AG_NODE_GRP;
#else
  static_assert(!FOUR || NKT == 1, "four waves per SIMD: one k-tile");
  static constexpr int GRP = (FOUR || NKT >= 3) ? 2 : 3;
#endif
#ifdef AG_NODECONV_WAVES
  static constexpr int WAVES = AG_NODECONV_WAVES;
#else
  static constexpr int WAVES = FOUR ? 16 : NKT >= 3 ? 8 : 12;       // (three / four k-tiles: 256 registers per lane)
  static_assert(PLAN >= 0 && PLAN <= 3, "poly_plan");
#endif
};
#ifndef AG_NODE_XD
#define AG_NODE_XD 2                        // x groups in flight (ring of buffers; must divide the number of groups: static indices)
#endif
#ifndef AG_NODE_XD_FOUR
#define AG_NODE_XD_FOUR AG_NODE_XD          // ... of the four-waves-per-SIMD shape (six groups of two channel tiles: 2 or 3)
#endif
#ifndef AG_QUAD_META_NT
#define AG_QUAD_META_NT 0     // radius rows' inputs of k_cfconv_quad: 1 non-temporal loads (as k_cfconv_node), 0 cached
#endif
#ifndef AG_QUAD_WAHEAD
#define AG_QUAD_WAHEAD 0       // k_cfconv_quad: 1 = coefficient blocks read one group ahead of their MFMAs (measured: no gain, 2 spills)
#endif
#ifndef AG_QUAD_FEATURES2
#define AG_QUAD_FEATURES2 1    // k_cfconv_quad: both feature sets of a tile from one routine (ag_poly_features2_mixed)
#endif
#ifndef AG_QUAD_STORE_NT
#define AG_QUAD_STORE_NT 1
#endif
#ifndef AG_QUAD_DYNAMIC
#define AG_QUAD_DYNAMIC 1      // k_cfconv_quad: a workgroup's quads dealt to its waves as they finish (0: quad p_begin + wave, + WAVES, ...)
#endif
// PLAN (agdiff_params_t.poly_plan): 0 three passes for every term; 1 one pass for the high terms, whose coefficients the
// host has bounded -- at NKT 1 two MFMAs per channel tile (hi x hi of all 32 terms, then both cross terms of terms 0..15 in
// one instruction: ag_poly_features<.., true> / the mixed unit 1 of the blocks), at NKT >= 2 every k-tile but the first by its hi x hi
// pass alone; 2 (NKT >= 3) / 3 (NKT 4): the same from k-tile 2 / 3 on -- sharp networks, whose terms 32..63 (..95) still carry weight.
// (k-tile t of a set takes all passes of the split arithmetic, or the hi x hi pass alone)
__host__ __device__ constexpr bool ag_plan_full(int plan, int t) { return plan == 0 || t < plan; }

template <int MODE, int NKT, int WAVES, int PLAN, int GRP>
__global__ void __launch_bounds__(64 * WAVES, WAVES / 4) k_cfconv_node(NodeConvArgs a) {
  static_assert(PLAN == 0 || MODE != AG_F32, "poly_plan needs a split mode");
  constexpr bool MIXED = PLAN == 1 && NKT == 1;           // unit 1 of a block / the operand's `lo` hold the mixed halves
  extern __shared__ u32x4 ag_nodeconv_smem[];
  lds_u32x4* wl = (lds_u32x4*)ag_nodeconv_smem;
  constexpr int SET = AG_CONV_NCH * NKT * 128;          // 16-byte units per coefficient set
  constexpr int NG = AG_CONV_NCH / GRP;
  ag_copy_lds(wl, reinterpret_cast<const u32x4*>(a.poly_rad), SET);
  if (a.lds_slots > 0) ag_copy_lds(wl + SET, reinterpret_cast<const u32x4*>(a.poly_typed), a.lds_slots * SET);
  __syncthreads();
  int lane = ag_lane();
  asm volatile("" : "+v"(lane));
  const int q = lane >> 4, col = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wg = (gridDim.x % 8 == 0) ? (int)((blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8) : (int)blockIdx.x;
  const int per_wg = (a.num_quads + (int)gridDim.x - 1) / (int)gridDim.x;
  const int p_begin = wg * per_wg;
  const int p_end = (p_begin + per_wg < a.num_quads) ? p_begin + per_wg : a.num_quads;
  const bool with_local = a.num_slots > 0;

  // A quad's description in plain scalars (a struct of them ended up as a dynamically indexed stack object in scratch):
  // nL local tiles from lt0, targets t0..t3 with n0..n3 radius tiles.
#define AG_QUAD_DECL(P) int P##nL = 0, P##lt0 = 0, P##t0 = -1, P##t1 = -1, P##t2 = -1, P##t3 = -1, P##n0 = 0, P##n1 = 0, P##n2 = 0, P##n3 = 0
#define AG_QUAD_LOAD(P, p)                                                              \
  do {                                                                                  \
    P##t0 = a.quad_tgt[4 * (p)];                                                        \
    P##t1 = a.quad_tgt[4 * (p) + 1];                                                    \
    P##t2 = a.quad_tgt[4 * (p) + 2];                                                    \
    P##t3 = a.quad_tgt[4 * (p) + 3];                                                    \
    P##n0 = tiles_of(P##t0);                                                            \
    P##n1 = tiles_of(P##t1);                                                            \
    P##n2 = tiles_of(P##t2);                                                            \
    P##n3 = tiles_of(P##t3);                                                            \
    P##lt0 = with_local ? a.lt_ptr[(p)] : 0;                                            \
    P##nL = with_local ? a.lt_ptr[(p) + 1] - P##lt0 : 0;                                \
  } while (0)
#define AG_QUAD_ARGS(P) P##nL, P##lt0, P##t0, P##t1, P##t2, P##t3, P##n0, P##n1, P##n2
  auto tiles_of = [&](int t) { return (t >= 0) ? (a.rad_cnt[t] + AG_TW - 1) / AG_TW : 0; };
  // first row of tile j of a quad (order: local tiles, radius tiles of its first .. fourth target)
  auto tile_rows = [&](int nL, int lt0, int t0, int t1, int t2, int t3, int n0, int n1, int n2, int j, bool& local) -> int {
    local = j < nL;
    if (local) return (lt0 + j) * AG_TW;
    j -= nL;
    int t = t0;
    if (j >= n0) {
      j -= n0;
      t = t1;
      if (j >= n1) {
        j -= n1;
        t = t2;
        if (j >= n2) {
          j -= n2;
          t = t3;
        }
      }
    }
    return t * AGDIFF_RAD_STRIDE + j * AG_TW;
  };
  // per-row inputs of the wave's NEXT tile: length and the two scales of row `col`, the tile's type slot, the sources of the
  // lane's four rows 4 q .. 4 q + 3
  float pf_d = 0.f, pf_s1 = 0.f, pf_s2 = 0.f;
  int pf_slot = -1;
  int pf_src[4] = {0, 0, 0, 0};
  // (uniform base pointer + 32-bit lane offset: the saddr form of global_load; 64-bit lane pointers per array cost a register
  // pair each and spilled)
  auto ldf = [](const float* base, uint32_t byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
  auto ldi = [](const int32_t* base, uint32_t byte_off) { return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(base) + byte_off); };
  // (the radius rows are written by the front kernel and read once per launch: non-temporal, so that they do not evict x rows)
  [[maybe_unused]] auto ldf_nt = [](const float* base, uint32_t byte_off) { return __builtin_nontemporal_load(reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off)); };
  auto prefetch_meta = [&](int rows, bool local) {
    const uint32_t e4 = (uint32_t)(rows + col) * 4u;
    const uint32_t r16 = (uint32_t)(rows + 4 * q) * 4u;
    u32x4 s4;
    if (!local) s4 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.rad_src) + r16));
    else s4 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.lt_src) + r16);
#pragma unroll
    for (int r = 0; r < 4; ++r) pf_src[r] = (int)s4[r];
    if (local) {
      const int sl = ldi(a.type_slot, (uint32_t)ldi(a.lt_type, e4) * 4u);  // (the same for the 16 rows of a tile)
      pf_slot = sl < 0 ? -2 : sl;                                           // (-2: a type without a polynomial)
      pf_d = ldf(a.lt_len, e4);
      pf_s1 = ldf(a.l_scale1, e4);
      pf_s2 = ldf(a.l_scale2, e4);
    } else {
      pf_slot = -1;
      pf_d = ldf_nt(a.rad_len, e4);
      pf_s1 = ldf_nt(a.r_scale1, e4);
      pf_s2 = ldf_nt(a.r_scale2, e4);
    }
  };
  // x[src] values of a group of GRP channel tiles, two groups in flight
  constexpr int XD = (GRP == 2) ? AG_NODE_XD_FOUR : AG_NODE_XD;
  static_assert((AG_CONV_NCH / GRP) % XD == 0 && XD >= 2, "ring of x buffers");
  f32x4 xg[XD][GRP];
  uint32_t xoff[4];
  auto set_xoff = [&]() {
#pragma unroll
    for (int r = 0; r < 4; ++r) xoff[r] = __umul24((uint32_t)pf_src[r], 768u) + (uint32_t)col * 4u;      // (v_mad_u32_u24: full rate.  N * 768 < 2^32: topology.py)
  };
  auto fetch_xg = [&](auto BUF, int g) {
    constexpr int kb = decltype(BUF)::value;
    const char* xb = reinterpret_cast<const char*>(a.xs);
#pragma unroll
    for (int j = 0; j < GRP; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xg[kb][j][r] = *reinterpret_cast<const float*>(xb + (size_t)xoff[r] + 64 * (GRP * g + j));
      }
    }
  };
  // the first XD - 1 x groups of a tile (requested before the tile starts)
  auto fetch_first_groups = [&]() {
    ag_static_for<0, XD - 1>([&](auto G) { fetch_xg(G, decltype(G)::value); });
  };
  const lds_u32x4* wl_l = wl + lane;
  // CN channel tiles C0 .. C0 + CN - 1 of one coefficient set (pk [12][NKT]: block nt * NKT + t) times the features:
  // independent accumulator chains with their MFMA passes interleaved; z starts from zero (the first MFMA takes the literal 0)
  auto mma_tiles = [&](auto base, auto C0_, const AgIn<MODE> (&ph)[NKT], auto& z) {
    constexpr int C0 = decltype(C0_)::value;
    constexpr int CN = sizeof(z) / sizeof(f32x4);
    u32x4 w[CN][NKT][2];
#pragma unroll
    for (int j = 0; j < CN; ++j) {
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        w[j][t][0] = base[(((C0 + j) * NKT + t) * 2) * 64];
        if (ag_plan_full(PLAN, t)) w[j][t][1] = base[(((C0 + j) * NKT + t) * 2 + 1) * 64];
      }
    }
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      const int parts = MIXED ? 2 : ag_plan_full(PLAN, t) ? AgParts<MODE>::n : 1;
#pragma unroll
      for (int part = 0; part < parts; ++part) {
#pragma unroll
        for (int j = 0; j < CN; ++j) {
          if constexpr (MIXED) {
            if (part == 1) {
              ag_block_mma_mixed<MODE, true>(z[j], ph[0], w[j][0]);
              continue;
            }
          }
          if (t == 0 && part == 0) z[j] = ag_block_mma_first<MODE, true>(ph[0], w[j][0]);
          else ag_block_mma_part<MODE, true>(z[j], ph[t], w[j][t], part);
        }
      }
    }
  };

  // features of the wave's next tile (its rows' inputs are in pf_*): channel tiles 0..7 are conv1 (features x its lw C),
  // 8..11 conv2.  A local tile whose type has no polynomial contributes nothing (scale 0)
  AgIn<MODE> ph1[NKT], ph2[NKT];
  auto next_features = [&]() {
    const bool dead = with_local && pf_slot < -1;
    const float s1 = dead ? 0.0f : pf_s1, s2 = dead ? 0.0f : pf_s2;
    ag_poly_features<MODE, NKT, MIXED>(pf_d, a.two_over_rc, q, ph1, s1);
    ag_poly_features<MODE, NKT, MIXED>(pf_d, a.two_over_rc, q, ph2, s2);
  };
  float acc[AG_CONV_NCH], accL[AG_CONV_NCH];
  // the sums over the wave's quarters, once per target: quarter j of a reduce-scatter ends up with channel tile 4 g + j.
  // `k`: the target's place in its quad -- its local rows are the quarters q with q >> qshift == k of the quad's local tiles
  auto finalize = [&](int tgt, int k) {
    char* dp = reinterpret_cast<char*>(a.agg + (size_t)tgt * 192);       // (uniform)
#pragma unroll
    for (int g = 0; g < AG_CONV_NCH / 4; ++g) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[4 * g + j];
      // (one 768-byte row per node, read once by the node stage: streamed past the caches like the radius rows)
      __builtin_nontemporal_store(a.unscale * ag_quarter_reduce_scatter4(v[0], v[1], v[2], v[3]), reinterpret_cast<float*>(dp + (uint32_t)(16 * (4 * g + q) + col) * 4u));
    }
  };

  int p = p_begin + wave;
  if (p >= p_end) return;                       // (no barrier below)
  AG_QUAD_DECL(c_);                             // the wave's current quad
  AG_QUAD_LOAD(c_, p);
  bool have_pf = false;
  while (p < p_end) {
    const int pn = p + WAVES;
    AG_QUAD_DECL(x_);                           // ... and its next one
    if (pn < p_end) AG_QUAD_LOAD(x_, pn);
    const int ntiles = c_nL + c_n0 + c_n1 + c_n2 + c_n3;
    const int ntiles_next = x_nL + x_n0 + x_n1 + x_n2 + x_n3;
    if (ntiles > 0 && !have_pf) {               // cold start (first quad of the wave, or the quad before had no tile)
      bool loc;
      const int rows = tile_rows(AG_QUAD_ARGS(c_), 0, loc);
      prefetch_meta(rows, loc);
      set_xoff();
      fetch_first_groups();
      next_features();
    }
#pragma unroll
    for (int i = 0; i < AG_CONV_NCH; ++i) accL[i] = 0.0f;
    // the wave's next tile after tile j (of this quad, or the first one of its next quad)
    auto next_tile = [&](int j, bool& nloc, bool& has_next) -> int {
      has_next = true;
      nloc = false;
      if (j + 1 < ntiles) return tile_rows(AG_QUAD_ARGS(c_), j + 1, nloc);
      if (ntiles_next > 0) return tile_rows(AG_QUAD_ARGS(x_), 0, nloc);
      has_next = false;
      return 0;
    };
    // One tile as a software pipeline over its four groups of three channel tiles: the MFMAs of group g + 1 are issued BEFORE
    // the sums of group g (sum += z x), so that the matrix pipe works while the wave's VALU does the sums; the x values of
    // group g + 2 are requested into the buffer the sums have just freed; the next tile's per-row inputs are requested at the
    // start, its first x group and -- behind the last group's MFMAs -- its features (ph1 / ph2 are carried from tile to
    // tile) at the end.  `base`: the tile's coefficient set (LDS or global), `S`: the sums it adds to.
    auto tile = [&](int j, auto base, float (&S)[AG_CONV_NCH]) {
      bool nloc, has_next;
      const int nrows = next_tile(j, nloc, has_next);
      if (has_next) prefetch_meta(nrows, nloc);
      auto mma_g = [&](auto GG, f32x4 (&z)[GRP]) {
        constexpr int c0 = GRP * decltype(GG)::value;
        // channel tiles 0..7 take conv1's features, 8..11 conv2's (a group of three straddles the boundary once: 6, 7 | 8)
        if constexpr (c0 + GRP <= 8) {
          mma_tiles(base, std::integral_constant<int, c0>{}, ph1, z);
        } else if constexpr (c0 >= 8) {
          mma_tiles(base, std::integral_constant<int, c0>{}, ph2, z);
        } else {
          static_assert(GRP == 3 && c0 == 6, "group layout");
          f32x4 (&za)[2] = *reinterpret_cast<f32x4 (*)[2]>(&z[0]);
          f32x4 (&zb)[1] = *reinterpret_cast<f32x4 (*)[1]>(&z[2]);
          mma_tiles(base, std::integral_constant<int, 6>{}, ph1, za);
          mma_tiles(base, std::integral_constant<int, 8>{}, ph2, zb);
        }
      };
      auto sums = [&](auto GG, const f32x4 (&z)[GRP]) {
        constexpr int gg = decltype(GG)::value;
#pragma unroll
        for (int jj = 0; jj < GRP; ++jj) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            S[GRP * gg + jj] = fmaf(z[jj][r], xg[gg % XD][jj][r], S[GRP * gg + jj]);
          }
          // (pins the sum to this step: the optimiser otherwise sinks all 48 FMAs of a tile below its last MFMA -- nothing
          // needs the sums before the target is complete -- and the wave then waits for x loads and MFMAs with nothing to do)
          asm volatile("" : "+v"(S[GRP * gg + jj]));
        }
      };
      f32x4 z[2][GRP];
      // (fences between the steps: the scheduler otherwise hoists every group's coefficient reads to the top of the tile and
      // spills; inside a step it is free to run the sums beside the MFMAs)
      fetch_xg(std::integral_constant<int, XD - 1>{}, XD - 1);
      mma_g(std::integral_constant<int, 0>{}, z[0]);
      __builtin_amdgcn_sched_barrier(0);
      ag_static_for<1, NG>([&](auto G) {
        constexpr int g = decltype(G)::value;
        mma_g(G, z[g & 1]);
        sums(std::integral_constant<int, g - 1>{}, z[(g - 1) & 1]);
        // the buffer the sums have just freed takes the group XD - 1 steps ahead: of this tile, or -- once all of this tile's
        // gathers are out and xoff is free -- of the wave's next tile (whose features follow the last request)
        if constexpr (g + XD - 1 < NG) {
          fetch_xg(std::integral_constant<int, (g - 1) % XD>{}, g + XD - 1);
        } else if (has_next) {
          if constexpr (g + XD - 1 == NG) set_xoff();
          fetch_xg(std::integral_constant<int, (g - 1) % XD>{}, g + XD - 1 - NG);
          if constexpr (g == NG - 1) next_features();
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      sums(std::integral_constant<int, NG - 1>{}, z[(NG - 1) & 1]);
      have_pf = has_next;
    };
    int j = 0;
#ifndef AG_NODE_NO_LOCAL        // (timing experiment: the kernel without its local tiles)
    for (; j < c_nL; ++j) {
      // the tile's set: one of the LDS-resident ones, or -- a rare type -- straight from L2; a type without a polynomial ran
      // with scale 0 (next_features) against the radius set
      const int slot = __builtin_amdgcn_readfirstlane(pf_slot);
      if (slot >= a.lds_slots) tile(j, reinterpret_cast<const u32x4*>(a.poly_typed) + (size_t)slot * SET + lane, accL);
      else tile(j, wl_l + (size_t)(slot >= 0 ? 1 + slot : 0) * SET, accL);
    }
#endif
    // (the four targets one after the other through shifting copies: indexing the quad's fields by k would put it in scratch)
    int ta = c_t0, tb = c_t1, tc = c_t2, td = c_t3, na = c_n0, nb = c_n1, nc = c_n2, nd = c_n3;
#pragma nounroll
    for (int k = 0; k < 4; ++k) {
      if (ta >= 0) {
        // the target's sums start from its local rows' (the quarters q with q >> qshift == k of the quad's local tiles)
        const bool mine = (q >> a.qshift) == k;
#pragma unroll
        for (int i = 0; i < AG_CONV_NCH; ++i) acc[i] = mine ? accL[i] : 0.0f;
        for (int u = 0; u < na; ++u, ++j) tile(j, wl_l, acc);
        finalize(ta, k);                        // the quad's k-th target is complete: write it
      }
      ta = tb, tb = tc, tc = td, td = -1;
      na = nb, nb = nc, nc = nd, nd = 0;
    }
    if (ntiles == 0) have_pf = false;
    p = pn;
    c_nL = x_nL, c_lt0 = x_lt0, c_t0 = x_t0, c_t1 = x_t1, c_t2 = x_t2, c_t3 = x_t3, c_n0 = x_n0, c_n1 = x_n1, c_n2 = x_n2, c_n3 = x_n3;
  }
#undef AG_QUAD_DECL
#undef AG_QUAD_LOAD
#undef AG_QUAD_ARGS
}


// ------------------------------------------------------------------------------ the same CFConv, radius rows in QUAD tiles too
// k_cfconv_node gives every target its own radius tiles (ceil(cnt / 16) of them, the last one padded: 24 % of the radius rows a
// launch of the default job executes are pads) and keeps two sets of sums (a target's, the quad's local rows').  Here a radius
// tile is laid out like a local one: quarter k of tile t holds rows 4 t .. 4 t + 3 of the quad's k-th target (the target's own
// rows AGDIFF_RAD_STRIDE i + 4 t + r of ws->rad_*: the memory layout does not change, only which rows a wave puts into one MFMA
// tile), a quad walks max_k ceil(cnt_k / 4) radius tiles (-8 % on the default job: counted by tools/tile_layouts.py), every lane
// keeps ONE set of sums -- its quarter's target, local and radius rows alike -- and a finished quad is written without any
// exchange between the quarters: lane (q, col) stores channel 16 ct + col of target q.  Rows 4 t + r >= cnt_k run with scale 0
// (the front kernel pads a target's rows only to the end of the last 16-row tile it used: what lies beyond is older, finite
// data of the same molecule).  Needs quads (topo->group_targets == 4); summation order per target: its local rows by tile,
// then its radius rows by source -- fixed, hence bitwise reproducible, but not the order of k_cfconv_node.
template <int MODE, int NKT, int WAVES, int PLAN, int GRP>
__global__ void __launch_bounds__(64 * WAVES, WAVES / 4) k_cfconv_quad(NodeConvArgs a) {
  static_assert(PLAN == 0 || MODE != AG_F32, "poly_plan needs a split mode");
  constexpr bool MIXED = PLAN == 1 && NKT == 1;
  extern __shared__ u32x4 ag_nodeconv_smem[];
  lds_u32x4* wl = (lds_u32x4*)ag_nodeconv_smem;
  constexpr int SET = AG_CONV_NCH * NKT * 128;
  constexpr int NG = AG_CONV_NCH / GRP;
  ag_copy_lds(wl, reinterpret_cast<const u32x4*>(a.poly_rad), SET);
  if (a.lds_slots > 0) ag_copy_lds(wl + SET, reinterpret_cast<const u32x4*>(a.poly_typed), a.lds_slots * SET);
#if AG_QUAD_DYNAMIC
  // the workgroup's quads are DEALT to its waves as they finish (one LDS counter): a quad has 5 to 13 tiles, and with the static
  // deal (quad p_begin + wave, + WAVES, ...) the slowest wave of a workgroup walked 18..35 % more tiles than the average one
  // (counted on the default job's batches).  A quad's result does not depend on the wave that computes it.
  typedef __attribute__((address_space(3))) int lds_ctr_t;
  lds_ctr_t* next_quad = reinterpret_cast<lds_ctr_t*>(wl + (size_t)(1 + (a.lds_slots > 0 ? a.lds_slots : 0)) * SET) + WAVES * 32;
  if (threadIdx.x == 0) *next_quad = WAVES;
#endif
  // edge type -> coefficient set, in LDS: a local tile's set is looked up when its features are made, by an LDS read.  (As a second,
  // dependent global load in the tile's prefetch it had to wait for the type to arrive -- and with it, in order, for every gather
  // in flight: a local tile cost 1.5 x a radius tile, tools/quad_stamps.py.)
  typedef __attribute__((address_space(3))) int lds_tab_t;
  lds_tab_t* tslot = reinterpret_cast<lds_tab_t*>(wl + (size_t)(1 + (a.lds_slots > 0 ? a.lds_slots : 0)) * SET) + WAVES * 32 + 16;
  if (threadIdx.x < 100) tslot[threadIdx.x] = a.num_slots > 0 ? a.type_slot[threadIdx.x] : -1;
  __syncthreads();
  int lane = ag_lane();
  asm volatile("" : "+v"(lane));
  const int q = lane >> 4, col = lane & 15;
  const int cq = col >> 2, cr = col & 3;              // the lane's row `col` of a tile: quarter cq, row cr of it
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wg = (gridDim.x % 8 == 0) ? (int)((blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8) : (int)blockIdx.x;
  const int per_wg = (a.num_quads + (int)gridDim.x - 1) / (int)gridDim.x;
  // (the host's ranges of like tile counts where it made them, else equal quad counts)
  const bool ranges = a.wg_ptr != nullptr && gridDim.x == 256;
  const int p_begin = ranges ? a.wg_ptr[wg] : wg * per_wg;
  const int p_end = ranges ? a.wg_ptr[wg + 1] : (p_begin + per_wg < a.num_quads) ? p_begin + per_wg : a.num_quads;
  const bool with_local = a.num_slots > 0;

  // a quad: nL local tiles from lt0 and nR radius tiles in scalars.  What differs by quarter -- the first radius row of the
  // quarter's target (a missing target borrows the quad's first: its count is 0), its number of radius rows, its row of agg -- lives
  // in a few wave-private LDS words (two slots: the current quad's and the next one's), written once per quad by lanes 0..3 and
  // read back per tile with the quarter as the index: as per-lane registers these values were spilled around every tile at the
  // 128-VGPR cap, and a scratch reload waits for every outstanding gather
  typedef __attribute__((address_space(3))) int lds_int;
  lds_int* qi = reinterpret_cast<lds_int*>(wl + (size_t)(1 + (a.lds_slots > 0 ? a.lds_slots : 0)) * SET) + wave * 32;
#define AG_RQ_DECL(P) int P##nL = 0, P##lt0 = 0, P##nR = 0
#define AG_RQ_LOAD(P, p, slot)                                                          \
  do {                                                                                  \
    const int t0_ = a.quad_tgt[4 * (p)], t1_ = a.quad_tgt[4 * (p) + 1], t2_ = a.quad_tgt[4 * (p) + 2], t3_ = a.quad_tgt[4 * (p) + 3]; \
    const int c0_ = cnt_of(t0_), c1_ = cnt_of(t1_), c2_ = cnt_of(t2_), c3_ = cnt_of(t3_); \
    const int m01 = c0_ > c1_ ? c0_ : c1_, m23 = c2_ > c3_ ? c2_ : c3_;               \
    P##nR = ((m01 > m23 ? m01 : m23) + 3) >> 2;                                         \
    P##lt0 = with_local ? a.lt_ptr[(p)] : 0;                                            \
    P##nL = with_local ? a.lt_ptr[(p) + 1] - P##lt0 : 0;                                \
    if (lane < 4) {                                                                     \
      const int tk = lane == 0 ? t0_ : lane == 1 ? t1_ : lane == 2 ? t2_ : t3_;         \
      const int ck = lane == 0 ? c0_ : lane == 1 ? c1_ : lane == 2 ? c2_ : c3_;         \
      qi[(slot) * 16 + lane] = (tk < 0 ? t0_ : tk) * AGDIFF_RAD_STRIDE;                 \
      qi[(slot) * 16 + 4 + lane] = ck;                                                  \
      qi[(slot) * 16 + 8 + lane] = tk < 0 ? -1 : tk * 768;                              \
    }                                                                                   \
  } while (0)
  auto cnt_of = [&](int t) { return (t >= 0) ? a.rad_cnt[t] : 0; };
  // per-row inputs of the wave's NEXT tile: length and the two scales of row `col`, the tile's type slot (-1: radius rows),
  // whether the row is live, the sources of the lane's four rows 4 q .. 4 q + 3
  float pf_d = 0.f, pf_s1 = 0.f, pf_s2 = 0.f;
  int pf_slot = -1, pf_type = 0;
  bool pf_dead = false;
  int pf_src[4] = {0, 0, 0, 0};
  auto ldf = [](const float* base, uint32_t byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
  auto ldi = [](const int32_t* base, uint32_t byte_off) { return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(base) + byte_off); };
  [[maybe_unused]] auto ldf_nt = [](const float* base, uint32_t byte_off) { return __builtin_nontemporal_load(reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off)); };
  auto prefetch_local = [&](int rows) {
    const uint32_t e4 = (uint32_t)(rows + col) * 4u;
    const uint32_t r16 = (uint32_t)(rows + 4 * q) * 4u;
    const u32x4 s4 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.lt_src) + r16);
#pragma unroll
    for (int r = 0; r < 4; ++r) pf_src[r] = (int)s4[r];
    pf_type = ldi(a.lt_type, e4);          // (the same for the 16 rows of a tile; its set: resolve_slot, once it has landed)
    pf_slot = -3;
    pf_d = ldf(a.lt_len, e4);
    pf_s1 = ldf(a.l_scale1, e4);
    pf_s2 = ldf(a.l_scale2, e4);
  };
  // radius tile t of the quad in slot `slot`
  auto prefetch_radius = [&](int t, int slot) {
    const int rbq = qi[slot * 16 + q], rbc = qi[slot * 16 + cq], cnt = qi[slot * 16 + 4 + cq];
    const uint32_t e4 = (uint32_t)(rbc + 4 * t + cr) * 4u;
    const uint32_t r16 = (uint32_t)(rbq + 4 * t) * 4u;
#if AG_QUAD_META_NT
    const u32x4 s4 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.rad_src) + r16));
    pf_d = ldf_nt(a.rad_len, e4);
    pf_s1 = ldf_nt(a.r_scale1, e4);
    pf_s2 = ldf_nt(a.r_scale2, e4);
#else
    // (cached: a quarter's four rows are a quarter of a line, the next three tiles read the rest of it)
    const u32x4 s4 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.rad_src) + r16);
    pf_d = ldf(a.rad_len, e4);
    pf_s1 = ldf(a.r_scale1, e4);
    pf_s2 = ldf(a.r_scale2, e4);
#endif
#pragma unroll
    for (int r = 0; r < 4; ++r) pf_src[r] = (int)s4[r];
    pf_slot = -1;
    pf_dead = 4 * t + cr >= cnt;
  };
  constexpr int XD = (GRP == 2) ? AG_NODE_XD_FOUR : AG_NODE_XD;
  static_assert((AG_CONV_NCH / GRP) % XD == 0 && XD >= 2, "ring of x buffers");
  f32x4 xg[XD][GRP];
  uint32_t xoff[4];
  auto set_xoff = [&]() {
#pragma unroll
    for (int r = 0; r < 4; ++r) xoff[r] = __umul24((uint32_t)pf_src[r], 768u) + (uint32_t)col * 4u;      // (v_mad_u32_u24: full rate.  N * 768 < 2^32: topology.py)
  };
  auto fetch_xg = [&](auto BUF, int g) {
    constexpr int kb = decltype(BUF)::value;
    const char* xb = reinterpret_cast<const char*>(a.xs);
#pragma unroll
    for (int j = 0; j < GRP; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xg[kb][j][r] = *reinterpret_cast<const float*>(xb + (size_t)xoff[r] + 64 * (GRP * g + j));
      }
    }
  };
  auto fetch_first_groups = [&]() {
    ag_static_for<0, XD - 1>([&](auto G) { fetch_xg(G, decltype(G)::value); });
  };
  const lds_u32x4* wl_l = wl + lane;
  // The coefficient blocks of a group of GRP channel tiles (pk [12][NKT]: block ct * NKT + t) are read into `w` ONE GROUP AHEAD of
  // their MFMAs -- straight after the MFMAs of the group before have issued, into the registers those have just read -- so that the
  // LDS round trip runs beside the sums of the group before instead of in front of the MFMAs that wait for it.
  u32x4 w[GRP][NKT][2];
  auto load_w = [&](auto base, auto GG) {
    constexpr int C0 = GRP * decltype(GG)::value;
#pragma unroll
    for (int j = 0; j < GRP; ++j) {
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        w[j][t][0] = base[(((C0 + j) * NKT + t) * 2) * 64];
        if (ag_plan_full(PLAN, t)) w[j][t][1] = base[(((C0 + j) * NKT + t) * 2 + 1) * 64];
      }
    }
  };
  AgIn<MODE> ph1[NKT], ph2[NKT];
  // z[j] = features x block of channel tile C0 + j (channel tiles 0..7: conv1's features, 8..11: conv2's); independent accumulator
  // chains with their MFMA passes interleaved, the first MFMA of a chain takes the literal 0
  auto mma_w = [&](auto GG, f32x4 (&z)[GRP]) {
    constexpr int C0 = GRP * decltype(GG)::value;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      const int parts = MIXED ? 2 : ag_plan_full(PLAN, t) ? AgParts<MODE>::n : 1;
#pragma unroll
      for (int part = 0; part < parts; ++part) {
#pragma unroll
        for (int j = 0; j < GRP; ++j) {
          const AgIn<MODE> (&ph)[NKT] = (C0 + j < 8) ? ph1 : ph2;
          if constexpr (MIXED) {
            if (part == 1) {
              ag_block_mma_mixed<MODE, true>(z[j], ph[0], w[j][0]);
              continue;
            }
          }
          if (t == 0 && part == 0) z[j] = ag_block_mma_first<MODE, true>(ph[0], w[j][0]);
          else ag_block_mma_part<MODE, true>(z[j], ph[t], w[j][t], part);
        }
      }
    }
  };
  auto next_features = [&]() {
    {   // a local tile's coefficient set (-2: its type has none -- the tile runs with scale 0)
      const bool loc = pf_slot == -3;
      const int sl = tslot[loc ? ((uint32_t)pf_type < 100u ? pf_type : 0) : 0];
      pf_dead = loc ? sl < 0 : pf_dead;
      pf_slot = loc ? (sl < 0 ? -2 : sl) : pf_slot;
    }
    const float s1 = pf_dead ? 0.0f : pf_s1, s2 = pf_dead ? 0.0f : pf_s2;
    if constexpr (MIXED && AG_QUAD_FEATURES2) {
      ag_poly_features2_mixed<MODE>(pf_d, a.two_over_rc, q, ph1[0], s1, ph2[0], s2);
    } else {
      ag_poly_features<MODE, NKT, MIXED>(pf_d, a.two_over_rc, q, ph1, s1);
      ag_poly_features<MODE, NKT, MIXED>(pf_d, a.two_over_rc, q, ph2, s2);
    }
  };
  float acc[AG_CONV_NCH];

#ifdef AG_QUAD_STAMPS
  if (threadIdx.x == 0 && gridDim.x <= 256) ag_quad_stamp[2 * blockIdx.x] = wall_clock64();
#endif
  int p = p_begin + wave;
  if (p >= p_end) return;                       // (no barrier below)
  AG_RQ_DECL(c_);
  int cslot = 0;                                // (uniform) the LDS slot of the current quad
  AG_RQ_LOAD(c_, p, 0);
  bool have_pf = false;
  while (p < p_end) {
#if AG_QUAD_DYNAMIC
    int take = 0;
    if (lane == 0) take = __hip_atomic_fetch_add(next_quad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int pn = p_begin + __builtin_amdgcn_readfirstlane(take);
#else
    const int pn = p + WAVES;
#endif
    AG_RQ_DECL(x_);
    if (pn < p_end) AG_RQ_LOAD(x_, pn, cslot ^ 1);
    const int ntiles = c_nL + c_nR;
    const int ntiles_next = x_nL + x_nR;
    // requests the per-row inputs of tile j of the current quad (j < ntiles) or of the first tile of the next one
    // (ONE call of either kind whose arguments are selected by the uniform `nxt`: with a branch per quad the compiler hoisted the next
    // quad's loop-invariant row addresses out of the tile loop as nine 64-bit lane pointers and spilled)
    auto prefetch_tile = [&](int j) {
      const bool nxt = j >= ntiles;
      const int jj = nxt ? 0 : j, nL = nxt ? x_nL : c_nL;
      if (jj < nL) prefetch_local(((nxt ? x_lt0 : c_lt0) + jj) * AG_TW);
      else prefetch_radius(jj - nL, nxt ? (cslot ^ 1) : cslot);
    };
    if (ntiles > 0 && !have_pf) {               // cold start
      prefetch_tile(0);
      set_xoff();
      fetch_first_groups();
      next_features();
    }
#pragma unroll
    for (int i = 0; i < AG_CONV_NCH; ++i) acc[i] = 0.0f;
    // One tile: group 0's coefficient blocks are requested first, then the per-row inputs of the wave's next tile and the last x
    // group of this one; per group g: its MFMAs, the blocks of group g + 1, the sums of group g - 1 (sum += z x), the x values XD - 1
    // groups ahead -- of this tile or, once all of its gathers are out, of the next one, whose features follow the last request.
    // The last tile of a wave's last quad "prefetches" itself again (no branch, hence no join in front of the last sums at which
    // every outstanding load would have to land).
    auto tile = [&](int j, auto base) {
      const bool has_next = (j + 1 < ntiles) || ntiles_next > 0;
      if (AG_QUAD_WAHEAD) load_w(base, std::integral_constant<int, 0>{});
      prefetch_tile(has_next ? j + 1 : j);
      auto sums = [&](auto GG, const f32x4 (&z)[GRP]) {
        constexpr int gg = decltype(GG)::value;
#pragma unroll
        for (int jj = 0; jj < GRP; ++jj) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[GRP * gg + jj] = fmaf(z[jj][r], xg[gg % XD][jj][r], acc[GRP * gg + jj]);
          }
          asm volatile("" : "+v"(acc[GRP * gg + jj]));
        }
      };
      f32x4 z[2][GRP];
      fetch_xg(std::integral_constant<int, XD - 1>{}, XD - 1);
      if (!AG_QUAD_WAHEAD) load_w(base, std::integral_constant<int, 0>{});
      mma_w(std::integral_constant<int, 0>{}, z[0]);
      if (AG_QUAD_WAHEAD) load_w(base, std::integral_constant<int, 1>{});
      __builtin_amdgcn_sched_barrier(0);
      ag_static_for<1, NG>([&](auto G) {
        constexpr int g = decltype(G)::value;
        if (!AG_QUAD_WAHEAD) load_w(base, G);
        mma_w(G, z[g & 1]);
        if constexpr (g + 1 < NG) {
          if (AG_QUAD_WAHEAD) load_w(base, std::integral_constant<int, g + 1>{});
        }
        sums(std::integral_constant<int, g - 1>{}, z[(g - 1) & 1]);
        if constexpr (g + XD - 1 < NG) {
          fetch_xg(std::integral_constant<int, (g - 1) % XD>{}, g + XD - 1);
        } else {
          if constexpr (g + XD - 1 == NG) set_xoff();
          fetch_xg(std::integral_constant<int, (g - 1) % XD>{}, g + XD - 1 - NG);
          if constexpr (g == NG - 1) next_features();
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      sums(std::integral_constant<int, NG - 1>{}, z[(NG - 1) & 1]);
      have_pf = has_next;
    };
    int j = 0;
    for (; j < c_nL; ++j) {
      const int slot = __builtin_amdgcn_readfirstlane(pf_slot);
      if (slot >= a.lds_slots) tile(j, reinterpret_cast<const u32x4*>(a.poly_typed) + (size_t)slot * SET + lane);
      else tile(j, wl_l + (size_t)(slot >= 0 ? 1 + slot : 0) * SET);
    }
    for (; j < ntiles; ++j) tile(j, wl_l);
    // the quad is complete: lane (q, col) holds channel 16 ct + col of its quarter's target (zeros for a target without edges)
    const int aoff = qi[cslot * 16 + 8 + q];
    if (aoff >= 0) {
      char* dp = reinterpret_cast<char*>(a.agg) + (uint32_t)(aoff + col * 4);
#pragma unroll
      for (int ct = 0; ct < AG_CONV_NCH; ++ct)
#if AG_QUAD_STORE_NT
        __builtin_nontemporal_store(a.unscale * acc[ct], reinterpret_cast<float*>(dp + 64 * ct));
#else
        *reinterpret_cast<float*>(dp + 64 * ct) = a.unscale * acc[ct];
#endif
    }
    if (ntiles == 0) have_pf = false;
    p = pn;
    c_nL = x_nL, c_lt0 = x_lt0, c_nR = x_nR;
    cslot ^= 1;
  }
#ifdef AG_QUAD_STAMPS      // (the workgroup's LAST wave to pass here leaves the latest clock)
  if (lane == 0 && gridDim.x <= 256) atomicMax(&ag_quad_stamp[2 * blockIdx.x + 1], (unsigned long long)wall_clock64());
#endif
#undef AG_RQ_DECL
#undef AG_RQ_LOAD
}

template <int MODE, int NKT, int PLAN, bool FOUR, bool QUAD>
int launch_cfconv_node_s(const NodeConvArgs& a, size_t smem, void* stream) {
  using Shape = NodeConvShape<NKT, FOUR, PLAN>;
  constexpr int WAVES = Shape::WAVES;
  static std::atomic<uint64_t> attr_done{0};
  auto kern = QUAD ? k_cfconv_quad<MODE, NKT, WAVES, PLAN, Shape::GRP> : k_cfconv_node<MODE, NKT, WAVES, PLAN, Shape::GRP>;
  if (!ag_allow_big_lds(attr_done, (size_t)160 * 1024, kern)) return AGDIFF_ERR_LAUNCH;
  int64_t wgs = (a.num_quads + WAVES - 1) / WAVES;        // persistent: one workgroup per CU, quads dealt in contiguous ranges
  if (wgs > 256) wgs = 256;
  kern<<<dim3((unsigned)wgs), dim3(64 * WAVES), smem, (hipStream_t)stream>>>(a);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}
template <int MODE, int NKT, int PLAN>
int launch_cfconv_node_p(const NodeConvArgs& a, bool four, bool quad, size_t smem, void* stream) {
  if constexpr (NKT == 1) {
    if (four) return quad ? launch_cfconv_node_s<MODE, NKT, PLAN, true, true>(a, smem, stream)
                          : launch_cfconv_node_s<MODE, NKT, PLAN, true, false>(a, smem, stream);
  }
  return quad ? launch_cfconv_node_s<MODE, NKT, PLAN, false, true>(a, smem, stream)
              : launch_cfconv_node_s<MODE, NKT, PLAN, false, false>(a, smem, stream);
}
template <int MODE, int NKT>
int launch_cfconv_node_t(const NodeConvArgs& a, int plan, bool four, bool quad, size_t smem, void* stream) {
  if constexpr (MODE != AG_F32) {
    if (plan == 1) return launch_cfconv_node_p<MODE, NKT, 1>(a, four, quad, smem, stream);
    if constexpr (NKT >= 3) {
      if (plan == 2) return launch_cfconv_node_p<MODE, NKT, 2>(a, four, quad, smem, stream);
    }
    if constexpr (NKT >= 4) {
      if (plan == 3) return launch_cfconv_node_p<MODE, NKT, 3>(a, four, quad, smem, stream);
    }
  }
  return launch_cfconv_node_p<MODE, NKT, 0>(a, four, quad, smem, stream);
}
}  // namespace

extern "C" int agdiff_cfconv_node(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, int32_t k,
                                  void* stream) {
  if (!p || !topo || !ws || k < 0 || k >= p->num_convs) return AGDIFF_ERR_ARG;
  if (p->poly_kt < 1 || p->poly_kt > AGDIFF_POLY_MAX_KT || !p->conv[k].filt_poly_pk) return AGDIFF_ERR_ARG;
  if (p->poly_plan < 0 || p->poly_plan > 3 || (p->poly_plan && p->precision == AG_F32) || (p->poly_plan >= 2 && p->poly_kt <= p->poly_plan))
    return AGDIFF_ERR_ARG;
  if (!(p->conv[k].filt_poly_unscale > 0.0f)) return AGDIFF_ERR_ARG;
  if (!ws->rad_cnt || !ws->rad_src || !ws->rad_len || !ws->r_scale || !ws->xs || !ws->agg || !topo->quad_tgt ||
      topo->num_quads <= 0 || (topo->group_targets != 4 && topo->group_targets != 2 && topo->group_targets != 1))
    return AGDIFF_ERR_ARG;
  if (topo->num_nodes <= 0) return AGDIFF_OK;
  if (topo->num_nodes * (int64_t)AGDIFF_RAD_STRIDE >= (1ll << 31)) return AGDIFF_ERR_LIMIT;
  const bool local = topo->num_local > 0 && agdiff_local_poly_enabled(p, topo, ws) != 0;   // (1 all, 2 the slotted types' edges)
  if (local && !p->conv[k].filt_poly_typed_pk) return AGDIFF_ERR_ARG;
  NodeConvArgs a;
  a.poly_rad = p->conv[k].filt_poly_pk;
  a.poly_typed = local ? p->conv[k].filt_poly_typed_pk : nullptr;
  a.type_slot = p->poly_type_slot;
  a.num_slots = local ? p->poly_num_slots : 0;
  // coefficient sets in LDS: the radius edges' one, then as many typed ones as fit (5 of 24 KiB at poly_kt 1, 2 of 48 KiB at 2, 1 of 72 KiB
  // at 3, none beside the radius set's 96 KiB at 4)
  const size_t set_bytes = (size_t)AG_CONV_NCH * p->poly_kt * 2048;
  int max_sets = (int)(((size_t)160 * 1024) / set_bytes);
  if (p->tune_poly_lds_sets > 0 && p->tune_poly_lds_sets < max_sets) max_sets = p->tune_poly_lds_sets;
  a.lds_slots = a.num_slots < max_sets - 1 ? a.num_slots : max_sets - 1;
  a.rad_cnt = ws->rad_cnt;
  a.rad_src = ws->rad_src;
  a.rad_len = ws->rad_len;
  const size_t rpad = (size_t)topo->num_nodes * AGDIFF_RAD_STRIDE;
  a.r_scale1 = ws->r_scale + (size_t)(2 * k) * rpad;
  a.r_scale2 = ws->r_scale + (size_t)(2 * k + 1) * rpad;
  a.quad_tgt = topo->quad_tgt;
  a.lt_ptr = topo->lt_ptr;
  a.lt_src = topo->lt_src;
  a.lt_type = topo->lt_type;
  a.lt_len = ws->lt_len;
  const size_t tpad = (size_t)topo->num_local_tiles * AG_TW;
  a.l_scale1 = local ? ws->lt_scale + (size_t)(2 * k) * tpad : nullptr;
  a.l_scale2 = local ? ws->lt_scale + (size_t)(2 * k + 1) * tpad : nullptr;
  a.xs = ws->xs;
  a.agg = ws->agg;
  a.n = (int32_t)topo->num_nodes;
  a.wg_ptr = topo->quad_wg_ptr;
  a.num_quads = (int32_t)topo->num_quads;
  a.qshift = topo->group_targets == 4 ? 0 : topo->group_targets == 2 ? 1 : 2;
  a.two_over_rc = 2.0f / p->cutoff;
  a.unscale = p->conv[k].filt_poly_unscale;
  const size_t smem = (size_t)(1 + a.lds_slots) * set_bytes + 2048 + 64 + 448;  // (+ k_cfconv_quad's per-wave quad words, quad counter, type table)
  // shape: four waves per SIMD pay from two quads per wave of a full grid on (below, 12-wave workgroups spread the quads wider)
  const int64_t four_min = p->tune_cfconv_four_min_quads ? p->tune_cfconv_four_min_quads : 8192;
  const bool four = p->poly_kt == 1 && four_min >= 0 && a.num_quads >= four_min;
  // radius rows in quad tiles (k_cfconv_quad) wherever the topology has quads
  const bool quad = topo->group_targets == 4 && p->tune_cfconv_quad_tiles >= 0;
  ag_log_variant(ws, AGDIFF_VAR_CFCONV_NODE | (local ? AGDIFF_VAR_CFCONV_NODE_LOCAL : 0) |
                         (a.lds_slots < a.num_slots ? AGDIFF_VAR_POLY_L2_SETS : 0) | (four ? AGDIFF_VAR_CFCONV_NODE_FOUR : 0) |
                         (quad ? AGDIFF_VAR_CFCONV_NODE_QUAD : 0));
  const int plan = p->poly_plan;
  auto by_terms = [&](auto MODE_, int pl) {
    constexpr int MODE = decltype(MODE_)::value;
    switch (p->poly_kt) {
      case 1: return launch_cfconv_node_t<MODE, 1>(a, pl, four, quad, smem, stream);
      case 2: return launch_cfconv_node_t<MODE, 2>(a, pl, four, quad, smem, stream);
      case 3: return launch_cfconv_node_t<MODE, 3>(a, pl, four, quad, smem, stream);
      default: return launch_cfconv_node_t<MODE, 4>(a, pl, four, quad, smem, stream);
    }
  };
  if (p->precision == AG_H3) return by_terms(std::integral_constant<int, AG_H3>{}, plan);
  if (p->precision == AG_BF3) return by_terms(std::integral_constant<int, AG_BF3>{}, plan);
  return by_terms(std::integral_constant<int, AG_F32>{}, 0);
}

