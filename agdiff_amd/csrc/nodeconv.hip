// The two CFConvs of an InteractionBlock with their filters from d-polynomials, one wave per pair of targets
// (include/agdiff_hip.h: agdiff_cfconv_node).  Its own translation unit: built with -fno-slp-vectorize -- the SLP
// vectoriser turns the per-row accumulation FMAs into v_pk_fma_f32 fed by register shuffles (358 v_mov per kernel), and
// packed fp32 next to MFMAs costs issue time instead of saving it (MI355X_MICROARCH.md, per-instruction cycle constants).
#include "common.hpp"
#include <type_traits>

#define AG_CONV_NCH 12          // 16-channel tiles of the 192 filter channels (conv1: 0..7, conv2: 8..11)

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
  // local pair tiles (agdiff_topo_t.lt_*, agdiff_ws_t.lt_*)
  const int32_t* pair_tgt;    // [2 P]: the two targets of a pair (second: -1 for none)
  const int32_t* lt_ptr;
  const int32_t* lt_src;
  const int32_t* lt_type;
  const float* lt_len;
  const float* l_scale1;
  const float* l_scale2;
  const float* xs;            // [N][192]
  float* agg;                 // [N][192]
  int32_t n;                  // N
  int32_t num_pairs;          // ceil(N / 2)
  float two_over_rc;
};

#ifdef AG_NODE_STAMPS
// diagnostic build (make EXTRA=-DAG_NODE_STAMPS): where a wave's time goes inside a radius tile -- s_memtime deltas of the
// five steps summed over all waves ([0..4]), tiles stamped ([5]), s_memrealtime total ([6]) and s_memtime total ([7])
__device__ unsigned long long ag_node_stamp_acc[8];
#define AG_NSTAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define AG_NSTAMP(var) do { } while (0)
#endif

// encoder/schnet.py:136-162 for conv1 and conv2 of one InteractionBlock, filters from d-polynomials:
//   W_e = nn(MLPEdgeEncoder(d_e, type_e)) = P_type(d_e);   agg[dst] += x[src] * W_e * (lw(d_e) C(d_e)).
// One wave owns a PAIR of targets (topo->pair_tgt: two atoms of one molecule with like local in-lists) and walks, in this
// order, the pair's local tiles (rows 0..7 = in-edges of its first target, rows 8..15 = of its second; static,
// topo->lt_*), the radius tiles of the first, the radius tiles of the second
// (every 16-row tile of the radius list belongs to one target, ws->rad_*).  Per tile: the K = 32 NKT polynomial features
// of each row, SCALED by the row's lw C (one set per conv: the per-edge scale rides through the MFMAs), times the
// LDS-resident coefficient blocks (flipped product: rows = edges, lanes = channels), then x[src] gathered per (row,
// channel) and  acc[channel tile] += sum_r z[r] x[r]  -- four FMAs per channel tile, no masks, no list bounds, no
// carries between waves: a lane's four rows (4 q + r) always belong to one target.  When a target's tiles are done the
// sums over the wave's quarters are taken once (reduce-scatter over the quarters, three lane swaps per four channel
// tiles) and the target's row of agg is written once, complete (zeros for a target without edges): no agg_first, no second
// aggregate for the node stage to add, no atomics, fixed order => bitwise reproducible.
// Local tiles: rows of several types; the wave loops over the types present (typically three), each adding its masked
// features times its own coefficient set -- sets 0..lds_slots-1 from LDS, rarer ones straight from L2.
// Everything a tile needs from memory (sources, lengths, the two scales, type slots; then the first x group) is requested
// during the wave's previous tile; x groups are double-buffered inside a tile.
// Lengths beyond the cutoff are clamped into the fitted range: their CFConv scale is exactly 0 (schnet.py:140-146).
#ifndef AG_NODE_GRP
#define AG_NODE_GRP 3                       // channel tiles per x / MFMA group
#endif
#ifndef AG_NODE_XD
#define AG_NODE_XD 2                        // x groups in flight (ring of buffers; must divide the number of groups: static indices)
#endif
#ifndef AG_NODE_ABL
#define AG_NODE_ABL 0       // timing experiments only (wrong results): 1 no x gathers, 2 no filter MFMAs, 4 no features, 8 no sums
#endif
template <int MODE, int NKT, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, WAVES / 4) k_cfconv_node(NodeConvArgs a) {
  extern __shared__ u32x4 ag_nodeconv_smem[];
  lds_u32x4* wl = (lds_u32x4*)ag_nodeconv_smem;
  constexpr int SET = AG_CONV_NCH * NKT * 128;          // 16-byte units per coefficient set
  constexpr int NG = AG_CONV_NCH / AG_NODE_GRP;
  ag_copy_lds(wl, reinterpret_cast<const u32x4*>(a.poly_rad), SET);
  if (a.lds_slots > 0) ag_copy_lds(wl + SET, reinterpret_cast<const u32x4*>(a.poly_typed), a.lds_slots * SET);
  __syncthreads();
  int lane = ag_lane();
  asm volatile("" : "+v"(lane));
  const int q = lane >> 4, col = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wg = (gridDim.x % 8 == 0) ? (int)((blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8) : (int)blockIdx.x;
  const int per_wg = (a.num_pairs + (int)gridDim.x - 1) / (int)gridDim.x;
  const int p_begin = wg * per_wg;
  const int p_end = (p_begin + per_wg < a.num_pairs) ? p_begin + per_wg : a.num_pairs;
  const bool with_local = a.num_slots > 0;

  struct PairInfo { int nL, nA, nB, lt0, tA, tB; };
  auto pair_info = [&](int p) -> PairInfo {
    PairInfo r;
    r.tA = a.pair_tgt[2 * p];
    r.tB = a.pair_tgt[2 * p + 1];
    const int cA = a.rad_cnt[r.tA];
    const int cB = (r.tB >= 0) ? a.rad_cnt[r.tB] : 0;
    r.nA = (cA + AG_TW - 1) / AG_TW;
    r.nB = (cB + AG_TW - 1) / AG_TW;
    r.lt0 = with_local ? a.lt_ptr[p] : 0;
    r.nL = with_local ? a.lt_ptr[p + 1] - r.lt0 : 0;
    return r;
  };
  // first row of tile j of a pair (order: local tiles, radius tiles of target 2 p, radius tiles of target 2 p + 1)
  auto tile_rows = [&](const PairInfo& pi, int p, int j, bool& local) -> int {
    local = j < pi.nL;
    if (local) return (pi.lt0 + j) * AG_TW;
    j -= pi.nL;
    return (j < pi.nA) ? pi.tA * AGDIFF_RAD_STRIDE + j * AG_TW : pi.tB * AGDIFF_RAD_STRIDE + (j - pi.nA) * AG_TW;
  };
  // per-row inputs of the wave's NEXT tile: length, the two scales and the type slot of row `col`, the sources of the lane's
  // four rows 4 q .. 4 q + 3
  float pf_d = 0.f, pf_s1 = 0.f, pf_s2 = 0.f;
  int pf_slot = -1;
  int pf_src[4] = {0, 0, 0, 0};
  // (uniform base pointer + 32-bit lane offset: the saddr form of global_load; 64-bit lane pointers per array cost a register
  // pair each and spilled)
  auto ldf = [](const float* base, uint32_t byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
  auto ldi = [](const int32_t* base, uint32_t byte_off) { return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(base) + byte_off); };
  auto prefetch_meta = [&](int rows, bool local) {
    const uint32_t e4 = (uint32_t)(rows + col) * 4u;
    const uint32_t r16 = (uint32_t)(rows + 4 * q) * 4u;
    const int32_t* srcs = local ? a.lt_src : a.rad_src;
    const u32x4 s4 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(srcs) + r16);
#pragma unroll
    for (int r = 0; r < 4; ++r) pf_src[r] = (int)s4[r];
    if (local) {
      pf_slot = ldi(a.type_slot, (uint32_t)ldi(a.lt_type, e4) * 4u);
      pf_d = ldf(a.lt_len, e4);
      pf_s1 = ldf(a.l_scale1, e4);
      pf_s2 = ldf(a.l_scale2, e4);
    } else {
      pf_slot = -1;
      pf_d = ldf(a.rad_len, e4);
      pf_s1 = ldf(a.r_scale1, e4);
      pf_s2 = ldf(a.r_scale2, e4);
    }
  };
  // x[src] values of a group of AG_NODE_GRP channel tiles, two groups in flight
  static_assert((AG_CONV_NCH / AG_NODE_GRP) % AG_NODE_XD == 0 && AG_NODE_XD >= 2, "ring of x buffers");
  f32x4 xg[AG_NODE_XD][AG_NODE_GRP];
  uint32_t xoff[4];
  auto set_xoff = [&]() {
#pragma unroll
    for (int r = 0; r < 4; ++r) xoff[r] = ((uint32_t)pf_src[r] * 192u + (uint32_t)col) * 4u;
  };
  auto fetch_xg = [&](auto BUF, int g) {
    constexpr int kb = decltype(BUF)::value;
    const char* xb = reinterpret_cast<const char*>(a.xs);
#pragma unroll
    for (int j = 0; j < AG_NODE_GRP; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (AG_NODE_ABL & 1) xg[kb][j][r] = __uint_as_float(xoff[r] + (uint32_t)(AG_NODE_GRP * g + j));
        else xg[kb][j][r] = *reinterpret_cast<const float*>(xb + (size_t)xoff[r] + 64 * (AG_NODE_GRP * g + j));
      }
    }
  };
  // the first AG_NODE_XD - 1 x groups of a radius tile (requested before the tile starts)
  auto fetch_first_groups = [&]() {
    ag_static_for<0, AG_NODE_XD - 1>([&](auto G) { fetch_xg(G, decltype(G)::value); });
  };
  const lds_u32x4* wl_l = wl + lane;
  // CN channel tiles C0 .. C0 + CN - 1 of one coefficient set (pk [12][NKT]: block nt * NKT + t) times the features:
  // independent accumulator chains with their MFMA passes interleaved
  auto mma_tiles = [&](auto base, auto C0_, const AgIn<MODE> (&ph)[NKT], auto& z, auto INIT_) {
    constexpr int C0 = decltype(C0_)::value;
    constexpr int CN = sizeof(z) / sizeof(f32x4);
    constexpr bool INIT = decltype(INIT_)::value;       // z starts from zero: the first MFMA takes the literal 0
    u32x4 w[CN][NKT][2];
#pragma unroll
    for (int j = 0; j < CN; ++j) {
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        w[j][t][0] = base[(((C0 + j) * NKT + t) * 2) * 64];
        w[j][t][1] = base[(((C0 + j) * NKT + t) * 2 + 1) * 64];
      }
    }
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
      for (int part = 0; part < AgParts<MODE>::n; ++part) {
#pragma unroll
        for (int j = 0; j < CN; ++j) {
          if (AG_NODE_ABL & 2) {
            if (t == 0 && part == 0) {
              u32x4 pu;
              __builtin_memcpy(&pu, &ph[0], 16);
              z[j] = (INIT ? f32x4{0.f, 0.f, 0.f, 0.f} : z[j]) + __builtin_bit_cast(f32x4, w[j][0][0]) * __uint_as_float(pu[0]);
            }
          } else if (INIT && t == 0 && part == 0) z[j] = ag_block_mma_first<MODE, true>(ph[0], w[j][0]);
          else ag_block_mma_part<MODE, true>(z[j], ph[t], w[j][t], part);
        }
      }
    }
  };
  // x[src] values of four channel tiles C0 .. C0 + 3 (local tiles fetch their own)
  auto fetch_x4 = [&](f32x4 (&x)[4], int c0) {
    const char* xb = reinterpret_cast<const char*>(a.xs);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) x[j][r] = *reinterpret_cast<const float*>(xb + (size_t)xoff[r] + 64 * (c0 + j));
    }
  };

  // features of the wave's next RADIUS tile (its rows' inputs are in pf_*): channel tiles 0..7 are conv1 (features x its lw C),
  // 8..11 conv2
  AgIn<MODE> ph1[NKT], ph2[NKT];
  auto next_features = [&]() {
    if (AG_NODE_ABL & 4) {
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        f32x4 v1[2] = {{pf_d, pf_s1, pf_d, pf_s1}, {pf_s1, pf_d, pf_s1, pf_d}}, v2[2] = {{pf_d, pf_s2, pf_d, pf_s2}, {pf_s2, pf_d, pf_s2, pf_d}};
        __builtin_memcpy(&ph1[t], v1, 32);
        __builtin_memcpy(&ph2[t], v2, 32);
      }
    } else {
      ag_poly_features<MODE, NKT>(pf_d, a.two_over_rc, q, ph1, pf_s1);
      ag_poly_features<MODE, NKT>(pf_d, a.two_over_rc, q, ph2, pf_s2);
    }
  };
  float acc[AG_CONV_NCH], accL[AG_CONV_NCH];
  // the sums over the wave's quarters, once per target: quarter j of a reduce-scatter ends up with channel tile 4 g + j.
  // `upper`: target 2 p + 1, whose local rows are rows 8..15 = quarters 2, 3 of the pair's local tiles
  auto finalize = [&](int tgt, bool upper) {
    char* dp = reinterpret_cast<char*>(a.agg + (size_t)tgt * 192);       // (uniform)
    const bool mine = (q >= 2) == upper;
#pragma unroll
    for (int g = 0; g < AG_CONV_NCH / 4; ++g) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[4 * g + j] + (mine ? accL[4 * g + j] : 0.0f);
      *reinterpret_cast<float*>(dp + (uint32_t)(16 * (4 * g + q) + col) * 4u) = ag_quarter_reduce_scatter4(v[0], v[1], v[2], v[3]);
    }
  };

  int p = p_begin + wave;
  if (p >= p_end) return;                       // (no barrier below)
#ifdef AG_NODE_STAMPS
  const unsigned long long k_rt0 = __builtin_amdgcn_s_memrealtime(), k_t0 = __builtin_amdgcn_s_memtime();
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
  PairInfo pi = pair_info(p);
  bool have_pf = false;
  while (p < p_end) {
    const int pn = p + WAVES;
    PairInfo pin = {0, 0, 0, 0, 0, -1};
    if (pn < p_end) pin = pair_info(pn);
    const int ntiles = pi.nL + pi.nA + pi.nB;
    if (ntiles > 0 && !have_pf) {               // cold start (first pair of the wave, or the pair before had no tile)
      bool loc;
      const int rows = tile_rows(pi, p, 0, loc);
      prefetch_meta(rows, loc);
      if (!loc) {                               // (a local tile fetches its own x values and evaluates its own features)
        set_xoff();
        fetch_first_groups();
        next_features();
      }
    }
#pragma unroll
    for (int i = 0; i < AG_CONV_NCH; ++i) accL[i] = 0.0f;
    // the wave's next tile after tile j (of this pair, or the first one of its next pair)
    auto next_tile = [&](int j, bool& nloc, bool& has_next) -> int {
      has_next = true;
      nloc = false;
      if (j + 1 < ntiles) return tile_rows(pi, p, j + 1, nloc);
      if (pin.nL + pin.nA + pin.nB > 0) return tile_rows(pin, pn, 0, nloc);
      has_next = false;
      return 0;
    };
    // A radius tile (one target) as a software pipeline over its four groups of three channel tiles: the MFMAs of group g + 1
    // are issued BEFORE the sums of group g (acc += z x), so that the matrix pipe works while the wave's VALU does the
    // sums; the x values of group g + 2 are requested into the buffer the sums have just freed; the next tile's per-row
    // inputs are requested at the start, its first x group and -- behind the last group's MFMAs -- its features
    // (ph1 / ph2 are carried from tile to tile) at the end.  A local tile fetches / evaluates its own.
    auto radius_tile = [&](int j) {
      bool nloc, has_next;
      const int nrows = next_tile(j, nloc, has_next);
      if (has_next) prefetch_meta(nrows, nloc);
      auto mma_g = [&](auto GG, f32x4 (&z)[AG_NODE_GRP]) {
        constexpr int c0 = AG_NODE_GRP * decltype(GG)::value;
        // channel tiles 0..7 take conv1's features, 8..11 conv2's (a group of three straddles the boundary once: 6, 7 | 8)
        if constexpr (c0 + AG_NODE_GRP <= 8) {
          mma_tiles(wl_l, std::integral_constant<int, c0>{}, ph1, z, std::true_type{});
        } else if constexpr (c0 >= 8) {
          mma_tiles(wl_l, std::integral_constant<int, c0>{}, ph2, z, std::true_type{});
        } else {
          static_assert(AG_NODE_GRP == 3 && c0 == 6, "group layout");
          f32x4 (&za)[2] = *reinterpret_cast<f32x4 (*)[2]>(&z[0]);
          f32x4 (&zb)[1] = *reinterpret_cast<f32x4 (*)[1]>(&z[2]);
          mma_tiles(wl_l, std::integral_constant<int, 6>{}, ph1, za, std::true_type{});
          mma_tiles(wl_l, std::integral_constant<int, 8>{}, ph2, zb, std::true_type{});
        }
      };
      auto sums = [&](auto GG, const f32x4 (&z)[AG_NODE_GRP]) {
        constexpr int gg = decltype(GG)::value;
#pragma unroll
        for (int jj = 0; jj < AG_NODE_GRP; ++jj) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if ((AG_NODE_ABL & 8) && r) continue;
            acc[AG_NODE_GRP * gg + jj] = fmaf(z[jj][r], xg[gg % AG_NODE_XD][jj][r], acc[AG_NODE_GRP * gg + jj]);
          }
          // (pins the sum to this step: the optimiser otherwise sinks all 48 FMAs of a tile below its last MFMA -- nothing
          // needs acc before the target is complete -- and the wave then waits for x loads and MFMAs with nothing to do)
          asm volatile("" : "+v"(acc[AG_NODE_GRP * gg + jj]));
        }
      };
      f32x4 z[2][AG_NODE_GRP];
      // (fences between the steps: the scheduler otherwise hoists every group's coefficient reads to the top of the tile and
      // spills; inside a step it is free to run the sums beside the MFMAs)
      AG_NSTAMP(t0);
      constexpr int XD = AG_NODE_XD;
      fetch_xg(std::integral_constant<int, XD - 1>{}, XD - 1);
      mma_g(std::integral_constant<int, 0>{}, z[0]);
      __builtin_amdgcn_sched_barrier(0);
      AG_NSTAMP(t1);
      ag_static_for<1, NG>([&](auto G) {
        constexpr int g = decltype(G)::value;
        mma_g(G, z[g & 1]);
        sums(std::integral_constant<int, g - 1>{}, z[(g - 1) & 1]);
        // the buffer the sums have just freed takes the group XD - 1 steps ahead: of this tile, or -- once all of this tile's
        // gathers are out and xoff is free -- of the wave's next radius tile (whose features follow the last request)
        if constexpr (g + XD - 1 < NG) {
          fetch_xg(std::integral_constant<int, (g - 1) % XD>{}, g + XD - 1);
        } else if (has_next && !nloc) {
          if constexpr (g + XD - 1 == NG) set_xoff();
          fetch_xg(std::integral_constant<int, (g - 1) % XD>{}, g + XD - 1 - NG);
          if constexpr (g == NG - 1) next_features();
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      AG_NSTAMP(t4);
      sums(std::integral_constant<int, NG - 1>{}, z[(NG - 1) & 1]);
      have_pf = has_next;
#ifdef AG_NODE_STAMPS
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t5 = __builtin_amdgcn_s_memtime();
      st_acc[0] += t1 - t0;
      st_acc[1] += t4 - t1;
      st_acc[4] += t5 - t4;
      st_acc[5] += 1ull;
#endif
    };
    // A local tile (rows 0..7: target 2 p, rows 8..15: target 2 p + 1; several edge types): conv2's four channel tiles
    // first, then conv1's eight; per conv the wave loops over the types present in the tile, each adding its masked
    // features times its own coefficient set.  The tile is long enough to fetch its own x values behind its MFMAs.
    auto local_tile = [&](int j) {
      const float d = pf_d, s1 = pf_s1, s2 = pf_s2;
      const int my_slot = pf_slot;
      bool nloc, has_next;
      const int nrows = next_tile(j, nloc, has_next);
      set_xoff();
      f32x4 xa[4];                              // one buffer: a phase's rounds are long enough for the next phase's values to land
      fetch_x4(xa, 8);
      if (has_next) prefetch_meta(nrows, nloc);
      const uint64_t rows_mask = __ballot(my_slot >= 0) & 0xFFFFull;      // one lane per row (the quarters hold copies)
      // z[...] = sum over the types present of (features of that type's rows) x (that type's coefficient blocks C0 ..)
      auto typed_rounds = [&](const AgIn<MODE> (&ph)[NKT], auto C0_, auto& z) {
        constexpr int CN = sizeof(z) / sizeof(f32x4);
#pragma unroll
        for (int i = 0; i < CN; ++i) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        uint64_t todo = rows_mask;
        while (todo) {
          const int g = __builtin_amdgcn_readlane(my_slot, (int)__builtin_ctzll(todo));
          const bool in = my_slot == g;
          todo &= ~__ballot(in);
          AgIn<MODE> m[NKT];                     // the group's operand: a copy with the other rows zeroed
#pragma unroll
          for (int t = 0; t < NKT; ++t) {
            const u32x4 zero = {0u, 0u, 0u, 0u};
            if constexpr (MODE == AG_F32) {
              m[t].v[0] = in ? ph[t].v[0] : f32x4{0.f, 0.f, 0.f, 0.f};
              m[t].v[1] = in ? ph[t].v[1] : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
              m[t].hi = __builtin_bit_cast(bf16x8, in ? __builtin_bit_cast(u32x4, ph[t].hi) : zero);
              m[t].lo = __builtin_bit_cast(bf16x8, in ? __builtin_bit_cast(u32x4, ph[t].lo) : zero);
            }
          }
          static_assert(CN == 4, "four channel tiles per phase");
          auto run = [&](auto base) {            // (two blocks pairs at a time: 16 coefficient registers in flight, not 32)
            f32x4 (&za)[2] = *reinterpret_cast<f32x4 (*)[2]>(&z[0]);
            f32x4 (&zb)[2] = *reinterpret_cast<f32x4 (*)[2]>(&z[2]);
            mma_tiles(base, std::integral_constant<int, decltype(C0_)::value>{}, m, za, std::false_type{});
            __builtin_amdgcn_sched_barrier(0);
            mma_tiles(base, std::integral_constant<int, decltype(C0_)::value + 2>{}, m, zb, std::false_type{});
            __builtin_amdgcn_sched_barrier(0);
          };
          if (g < a.lds_slots) run(wl_l + (size_t)(1 + g) * SET);
          else run(reinterpret_cast<const u32x4*>(a.poly_typed) + (size_t)g * SET + lane);   // a set that did not fit in LDS: from L2
        }
      };
      // three phases of four channel tiles (conv2: 8..11; conv1: 0..3, 4..7): the accumulators of all twelve at once, next
      // to the coefficient blocks in flight, do not fit the register budget of three waves per SIMD
      auto phase = [&](const AgIn<MODE> (&ph)[NKT], auto C0_, const f32x4 (&x)[4]) {
        constexpr int C0 = decltype(C0_)::value;
        f32x4 z[4];
        typed_rounds(ph, C0_, z);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
          for (int r = 0; r < 4; ++r) accL[C0 + jj] = fmaf(z[jj][r], x[jj][r], accL[C0 + jj]);
          asm volatile("" : "+v"(accL[C0 + jj]));
        }
      };
      {
        AgIn<MODE> ph2[NKT];
        ag_poly_features<MODE, NKT>(d, a.two_over_rc, q, ph2, s2);
        phase(ph2, std::integral_constant<int, 8>{}, xa);
      }
      fetch_x4(xa, 0);
      {
        AgIn<MODE> ph1[NKT];
        ag_poly_features<MODE, NKT>(d, a.two_over_rc, q, ph1, s1);
        phase(ph1, std::integral_constant<int, 0>{}, xa);
        fetch_x4(xa, 4);
        phase(ph1, std::integral_constant<int, 4>{}, xa);
      }
      if (has_next && !nloc) {
        set_xoff();
        fetch_first_groups();
        next_features();
      } else {                                  // (definite writes: keep the buffers and the features out of this tile's live registers)
#pragma unroll
        for (int b = 0; b < AG_NODE_XD - 1; ++b)
#pragma unroll
          for (int jj = 0; jj < AG_NODE_GRP; ++jj) xg[b][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        const u32x4 zero = {0u, 0u, 0u, 0u};
        u32x4 zz[2] = {zero, zero};
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
          __builtin_memcpy(&ph1[t], zz, 32);
          __builtin_memcpy(&ph2[t], zz, 32);
        }
      }
      have_pf = has_next;
    };
    int j = 0;
    for (; j < pi.nL; ++j) local_tile(j);
#pragma unroll
    for (int i = 0; i < AG_CONV_NCH; ++i) acc[i] = 0.0f;
    bool first_done = false;
    for (; j < ntiles; ++j) {
      if (j == pi.nL + pi.nA) {                 // target 2 p is complete: write it, start target 2 p + 1
        finalize(pi.tA, false);
        first_done = true;
#pragma unroll
        for (int i = 0; i < AG_CONV_NCH; ++i) acc[i] = 0.0f;
      }
      radius_tile(j);
    }
    if (!first_done) {
      finalize(pi.tA, false);
#pragma unroll
      for (int i = 0; i < AG_CONV_NCH; ++i) acc[i] = 0.0f;
    }
    if (pi.tB >= 0) finalize(pi.tB, true);
    if (ntiles == 0) have_pf = false;
    p = pn;
    pi = pin;
  }
#ifdef AG_NODE_STAMPS
  if (lane == 0) {
    for (int i = 0; i < 6; ++i) atomicAdd(&ag_node_stamp_acc[i], st_acc[i]);
    atomicAdd(&ag_node_stamp_acc[6], __builtin_amdgcn_s_memrealtime() - k_rt0);
    atomicAdd(&ag_node_stamp_acc[7], __builtin_amdgcn_s_memtime() - k_t0);
  }
#endif
}

#ifndef AG_NODECONV_WAVES
#define AG_NODECONV_WAVES 12     // ~165 VGPRs: three waves per SIMD
#endif
template <int MODE, int NKT>
int launch_cfconv_node_t(const NodeConvArgs& a, int64_t wgs, size_t smem, void* stream) {
  static std::atomic<uint64_t> attr_done{0};
  if (!ag_allow_big_lds(attr_done, (size_t)160 * 1024, k_cfconv_node<MODE, NKT, AG_NODECONV_WAVES>)) return AGDIFF_ERR_LAUNCH;
  k_cfconv_node<MODE, NKT, AG_NODECONV_WAVES><<<dim3((unsigned)wgs), dim3(64 * AG_NODECONV_WAVES), smem, (hipStream_t)stream>>>(a);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}
}  // namespace

extern "C" int agdiff_cfconv_node(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, int32_t k,
                                  void* stream) {
  if (!p || !topo || !ws || k < 0 || k >= p->num_convs) return AGDIFF_ERR_ARG;
  if (p->poly_kt < 1 || p->poly_kt > AGDIFF_POLY_MAX_KT || !p->conv[k].filt_poly_pk) return AGDIFF_ERR_ARG;
  if (!ws->rad_cnt || !ws->rad_src || !ws->rad_len || !ws->r_scale || !ws->xs || !ws->agg || !topo->pair_tgt ||
      topo->num_pairs <= 0)
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
  // coefficient sets in LDS: the radius edges' one, then as many typed ones as fit (5 of 24 KiB at poly_kt 1, 2 of 48 KiB at 2)
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
  a.pair_tgt = topo->pair_tgt;
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
  a.num_pairs = (int32_t)topo->num_pairs;
  a.two_over_rc = 2.0f / p->cutoff;
  int64_t wgs = (a.num_pairs + AG_NODECONV_WAVES - 1) / AG_NODECONV_WAVES;
  if (wgs > 256) wgs = 256;
  const size_t smem = (size_t)(1 + a.lds_slots) * set_bytes;
  ag_log_variant(ws, AGDIFF_VAR_CFCONV_NODE | (local ? AGDIFF_VAR_CFCONV_NODE_LOCAL : 0) |
                         (a.lds_slots < a.num_slots ? AGDIFF_VAR_POLY_L2_SETS : 0));
  if (p->precision == AG_BF3)
    return p->poly_kt == 1 ? launch_cfconv_node_t<AG_BF3, 1>(a, wgs, smem, stream) : launch_cfconv_node_t<AG_BF3, 2>(a, wgs, smem, stream);
  return p->poly_kt == 1 ? launch_cfconv_node_t<AG_F32, 1>(a, wgs, smem, stream) : launch_cfconv_node_t<AG_F32, 2>(a, wgs, smem, stream);
}

#ifdef AG_NODE_STAMPS
extern "C" int agdiff_debug_node_stamps(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(ag_node_stamp_acc), sizeof(ag_node_stamp_acc)) != hipSuccess) return AGDIFF_ERR_LAUNCH;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(ag_node_stamp_acc), z, sizeof(z)) != hipSuccess) return AGDIFF_ERR_LAUNCH;
  }
  return AGDIFF_OK;
}
#endif
