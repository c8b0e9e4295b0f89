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
#define AG_NODE_GRP 2                       // channel tiles per x / MFMA group
#define AG_NODE_XD 2                        // x groups in flight per tile
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
  // Work items of a pair, in this order: its local tiles, then the radius UNITS of its first target, then those of its second.
  // A unit is two consecutive 16-row tiles of ONE target (rows 32 u .. 32 u + 31 of its radius rows; the second tile absent
  // when the target has no more rows): both tiles run against each coefficient block read from LDS -- the LDS reads of the
  // coefficient sets were this kernel's busiest pipe with one tile per read -- and the MFMAs of one tile overlap the sums
  // of the other inside the wave.
  struct Item { bool local, hasY; int rowsX, rowsY; };
  auto n_items = [&](const PairInfo& pi) { return pi.nL + (pi.nA + 1) / 2 + (pi.nB + 1) / 2; };
  auto item_at = [&](const PairInfo& pi, int j) -> Item {
    Item it;
    it.local = j < pi.nL;
    it.hasY = false;
    it.rowsY = 0;
    if (it.local) {
      it.rowsX = (pi.lt0 + j) * AG_TW;
      return it;
    }
    j -= pi.nL;
    const int uA = (pi.nA + 1) / 2;
    const int tgt = (j < uA) ? pi.tA : pi.tB, nt = (j < uA) ? pi.nA : pi.nB, u = (j < uA) ? j : j - uA;
    it.rowsX = tgt * AGDIFF_RAD_STRIDE + 2 * u * AG_TW;
    it.hasY = 2 * u + 1 < nt;
    it.rowsY = it.rowsX + AG_TW;
    return it;
  };
  // per-row inputs of one tile: length and the two scales of row `col`, the sources of the lane's four rows 4 q .. 4 q + 3
  struct Meta { float d, s1, s2; int src[4]; };
  Meta pfX = {0.f, 0.f, 0.f, {0, 0, 0, 0}}, pfY = {0.f, 0.f, 0.f, {0, 0, 0, 0}};
  int pf_slot = -1;
  // (uniform base pointer + 32-bit lane offset: the saddr form of global_load; 64-bit lane pointers per array cost a register
  // pair each and spilled)
  auto ldf = [](const float* base, uint32_t byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
  auto ldi = [](const int32_t* base, uint32_t byte_off) { return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(base) + byte_off); };
  auto prefetch_rows = [&](Meta& m, int rows, bool local) {
    const uint32_t e4 = (uint32_t)(rows + col) * 4u;
    const uint32_t r16 = (uint32_t)(rows + 4 * q) * 4u;
    const int32_t* srcs = local ? a.lt_src : a.rad_src;
    const u32x4 s4 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(srcs) + r16);
#pragma unroll
    for (int r = 0; r < 4; ++r) m.src[r] = (int)s4[r];
    if (local) {
      pf_slot = ldi(a.type_slot, (uint32_t)ldi(a.lt_type, e4) * 4u);
      m.d = ldf(a.lt_len, e4);
      m.s1 = ldf(a.l_scale1, e4);
      m.s2 = ldf(a.l_scale2, e4);
    } else {
      m.d = ldf(a.rad_len, e4);
      m.s1 = ldf(a.r_scale1, e4);
      m.s2 = ldf(a.r_scale2, e4);
    }
  };
  auto prefetch_item = [&](const Item& it) {
    prefetch_rows(pfX, it.rowsX, it.local);
    if (it.hasY) prefetch_rows(pfY, it.rowsY, false);
  };
  // x[src] values of a group of AG_NODE_GRP channel tiles, two groups in flight per tile of the unit
  static_assert(AG_NODE_GRP == 2 && AG_NODE_XD == 2, "unit pipeline: groups of two channel tiles, two x groups in flight");
  f32x4 xX[2][AG_NODE_GRP], xY[2][AG_NODE_GRP];
  uint32_t xoffX[4], xoffY[4];
  auto set_xoff = [&](uint32_t (&xo)[4], const Meta& m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) xo[r] = ((uint32_t)m.src[r] * 192u + (uint32_t)col) * 4u;
  };
  auto fetch_g = [&](f32x4 (&x)[AG_NODE_GRP], const uint32_t (&xo)[4], int g) {
    const char* xb = reinterpret_cast<const char*>(a.xs);
#pragma unroll
    for (int jj = 0; jj < AG_NODE_GRP; ++jj) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (AG_NODE_ABL & 1) x[jj][r] = __uint_as_float(xo[r] + (uint32_t)(AG_NODE_GRP * g + jj));
        else x[jj][r] = *reinterpret_cast<const float*>(xb + (size_t)xo[r] + 64 * (AG_NODE_GRP * g + jj));
      }
    }
  };
  const lds_u32x4* wl_l = wl + lane;
  // coefficient blocks of CN channel tiles C0 .. (pk [12][NKT]: block nt * NKT + t) of one set
  auto load_w = [&](auto base, auto C0_, auto& w) {
    constexpr int C0 = decltype(C0_)::value;
    constexpr int CN = sizeof(w) / (sizeof(u32x4) * NKT * 2);
#pragma unroll
    for (int jj = 0; jj < CN; ++jj) {
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        if ((AG_NODE_ABL & 16) && jj > 0) {            // (timing experiment: one coefficient block read per group)
          w[jj][t][0] = w[0][t][0];
          w[jj][t][1] = w[0][t][1];
          continue;
        }
        w[jj][t][0] = base[(((C0 + jj) * NKT + t) * 2) * 64];
        w[jj][t][1] = base[(((C0 + jj) * NKT + t) * 2 + 1) * 64];
      }
    }
  };
  // ... times the features: independent accumulator chains with their MFMA passes interleaved.  INIT: z starts from zero (the
  // first MFMA takes the literal 0)
  auto mma_w = [&](const auto& w, const AgIn<MODE> (&ph)[NKT], auto& z, auto INIT_) {
    constexpr int CN = sizeof(z) / sizeof(f32x4);
    constexpr bool INIT = decltype(INIT_)::value;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
      for (int part = 0; part < AgParts<MODE>::n; ++part) {
#pragma unroll
        for (int jj = 0; jj < CN; ++jj) {
          if (AG_NODE_ABL & 2) {
            if (t == 0 && part == 0) {
              u32x4 pu;
              __builtin_memcpy(&pu, &ph[0], 16);
              z[jj] = (INIT ? f32x4{0.f, 0.f, 0.f, 0.f} : z[jj]) + __builtin_bit_cast(f32x4, w[jj][0][0]) * __uint_as_float(pu[0]);
            }
          } else if (INIT && t == 0 && part == 0) z[jj] = ag_block_mma_first<MODE, true>(ph[0], w[jj][0]);
          else ag_block_mma_part<MODE, true>(z[jj], ph[t], w[jj][t], part);
        }
      }
    }
  };
  auto mma_tiles = [&](auto base, auto C0_, const AgIn<MODE> (&ph)[NKT], auto& z, auto INIT_) {
    constexpr int CN = sizeof(z) / sizeof(f32x4);
    u32x4 w[CN][NKT][2];
    load_w(base, C0_, w);
    mma_w(w, ph, z, INIT_);
  };
  // x[src] values of four channel tiles C0 .. C0 + 3 (local tiles fetch their own)
  auto fetch_x4 = [&](f32x4 (&x)[4], int c0) {
    const char* xb = reinterpret_cast<const char*>(a.xs);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
      for (int r = 0; r < 4; ++r) x[jj][r] = *reinterpret_cast<const float*>(xb + (size_t)xoffX[r] + 64 * (c0 + jj));
    }
  };

  // features of the wave's next radius unit (its rows' inputs are in pfX / pfY): channel tiles 0..7 are conv1 (features x its
  // lw C), 8..11 conv2
  AgIn<MODE> phX1[NKT], phX2[NKT], phY1[NKT], phY2[NKT];
  auto features_of = [&](const Meta& m, AgIn<MODE> (&p1)[NKT], AgIn<MODE> (&p2)[NKT]) {
    if (AG_NODE_ABL & 4) {
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        f32x4 v1[2] = {{m.d, m.s1, m.d, m.s1}, {m.s1, m.d, m.s1, m.d}}, v2[2] = {{m.d, m.s2, m.d, m.s2}, {m.s2, m.d, m.s2, m.d}};
        __builtin_memcpy(&p1[t], v1, 32);
        __builtin_memcpy(&p2[t], v2, 32);
      }
    } else {
      ag_poly_features<MODE, NKT>(m.d, a.two_over_rc, q, p1, m.s1);
      ag_poly_features<MODE, NKT>(m.d, a.two_over_rc, q, p2, m.s2);
    }
  };
  // everything the next radius unit needs before it starts: gather offsets, its first two x groups, its features
  auto start_unit = [&](const Item& it) {
    set_xoff(xoffX, pfX);
    fetch_g(xX[0], xoffX, 0);
    fetch_g(xX[1], xoffX, 1);
    if (it.hasY) {
      set_xoff(xoffY, pfY);
      fetch_g(xY[0], xoffY, 0);
      fetch_g(xY[1], xoffY, 1);
    }
    features_of(pfX, phX1, phX2);
    if (it.hasY) features_of(pfY, phY1, phY2);
  };
  float acc[AG_CONV_NCH], accL[AG_CONV_NCH];
  // the sums over the wave's quarters, once per target: quarter j of a reduce-scatter ends up with channel tile 4 g + j.
  // `upper`: the pair's second target, whose local rows are rows 8..15 = quarters 2, 3 of the pair's local tiles
  auto finalize = [&](int tgt, bool upper) {
    char* dp = reinterpret_cast<char*>(a.agg + (size_t)tgt * 192);       // (uniform)
    const bool mine = (q >= 2) == upper;
#pragma unroll
    for (int g = 0; g < AG_CONV_NCH / 4; ++g) {
      float v[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) v[jj] = acc[4 * g + jj] + (mine ? accL[4 * g + jj] : 0.0f);
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
    const int nitems = n_items(pi);
    if (nitems > 0 && !have_pf) {               // cold start (first pair of the wave, or the pair before had no item)
      const Item it = item_at(pi, 0);
      prefetch_item(it);
      if (!it.local) start_unit(it);           // (a local tile fetches its own x values and evaluates its own features)
    }
#pragma unroll
    for (int i = 0; i < AG_CONV_NCH; ++i) accL[i] = 0.0f;
    // the wave's next item after item j (of this pair, or the first one of its next pair)
    auto next_item = [&](int j, bool& has_next) -> Item {
      has_next = true;
      if (j + 1 < nitems) return item_at(pi, j + 1);
      if (n_items(pin) > 0) return item_at(pin, 0);
      has_next = false;
      return Item{false, false, 0, 0};
    };
    // A radius unit: per group of two channel tiles ONE read of the coefficient blocks, the MFMAs of tile X, those of tile Y,
    // then the sums acc += z x of X (while Y's MFMAs run) and of Y; the x values two groups ahead go into the buffer the
    // sums have just freed -- of this unit, or, for the last two groups, of the wave's next unit, whose per-row inputs were
    // requested at the start and whose features follow at the end.
    auto radius_unit = [&](int j, bool hasY) {
      bool has_next;
      const Item nx = next_item(j, has_next);
      AG_NSTAMP(t0);
      // (the current unit's inputs sit in xoff / x / ph already: pfX / pfY are free for the next item's)
      if (has_next) prefetch_item(nx);
      const bool next_unit = has_next && !nx.local;
      ag_static_for<0, NG>([&](auto G) {
        constexpr int g = decltype(G)::value;
        constexpr int c0 = AG_NODE_GRP * g;
        u32x4 w[AG_NODE_GRP][NKT][2];
        load_w(wl_l, std::integral_constant<int, c0>{}, w);
        f32x4 zX[AG_NODE_GRP], zY[AG_NODE_GRP];
        mma_w(w, (c0 < 8) ? phX1 : phX2, zX, std::true_type{});
        if (hasY) mma_w(w, (c0 < 8) ? phY1 : phY2, zY, std::true_type{});
#pragma unroll
        for (int jj = 0; jj < AG_NODE_GRP; ++jj) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if ((AG_NODE_ABL & 8) && r) continue;
            acc[c0 + jj] = fmaf(zX[jj][r], xX[g & 1][jj][r], acc[c0 + jj]);
          }
          // (pins the sum to this step: the optimiser otherwise sinks every FMA of a unit below its last MFMA -- nothing
          // needs acc before the target is complete -- and the wave then waits for x loads and MFMAs with nothing to do)
          asm volatile("" : "+v"(acc[c0 + jj]));
        }
        if constexpr (g + 2 < NG) fetch_g(xX[g & 1], xoffX, g + 2);
        if (hasY) {
#pragma unroll
          for (int jj = 0; jj < AG_NODE_GRP; ++jj) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if ((AG_NODE_ABL & 8) && r) continue;
              acc[c0 + jj] = fmaf(zY[jj][r], xY[g & 1][jj][r], acc[c0 + jj]);
            }
            asm volatile("" : "+v"(acc[c0 + jj]));
          }
          if constexpr (g + 2 < NG) fetch_g(xY[g & 1], xoffY, g + 2);
        }
        if constexpr (g + 2 >= NG) {             // (all of this unit's gathers of the buffer are out: the next unit's first groups)
          if (next_unit) {
            if constexpr (g + 2 == NG) {
              set_xoff(xoffX, pfX);
              if (nx.hasY) set_xoff(xoffY, pfY);
            }
            fetch_g(xX[g & 1], xoffX, g + 2 - NG);
            if (nx.hasY) fetch_g(xY[g & 1], xoffY, g + 2 - NG);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      AG_NSTAMP(t4);
      if (next_unit) {
        features_of(pfX, phX1, phX2);
        if (nx.hasY) features_of(pfY, phY1, phY2);
      }
      have_pf = has_next;
#ifdef AG_NODE_STAMPS
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t5 = __builtin_amdgcn_s_memtime();
      st_acc[0] += t4 - t0;
      st_acc[4] += t5 - t4;
      st_acc[5] += 1ull;
#endif
    };
    // A local tile (rows 0..7: the pair's first target, rows 8..15: its second; several edge types): conv2's four channel
    // tiles first, then conv1's eight; per phase the wave loops over the types present in the tile, each adding its masked
    // features times its own coefficient set.  The tile is long enough to fetch its own x values behind its MFMAs.
    auto local_tile = [&](int j) {
      const float d = pfX.d, s1 = pfX.s1, s2 = pfX.s2;
      const int my_slot = pf_slot;
      bool has_next;
      const Item nx = next_item(j, has_next);
      set_xoff(xoffX, pfX);
      f32x4 xa[4];                              // one buffer: a phase's rounds are long enough for the next phase's values to land
      fetch_x4(xa, 8);
      if (has_next) prefetch_item(nx);
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
      if (has_next && !nx.local) {
        start_unit(nx);
        if (!nx.hasY) {                         // (definite writes keep the unused half out of this tile's live registers)
          const u32x4 zero = {0u, 0u, 0u, 0u};
          u32x4 zz[2] = {zero, zero};
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int jj = 0; jj < AG_NODE_GRP; ++jj) xY[b][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < NKT; ++t) {
            __builtin_memcpy(&phY1[t], zz, 32);
            __builtin_memcpy(&phY2[t], zz, 32);
          }
        }
      } else {                                  // (definite writes: keep the buffers and the features out of this tile's live registers)
        const u32x4 zero = {0u, 0u, 0u, 0u};
        u32x4 zz[2] = {zero, zero};
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int jj = 0; jj < AG_NODE_GRP; ++jj) {
            xX[b][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
            xY[b][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
          __builtin_memcpy(&phX1[t], zz, 32);
          __builtin_memcpy(&phX2[t], zz, 32);
          __builtin_memcpy(&phY1[t], zz, 32);
          __builtin_memcpy(&phY2[t], zz, 32);
        }
      }
      have_pf = has_next;
    };
    int j = 0;
#ifndef AG_NODE_NO_LOCAL        // (timing experiment: the kernel without its local-tile code -- fewer registers, more waves)
    for (; j < pi.nL; ++j) local_tile(j);
#endif
#pragma unroll
    for (int i = 0; i < AG_CONV_NCH; ++i) acc[i] = 0.0f;
    const int first_b = pi.nL + (pi.nA + 1) / 2;
    bool first_done = false;
    for (; j < nitems; ++j) {
      if (j == first_b) {                       // the pair's first target is complete: write it, start the second
        finalize(pi.tA, false);
        first_done = true;
#pragma unroll
        for (int i = 0; i < AG_CONV_NCH; ++i) acc[i] = 0.0f;
      }
      const int jr = j - pi.nL;
      const int uA = (pi.nA + 1) / 2;
      const bool hasY = (jr < uA) ? (2 * jr + 1 < pi.nA) : (2 * (jr - uA) + 1 < pi.nB);
      radius_unit(j, hasY);
    }
    if (!first_done) {
      finalize(pi.tA, false);
#pragma unroll
      for (int i = 0; i < AG_CONV_NCH; ++i) acc[i] = 0.0f;
    }
    if (pi.tB >= 0) finalize(pi.tB, true);
    if (nitems == 0) have_pf = false;
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
