// Per-step graph build: torch_cluster.radius_graph (CUDA-kernel semantics) + union with the static
// bond/2-hop/3-hop edges + coalesce order, restated for one workgroup per molecule.
// Replaces models/common.py:208-233 (_extend_to_radius_graph) and geometry.py:5-6 (get_distance).
//
// Output (agdiff_ws_t): a destination-sorted CSR (edges of target i contiguous, sources ascending)
// plus `ref2dst`, the permutation from the reference's (src, dst)-sorted order.  Three launches:
//   count (per graph) -> scan over graphs (one workgroup) -> fill (per graph, lists recomputed).
#include "common.hpp"

namespace {

struct GraphArgs {
  const int32_t* graph_ptr;
  const int32_t* loc_in_ptr;
  const int32_t* loc_in_eid;
  const int32_t* loc_src;
  const int32_t* loc_type;
  const int32_t* loc_row;
  const float* pos;
  float r2;
  int32_t words;  // ceil(max_atoms_per_graph / 32)
  // outputs
  int32_t* graph_edge_cnt;
  const int32_t* graph_edge_ptr;
  int32_t* in_ptr;
  int32_t* out_ptr;
  int32_t* e_src;
  int32_t* e_dst;
  int32_t* e_type;
  float* e_len;
  int32_t* ref2dst;
  int32_t* e_loc;
  // canonical edges (one per undirected pair whose two directions carry the same type, plus every unpaired edge)
  int32_t* graph_canon_cnt;
  const int32_t* graph_canon_ptr;
  float* c_len;
  int32_t* c_type;
  int32_t* c_src;
  int32_t* c_dst;
  int32_t* c_pos;
  int32_t* c_mir;
  // radius edges (type 0) by target, AGDIFF_RAD_STRIDE rows per target (optional: rad_cnt may be null)
  int32_t* rad_cnt;
  int32_t* rad_src;
  float* rad_len;
  // canon_radius_only != 0: the canonical list holds radius edges only (the denoising loop: nothing but the global head
  // walks it then)
  int32_t canon_radius_only;
  // hand-over from the count pass to the fill pass (optional): the in-adjacency masks and the two degree arrays of every
  // molecule, so that the fill pass does not repeat the distance tests and the mirror look-ups
  uint32_t* g_inbits;    // [N][words]
  int32_t* g_deg;        // [N] in-degrees
  int32_t* g_cdeg;       // [N] canonical in-degrees
  int32_t num_graphs;
  // optional (r_scale != null): lw(d) * C(d) of the 2 * num_convs CFConvs for the radius rows, and the pad rows that complete
  // every target's last 16-row tile, straight from the fill pass (what agdiff_edge_scales_split(which = 0) does as a launch
  // of its own: 36 us alone, 70 us beside the local branch's kernels)
  const float* dw[2 * AGDIFF_MAX_CONVS];
  float* r_scale;
  int64_t rpad;
  int32_t n_scales;
  float cutoff;
  int32_t smooth;
};

// Edge lengths must come out bit-identical wherever they are computed (the per-step graph build and the local-edge
// pass feed the same encoder): hipcc lowers sqrtf differently from kernel to kernel (1 ulp apart), so the root is
// taken in double and rounded once -- correctly rounded for every float input.
__device__ __forceinline__ float ag_sqrt_rn(float x) { return (float)sqrt((double)x); }

// d^2 exactly as the restated rule: ((dx*dx + dy*dy) + dz*dz), no FMA contraction.
__device__ __forceinline__ float dist2_nofma(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
  return (xx + yy) + zz;
}

extern __shared__ uint32_t ag_graph_smem[];

// One workgroup per molecule, one WAVE per target atom at a time: the 64 lanes test 64 candidate sources at once,
// wave ballots replace the serial scan of the reference rule --
//   radius_graph: source j is kept for target i if d2(i, j) < r^2 and fewer than 33 such candidates (self
//   included) precede it in ascending j; self is then dropped; bond / 2-hop / 3-hop edges are always present
// -- and the position of an edge inside the target's list is the population count of the kept mask below its
// lane, so all per-edge stores are contiguous.  The in-adjacency masks stay in LDS (row i, bit j <=> edge j -> i);
// out-degrees and the (src, dst)-order permutation ref2dst come from their columns.
#define AG_GRAPH_THREADS 1024     // launch bound; the launcher picks 512 or 1024 threads per molecule
template <bool FILL>
__global__ void __launch_bounds__(AG_GRAPH_THREADS) k_graph(GraphArgs a) {
  const int g = blockIdx.x;
  const int g0 = a.graph_ptr[g];
  const int n = a.graph_ptr[g + 1] - g0;
  const int words = a.words;          // 32-bit words per mask row, even (whole 64-lane chunks)
  const int nmax = words * 32;
  // LDS carve: pos[3 nmax] | indeg[nmax] | outdeg[nmax] | canonical indeg[nmax] | inbits[nmax][words] | locbits[nmax][words]
  float* spos = reinterpret_cast<float*>(ag_graph_smem);
  int* sin = reinterpret_cast<int*>(ag_graph_smem + 3 * nmax);
  int* sout = sin + nmax;
  int* scan_c = sout + nmax;
  uint32_t* inbits = reinterpret_cast<uint32_t*>(scan_c + nmax);
  uint32_t* locbits = inbits + nmax * words;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));     // lanes below mine
  auto bit = [&](const uint32_t* rows, int r, int c) -> bool { return (rows[r * words + (c >> 5)] >> (c & 31)) & 1u; };
  auto rank_below = [&](const uint32_t* rows, int r, int c) -> int {   // set bits of row r below column c
    const uint32_t* row = rows + r * words;
    int k = __popc(row[c >> 5] & ((1u << (c & 31)) - 1u));
    for (int w = 0; w < (c >> 5); ++w) k += __popc(row[w]);
    return k;
  };
  auto local_type = [&](int i, int j) -> int {      // type of the local edge j -> i (caller checked its locbits bit)
    return a.loc_type[a.loc_in_eid[a.loc_in_ptr[g0 + i] + rank_below(locbits, i, j)]];
  };
  // The directed edges j -> i and i -> j are mirrors when both exist with the same type: same length, same type,
  // hence bit-identical edge_attr, filter and pair-head output (dualenc.py:189-211).  Canonical = the one with
  // src < dst, or any edge without a mirror.
  auto has_mirror = [&](int i, int j) -> bool {     // for an existing edge j -> i
    if (!bit(inbits, j, i)) return false;
    const bool li = bit(locbits, i, j), lj = bit(locbits, j, i);
    if (li != lj) return false;
    return !li || local_type(i, j) == local_type(j, i);
  };

  __shared__ float sseg[FILL ? 2 * AGDIFF_MAX_CONVS * 100 : 1];         // the CFConvs' distance-weighting segment tables
  if (FILL && a.r_scale)
    for (int i = threadIdx.x; i < a.n_scales * 100; i += blockDim.x) sseg[i] = a.dw[i / 100][i % 100];
  for (int i = threadIdx.x; i < 3 * n; i += blockDim.x) spos[i] = a.pos[3 * (size_t)g0 + i];
  // static local in-adjacency of the molecule: the thread that owns target i sets its row
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    for (int w = 0; w < words; ++w) locbits[i * words + w] = 0u;
    for (int k = a.loc_in_ptr[g0 + i]; k < a.loc_in_ptr[g0 + i + 1]; ++k) {
      const int j = a.loc_src[a.loc_in_eid[k]] - g0;
      locbits[i * words + (j >> 5)] |= 1u << (j & 31);
    }
  }
  __syncthreads();

  int wave_total = 0, wave_ctotal = 0;
  if (FILL && a.g_inbits) {
    // the count pass of this build left the masks and degrees in global memory
    for (int k = threadIdx.x; k < n * words; k += blockDim.x) inbits[k] = a.g_inbits[(size_t)g0 * words + k];
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      sin[i] = a.g_deg[g0 + i];
      scan_c[i] = a.g_cdeg[g0 + i];
    }
    __syncthreads();
  } else {
  // pass 1: kept masks and in-degrees
  for (int i = wave; i < n; i += nwaves) {
    const float xi = spos[3 * i], yi = spos[3 * i + 1], zi = spos[3 * i + 2];
    int cnt_r = 0, deg = 0;
    for (int c = 0; 64 * c < n; ++c) {
      const int j = 64 * c + lane;
      const bool valid = j < n;
      const int jj = valid ? j : 0;
      const float d2 = dist2_nofma(xi, yi, zi, spos[3 * jj], spos[3 * jj + 1], spos[3 * jj + 2]);
      const bool within = valid && d2 < a.r2;
      const uint64_t wmask = __ballot(within);
      const bool rad = within && (cnt_r + __popcll(wmask & lt) < AGDIFF_RADIUS_CAP) && j != i;
      cnt_r += __popcll(wmask);
      const bool loc = valid && ((locbits[i * words + 2 * c + (lane >> 5)] >> (lane & 31)) & 1u);
      const uint64_t emask = __ballot(rad || loc);
      if (lane == 0) {
        inbits[i * words + 2 * c] = (uint32_t)emask;
        inbits[i * words + 2 * c + 1] = (uint32_t)(emask >> 32);
      }
      deg += __popcll(emask);
    }
    if (lane == 0) sin[i] = deg;
    wave_total += deg;
  }
  __syncthreads();

  // canonical in-degrees (needs every row of inbits: after the barrier)
  for (int i = wave; i < n; i += nwaves) {
    int cdeg = 0;
    for (int c = 0; 64 * c < n; ++c) {
      const int j = 64 * c + lane;
      const bool e = j < n && bit(inbits, i, j);
      bool canon;
      if (a.canon_radius_only) {       // radius edges: a mirror is the radius edge the other way round (same length)
        const bool rad = e && !bit(locbits, i, j);
        canon = rad && (j < i || !(bit(inbits, j, i) && !bit(locbits, j, i)));
      } else {
        canon = e && (j < i || !has_mirror(i, j));
      }
      cdeg += __popcll(__ballot(canon));
    }
    if (lane == 0) scan_c[i] = cdeg;
    wave_ctotal += cdeg;
  }
  __syncthreads();
  if (!FILL && a.g_inbits) {        // hand the masks and degrees over to the fill pass
    for (int k = threadIdx.x; k < n * words; k += blockDim.x) a.g_inbits[(size_t)g0 * words + k] = inbits[k];
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      a.g_deg[g0 + i] = sin[i];
      a.g_cdeg[g0 + i] = scan_c[i];
    }
  }
  }

  if (!FILL) {
    __shared__ int wsum[2][AG_GRAPH_THREADS / 64];
    if (lane == 0) { wsum[0][wave] = wave_total; wsum[1][wave] = wave_ctotal; }
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0, ctot = 0;
      for (int w = 0; w < nwaves; ++w) { tot += wsum[0][w]; ctot += wsum[1][w]; }
      a.graph_edge_cnt[g] = tot;
      a.graph_canon_cnt[g] = ctot;
    }
    return;
  }

  // out-degrees = column counts of the in-adjacency
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    int c = 0;
    for (int i = 0; i < n; ++i) c += (inbits[i * words + (j >> 5)] >> (j & 31)) & 1u;
    sout[j] = c;
  }
  __syncthreads();
  // inclusive scans of sin / sout / scan_c (n <= 512): Hillis-Steele in place, three arrays together
  for (int off = 1; off < n; off <<= 1) {
    int vi[2], vo[2], vc[2], cnt = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x, ++cnt) {
      vi[cnt] = sin[i] + (i >= off ? sin[i - off] : 0);
      vo[cnt] = sout[i] + (i >= off ? sout[i - off] : 0);
      vc[cnt] = scan_c[i] + (i >= off ? scan_c[i - off] : 0);
    }
    __syncthreads();
    cnt = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x, ++cnt) {
      sin[i] = vi[cnt];
      sout[i] = vo[cnt];
      scan_c[i] = vc[cnt];
    }
    __syncthreads();
  }
  const int base = a.graph_edge_ptr[g];
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    a.in_ptr[g0 + i] = base + (i ? sin[i - 1] : 0);
    a.out_ptr[g0 + i] = base + (i ? sout[i - 1] : 0);
  }
  if (g == a.num_graphs - 1 && threadIdx.x == 0) {
    a.in_ptr[g0 + n] = base + (n ? sin[n - 1] : 0);
    a.out_ptr[g0 + n] = base + (n ? sout[n - 1] : 0);
  }
  // pass 2: emit the lists, one wave per target, contiguous stores
  for (int i = wave; i < n; i += nwaves) {
    const float xi = spos[3 * i], yi = spos[3 * i + 1], zi = spos[3 * i + 2];
    int p0 = base + (i ? sin[i - 1] : 0);
    int cp0 = a.graph_canon_ptr[g] + (i ? scan_c[i - 1] : 0);
    int lk = a.loc_in_ptr[g0 + i];
    int rp0 = (g0 + i) * AGDIFF_RAD_STRIDE;        // the target's own rows of the radius list (at most AGDIFF_RADIUS_CAP used)
    for (int c = 0; 64 * c < n; ++c) {
      const int j = 64 * c + lane;
      const int jj = (j < n) ? j : 0;
      const uint64_t emask = (uint64_t)inbits[i * words + 2 * c] | ((uint64_t)inbits[i * words + 2 * c + 1] << 32);
      const uint64_t lmask = (uint64_t)locbits[i * words + 2 * c] | ((uint64_t)locbits[i * words + 2 * c + 1] << 32);
      const uint64_t rmask = emask & ~lmask;
      const bool e = (emask >> lane) & 1ull;
      bool mir, canon;
      if (a.canon_radius_only) {
        const bool rad = (rmask >> lane) & 1ull;
        mir = rad && bit(inbits, j, i) && !bit(locbits, j, i);
        canon = rad && (j < i || !mir);
      } else {
        mir = e && has_mirror(i, j);
        canon = e && (j < i || !mir);
      }
      const uint64_t cmask = __ballot(canon);
      if (e) {
        const int p = p0 + __popcll(emask & lt);
        int ty = 0, eid = -1;
        if ((lmask >> lane) & 1ull) {
          eid = a.loc_in_eid[lk + __popcll(lmask & lt)];
          ty = a.loc_type[eid];
        }
        const float len = ag_sqrt_rn(dist2_nofma(xi, yi, zi, spos[3 * jj], spos[3 * jj + 1], spos[3 * jj + 2]));
        a.e_loc[p] = eid >= 0 ? a.loc_row[eid] : -1;
        a.e_src[p] = g0 + j;
        a.e_dst[p] = g0 + i;
        a.e_type[p] = ty;
        a.e_len[p] = len;
        if (a.rad_cnt && eid < 0) {
          const int rp = rp0 + __popcll(rmask & lt);
          a.rad_src[rp] = g0 + j;
          a.rad_len[rp] = len;
          if (a.r_scale) {
            const float C = cf_envelope(len, a.cutoff, a.smooth);
            for (int c = 0; c < a.n_scales; ++c) a.r_scale[(size_t)c * a.rpad + rp] = cf_dist_weight(sseg + c * 100, len) * C;
          }
        }
        if (canon) {
          const int cp = cp0 + __popcll(cmask & lt);
          a.c_len[cp] = len;
          a.c_type[cp] = ty;
          a.c_src[cp] = g0 + j;
          a.c_dst[cp] = g0 + i;
          a.c_pos[cp] = p;
          a.c_mir[cp] = mir ? base + (j ? sin[j - 1] : 0) + rank_below(inbits, j, i) : -1;
        }
      }
      p0 += __popcll(emask);
      cp0 += __popcll(cmask);
      lk += __popcll(lmask);
      rp0 += __popcll(rmask);
    }
    if (a.rad_cnt) {
      const int cnt = rp0 - (g0 + i) * AGDIFF_RAD_STRIDE;
      if (lane == 0) a.rad_cnt[g0 + i] = cnt;
      if (a.r_scale && lane < ((cnt + AG_TW - 1) / AG_TW) * AG_TW - cnt) {     // pad rows: src = the target itself, length 0, scale 0
        const int rp = rp0 + lane;
        a.rad_src[rp] = g0 + i;
        a.rad_len[rp] = 0.0f;
        for (int c = 0; c < a.n_scales; ++c) a.r_scale[(size_t)c * a.rpad + rp] = 0.0f;
      }
    }
  }
  // pass 3: ref2dst, one thread per source walking its column (targets ascending = the (src, dst) order)
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    int q = base + (j ? sout[j - 1] : 0);
    const int wj = j >> 5;
    const uint32_t below = (1u << (j & 31)) - 1u;
    for (int i = 0; i < n; ++i) {
      const uint32_t* row = inbits + i * words;
      if ((row[wj] >> (j & 31)) & 1u) {
        int rank = __popc(row[wj] & below);
        for (int w = 0; w < wj; ++w) rank += __popc(row[w]);
        a.ref2dst[q++] = base + (i ? sin[i - 1] : 0) + rank;
      }
    }
  }
}

// exclusive scan of graph_edge_cnt[G] -> graph_edge_ptr[G+1]; total -> num_edges. One workgroup.
// Block 1 (if launched) does the same for the canonical counts.
__global__ void __launch_bounds__(1024) k_scan_graphs(const int32_t* __restrict__ cnt0, int32_t* __restrict__ ptr0,
                                                      int32_t* __restrict__ total0, const int32_t* __restrict__ cnt1,
                                                      int32_t* __restrict__ ptr1, int32_t* __restrict__ total1, int G) {
  const int32_t* cnt = blockIdx.x ? cnt1 : cnt0;
  int32_t* ptr = blockIdx.x ? ptr1 : ptr0;
  int32_t* total = blockIdx.x ? total1 : total0;
  __shared__ int wtot[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int b0 = 0; b0 < G; b0 += 1024) {
    const int i = b0 + threadIdx.x;
    int v = (i < G) ? cnt[i] : 0;
    int s = v;
    for (int o = 1; o < 64; o <<= 1) {
      int u = __shfl_up(s, o);
      if (lane >= o) s += u;
    }
    if (lane == 63) wtot[wv] = s;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wv; ++w) woff += wtot[w];
    const int carry = carry_s;
    if (i < G) ptr[i] = carry + woff + s - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + s;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    ptr[G] = carry_s;
    *total = carry_s;
  }
}

// one evaluation per canonical local edge (|p_i - p_j| is the same number for j -> i and i -> j: the differences only
// change sign before they are squared)
__global__ void k_local_lengths(const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                const int32_t* __restrict__ cpos, const int32_t* __restrict__ cmir,
                                const float* __restrict__ pos, float* __restrict__ len, float* __restrict__ clen, int Lc,
                                const int32_t* __restrict__ inpos, const int32_t* __restrict__ inmir,
                                float* __restrict__ len_in, const int32_t* __restrict__ tpos,
                                const int32_t* __restrict__ tmir, float* __restrict__ len_t) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Lc) return;
  int s = src[c], d = dst[c];
  const float v = ag_sqrt_rn(dist2_nofma(pos[3 * s], pos[3 * s + 1], pos[3 * s + 2], pos[3 * d], pos[3 * d + 1], pos[3 * d + 2]));
  clen[c] = v;
  len[cpos[c]] = v;
  if (cmir[c] >= 0) len[cmir[c]] = v;
  if (len_in) {          // the same by padded-list position (agdiff_topo_t.lp_*)
    len_in[inpos[c]] = v;
    if (inmir[c] >= 0) len_in[inmir[c]] = v;
  }
  if (len_t) {           // ... and by quad-tile row (agdiff_topo_t.lt_*)
    len_t[tpos[c]] = v;
    if (tmir[c] >= 0) len_t[tmir[c]] = v;
  }
}

}  // namespace

extern "C" int agdiff_graph_build(const agdiff_topo_t* topo, const agdiff_ws_t* ws, const float* pos, float cutoff,
                                  void* stream) {
  return agdiff_graph_build_ex(topo, ws, pos, cutoff, 0, stream);
}

namespace {
int graph_build_impl(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, const float* pos, float cutoff,
                     int32_t canon_radius_only, void* stream);
}

extern "C" int agdiff_graph_build_ex(const agdiff_topo_t* topo, const agdiff_ws_t* ws, const float* pos, float cutoff,
                                     int32_t canon_radius_only, void* stream) {
  return graph_build_impl(nullptr, topo, ws, pos, cutoff, canon_radius_only, stream);
}

extern "C" int agdiff_graph_build_scaled(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                                         const float* pos, float cutoff, int32_t canon_radius_only, void* stream) {
  if (!p || !ws || !ws->r_scale || !ws->rad_cnt || p->num_convs > AGDIFF_MAX_CONVS) return AGDIFF_ERR_ARG;
  return graph_build_impl(p, topo, ws, pos, cutoff, canon_radius_only, stream);
}

namespace {
int graph_build_impl(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, const float* pos, float cutoff,
                     int32_t canon_radius_only, void* stream) {
  if (!topo || !ws || !pos || topo->num_graphs <= 0 || topo->num_nodes <= 0 || (topo->num_local > 0 && !topo->loc_row))
    return AGDIFF_ERR_ARG;
  if (!ws->graph_edge_cnt || !ws->graph_edge_ptr || !ws->in_ptr || !ws->out_ptr || !ws->e_src || !ws->e_dst ||
      !ws->e_type || !ws->e_len || !ws->ref2dst || !ws->e_loc || !ws->num_edges || !ws->num_canon ||
      !ws->graph_canon_cnt || !ws->graph_canon_ptr || !ws->c_len || !ws->c_type || !ws->c_src || !ws->c_dst ||
      !ws->c_pos || !ws->c_mir)
    return AGDIFF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  // The host (agdiff_amd/topology.py) guarantees max atoms per graph <= AGDIFF_MAX_ATOMS_PER_GRAPH and
  // passes it through max_edges bookkeeping; the kernels size LDS for the compile-time limit's word count
  // actually needed, which the host stores in graph_edge_cnt capacity order: words from max atoms.
  GraphArgs a;
  a.graph_ptr = topo->graph_ptr;
  a.loc_in_ptr = topo->loc_in_ptr;
  a.loc_in_eid = topo->loc_in_eid;
  a.loc_src = topo->loc_src;
  a.loc_type = topo->loc_type;
  a.loc_row = topo->loc_row;
  a.pos = pos;
  a.r2 = cutoff * cutoff;
  a.graph_edge_cnt = ws->graph_edge_cnt;
  a.graph_edge_ptr = ws->graph_edge_ptr;
  a.in_ptr = ws->in_ptr;
  a.out_ptr = ws->out_ptr;
  a.e_src = ws->e_src;
  a.e_dst = ws->e_dst;
  a.e_type = ws->e_type;
  a.e_len = ws->e_len;
  a.ref2dst = ws->ref2dst;
  a.e_loc = ws->e_loc;
  a.graph_canon_cnt = ws->graph_canon_cnt;
  a.graph_canon_ptr = ws->graph_canon_ptr;
  a.c_len = ws->c_len;
  a.c_type = ws->c_type;
  a.c_src = ws->c_src;
  a.c_dst = ws->c_dst;
  a.c_pos = ws->c_pos;
  a.c_mir = ws->c_mir;
  const bool rad = ws->rad_cnt && ws->rad_src && ws->rad_len;
  a.rad_cnt = rad ? ws->rad_cnt : nullptr;
  a.rad_src = ws->rad_src;
  a.rad_len = ws->rad_len;
  if (canon_radius_only && !rad) return AGDIFF_ERR_ARG;
  a.canon_radius_only = canon_radius_only ? 1 : 0;
  const bool handover = ws->g_inbits && ws->g_deg && ws->g_cdeg;
  a.g_inbits = handover ? reinterpret_cast<uint32_t*>(ws->g_inbits) : nullptr;
  a.g_deg = ws->g_deg;
  a.g_cdeg = ws->g_cdeg;
  a.num_graphs = (int32_t)topo->num_graphs;
  a.r_scale = nullptr;
  a.rpad = 0;
  a.n_scales = 0;
  a.cutoff = cutoff;
  a.smooth = 0;
  if (p) {                 // the radius rows' CFConv scales from the fill pass (agdiff_graph_build_scaled)
    for (int k = 0; k < p->num_convs; ++k) {
      a.dw[2 * k] = p->conv[k].dist_seg;
      a.dw[2 * k + 1] = p->conv[k].dist_seg + 100;
    }
    a.r_scale = ws->r_scale;
    a.rpad = topo->num_nodes * (int64_t)AGDIFF_RAD_STRIDE;
    a.n_scales = 2 * p->num_convs;
    a.cutoff = p->cutoff;       // (the envelope's cutoff is the model's, also when the graph is built without radius edges)
    a.smooth = p->smooth;
  }
  const int max_atoms = (int)topo->max_atoms_per_graph;
  if (max_atoms <= 0 || max_atoms > AGDIFF_MAX_ATOMS_PER_GRAPH) return AGDIFF_ERR_LIMIT;
  a.words = 2 * ((max_atoms + 63) / 64);
  const int nmax = a.words * 32;
  // threads per molecule: a wave walks its targets one after the other (per target a chain of ballots and dependent
  // look-ups), so more waves per molecule shorten the launch until the workgroups no longer fit the chip at once
  // (1 x 100 conformers: 61 / 45 / 37 us with 256 / 512 / 1024 threads; 1024 molecules: 133 / 119 / 123)
  const int bd = topo->num_graphs >= 512 ? 512 : 1024;
  const size_t smem = (size_t)(3 * nmax + 3 * nmax + 2 * nmax * a.words) * 4;
  if (smem > 48 * 1024) {
    static std::atomic<uint64_t> attr_done{0};
    if (!ag_allow_big_lds(attr_done, (size_t)160 * 1024, k_graph<false>, k_graph<true>)) return AGDIFF_ERR_LAUNCH;
  }
  k_graph<false><<<dim3((unsigned)topo->num_graphs), dim3(bd), smem, st>>>(a);
  AG_CHECK_LAUNCH();
  k_scan_graphs<<<2, 1024, 0, st>>>(ws->graph_edge_cnt, ws->graph_edge_ptr, ws->num_edges, ws->graph_canon_cnt,
                                    ws->graph_canon_ptr, ws->num_canon, (int)topo->num_graphs);
  AG_CHECK_LAUNCH();
  k_graph<true><<<dim3((unsigned)topo->num_graphs), dim3(bd), smem, st>>>(a);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}
}  // namespace

extern "C" int agdiff_local_lengths(const agdiff_topo_t* topo, const agdiff_ws_t* ws, const float* pos, void* stream) {
  if (!topo || !ws || !pos || !ws->l_len || !ws->lc_len) return AGDIFF_ERR_ARG;
  if (topo->num_local == 0) return AGDIFF_OK;
  if (!topo->lc_src || !topo->lc_dst || !topo->lc_pos || !topo->lc_mir || topo->num_local_canon <= 0) return AGDIFF_ERR_ARG;
  const int Lc = (int)topo->num_local_canon;
  const bool by_slot = ws->l_len_p && topo->lc_ppos && topo->lc_pmir;
  const bool by_tile = ws->lt_len && topo->lc_tpos && topo->lc_tmir;
  k_local_lengths<<<(Lc + 255) / 256, 256, 0, (hipStream_t)stream>>>(topo->lc_src, topo->lc_dst, topo->lc_pos, topo->lc_mir,
                                                                     pos, ws->l_len, ws->lc_len, Lc, topo->lc_ppos,
                                                                     topo->lc_pmir, by_slot ? ws->l_len_p : nullptr,
                                                                     topo->lc_tpos, topo->lc_tmir, by_tile ? ws->lt_len : nullptr);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}
