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
  int32_t num_graphs;
};

// d^2 exactly as the restated rule: ((dx*dx + dy*dy) + dz*dz), no FMA contraction.
__device__ __forceinline__ float dist2_nofma(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  float s = __fmul_rn(dx, dx);
  s = __fadd_rn(s, __fmul_rn(dy, dy));
  s = __fadd_rn(s, __fmul_rn(dz, dz));
  return s;
}

// Enumerate the merged in-list of target i (graph-local index) in ascending source order and call
// emit(k, j, type, d2) for the k-th in-edge j -> i.  Returns the in-degree.
template <class Emit>
__device__ __forceinline__ int enumerate_in_edges(const GraphArgs& a, const float* spos, int n, int g0, int i, Emit emit) {
  const float xi = spos[3 * i], yi = spos[3 * i + 1], zi = spos[3 * i + 2];
  int lk = a.loc_in_ptr[g0 + i];
  const int lend = a.loc_in_ptr[g0 + i + 1];
  int lsrc = (lk < lend) ? a.loc_src[a.loc_in_eid[lk]] - g0 : n;
  int cnt_r = 0, k = 0;
  for (int j = 0; j < n; ++j) {
    float d2 = dist2_nofma(xi, yi, zi, spos[3 * j], spos[3 * j + 1], spos[3 * j + 2]);
    bool within = false;
    if (cnt_r < AGDIFF_RADIUS_CAP && d2 < a.r2) {  // first 33 in-radius candidates, self included
      within = true;
      ++cnt_r;
    }
    const bool rad = within && (j != i);
    const bool loc = (j == lsrc);
    if (rad || loc) {
      int ty = 0;
      if (loc) ty = a.loc_type[a.loc_in_eid[lk]];
      emit(k, j, ty, d2);
      ++k;
    }
    if (loc) {
      ++lk;
      lsrc = (lk < lend) ? a.loc_src[a.loc_in_eid[lk]] - g0 : n;
    }
  }
  return k;
}

extern __shared__ uint32_t ag_graph_smem[];

template <bool FILL>
__global__ void __launch_bounds__(512) k_graph(GraphArgs a) {
  const int g = blockIdx.x;
  const int g0 = a.graph_ptr[g];
  const int n = a.graph_ptr[g + 1] - g0;
  const int words = a.words;
  // LDS carve: pos[3n] | indeg[n] | outdeg[n] | bits[n][words]
  const int nmax = words * 32;
  float* spos = reinterpret_cast<float*>(ag_graph_smem);
  int* sin = reinterpret_cast<int*>(ag_graph_smem + 3 * nmax);
  int* sout = sin + nmax;
  uint32_t* bits = reinterpret_cast<uint32_t*>(sout + nmax);

  for (int i = threadIdx.x; i < 3 * n; i += blockDim.x) spos[i] = a.pos[3 * (size_t)g0 + i];
  for (int i = threadIdx.x; i < n * words; i += blockDim.x) bits[i] = 0u;
  __syncthreads();

  // pass 1: in-degree and out-adjacency bits (bit i of row j <=> edge j -> i exists)
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int deg = enumerate_in_edges(a, spos, n, g0, i, [&](int, int j, int, float) {
      atomicOr(&bits[j * words + (i >> 5)], 1u << (i & 31));
    });
    sin[i] = deg;
  }
  __syncthreads();

  if (!FILL) {
    // per-graph edge count = sum of in-degrees
    int s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += sin[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    __shared__ int wsum[8];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) tot += wsum[w];
      a.graph_edge_cnt[g] = tot;
    }
    return;
  }

  // out-degrees from the bit rows
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    int c = 0;
    for (int w = 0; w < words; ++w) c += __popc(bits[j * words + w]);
    sout[j] = c;
  }
  __syncthreads();
  // exclusive scans of sin / sout (n <= 512): Hillis-Steele in place, two arrays together
  for (int off = 1; off < n; off <<= 1) {
    int vi[4], vo[4], cnt = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x, ++cnt) {
      vi[cnt] = sin[i] + (i >= off ? sin[i - off] : 0);
      vo[cnt] = sout[i] + (i >= off ? sout[i - off] : 0);
    }
    __syncthreads();
    cnt = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x, ++cnt) {
      sin[i] = vi[cnt];
      sout[i] = vo[cnt];
    }
    __syncthreads();
  }
  // sin/sout now hold INCLUSIVE sums
  const int base = a.graph_edge_ptr[g];
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int in_excl = (i ? sin[i - 1] : 0), out_excl = (i ? sout[i - 1] : 0);
    a.in_ptr[g0 + i] = base + in_excl;
    a.out_ptr[g0 + i] = base + out_excl;
  }
  if (g == a.num_graphs - 1 && threadIdx.x == 0) {
    a.in_ptr[g0 + n] = base + (n ? sin[n - 1] : 0);
    a.out_ptr[g0 + n] = base + (n ? sout[n - 1] : 0);
  }
  // pass 2: emit
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int pbase = base + (i ? sin[i - 1] : 0);
    enumerate_in_edges(a, spos, n, g0, i, [&](int k, int j, int ty, float d2) {
      const int p = pbase + k;
      a.e_src[p] = g0 + j;
      a.e_dst[p] = g0 + i;
      a.e_type[p] = ty;
      a.e_len[p] = sqrtf(d2);
      // rank of target i among the out-edges of j (targets ascending) = set bits below i in row j
      int rank = 0;
      const uint32_t* row = bits + j * words;
      for (int w = 0; w < (i >> 5); ++w) rank += __popc(row[w]);
      rank += __popc(row[i >> 5] & ((1u << (i & 31)) - 1u));
      const int q = base + (j ? sout[j - 1] : 0) + rank;
      a.ref2dst[q] = p;
    });
  }
}

// exclusive scan of graph_edge_cnt[G] -> graph_edge_ptr[G+1]; total -> num_edges. One workgroup.
__global__ void __launch_bounds__(1024) k_scan_graphs(const int32_t* __restrict__ cnt, int32_t* __restrict__ ptr,
                                                      int32_t* __restrict__ total, int G) {
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

__global__ void k_local_lengths(const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                const float* __restrict__ pos, float* __restrict__ len, int L) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= L) return;
  int s = src[e], d = dst[e];
  float dx = pos[3 * s] - pos[3 * d], dy = pos[3 * s + 1] - pos[3 * d + 1], dz = pos[3 * s + 2] - pos[3 * d + 2];
  len[e] = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
}

}  // namespace

extern "C" int agdiff_graph_build(const agdiff_topo_t* topo, const agdiff_ws_t* ws, const float* pos, float cutoff,
                                  void* stream) {
  if (!topo || !ws || !pos || topo->num_graphs <= 0 || topo->num_nodes <= 0) return AGDIFF_ERR_ARG;
  if (!ws->graph_edge_cnt || !ws->graph_edge_ptr || !ws->in_ptr || !ws->out_ptr || !ws->e_src || !ws->e_dst ||
      !ws->e_type || !ws->e_len || !ws->ref2dst || !ws->num_edges)
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
  a.num_graphs = (int32_t)topo->num_graphs;
  const int max_atoms = (int)topo->max_atoms_per_graph;
  if (max_atoms <= 0 || max_atoms > AGDIFF_MAX_ATOMS_PER_GRAPH) return AGDIFF_ERR_LIMIT;
  a.words = (max_atoms + 31) / 32;
  const int nmax = a.words * 32;
  int bd = ((max_atoms + 63) / 64) * 64;
  if (bd > 512) bd = 512;
  const size_t smem = (size_t)(3 * nmax + 2 * nmax + nmax * a.words) * 4;
  k_graph<false><<<dim3((unsigned)topo->num_graphs), dim3(bd), smem, st>>>(a);
  AG_CHECK_LAUNCH();
  k_scan_graphs<<<1, 1024, 0, st>>>(ws->graph_edge_cnt, ws->graph_edge_ptr, ws->num_edges, (int)topo->num_graphs);
  AG_CHECK_LAUNCH();
  k_graph<true><<<dim3((unsigned)topo->num_graphs), dim3(bd), smem, st>>>(a);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}

extern "C" int agdiff_local_lengths(const agdiff_topo_t* topo, const agdiff_ws_t* ws, const float* pos, void* stream) {
  if (!topo || !ws || !pos || !ws->l_len) return AGDIFF_ERR_ARG;
  if (topo->num_local == 0) return AGDIFF_OK;
  const int L = (int)topo->num_local;
  k_local_lengths<<<(L + 255) / 256, 256, 0, (hipStream_t)stream>>>(topo->loc_src, topo->loc_dst, pos, ws->l_len, L);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}
