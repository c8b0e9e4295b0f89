// Whole-path entry points: ABI self-description, the score network (dualenc.py:142-251) and one
// denoising step (dualenc.py:478-545), composed from the per-op launchers.
#include "common.hpp"
#include <algorithm>
#include <mutex>
#include <vector>

extern "C" int agdiff_abi_version(void) { return AGDIFF_ABI_VERSION; }

// Tiles per chunk of the fused CFConv: every one of the 2,048 resident waves (256 CUs x 8) should walk at least
// 16 chunks, so that the uneven last round of the grid-stride loop stays a small tail (measured at 100 k tiles:
// 8-tile chunks = 6.1 chunks per wave cost 4 % over 2-tile chunks); at most AGDIFF_MAX_CHUNK_TILES.
extern "C" int agdiff_conv_chunk_tiles(int64_t max_edges) {
  const int64_t tiles = (max_edges + AG_TW - 1) / AG_TW;
  int c = 1;
  while (c < AGDIFF_MAX_CHUNK_TILES && tiles / (2 * c) >= 16 * 2048) c *= 2;
  return c;
}

// include/agdiff_hip.h: agdiff_group_order (host only).  Sizes are a molecule's: n <= 512 atoms, k <= ~8 types.
extern "C" int agdiff_group_order(const int32_t* need, int32_t n, int32_t k, int32_t gt, int32_t* order_out) {
  if (n < 0 || k < 0 || (gt != 1 && gt != 2 && gt != 4) || (n > 0 && (!need || !order_out))) return AGDIFF_ERR_ARG;
  if (n == 0) return AGDIFF_OK;
  auto row = [&](int i) { return need + (size_t)i * k; };
  std::vector<int> total(n, 0);
  for (int i = 0; i < n; ++i)
    for (int t = 0; t < k; ++t) total[i] += row(i)[t];
  // start r < k: ascending by the need vector read from type r on, cyclically; start k: ascending by total need (stable)
  auto start_order = [&](int r, std::vector<int>& o) {
    o.resize(n);
    for (int i = 0; i < n; ++i) o[i] = i;
    std::stable_sort(o.begin(), o.end(), [&](int a, int b) {
      if (r == k) return total[a] < total[b];
      for (int t = 0; t < k; ++t) {
        const int c = (r + t) % k;
        if (row(a)[c] != row(b)[c]) return row(a)[c] < row(b)[c];
      }
      return false;
    });
  };
  const int ng = (n + gt - 1) / gt;
  std::vector<int> grp((size_t)ng * gt), best;
  long best_cost = -1;
  auto cost = [&](const int* g) {            // sum over the types of the max over the group's atoms (-1: unused slot)
    long c = 0;
    for (int t = 0; t < k; ++t) {
      int m = 0;
      for (int s = 0; s < gt; ++s)
        if (g[s] >= 0 && row(g[s])[t] > m) m = row(g[s])[t];
      c += m;
    }
    return c;
  };
  std::vector<int> o;
  std::vector<long> costs(ng);
  for (int r = 0; r <= k; ++r) {
    if (r == k && k == 0) break;
    start_order(r, o);
    for (size_t i = 0; i < grp.size(); ++i) grp[i] = i < (size_t)n ? o[i] : -1;
    if (gt > 1 && n > gt) {
      for (int g = 0; g < ng; ++g) costs[g] = cost(&grp[(size_t)g * gt]);
      for (int pass = 0; pass < 16; ++pass) {
        bool improved = false;
        for (int a = 0; a < ng; ++a)
          for (int b = a + 1; b < ng; ++b) {
            int* ga = &grp[(size_t)a * gt];
            int* gb = &grp[(size_t)b * gt];
            for (int ia = 0; ia < gt; ++ia)
              for (int ib = 0; ib < gt; ++ib) {
                if (ga[ia] < 0 || gb[ib] < 0) continue;
                std::swap(ga[ia], gb[ib]);
                const long ca = cost(ga), cb = cost(gb);
                if (ca + cb < costs[a] + costs[b]) {
                  costs[a] = ca;
                  costs[b] = cb;
                  improved = true;
                } else {
                  std::swap(ga[ia], gb[ib]);
                }
              }
          }
        if (!improved) break;
      }
    }
    long c = 0;
    for (int g = 0; g < ng; ++g) c += cost(&grp[(size_t)g * gt]);
    if (best_cost < 0 || c < best_cost) {
      best_cost = c;
      best = grp;
    }
    if (k == 0) break;
  }
  int w = 0;
  for (size_t i = 0; i < best.size(); ++i)
    if (best[i] >= 0) order_out[w++] = best[i];
  return w == n ? AGDIFF_OK : AGDIFF_ERR_ARG;
}

extern "C" int agdiff_struct_sizes(int64_t* out) {
  if (!out) return AGDIFF_ERR_ARG;
  out[0] = sizeof(agdiff_conv_params_t);
  out[1] = sizeof(agdiff_gin_params_t);
  out[2] = sizeof(agdiff_head_params_t);
  out[3] = sizeof(agdiff_params_t);
  out[4] = sizeof(agdiff_topo_t);
  out[5] = sizeof(agdiff_ws_t);
  out[6] = sizeof(agdiff_step_args_t);
  return AGDIFF_OK;
}

// ---- in-step timing of the CFConv launches (bench.py's roofline object): when switched on, agdiff_score_forward brackets
// every CFConv launch of the global branch (agdiff_cfconv_node [+ agdiff_cfconv_local], or agdiff_cfconv_fused) with a pair
// of HIP events on the stream it launches them on; agdiff_profile_cfconv_read synchronises them and returns the sum.
namespace {
struct CfconvProfile {
  std::mutex mu;
  bool on = false;
  std::vector<hipEvent_t> pool;      // pairs: [2 i] before, [2 i + 1] after
  size_t used = 0;
};
CfconvProfile g_prof;
struct ProfScope {                   // records the pair around one block's CFConv launches when profiling is on
  hipStream_t st;
  hipEvent_t after = nullptr;
  explicit ProfScope(void* stream) : st((hipStream_t)stream) {
    std::lock_guard<std::mutex> lock(g_prof.mu);
    if (!g_prof.on) return;
    if (g_prof.used + 2 > g_prof.pool.size()) {
      hipEvent_t a = nullptr, b = nullptr;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      g_prof.pool.push_back(a);
      g_prof.pool.push_back(b);
    }
    if (hipEventRecord(g_prof.pool[g_prof.used], st) != hipSuccess) return;
    after = g_prof.pool[g_prof.used + 1];
    g_prof.used += 2;
  }
  ~ProfScope() {
    if (after) (void)hipEventRecord(after, st);
  }
};
}  // namespace

extern "C" int agdiff_profile_cfconv(int32_t enable) {
  std::lock_guard<std::mutex> lock(g_prof.mu);
  g_prof.on = enable != 0;
  if (enable) g_prof.used = 0;
  return AGDIFF_OK;
}

extern "C" int agdiff_profile_cfconv_read(double* total_ms, int64_t* launches) {
  if (!total_ms || !launches) return AGDIFF_ERR_ARG;
  std::lock_guard<std::mutex> lock(g_prof.mu);
  double sum = 0.0;
  int64_t n = 0;
  for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
    float ms = 0.0f;
    if (hipEventSynchronize(g_prof.pool[i + 1]) != hipSuccess) return AGDIFF_ERR_LAUNCH;
    if (hipEventElapsedTime(&ms, g_prof.pool[i], g_prof.pool[i + 1]) != hipSuccess) return AGDIFF_ERR_LAUNCH;
    sum += ms;
    ++n;
  }
  g_prof.used = 0;
  *total_ms = sum;
  *launches = n;
  return AGDIFF_OK;
}

#define AG_TRY(call)            \
  do {                          \
    int _rc = (call);           \
    if (_rc != AGDIFF_OK) return _rc; \
  } while (0)

namespace {
// The local branch (bond graph: lengths -> encoder -> GIN -> local head) and the global branch (radius graph ->
// encoder -> SchNet -> global head) only share `pos`, so they are forked onto two HIP streams and joined before
// returning: the local branch's small, latency-bound launches fill the gaps the global branch's node stages and
// graph kernels leave.  One side stream and two events per device, created on first use (HIP objects only; no
// device memory).  The fork/join is event-based, hence also capturable into a hipGraph.
struct ForkJoin {
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr, rows = nullptr;
  bool ok = false;
};
ForkJoin& fork_join_for_current_device() {
  static ForkJoin fj[16];
  static std::mutex mu;                      // first use may come from several host threads
  int dev = 0;
  (void)hipGetDevice(&dev);
  ForkJoin& f = fj[dev & 15];
  std::lock_guard<std::mutex> lock(mu);
  if (!f.ok) {
    f.ok = hipStreamCreateWithFlags(&f.side, hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&f.fork, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&f.join, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&f.rows, hipEventDisableTiming) == hipSuccess;
  }
  return f;
}

// Local branch = lengths -> edge encoder (local rows) -> GIN -> local head   (dualenc.py:214-239).
// `rows_from_global`: the global branch's encoder pass writes ws->l_attr_rows itself (the local edges are a subset of
// the edge set it walks and dualenc.py:214-216 evaluates the SAME encoder on them), so the pass over the local list is
// skipped and everything after it waits for `rows_ready`.
// `split` (CFConv by filter polynomials): the local edges' CFConv inputs are prepared here too -- their scales by quad-tile row
// (ws->lt_scale) when agdiff_cfconv_node takes their filters from polynomials, else their operand-form attributes and
// scales by padded-list position (ws->l_attr_frag, ws->l_scale) for agdiff_cfconv_local; `split_ready` is recorded once
// they are enqueued (the CFConv launches on the other stream wait for it).
int local_branch(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, const float* pos, int flags,
                 bool rows_from_global, hipEvent_t rows_ready, void* stream, bool split = false,
                 hipEvent_t split_ready = nullptr) {
  const int64_t ltiles = (topo->num_local + AG_TW - 1) / AG_TW;
  const int64_t ctiles = (topo->num_local_canon + AG_TW - 1) / AG_TW;
  // caller-supplied lengths (forward(edge_length=...)) need not be symmetric: then every local edge is evaluated
  const bool canon = !(flags & AGDIFF_FWD_GRAPH_GIVEN) && topo->num_local_canon > 0;
  // (the fused sampler front has written the lengths -- and the quad-tile scales -- already)
  const bool front_did_local = (flags & AGDIFF_FWD_GRAPH_READY) != 0;
  if (!(flags & AGDIFF_FWD_GRAPH_GIVEN) && !front_did_local) AG_TRY(agdiff_local_lengths(topo, ws, pos, stream));
  if (rows_from_global) {
    if (hipStreamWaitEvent((hipStream_t)stream, rows_ready, 0) != hipSuccess) return AGDIFF_ERR_LAUNCH;
  } else if (topo->num_local > 0) {
    if (canon && split) {
      const int lp = agdiff_local_poly_enabled(p, topo, ws);
      if (lp != 0 && !front_did_local) AG_TRY(agdiff_edge_scales_split(p, topo, ws, 2, stream));   // scales by quad-tile row (agdiff_cfconv_node)
      if (lp == 1) {        // the local CFConv takes every filter from polynomials: only the rows are needed
        AG_TRY(agdiff_local_edge_rows(p, topo, ws, stream));
      } else {              // (some) local edges through the filter MLPs: rows and the operand-form copy at the padded-list
                            // positions of the edge and of its mirror; in a mixed batch the slotted types' scales stay 0 there
        AG_TRY(agdiff_edge_encoder(p, ws->num_local_canon, ctiles, ws->lc_len, topo->lc_type, ws->l_attr_frag, ws->l_attr_rows,
                                   topo->lp_row, topo->lc_ppos, topo->lc_pmir, stream));
        AG_TRY(agdiff_edge_scales_split(p, topo, ws, 1, stream));
      }
    } else if (canon)        // one evaluation and one row per mirror pair of local edges (polynomials where they apply)
      AG_TRY(agdiff_local_edge_rows(p, topo, ws, stream));
    else
      AG_TRY(agdiff_edge_encoder(p, ws->num_local, ltiles, ws->l_len, topo->loc_type, nullptr, ws->l_attr_rows, nullptr,
                                 nullptr, nullptr, stream));
  }
  if (split && split_ready && hipEventRecord(split_ready, (hipStream_t)stream) != hipSuccess) return AGDIFF_ERR_LAUNCH;
  AG_TRY(agdiff_gin_encoder(p, topo, ws, canon ? 1 : 0, stream));
  if (topo->num_local > 0) {
    agdiff_head_params_t hl = p->head_local;       // (the hidden layer's range flags go to this workspace)
    hl.range_rows = ws->range_rows;
    if (canon)
      AG_TRY(agdiff_pair_head(&hl, ws->num_local_canon, ctiles, topo->lc_src, topo->lc_dst, ws->hl, nullptr,
                              ws->l_attr_rows, topo->lc_pos, topo->lc_mir, ws->l_inv, stream));
    else
      AG_TRY(agdiff_pair_head(&hl, ws->num_local, ltiles, topo->loc_src, topo->loc_dst, ws->hl, nullptr,
                              ws->l_attr_rows, nullptr, nullptr, ws->l_inv, stream));
  }
  return AGDIFF_OK;
}

// radius graph -> scales -> edge encoder   (dualenc.py:167-191)
int global_front(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, const float* pos, int flags,
                 bool rows_for_local, void* stream) {
  const int64_t etiles = (topo->max_edges + AG_TW - 1) / AG_TW;
  if (!(flags & AGDIFF_FWD_GRAPH_GIVEN))     // cutoff 0 admits no radius edge: the bond graph alone (extend_radius=False)
    AG_TRY(agdiff_graph_build(topo, ws, pos, (flags & AGDIFF_FWD_NO_RADIUS) ? 0.0f : p->cutoff, stream));
  AG_TRY(agdiff_edge_scales(p, topo, ws, (flags & AGDIFF_FWD_GRAPH_GIVEN) ? 0 : 1, stream));
  if (flags & AGDIFF_FWD_GRAPH_GIVEN) {       // caller's edge list: no canonical list, one encoder evaluation per edge
    AG_TRY(agdiff_edge_encoder(p, ws->num_edges, etiles, ws->e_len, ws->e_type, ws->e_attr, nullptr, nullptr, nullptr,
                               nullptr, stream));
  } else {
    // one evaluation per canonical edge, written to its own and its mirror's slot of e_attr (and, for local edges,
    // to their rows of l_attr_rows)
    AG_TRY(agdiff_edge_encoder(p, ws->num_canon, etiles, ws->c_len, ws->c_type, ws->e_attr,
                               rows_for_local ? ws->l_attr_rows : nullptr, rows_for_local ? ws->e_loc : nullptr, ws->c_pos,
                               ws->c_mir, stream));
  }
  return AGDIFF_OK;
}

// SchNet -> global head   (dualenc.py:193-211)
int global_back(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, int flags, void* stream) {
  const int64_t etiles = (topo->max_edges + AG_TW - 1) / AG_TW;
  for (int k = 0; k <= p->num_convs; ++k) {
    AG_TRY(agdiff_schnet_node_stage(p, topo, ws, k, stream));
    if (k < p->num_convs) {
      ProfScope prof(stream);
      AG_TRY(agdiff_cfconv_fused(p, topo, ws, k, stream));
    }
  }
  agdiff_head_params_t hg = p->head_global;
  hg.range_rows = ws->range_rows;
  if (flags & AGDIFF_FWD_GRAPH_GIVEN) {
    AG_TRY(agdiff_pair_head(&hg, ws->num_edges, etiles, ws->e_src, ws->e_dst, ws->h, ws->e_attr, nullptr,
                            nullptr, nullptr, ws->e_inv_global, stream));
  } else {
    AG_TRY(agdiff_pair_head(&hg, ws->num_canon, etiles, ws->c_src, ws->c_dst, ws->h, ws->e_attr, nullptr,
                            ws->c_pos, ws->c_mir, ws->e_inv_global, stream));
  }
  return AGDIFF_OK;
}

// CFConv by filter polynomials (p->poly_kt > 0, graph built here): one agdiff_cfconv_node launch per block (include/agdiff_hip.h).
// front: radius graph -> radius-row scales (-> the encoder over all canonical edges when the full head will need edge_attr)
int global_front_split(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, const float* pos, int flags,
                       void* stream) {
  const int64_t etiles = (topo->max_edges + AG_TW - 1) / AG_TW;
  // inside the denoising loop nothing but the polynomial head walks the canonical list: radius edges only then (one head
  // evaluation per mirror pair of radius edges)
  const bool ronly = (flags & AGDIFF_FWD_SAMPLER) != 0;
  if ((flags & AGDIFF_FWD_GRAPH_READY) && ronly) {
    if (!(flags & AGDIFF_FWD_GRAPH_PENDING)) return AGDIFF_OK;          // agdiff_sampler_front has built this step's graph
    agdiff_step_args_t sa = {};                                          // ... or left its graph phase to this side of the fork
    sa.pos_in = pos;
    return agdiff_sampler_front(p, topo, ws, &sa, 2 | ((flags & AGDIFF_FWD_PARITY) ? 16 : 0),
                                (flags & AGDIFF_FWD_NO_RADIUS) ? 0.0f : p->cutoff, stream);
  }
  // (the radius rows' scales and pad rows come out of the graph build's fill pass)
  AG_TRY(agdiff_graph_build_scaled(p, topo, ws, pos, (flags & AGDIFF_FWD_NO_RADIUS) ? 0.0f : p->cutoff, ronly ? 1 : 0, stream));
  if (!(flags & AGDIFF_FWD_SAMPLER))
    AG_TRY(agdiff_edge_encoder(p, ws->num_canon, etiles, ws->c_len, ws->c_type, ws->e_attr, nullptr, nullptr, ws->c_pos,
                               ws->c_mir, stream));
  return AGDIFF_OK;
}
// back: SchNet with polynomial CFConvs -> global head.  `local_ready`: the local edges' CFConv inputs (ws->lt_len / lt_scale,
// or ws->l_attr_frag / l_scale) are complete (recorded on the local branch's stream), or null when the local branch ran on
// this stream.
int global_back_split(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, int flags,
                      hipEvent_t local_ready, void* stream) {
  const int64_t etiles = (topo->max_edges + AG_TW - 1) / AG_TW;
  // stage 0 (embeddings + block 0's lin1) is the same on every step of a run: written to ws->h0 / ws->xs0 once, block 0
  // reads it from there
  const bool cache = ws->h0 && ws->xs0;
  // local edges: inside agdiff_cfconv_node (per-type polynomials), or agdiff_cfconv_local's second aggregate (filter MLPs)
  const bool local_mlp = topo->num_local > 0 && agdiff_local_poly_enabled(p, topo, ws) != 1;    // none or only some by polynomials
  const int sp = 1 | (local_mlp ? 8 : 0);
  if (!(cache && (flags & AGDIFF_FWD_STAGE0_CACHED))) AG_TRY(agdiff_schnet_node_stage_split(p, topo, ws, 0, sp | 4, stream));
  agdiff_ws_t ws0 = *ws;
  if (cache) ws0.xs = ws->xs0;
  for (int k = 0; k < p->num_convs; ++k) {
    const agdiff_ws_t* wk = (k == 0) ? &ws0 : ws;
    // (the local edges' inputs come from the side stream -- unless the fused front wrote them on this one and no MLP pass needs more)
    if (k == 0 && local_ready && (local_mlp || !(flags & AGDIFF_FWD_GRAPH_READY)) &&
        hipStreamWaitEvent((hipStream_t)stream, local_ready, 0) != hipSuccess)
      return AGDIFF_ERR_LAUNCH;
    {
      ProfScope prof(stream);
      AG_TRY(agdiff_cfconv_node(p, topo, wk, k, stream));
      if (local_mlp) AG_TRY(agdiff_cfconv_local(p, topo, wk, k, stream));
    }
    AG_TRY(agdiff_schnet_node_stage_split(p, topo, ws, k + 1, sp | (k == 0 ? 2 : 0), stream));
  }
  if (flags & AGDIFF_FWD_SAMPLER) {
    // only the radius edges' outputs are used (dualenc.py:516-518): the head's edge_attr half from the d-polynomial, over
    // the canonical list (a mirror pair of radius edges has one length and h_i * h_j is symmetric)
    ag_log_variant(ws, AGDIFF_VAR_HEAD_POLY);
    if (flags & AGDIFF_FWD_GRAPH_READY)       // segmented canonical list, results by radius row (ws->inv_r)
      AG_TRY(agdiff_pair_head_poly_rows(p, topo, ws, (flags & AGDIFF_FWD_PARITY) ? 1 : 0, stream));
    else {
      agdiff_params_t pp = *p;                      // (the hidden layer's range flags go to this workspace)
      pp.head_global.range_rows = ws->range_rows;
      AG_TRY(agdiff_pair_head_poly(&pp, ws->num_canon, etiles, ws->c_src, ws->c_dst, ws->c_len, ws->h, ws->c_pos, ws->c_mir,
                                   ws->e_inv_global, stream));
    }
  } else {
    agdiff_head_params_t hg = p->head_global;
    hg.range_rows = ws->range_rows;
    AG_TRY(agdiff_pair_head(&hg, ws->num_canon, etiles, ws->c_src, ws->c_dst, ws->h, ws->e_attr, nullptr,
                            ws->c_pos, ws->c_mir, ws->e_inv_global, stream));
  }
  return AGDIFF_OK;
}
}  // namespace

extern "C" int agdiff_score_forward(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                                    const float* pos, int32_t flags, void* stream) {
  if (!p || !topo || !ws || !pos) return AGDIFF_ERR_ARG;
  if (!(flags & AGDIFF_FWD_GLOBAL)) return local_branch(p, topo, ws, pos, flags, false, nullptr, stream);
  // the encoder pass over all edges also fills the local rows when the graph came from agdiff_graph_build (e_loc) ...
  // ... for batches above ~8 k atoms.  Below, the local branch runs its own encoder pass over the canonical local list:
  // it then does not wait for the global encoder and overlaps the graph build and the encoder instead of the first
  // CFConv launches, whose persistent workgroups it would delay (1 molecule x 25 / 100 / 400 conformers:
  // 0.415 / 0.645 / 1.82 ms per step against 0.441 / 0.668 / 1.79): agdiff_params_t.tune_share_rows_min_nodes.
  const int64_t share_min = ag_tune(p->tune_share_rows_min_nodes, 8192);
  const bool share_rows = !(flags & AGDIFF_FWD_GRAPH_GIVEN) && topo->num_local > 0 && ws->e_loc && topo->num_nodes >= share_min;
  const bool serial = p->tune_serial_branches != 0;
  ForkJoin& fj = fork_join_for_current_device();
  hipStream_t main = (hipStream_t)stream;
  const bool split = p->poly_kt > 0 && !(flags & AGDIFF_FWD_GRAPH_GIVEN) && ws->rad_cnt != nullptr;
  if (!(serial || !fj.ok)) ag_log_variant(ws, AGDIFF_VAR_SIDE_STREAM);
  if (split) {
    if (serial || !fj.ok) {
      AG_TRY(global_front_split(p, topo, ws, pos, flags, stream));
      AG_TRY(local_branch(p, topo, ws, pos, flags, false, nullptr, stream, true, nullptr));
      return global_back_split(p, topo, ws, flags, nullptr, stream);
    }
    if (hipEventRecord(fj.fork, main) != hipSuccess || hipStreamWaitEvent(fj.side, fj.fork, 0) != hipSuccess)
      return AGDIFF_ERR_LAUNCH;
    AG_TRY(local_branch(p, topo, ws, pos, flags, false, nullptr, (void*)fj.side, true, fj.rows));
    AG_TRY(global_front_split(p, topo, ws, pos, flags, stream));
    if (hipEventRecord(fj.join, fj.side) != hipSuccess) return AGDIFF_ERR_LAUNCH;
    AG_TRY(global_back_split(p, topo, ws, flags, topo->num_local > 0 ? fj.rows : nullptr, stream));
    if (hipStreamWaitEvent(main, fj.join, 0) != hipSuccess) return AGDIFF_ERR_LAUNCH;
    return AGDIFF_OK;
  }
  if (share_rows) ag_log_variant(ws, AGDIFF_VAR_SHARE_ROWS);
  if (serial || !fj.ok) {
    AG_TRY(global_front(p, topo, ws, pos, flags, share_rows, stream));
    if (share_rows && fj.ok && hipEventRecord(fj.rows, main) != hipSuccess) return AGDIFF_ERR_LAUNCH;
    AG_TRY(local_branch(p, topo, ws, pos, flags, share_rows && fj.ok, fj.rows, stream));
    return global_back(p, topo, ws, flags, stream);
  }
  if (hipEventRecord(fj.fork, main) != hipSuccess || hipStreamWaitEvent(fj.side, fj.fork, 0) != hipSuccess)
    return AGDIFF_ERR_LAUNCH;
  AG_TRY(global_front(p, topo, ws, pos, flags, share_rows, stream));
  if (share_rows && hipEventRecord(fj.rows, main) != hipSuccess) return AGDIFF_ERR_LAUNCH;
  AG_TRY(local_branch(p, topo, ws, pos, flags, share_rows, fj.rows, (void*)fj.side));
  if (hipEventRecord(fj.join, fj.side) != hipSuccess) return AGDIFF_ERR_LAUNCH;
  AG_TRY(global_back(p, topo, ws, flags, stream));
  if (hipStreamWaitEvent(main, fj.join, 0) != hipSuccess) return AGDIFF_ERR_LAUNCH;
  return AGDIFF_OK;
}

extern "C" int agdiff_langevin_step(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                                    const agdiff_step_args_t* a, void* stream) {
  if (!a) return AGDIFF_ERR_ARG;
  AG_TRY(agdiff_score_forward(p, topo, ws, a->pos_in, (a->use_global ? AGDIFF_FWD_GLOBAL : 0) | AGDIFF_FWD_SAMPLER, stream));
  return agdiff_langevin_update(topo, ws, a, stream);
}

// ---- one denoising step as a replayable HIP graph (include/agdiff_hip.h: agdiff_step_graph_*)
namespace {
__global__ void k_bump_word(int32_t* p) {
  if (threadIdx.x == 0) ++*p;
}
}  // namespace

extern "C" int agdiff_step_graph_capture(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                                         const agdiff_step_args_t* host_step, const agdiff_step_args_t* step_table, int32_t* step_index,
                                         int32_t front_mode, float cutoff, int32_t fwd_flags, void* stream, void** graph_out) {
  if (!p || !topo || !ws || !host_step || !step_table || !step_index || !graph_out || !stream) return AGDIFF_ERR_ARG;
  {
    std::lock_guard<std::mutex> lock(g_prof.mu);       // (event pairs around the CFConv launches are not replayable: no graphs while
    if (g_prof.on) return AGDIFF_ERR_ARG;              // agdiff_profile_cfconv is on)
  }
  hipStream_t st = (hipStream_t)stream;
  *graph_out = nullptr;
  if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) return AGDIFF_ERR_LAUNCH;
  int rc = ag_sampler_front_table(p, topo, ws, host_step, step_table, step_index, front_mode, cutoff, stream);
  if (rc == AGDIFF_OK) rc = agdiff_score_forward(p, topo, ws, host_step->pos_in, fwd_flags, stream);
  if (rc == AGDIFF_OK) {
    k_bump_word<<<1, 64, 0, st>>>(step_index);
    if (hipGetLastError() != hipSuccess) rc = AGDIFF_ERR_LAUNCH;
  }
  hipGraph_t graph = nullptr;
  const hipError_t ec = hipStreamEndCapture(st, &graph);
  if (rc != AGDIFF_OK || ec != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc != AGDIFF_OK ? rc : AGDIFF_ERR_LAUNCH;
  }
  hipGraphExec_t exec = nullptr;
  const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ei != hipSuccess || !exec) return AGDIFF_ERR_LAUNCH;
  *graph_out = (void*)exec;
  return AGDIFF_OK;
}

extern "C" int agdiff_step_graph_launch(void* graph, void* stream) {
  if (!graph) return AGDIFF_ERR_ARG;
  return hipGraphLaunch((hipGraphExec_t)graph, (hipStream_t)stream) == hipSuccess ? AGDIFF_OK : AGDIFF_ERR_LAUNCH;
}

extern "C" int agdiff_step_graph_destroy(void* graph) {
  if (!graph) return AGDIFF_OK;
  return hipGraphExecDestroy((hipGraphExec_t)graph) == hipSuccess ? AGDIFF_OK : AGDIFF_ERR_LAUNCH;
}
