// The serial front of a denoising step as ONE launch per step (include/agdiff_hip.h: agdiff_sampler_front): one workgroup
// per molecule does the Langevin update of step t (eq_transform x 2, clip_norm, move, NaN check, center_pos, clamp:
// geometry.py:9-17, dualenc.py:506-545, 581-589) with the molecule's positions and radius adjacency in LDS, and then -- on
// the positions it has just written -- the radius graph of step t + 1 (torch_cluster.radius_graph + union with the static
// local edges, models/common.py:208-233) in the lean form the polynomial path consumes: radius rows by target with their
// CFConv scales and pad rows (agdiff_cfconv_node), and the canonical radius list (polynomial global head).
// Nothing else of the full edge list (in_ptr / out_ptr / ref2dst / e_*) is needed inside the loop: the update finds a node's
// in-edges in its own radius rows and its out-edges in the columns of the adjacency masks (the rank of a source inside a
// row is a population count: rows hold their sources in ascending order), with the edge lengths recomputed from the
// positions in LDS by the very expression the graph build used (bit-identical).
// Replaces, per step, k_langevin_update + k_graph<count> + k_scan_graphs + k_graph<fill> + k_rad_scales.
#include "common.hpp"

namespace {

struct FrontArgs {
  agdiff_step_args_t s;        // the update's step (valid with do_update) ...
  const agdiff_step_args_t* step_table;   // ... or, when non-null, entry *step_index of this DEVICE table: a launch replayed from a
  const int32_t* step_index;              // HIP graph (agdiff_step_graph_capture) takes its step from memory, not from its arguments
  int32_t do_update;
  int32_t do_graph;
  const int32_t* graph_ptr;
  // static local edges
  const int32_t* loc_src;
  const int32_t* loc_dst;
  const int32_t* loc_out_ptr;
  const int32_t* loc_in_ptr;
  const int32_t* loc_in_eid;
  const float* l_len;
  const float* l_inv;
  // radius rows (AGDIFF_RAD_STRIDE per target): read by the update (graph of step t), rewritten by the graph phase
  int32_t* rad_cnt;
  int32_t* rad_src;
  float* rad_len;
  float* r_scale;
  const float* dist_union;   // agdiff_params_t.dist_union or null (then: one search per conv in dw[])
  int32_t union_kinks;       // K, S of agdiff_params_t
  int32_t union_segments;
  int64_t rpad;
  int32_t pad_rows;            // write the pad rows that complete a target's last 16-row radius tile
  const uint32_t* loc_bits;    // topo->loc_bits or null: [N][words] static local in-adjacency masks (bit j of row i: local edge j -> i)
  const float* inv_r;          // [N * AGDIFF_RAD_STRIDE] global head output by radius row
  const float* dw[2 * AGDIFF_MAX_CONVS];
  int32_t n_scales;
  float cutoff;
  float r2;                    // (0 with extend_radius = False: no radius edge)
  int32_t smooth;
  // canonical radius list: every molecule claims a contiguous range of the list with ONE atomic add on the step's counter (the
  // order of the molecules in the list then depends on the order the workgroups get there -- harmless: each entry's result
  // goes to fixed radius rows, nothing depends on where in the list it sits; inside a range the order is fixed)
  int32_t* canon_counter;      // [2]: this step's live count (becomes ws->num_canon), and the other parity's, zeroed here
  int32_t parity;
  float* c_len;
  int32_t* c_src;
  int32_t* c_dst;
  int32_t* c_pos;              // radius row of the entry's edge
  int32_t* c_mir;              // radius row of its mirror, or -1
  // local phase (do_local): lengths of the molecule's canonical local edges to every layout that holds them, and their CFConv
  // scales by quad-tile row -- everything the first CFConv of the next forward needs from the local edges
  int32_t do_local;
  const int32_t* lcm_ptr;      // [G + 1] canonical local edges of a molecule
  const int32_t* lc_src;
  const int32_t* lc_dst;
  const int32_t* lc_pos;
  const int32_t* lc_mir;
  const int32_t* lc_ppos;      // (with l_len_p) padded-list positions
  const int32_t* lc_pmir;
  const int32_t* lc_tpos;      // (with lt_len / lt_scale) quad-tile rows
  const int32_t* lc_tmir;
  float* l_len_w;
  float* lc_len;
  float* l_len_p;
  float* lt_len;
  float* lt_scale;
  int64_t tpad;                // 16 T
  int32_t* nan_flag;
  int32_t parts;               // lanes per atom in the update (power of two, 1..16)
  int32_t words;               // 32-bit words per mask row (even)
};

__device__ __forceinline__ float fr_sqrt_rn(float x) { return (float)sqrt((double)x); }
__device__ __forceinline__ float fr_dist2(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
  return (xx + yy) + zz;
}
__device__ __forceinline__ void fr_clip3(float& x, float& y, float& z, float limit) {  // dualenc.py:586-589
  const float nrm = sqrtf(x * x + y * y + z * z);
  if (nrm > limit) {
    const float d = limit / nrm;
    x *= d; y *= d; z *= d;
  }
}

extern __shared__ uint32_t ag_front_smem[];

#define AG_FRONT_THREADS 1024
__global__ void __launch_bounds__(AG_FRONT_THREADS) k_sampler_front(FrontArgs a) {
  const agdiff_step_args_t S = a.step_table ? a.step_table[*a.step_index] : a.s;
  const int g = blockIdx.x;
  const int g0 = a.graph_ptr[g];
  const int n = a.graph_ptr[g + 1] - g0;
  const int words = a.words;
  const int nmax = words * 32;
  // LDS: pos[3 nmax] | new pos[3 nmax] | canonical counts / scan[nmax] | radbits[nmax][words] | locbits[nmax][words]
  float* spos = reinterpret_cast<float*>(ag_front_smem);
  float* snew = spos + 3 * nmax;
  int* scan_c = reinterpret_cast<int*>(snew + 3 * nmax);
  uint32_t* radbits = reinterpret_cast<uint32_t*>(scan_c + nmax);
  uint32_t* locbits = radbits + nmax * words;
  // the distance-weighting networks over their common segments (a.dist_union): K kinks + S x n_scales lines
  float* skink = reinterpret_cast<float*>(locbits + nmax * words);
  const float2* sline = reinterpret_cast<const float2*>(skink + a.union_kinks);
  const bool by_union = a.dist_union != nullptr;
  __shared__ float red[3][AG_FRONT_THREADS / 64];
  __shared__ int nanw[AG_FRONT_THREADS / 64];
  float* sseg = skink;          // (without the union table: the per-conv tables, n_scales x 100 floats, in its place)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));     // lanes below mine
  auto bit = [&](const uint32_t* rows, int r, int c) -> bool { return (rows[r * words + (c >> 5)] >> (c & 31)) & 1u; };
  auto rank_below = [&](const uint32_t* rows, int r, int c) -> int {   // set bits of row r below column c
    const uint32_t* row = rows + r * words;
    int k = __popc(row[c >> 5] & ((1u << (c & 31)) - 1u));
    for (int w = 0; w < (c >> 5); ++w) k += __popc(row[w]);
    return k;
  };

  for (int i = threadIdx.x; i < 3 * n; i += blockDim.x) spos[i] = S.pos_in[3 * (size_t)g0 + i];
  if ((a.do_graph || a.do_local) && !by_union)
    for (int i = threadIdx.x; i < a.n_scales * 100; i += blockDim.x) sseg[i] = a.dw[i / 100][i % 100];
  if ((a.do_graph || a.do_local) && by_union) {
    const int words16 = (a.union_kinks + a.union_segments * 2 * a.n_scales + 3) / 4;
    ag_copy_lds<4>((lds_u32x4*)(ag_front_smem + (7 * nmax + 2 * nmax * words)), reinterpret_cast<const u32x4*>(a.dist_union), words16);
  }
  // lw_cc(d) C(d) of every CFConv for one length: ONE search among the union of the networks' kinks (log2 K steps), then per conv
  // the line its own table would have selected (the same floats: bit-identical to cf_dist_weight) and the sigmoid
  auto union_segment = [&](float d) -> int {
    int u = 0;
    for (int step = a.union_kinks >> 1; step >= 1; step >>= 1) u += (skink[u + step - 1] <= d) ? step : 0;
    return u;
  };
  auto scale_of = [&](int u, int cc, float d) -> float {
    const float2 ab = sline[u * a.n_scales + cc];
    return ag_sigmoid(fmaf(ab.x, d, ab.y));
  };

  // ================================================================== update of step t
  if (a.do_update) {
    const bool use_global = S.use_global != 0;
    if (use_global) {
      // radius in-adjacency of the graph the global scores were computed on, from its stored rows
      for (int k = threadIdx.x; k < n * words; k += blockDim.x) radbits[k] = 0u;
      __syncthreads();
      for (int i = wave; i < n; i += nwaves) {
        const int cnt = a.rad_cnt[g0 + i];
        if (lane < cnt) {
          const int j = a.rad_src[(size_t)(g0 + i) * AGDIFF_RAD_STRIDE + lane] - g0;
          atomicOr(&radbits[i * words + (j >> 5)], 1u << (j & 31));       // (bits: the order does not matter)
        }
      }
    }
    __syncthreads();
    const int was_bad = a.nan_flag[1 + g];       // quarantine of graphs that went NaN: see k_langevin_update (node.hip)
    float sx = 0.f, sy = 0.f, sz = 0.f;
    int bad = 0;
    // lanes per atom: as many as the workgroup has for THIS molecule (a power of two <= 16; the launcher's a.parts is the value
    // for the batch's largest molecule -- with it a 44-atom molecule in a batch that also holds a 181-atom one kept 468 of its 512
    // threads idle while 44 walked their edge lists alone)
    int P = 1;
    while (P < 16 && 2 * P * n <= (int)blockDim.x) P *= 2;
    if (P < a.parts) P = a.parts;
    const int part = threadIdx.x & (P - 1);
    const int per_pass = blockDim.x / P;
    for (int base = 0; base < n; base += per_pass) {
      const int li = base + (int)(threadIdx.x / P);
      const bool on = li < n;
      const int lic = on ? li : 0;
      const int i = g0 + lic;
      const float px = spos[3 * lic], py = spos[3 * lic + 1], pz = spos[3 * lic + 2];
      float lx = 0.f, ly = 0.f, lz = 0.f;
      if (on) {
        // local edges (static lists; lengths and scores of this step from the local branch): as k_langevin_update
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
            j[u] = ok[u] ? a.loc_dst[e] - g0 : lic;
            ln[u] = ok[u] ? a.l_len[e] : 1.0f;
            sc[u] = ok[u] ? a.l_inv[e] : 0.0f;
          }
#pragma unroll
          for (int u = 0; u < UL; ++u) {
            if (!ok[u]) continue;
            const float w = 1.0f / ln[u];
            lx += (w * (px - spos[3 * j[u]])) * sc[u];
            ly += (w * (py - spos[3 * j[u] + 1])) * sc[u];
            lz += (w * (pz - spos[3 * j[u] + 2])) * sc[u];
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
            j[u] = ok[u] ? a.loc_src[e[u]] - g0 : lic;
            ln[u] = ok[u] ? a.l_len[e[u]] : 1.0f;
            sc[u] = ok[u] ? a.l_inv[e[u]] : 0.0f;
          }
#pragma unroll
          for (int u = 0; u < UL; ++u) {
            if (!ok[u]) continue;
            const float w = 1.0f / ln[u];
            lx -= (w * (spos[3 * j[u]] - px)) * sc[u];
            ly -= (w * (spos[3 * j[u] + 1] - py)) * sc[u];
            lz -= (w * (spos[3 * j[u] + 2] - pz)) * sc[u];
          }
        }
      }
      for (int o = P >> 1; o > 0; o >>= 1) { lx += __shfl_xor(lx, o); ly += __shfl_xor(ly, o); lz += __shfl_xor(lz, o); }
      if (S.clip_local >= 0.0f) fr_clip3(lx, ly, lz, S.clip_local);
      float gx = 0.f, gy = 0.f, gz = 0.f;
      if (use_global) {
        if (on) {
          // radius edges only: edge_inv_global * (1 - local_edge_mask), dualenc.py:516-518.  Out-edges i -> k sit in row k
          // at the rank of i among its sources; in-edges j -> i are row i itself.  Lengths from the positions in LDS.
          for (int k = part; k < n; k += P) {
            if (!bit(radbits, k, lic)) continue;
            const float sc = a.inv_r[(size_t)(g0 + k) * AGDIFF_RAD_STRIDE + rank_below(radbits, k, lic)];
            const float qx = spos[3 * k], qy = spos[3 * k + 1], qz = spos[3 * k + 2];
            const float w = 1.0f / fr_sqrt_rn(fr_dist2(qx, qy, qz, px, py, pz));
            gx += (w * (px - qx)) * sc;
            gy += (w * (py - qy)) * sc;
            gz += (w * (pz - qz)) * sc;
          }
          const int cnt = a.rad_cnt[i];
          const size_t row = (size_t)i * AGDIFF_RAD_STRIDE;
          for (int k = part; k < cnt; k += P) {
            const int j = a.rad_src[row + k] - g0;
            const float sc = a.inv_r[row + k];
            const float qx = spos[3 * j], qy = spos[3 * j + 1], qz = spos[3 * j + 2];
            const float w = 1.0f / fr_sqrt_rn(fr_dist2(px, py, pz, qx, qy, qz));
            gx -= (w * (qx - px)) * sc;
            gy -= (w * (qy - py)) * sc;
            gz -= (w * (qz - pz)) * sc;
          }
        }
        for (int o = P >> 1; o > 0; o >>= 1) { gx += __shfl_xor(gx, o); gy += __shfl_xor(gy, o); gz += __shfl_xor(gz, o); }
        fr_clip3(gx, gy, gz, S.clip);
      }
      if (on && part == 0) {
        const float ex = lx + gx * S.w_global, ey = ly + gy * S.w_global, ez = lz + gz * S.w_global;
        const float nx = (px + (S.step_size * ex) / S.sigma) + S.noise[3 * i] * S.noise_scale;
        const float ny = (py + (S.step_size * ey) / S.sigma) + S.noise[3 * i + 1] * S.noise_scale;
        const float nz = (pz + (S.step_size * ez) / S.sigma) + S.noise[3 * i + 2] * S.noise_scale;
        bad |= (nx != nx) | (ny != ny) | (nz != nz);
        sx += nx; sy += ny; sz += nz;
        snew[3 * li] = nx; snew[3 * li + 1] = ny; snew[3 * li + 2] = nz;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o);
      bad |= __shfl_xor(bad, o);
    }
    if (lane == 0) { red[0][wave] = sx; red[1][wave] = sy; red[2][wave] = sz; nanw[wave] = bad; }
    __syncthreads();
    float cx = 0.f, cy = 0.f, cz = 0.f;
    int anybad = 0;
    for (int w = 0; w < nwaves; ++w) { cx += red[0][w]; cy += red[1][w]; cz += red[2][w]; anybad |= nanw[w]; }
    const float inv_n = 1.0f / (float)(n > 0 ? n : 1);
    cx *= inv_n; cy *= inv_n; cz *= inv_n;
    if (anybad && threadIdx.x == 0) {
      a.nan_flag[0] = 1;
      a.nan_flag[1 + g] = 1;
    }
    const bool frozen = was_bad || anybad;
    for (int li = threadIdx.x; li < n; li += blockDim.x) {
      const int i = g0 + li;
      float x = snew[3 * li] - cx, y = snew[3 * li + 1] - cy, z = snew[3 * li + 2] - cz;
      if (S.clip_pos >= 0.0f) {
        x = fminf(fmaxf(x, -S.clip_pos), S.clip_pos);
        y = fminf(fmaxf(y, -S.clip_pos), S.clip_pos);
        z = fminf(fmaxf(z, -S.clip_pos), S.clip_pos);
      }
      float tx = x, ty = y, tz = z;
      if (frozen) {          // placeholder: a centred straight chain, 1.5 apart (finite, no two atoms at one place)
        x = ((float)li - 0.5f * (float)(n - 1)) * 1.5f;
        y = z = 0.0f;
        tx = ty = tz = __uint_as_float(0x7FC00000u);
      }
      S.pos_out[3 * i] = x; S.pos_out[3 * i + 1] = y; S.pos_out[3 * i + 2] = z;
      if (S.traj_out) { S.traj_out[3 * i] = tx; S.traj_out[3 * i + 1] = ty; S.traj_out[3 * i + 2] = tz; }
      snew[3 * li] = x; snew[3 * li + 1] = y; snew[3 * li + 2] = z;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 3 * n; k += blockDim.x) spos[k] = snew[k];     // the graph below is that of the NEW positions
  }
  __syncthreads();
  // ================================================================== local edges of the next forward
  // get_distance on the static local edges (geometry.py:5-6), one evaluation per canonical local edge written to the edge's
  // and its mirror's slot of every layout, and lw(d) C(d) of all CFConvs by quad-tile row (agdiff_cfconv_node)
  if (a.do_local) {
    for (int c = a.lcm_ptr[g] + (int)threadIdx.x; c < a.lcm_ptr[g + 1]; c += blockDim.x) {
      const int sl = a.lc_src[c] - g0, dl = a.lc_dst[c] - g0;
      const float v = fr_sqrt_rn(fr_dist2(spos[3 * sl], spos[3 * sl + 1], spos[3 * sl + 2], spos[3 * dl], spos[3 * dl + 1], spos[3 * dl + 2]));
      const int mir = a.lc_mir[c];
      a.lc_len[c] = v;
      a.l_len_w[a.lc_pos[c]] = v;
      if (mir >= 0) a.l_len_w[mir] = v;
      if (a.l_len_p) {
        a.l_len_p[a.lc_ppos[c]] = v;
        if (a.lc_pmir[c] >= 0) a.l_len_p[a.lc_pmir[c]] = v;
      }
      if (a.lt_len) {
        const int tp = a.lc_tpos[c], tm = a.lc_tmir[c];
        a.lt_len[tp] = v;
        if (tm >= 0) a.lt_len[tm] = v;
        const float C = cf_envelope(v, a.cutoff, a.smooth);
        const int u = by_union ? union_segment(v) : 0;
        for (int cc = 0; cc < a.n_scales; ++cc) {
          const float sc = (by_union ? scale_of(u, cc, v) : cf_dist_weight(sseg + cc * 100, v)) * C;
          a.lt_scale[(size_t)cc * a.tpad + tp] = sc;
          if (tm >= 0) a.lt_scale[(size_t)cc * a.tpad + tm] = sc;
        }
      }
    }
  }
  if (!a.do_graph) return;

  // ================================================================== radius graph of step t + 1
  // static local in-adjacency of the molecule: the thread that owns target i sets its row
  if (a.loc_bits) {          // (the masks are static: the host built them with the topology, a coalesced copy instead of a chain of
                             // dependent index loads per target -- 15 % of the graph phase)
    for (int k = threadIdx.x; k < n * words; k += blockDim.x) locbits[k] = a.loc_bits[(size_t)g0 * words + k];
  } else
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    for (int w = 0; w < words; ++w) locbits[i * words + w] = 0u;
    for (int k = a.loc_in_ptr[g0 + i]; k < a.loc_in_ptr[g0 + i + 1]; ++k) {
      const int j = a.loc_src[a.loc_in_eid[k]] - g0;
      locbits[i * words + (j >> 5)] |= 1u << (j & 31);
    }
  }
  __syncthreads();
  // pass 1: radius_graph's rule (source j is kept for target i if d2(i, j) < r^2 and fewer than 33 such candidates, self
  // included, precede it in ascending j; self is then dropped), minus the local edges: the radius-only mask of every target
  for (int i = wave; i < n; i += nwaves) {
    const float xi = spos[3 * i], yi = spos[3 * i + 1], zi = spos[3 * i + 2];
    int cnt_r = 0;
    for (int c = 0; 64 * c < n; ++c) {
      const int j = 64 * c + lane;
      const bool valid = j < n;
      const int jj = valid ? j : 0;
      const float d2 = fr_dist2(xi, yi, zi, spos[3 * jj], spos[3 * jj + 1], spos[3 * jj + 2]);
      const bool within = valid && d2 < a.r2;
      const uint64_t wmask = __ballot(within);
      const bool rad = within && (cnt_r + __popcll(wmask & lt) < AGDIFF_RADIUS_CAP) && j != i;
      cnt_r += __popcll(wmask);
      const bool loc = valid && ((locbits[i * words + 2 * c + (lane >> 5)] >> (lane & 31)) & 1u);
      const uint64_t rmask = __ballot(rad && !loc);
      if (lane == 0) {
        radbits[i * words + 2 * c] = (uint32_t)rmask;
        radbits[i * words + 2 * c + 1] = (uint32_t)(rmask >> 32);
      }
    }
  }
  __syncthreads();
  // canonical radius edges per target: one of j -> i / i -> j when both are radius edges (the one with src < dst), and
  // every radius edge without such a mirror
  int wave_ctotal = 0;
  for (int i = wave; i < n; i += nwaves) {
    int cdeg = 0;
    for (int c = 0; 64 * c < n; ++c) {
      const int j = 64 * c + lane;
      const bool rad = j < n && bit(radbits, i, j);
      const bool canon = rad && (j < i || !bit(radbits, j, i));
      cdeg += __popcll(__ballot(canon));
    }
    if (lane == 0) scan_c[i] = cdeg;
    wave_ctotal += cdeg;
  }
  __syncthreads();
  // inclusive scan of scan_c (n <= 512) by ONE wave: eight consecutive counts per lane, a shuffle scan of the lanes' sums (the
  // Hillis-Steele scan over the whole workgroup took two barriers per doubling: 13 % of the graph phase)
  if (wave == 0) {
    int run[8], sum = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int idx = 8 * lane + t;
      sum += (idx < n) ? scan_c[idx] : 0;
      run[t] = sum;
    }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o);
      incl += (lane >= o) ? up : 0;
    }
    const int excl = incl - sum;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int idx = 8 * lane + t;
      if (idx < n) scan_c[idx] = run[t] + excl;
    }
  }
  __syncthreads();
  __shared__ int cbase_s;
  if (threadIdx.x == 0) {
    cbase_s = atomicAdd(&a.canon_counter[a.parity], n ? scan_c[n - 1] : 0);
    if (g == 0) a.canon_counter[a.parity ^ 1] = 0;       // (the previous step's head has long consumed it)
  }
  __syncthreads();
  // pass 2: emit, one wave per target, contiguous stores
  const int cbase = cbase_s;
  for (int i = wave; i < n; i += nwaves) {
    const float xi = spos[3 * i], yi = spos[3 * i + 1], zi = spos[3 * i + 2];
    const int row0 = (g0 + i) * AGDIFF_RAD_STRIDE;
    int rp0 = row0;
    int cp0 = cbase + (i ? scan_c[i - 1] : 0);
    for (int c = 0; 64 * c < n; ++c) {
      const int j = 64 * c + lane;
      const int jj = (j < n) ? j : 0;
      const uint64_t rmask = (uint64_t)radbits[i * words + 2 * c] | ((uint64_t)radbits[i * words + 2 * c + 1] << 32);
      const bool rad = (rmask >> lane) & 1ull;
      const bool mir = rad && bit(radbits, jj, i);
      const bool canon = rad && (j < i || !mir);
      const uint64_t cmask = __ballot(canon);
      if (rad) {
        const int rp = rp0 + __popcll(rmask & lt);
        const float len = fr_sqrt_rn(fr_dist2(xi, yi, zi, spos[3 * jj], spos[3 * jj + 1], spos[3 * jj + 2]));
        a.rad_src[rp] = g0 + j;
        a.rad_len[rp] = len;
        const float C = cf_envelope(len, a.cutoff, a.smooth);
        if (by_union) {
          const int u = union_segment(len);
          for (int cc = 0; cc < a.n_scales; ++cc) a.r_scale[(size_t)cc * a.rpad + rp] = scale_of(u, cc, len) * C;
        } else {
          for (int cc = 0; cc < a.n_scales; ++cc) a.r_scale[(size_t)cc * a.rpad + rp] = cf_dist_weight(sseg + cc * 100, len) * C;
        }
        if (canon) {
          const int cp = cp0 + __popcll(cmask & lt);
          a.c_len[cp] = len;
          a.c_src[cp] = g0 + j;
          a.c_dst[cp] = g0 + i;
          a.c_pos[cp] = rp;
          a.c_mir[cp] = mir ? (g0 + j) * AGDIFF_RAD_STRIDE + rank_below(radbits, j, i) : -1;
        }
      }
      rp0 += __popcll(rmask);
      cp0 += __popcll(cmask);
    }
    const int cnt = rp0 - row0;
    if (lane == 0) a.rad_cnt[g0 + i] = cnt;
    // pad rows (src = the target itself, length 0, scale 0) up to the end of the target's last 16-row tile: what k_cfconv_node
    // reads.  k_cfconv_quad runs the rows from a target's count on with scale 0 itself (what they hold is older, finite data of
    // the same molecule or the buffers' initial zeros), so on quads the 14 store instructions per target are not issued
    if (a.pad_rows && lane < ((cnt + AG_TW - 1) / AG_TW) * AG_TW - cnt) {
      const int rp = rp0 + lane;
      a.rad_src[rp] = g0 + i;
      a.rad_len[rp] = 0.0f;
      for (int cc = 0; cc < a.n_scales; ++cc) a.r_scale[(size_t)cc * a.rpad + rp] = 0.0f;
    }
  }
}

}  // namespace

static int sampler_front_impl(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, const agdiff_step_args_t* s,
                              const agdiff_step_args_t* step_table, const int32_t* step_index, int32_t mode, float cutoff, void* stream);

extern "C" int agdiff_sampler_front(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                                    const agdiff_step_args_t* s, int32_t mode, float cutoff, void* stream) {
  return sampler_front_impl(p, topo, ws, s, nullptr, nullptr, mode, cutoff, stream);
}

// (internal, csrc/common.hpp: the front launch with its step taken from a device table -- agdiff_step_graph_capture, api.hip)
int ag_sampler_front_table(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, const agdiff_step_args_t* s,
                           const agdiff_step_args_t* step_table, const int32_t* step_index, int32_t mode, float cutoff, void* stream) {
  return sampler_front_impl(p, topo, ws, s, step_table, step_index, mode, cutoff, stream);
}

static int sampler_front_impl(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, const agdiff_step_args_t* s,
                              const agdiff_step_args_t* step_table, const int32_t* step_index, int32_t mode, float cutoff, void* stream) {
  if (!p || !topo || !ws || !(mode & 7)) return AGDIFF_ERR_ARG;
  const bool do_update = (mode & 1) != 0, do_graph = (mode & 2) != 0, do_local = (mode & 4) != 0 && topo->num_local > 0;
  if (do_local && (!topo->lcm_ptr || !topo->lc_src || !topo->lc_dst || !topo->lc_pos || !topo->lc_mir || !ws->l_len || !ws->lc_len))
    return AGDIFF_ERR_ARG;
  if (do_update && (!s || !s->pos_in || !s->pos_out || !s->noise || !ws->nan_flag)) return AGDIFF_ERR_ARG;
  if (!do_update && (!s || !s->pos_in)) return AGDIFF_ERR_ARG;
  if (!ws->rad_cnt || !ws->rad_src || !ws->rad_len || !ws->r_scale || !ws->inv_r || !ws->canon_counter ||
      !ws->c_len || !ws->c_src || !ws->c_dst || !ws->c_pos || !ws->c_mir || p->num_convs > AGDIFF_MAX_CONVS)
    return AGDIFF_ERR_ARG;
  if (topo->num_graphs <= 0 || topo->num_nodes <= 0) return AGDIFF_OK;
  const int max_atoms = (int)topo->max_atoms_per_graph;
  if (max_atoms <= 0 || max_atoms > AGDIFF_MAX_ATOMS_PER_GRAPH) return AGDIFF_ERR_LIMIT;
  if (topo->num_nodes * (int64_t)AGDIFF_RAD_STRIDE >= (1ll << 31)) return AGDIFF_ERR_LIMIT;
  FrontArgs a;
  a.s = *s;
  a.step_table = step_table;
  a.step_index = step_index;
  a.do_update = do_update ? 1 : 0;
  a.do_graph = do_graph ? 1 : 0;
  a.graph_ptr = topo->graph_ptr;
  a.loc_src = topo->loc_src;
  a.loc_dst = topo->loc_dst;
  a.loc_out_ptr = topo->loc_out_ptr;
  a.loc_in_ptr = topo->loc_in_ptr;
  a.loc_in_eid = topo->loc_in_eid;
  a.l_len = ws->l_len;
  a.l_inv = ws->l_inv;
  a.rad_cnt = ws->rad_cnt;
  a.rad_src = ws->rad_src;
  a.rad_len = ws->rad_len;
  a.r_scale = ws->r_scale;
  a.rpad = topo->num_nodes * (int64_t)AGDIFF_RAD_STRIDE;
  a.pad_rows = !(topo->group_targets == 4 && p->tune_cfconv_quad_tiles >= 0);      // (agdiff_cfconv_node's choice of k_cfconv_quad)
  a.inv_r = ws->inv_r;
  for (int k = 0; k < p->num_convs; ++k) {
    a.dw[2 * k] = p->conv[k].dist_seg;
    a.dw[2 * k + 1] = p->conv[k].dist_seg + 100;
  }
  a.n_scales = 2 * p->num_convs;
  a.dist_union = p->dist_union;
  a.union_kinks = p->dist_union_kinks;
  a.union_segments = p->dist_union_segments;
  if (a.dist_union && (a.union_kinks < 2 || a.union_kinks > 512 || (a.union_kinks & (a.union_kinks - 1)) || a.union_segments < 1 ||
                       a.union_segments > a.union_kinks))
    return AGDIFF_ERR_ARG;
  a.cutoff = p->cutoff;
  a.r2 = cutoff * cutoff;
  a.smooth = p->smooth;
  a.canon_counter = ws->canon_counter;
  a.parity = (mode >> 4) & 1;
  a.c_len = ws->c_len;
  a.c_src = ws->c_src;
  a.c_dst = ws->c_dst;
  a.c_pos = ws->c_pos;
  a.c_mir = ws->c_mir;
  a.do_local = do_local ? 1 : 0;
  a.lcm_ptr = topo->lcm_ptr;
  a.lc_src = topo->lc_src;
  a.lc_dst = topo->lc_dst;
  a.lc_pos = topo->lc_pos;
  a.lc_mir = topo->lc_mir;
  const bool by_slot = ws->l_len_p && topo->lc_ppos && topo->lc_pmir;
  const bool by_tile = ws->lt_len && ws->lt_scale && topo->lc_tpos && topo->lc_tmir;
  a.lc_ppos = topo->lc_ppos;
  a.lc_pmir = topo->lc_pmir;
  a.lc_tpos = topo->lc_tpos;
  a.lc_tmir = topo->lc_tmir;
  a.l_len_w = ws->l_len;
  a.lc_len = ws->lc_len;
  a.l_len_p = by_slot ? ws->l_len_p : nullptr;
  a.lt_len = by_tile ? ws->lt_len : nullptr;
  a.lt_scale = ws->lt_scale;
  a.tpad = topo->num_local_tiles * (int64_t)AG_TW;
  a.nan_flag = ws->nan_flag;
  a.words = 2 * ((max_atoms + 63) / 64);
  a.loc_bits = reinterpret_cast<const uint32_t*>(topo->loc_bits);       // (rows of 2 ceil(max_atoms_per_graph / 64) words, as a.words)
  const int nmax = a.words * 32;
  // threads per molecule and lanes per atom: as k_graph / k_langevin_update (node.hip, graph.hip)
  const int bd = topo->num_graphs >= 512 ? 512 : 1024;
  int parts = 1;
  while (parts < 16 && 2 * parts * max_atoms <= bd) parts *= 2;
  a.parts = parts;
  const size_t smem = (size_t)(3 * nmax + 3 * nmax + nmax + 2 * nmax * a.words) * 4 +
                      (a.dist_union ? ((size_t)(a.union_kinks + a.union_segments * 2 * a.n_scales) * 4 + 15) / 16 * 16
                                    : (size_t)a.n_scales * 100 * 4);
  if (smem > 48 * 1024) {
    static std::atomic<uint64_t> attr_done{0};
    // (the kernel also has ~6 KiB of static LDS: the dynamic part may not claim all 160 KiB)
    if (smem > (size_t)152 * 1024) return AGDIFF_ERR_LIMIT;
    if (!ag_allow_big_lds(attr_done, (size_t)152 * 1024, k_sampler_front)) return AGDIFF_ERR_LAUNCH;
  }
  ag_log_variant(ws, AGDIFF_VAR_FUSED_FRONT);
  k_sampler_front<<<dim3((unsigned)topo->num_graphs), dim3(bd), smem, (hipStream_t)stream>>>(a);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}
