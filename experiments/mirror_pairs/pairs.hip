// CFConv with the filter network evaluated ONCE per mirror pair.
//
// encoder/schnet.py:136-162: W_e = nn(edge_attr_e) * (lw(d_e) * C(d_e)); agg[dst] += x[src] * W_e.  The two directed
// edges j -> i and i -> j of a mirror pair carry the same length and type, hence bit-identical edge_attr and filter; the
// canonical list of the graph build holds one of them (plus every unpaired edge).  Here that list is walked in PAIR-SWEEP
// order -- by (molecule, block of 16 sources, destination, source) -- so that
//   * the direct message  agg[dst] += W x[src]  is a destination-segmented sum inside a tile (whole segments per
//     tile: no carry between tiles), written to one row per (tile group) segment, and
//   * the mirror message  agg[src] += W x[dst]  goes to one of only 16 source slots for a whole sweep: it is
//     accumulated in registers by v_mfma_f32_16x16x4_f32 with a 0/1 selection matrix as the A operand
//     (D[slot][ch] += sum_row P[slot][row] * msg[row][ch]; products by 1.0 and 0.0 are exact, the order is fixed),
//     and flushed once per work item.
// No atomics: every output row has exactly one writer, so runs are bitwise reproducible.  The node side adds, per atom,
// its segment rows and its slot of the mirror rows of its block's items, in a fixed order.
// STATUS: experiment, not on the product path.  Measured on MI355X at the bench workload (tools/proto_run.py,
// profiles/r02_pairs_prototype.txt): aggregates equal the product kernel's (2e-7; the direct sums bit for bit), 64.9 k
// tiles instead of 100.4 k, but 0.56-0.57 ms per launch against 0.48 ms for k_cfconv_fused -- the product kernel's own
// code run on this row order (direct sums only) takes 0.375 ms, i.e. the order alone costs 21 % per tile, and the
// mirror path (48 f32 MFMAs, 48 more gathered x values and 48 live accumulators per tile) the rest.  DESIGN.md §8.2.
#include "common.hpp"
#include <type_traits>

namespace {

#define AG_PAIR_WAVES 8
// timing experiments (separate builds): -DPAIR_ABL=<bits>   1 no mirror MFMAs, 2 no x[dst] gathers, 4 no direct sums
// measured at the bench workload: full 0.578 ms, 1: 0.543, 2: 0.556, 3: 0.530, 7: 0.445 (profiles/r02_pairs_prototype.txt)
#ifndef PAIR_ABL
#define PAIR_ABL 0
#endif
#define AG_PABL(bit) ((PAIR_ABL) & (bit))

// ---------------------------------------------------------------------------------------------------------------
// Both convs in ONE launch.  Per tile the lighter conv2 (64 filter channels) runs first and conv1 (128) second, each as
// [first layer + softplus] -> [second layer per 16-channel tile + messages + sums]; only one conv's hidden activations
// are live at a time, which keeps the body (with all 48 mirror accumulators) inside 256 VGPRs = two waves per SIMD.
// LDS holds the fused first layer (96 KiB) and conv1's second layer (64 KiB); conv2's second layer (16 KiB per tile)
// streams from L2.
struct PairFusedArgs {
  agdiff_conv_params_t cp;
  const int32_t* p_src;
  const int32_t* p_dst;
  const int32_t* p_slot;
  const int32_t* p_seg;
  const int32_t* seg_ptr;
  const float* scale1;         // [R] conv1
  const float* scale2;         // [R] conv2
  const float* e_attr;
  const float* xs;
  float* agg_seg;
  float* mir_rows;
  const int32_t* item_row0;
  const int32_t* item_tiles;
  const int32_t* wave_item_ptr;
  int32_t num_waves;
};

template <int MODE>
__global__ void __launch_bounds__(64 * AG_PAIR_WAVES, 2) k_cfconv_pairs_fused(PairFusedArgs a) {
  extern __shared__ u32x4 ag_pair_smem[];
  lds_u32x4* w1 = (lds_u32x4*)ag_pair_smem;  // fused first layer, blocks [t][12]
  lds_u32x4* w2a = w1 + 48 * 128;            // conv1 second layer pk [8][4]
  {
    const u32x4* g1 = reinterpret_cast<const u32x4*>(a.cp.filt_w1_pk);
    const u32x4* ga = reinterpret_cast<const u32x4*>(a.cp.filt_w2a_pk);
    for (int i = threadIdx.x; i < 48 * 128; i += blockDim.x) w1[i] = g1[i];
    for (int i = threadIdx.x; i < 32 * 128; i += blockDim.x) w2a[i] = ga[i];
  }
  __syncthreads();
  const int lane0 = ag_lane();
  const int wave = threadIdx.x >> 6;
  const int wg = (gridDim.x % 8 == 0) ? (int)((blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8) : (int)blockIdx.x;
  const int w = wg * AG_PAIR_WAVES + wave;
  if (w >= a.num_waves) return;
  const int it_begin = a.wave_item_ptr[w], it_end = a.wave_item_ptr[w + 1];
  if (it_begin >= it_end) return;

  AgIn<MODE> ea[4];
  f32x4 nb0 = ag_ld4(a.cp.filt_b1 + 128 + 4 * (lane0 >> 4)), nb1 = ag_ld4(a.cp.filt_b1 + 128 + 16 + 4 * (lane0 >> 4));
  int pf_src = 0, pf_dst = 0, pf_slot = -1, pf_t0 = 0, pf_t1 = 0;
  float pf_s1 = 0.0f, pf_s2 = 0.0f;
  auto prefetch = [&](int64_t tl, int ln) {
#pragma unroll
    for (int t = 0; t < 4; ++t) ag_load_attr(ea[t], a.e_attr, tl, t, ln);
    const int64_t tb = tl * AG_TW, e = tb + (ln & 15);
    pf_src = a.p_src[e];
    pf_dst = a.p_dst[e];
    pf_slot = a.p_slot[e];
    pf_s1 = a.scale1[e];
    pf_s2 = a.scale2[e];
    pf_t0 = a.p_seg[tb];
    pf_t1 = a.p_seg[tb + AG_TW - 1];
  };
  prefetch(a.item_row0[it_begin] / AG_TW, lane0);

  for (int item = it_begin; item < it_end; ++item) {
    const int64_t tile0 = a.item_row0[item] / AG_TW;
    const int ntiles = a.item_tiles[item];
    f32x4 acc[12];          // mirror sums: acc[c][v] = slot 4 q + v, channel 16 c + col
#pragma unroll
    for (int c = 0; c < 12; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tt = 0; tt < ntiles; ++tt) {
      const int64_t tile = tile0 + tt;
      const int64_t tbase = tile * AG_TW;
      int lane = lane0;
      asm volatile("" : "+v"(lane));
      const int q = lane >> 4, col = lane & 15;
      const int my_src = pf_src, my_dst = pf_dst, my_slot = pf_slot;
      const float s1 = pf_s1, s2 = pf_s2;
      const int t0 = __builtin_amdgcn_readfirstlane(pf_t0);
      const int t1 = __builtin_amdgcn_readfirstlane(pf_t1);
      uint32_t xoff[4], xdoff[4];
      f32x4 selA;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xoff[r] = (uint32_t)__shfl(my_src, 4 * q + r) * 192u + (uint32_t)col;
        xdoff[r] = (uint32_t)__shfl(my_dst, 4 * q + r) * 192u + (uint32_t)col;
        selA[r] = (__shfl(my_slot, 4 * q + r) == col) ? 1.0f : 0.0f;
      }
      const int ntg = t1 - t0 + 1;
      const int ipl = a.seg_ptr[t0 + (lane <= ntg ? lane : ntg)];
      auto bound = [&](int i) -> int {
        return (i < 64) ? __builtin_amdgcn_readlane(ipl, i) : __builtin_amdgcn_readfirstlane(a.seg_ptr[t0 + i]);
      };
      float* const dp0 = a.agg_seg + (size_t)t0 * 192;
      const bool fast = ntg <= 2;
      f32x4 xg, xdg;
      auto fetch_x = [&](int c) {          // c = channel tile in the 192-wide rows
        const float* xb = a.xs + 16 * c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          xg[r] = xb[xoff[r]];
          xdg[r] = AG_PABL(2) ? xg[r] : xb[xdoff[r]];
        }
      };
      fetch_x(8);
      const lds_u32x4* w1_lo = w1 + lane;
      asm volatile("" : "+v"(w1_lo));
      const lds_u32x4* w1_hi = w1 + 32 * 128 + lane;
      asm volatile("" : "+v"(w1_hi));
      const lds_u32x4* w2a_l = w2a + lane;
      asm volatile("" : "+v"(w2a_l));
      f32x4 m0 = {0.f, 0.f, 0.f, 0.f}, m1 = {0.f, 0.f, 0.f, 0.f};
      bool masks_ready = false;

      // one conv: CONV 2 first (channel tiles 8..11, hidden k-tiles from first-layer tiles 8..11), then CONV 1
      auto phase = [&](auto CONVT, bool prefetch_next) {
        constexpr int CONV = decltype(CONVT)::value;
        constexpr int NCH = CONV == 1 ? 8 : 4;
        constexpr int NM = NCH / 2;
        constexpr int OT0 = CONV == 1 ? 0 : 8;
        f32x4 sr;
#pragma unroll
        for (int r = 0; r < 4; ++r) sr[r] = __shfl(CONV == 1 ? s1 : s2, 4 * q + r);
        AgIn<MODE> hidb[NM];
        {
          u32x4 wq[2][2][2];
          auto fetch_w = [&](u32x4 (&dst)[2][2], int step) {
            const int m = step >> 2, t = step & 3;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              const int bi = t * 12 + OT0 + 2 * m + b;
              const lds_u32x4* wb = (bi < 32) ? w1_lo + bi * 128 : w1_hi + (bi - 32) * 128;
              dst[b][0] = wb[0];
              dst[b][1] = wb[64];
            }
          };
          fetch_w(wq[0], 0);
          f32x4 hp0 = {0.f, 0.f, 0.f, 0.f}, hp1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int m = 0; m <= NM; ++m) {
            f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
            if (m < NM) {
              // bias requested a pair ahead (nb0 / nb1 persist across phases and tiles; order of pairs per tile:
              // conv2's two, then conv1's four, then the next tile's first)
              h0 = nb0;
              h1 = nb1;
              const int gm = (CONV == 2 ? m : 2 + m) + 1;              // global pair counter of the next pair (0..5)
              const int nm = gm % 6;                                    // 0,1: conv2 pairs (tiles 8..11); 2..5: conv1 pairs
              const int off = (nm < 2) ? 128 + 32 * nm : 32 * (nm - 2);
              nb0 = ag_ld4(a.cp.filt_b1 + off + 4 * q);
              nb1 = ag_ld4(a.cp.filt_b1 + off + 16 + 4 * q);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int step = m * 4 + t;
              if (m < NM) {
                if (step + 1 < 4 * NM) fetch_w(wq[(step + 1) & 1], step + 1);
                ag_block_mma<MODE, false>(h0, ea[t], wq[step & 1][0]);
                ag_block_mma<MODE, false>(h1, ea[t], wq[step & 1][1]);
              }
              if (m > 0) {
                float v0 = (t < 2) ? hp0[2 * t] : hp1[2 * t - 4], v1 = (t < 2) ? hp0[2 * t + 1] : hp1[2 * t - 3];
                v0 = ag_ssp_base2(v0);
                v1 = ag_ssp_base2(v1);
                asm volatile("" : "+v"(v0), "+v"(v1));
                ag_cvt_pair(hidb[m - 1], 2 * t, v0, v1);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            hp0 = h0; hp1 = h1;
          }
        }
        if (prefetch_next) {
          int64_t nxt = -1;
          if (tt + 1 < ntiles) nxt = tile + 1;
          else if (item + 1 < it_end) nxt = a.item_row0[item + 1] / AG_TW;
          if (nxt >= 0) prefetch(nxt, lane);
        }
        if (!masks_ready) {
          const int b0 = bound(0), b1 = bound(1), b2 = bound(ntg >= 2 ? 2 : 1);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int er = (int)tbase + 4 * q + r;
            m0[r] = (er >= b0 && er < b1) ? 1.0f : 0.0f;
            m1[r] = (ntg >= 2 && er >= b1 && er < b2) ? 1.0f : 0.0f;
          }
          masks_ready = true;
        }
        // conv2's second layer streams from L2 one channel tile ahead (pk [4][2])
        const u32x4* gl = reinterpret_cast<const u32x4*>(a.cp.filt_w2b_pk) + lane;
        u32x4 g[2][2];
        auto fetch_g = [&](int pair) {
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            g[b][0] = gl[((2 * pair + b) * 2) * 64];
            g[b][1] = gl[((2 * pair + b) * 2 + 1) * 64];
          }
        };
        if constexpr (CONV == 2) fetch_g(0);
        auto dense2 = [&](int nt) -> f32x4 {
          f32x4 z[1] = {{0.f, 0.f, 0.f, 0.f}};
          if constexpr (CONV == 1) {
            ag_dense_lds<MODE, true, false, 4, 1, 0, 0>(hidb, z, w2a_l + (nt * 4) * 128, 0);
          } else {
            ag_block_mma<MODE, true>(z[0], hidb[0], g[0]);
            ag_block_mma<MODE, true>(z[0], hidb[1], g[1]);
            if (nt + 1 < NCH) fetch_g(nt + 1);
          }
          return z[0];
        };
        constexpr int NEXT_FIRST = CONV == 2 ? 0 : -1;     // after conv2 comes conv1's channel tile 0
        if (fast) {
          f32x4 zp = {0.f, 0.f, 0.f, 0.f}, xp = {0.f, 0.f, 0.f, 0.f}, xdp = {0.f, 0.f, 0.f, 0.f};
          float bp = 0.0f;
          float p0[4], p1[4];
          const f32x4 w0 = m0 * sr, w1m = m1 * sr;
#pragma unroll
          for (int nt = 0; nt <= NCH; ++nt) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f}, xc = {0.f, 0.f, 0.f, 0.f}, xdc = {0.f, 0.f, 0.f, 0.f};
            float bb = 0.0f;
            if (nt < NCH) {
              z = dense2(nt);
              bb = a.cp.filt_b2[16 * (OT0 + nt) + col];
              xc = xg;
              xdc = xdg;
              if (nt + 1 < NCH) fetch_x(OT0 + nt + 1);
              else if (NEXT_FIRST >= 0) fetch_x(NEXT_FIRST);
            }
            if (nt > 0) {
              const int j = (nt - 1) & 3, g4 = (nt - 1) >> 2;
              f32x4 zb, t;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                zb[r] = zp[r] + bp;
                t[r] = zb[r] * xp[r];
              }
              p0[j] = t[0] * w0[0];
              p1[j] = t[0] * w1m[0];
#pragma unroll
              for (int r = 1; r < 4; ++r) {
                p0[j] = fmaf(t[r], w0[r], p0[j]);
                p1[j] = fmaf(t[r], w1m[r], p1[j]);
              }
              if (!AG_PABL(1)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const float u = zb[r] * (sr[r] * xdp[r]);
                  acc[OT0 + nt - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(selA[r], u, acc[OT0 + nt - 1], 0, 0, 0);
                }
              }
              if (j == 3 && !AG_PABL(4)) {
                const float r0 = ag_quarter_reduce_scatter4(p0[0], p0[1], p0[2], p0[3]);
                const float r1 = ag_quarter_reduce_scatter4(p1[0], p1[1], p1[2], p1[3]);
                dp0[16 * (OT0 + 4 * g4 + q) + col] = r0;
                if (ntg == 2) dp0[192 + 16 * (OT0 + 4 * g4 + q) + col] = r1;
              }
            }
            zp = z; xp = xc; xdp = xdc; bp = bb;
          }
        } else {
#pragma unroll
          for (int nt = 0; nt < NCH; ++nt) {
            f32x4 z = dense2(nt);
            const float bb = a.cp.filt_b2[16 * (OT0 + nt) + col];
            const f32x4 xc = xg, xdc = xdg;
            if (nt + 1 < NCH) fetch_x(OT0 + nt + 1);
            else if (NEXT_FIRST >= 0) fetch_x(NEXT_FIRST);
            f32x4 zb, t;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              zb[r] = z[r] + bb;
              t[r] = zb[r] * (sr[r] * xc[r]);
              const float u = zb[r] * (sr[r] * xdc[r]);
              acc[OT0 + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(selA[r], u, acc[OT0 + nt], 0, 0, 0);
            }
            const bool mine = q == (nt & 3);
            for (int i = 0; i < ntg; ++i) {
              const int lo = bound(i), hi = bound(i + 1);
              float p = 0.0f;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int er = (int)tbase + 4 * q + r;
                p += ((er >= lo) && (er < hi)) ? t[r] : 0.0f;
              }
              p = ag_quarter_sum(p);
              if (mine) dp0[(size_t)i * 192 + 16 * (OT0 + nt) + col] = p;
            }
          }
        }
      };
      phase(std::integral_constant<int, 2>{}, false);
      phase(std::integral_constant<int, 1>{}, true);
    }  // tiles
    {
      const int q = lane0 >> 4, col = lane0 & 15;
      float* mr = a.mir_rows + ((size_t)item * 16 + 4 * q) * 192 + col;
#pragma unroll
      for (int c = 0; c < 12; ++c)
#pragma unroll
        for (int v = 0; v < 4; ++v) mr[(size_t)v * 192 + 16 * c] = acc[c][v];
    }
  }  // items
}

}  // namespace

// Experimental entry point (not part of the public header): block k over a pair-sweep ordered list (tools/proto_pairs.py).
extern "C" int agdiff_proto_cfconv_pairs_fused(const agdiff_params_t* p, int32_t k, const int32_t* p_src, const int32_t* p_dst,
                                               const int32_t* p_slot, const int32_t* p_seg, const int32_t* seg_ptr,
                                               const float* scale1, const float* scale2, const float* e_attr,
                                               const float* xs, float* agg_seg, float* mir_rows, const int32_t* item_row0,
                                               const int32_t* item_tiles, const int32_t* wave_item_ptr, int32_t num_waves,
                                               void* stream) {
  if (!p || k < 0 || k >= p->num_convs || num_waves <= 0) return AGDIFF_ERR_ARG;
  PairFusedArgs a{p->conv[k], p_src, p_dst, p_slot, p_seg, seg_ptr, scale1, scale2, e_attr, xs, agg_seg, mir_rows,
                  item_row0, item_tiles, wave_item_ptr, num_waves};
  const int wgs = (num_waves + AG_PAIR_WAVES - 1) / AG_PAIR_WAVES;
  const size_t smem = (size_t)80 * 2048;
  static std::atomic<uint64_t> attr_done{0};
  if (!ag_allow_big_lds(attr_done, smem, k_cfconv_pairs_fused<AG_BF3>, k_cfconv_pairs_fused<AG_F32>)) return AGDIFF_ERR_LAUNCH;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)wgs), block(64 * AG_PAIR_WAVES);
  if (p->precision == AG_BF3) k_cfconv_pairs_fused<AG_BF3><<<grid, block, smem, st>>>(a);
  else k_cfconv_pairs_fused<AG_F32><<<grid, block, smem, st>>>(a);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}
