// Evaluation kernels: the RMSD confusion matrix of get_rmsd_confusion_matrix
// (utils/evaluation/covmat.py:16-35) and its row / column minima (covmat.py:135-136).
//
// rdkit's GetBestRMS (third party, not vendored by the reference) = min over the molecule's self-matches of the RMSD
// after AlignMol's optimal proper rotation + translation.  For one mapping, with both conformers centred,
//   RMSD^2 = (|X|^2 + |Y|^2 - 2 lambda_max(K)) / m,
// K = Horn's 4x4 symmetric key matrix of the 3x3 cross-covariance S = sum_k x_k y_k^T (largest eigenvalue = the best
// proper rotation as a unit quaternion; reflections are excluded by construction).  lambda_max comes from cyclic
// Jacobi sweeps in fp64: a few hundred flops per pair, robust for planar / collinear / identical conformers.
// Latency-bound, tiny next to the sampler: one thread per (reference, generated) pair, conformer tiles staged in LDS.
#include "common.hpp"

namespace {

// centred coordinates of the selected atoms, one wave per conformer (centroid in fp64, rounded once):
// out[c] = { x_0 y_0 z_0 ... x_{m-1} y_{m-1} z_{m-1} | unused }
__global__ void __launch_bounds__(64) k_center_selected(const float* __restrict__ pos, const int32_t* __restrict__ idx,
                                                        int n, int m, float* __restrict__ out) {
  const int c = blockIdx.x, lane = threadIdx.x;
  const float* p = pos + (size_t)c * n * 3;
  double sx = 0.0, sy = 0.0, sz = 0.0;
  for (int k = lane; k < m; k += 64) {
    const int a = idx[k];
    sx += p[3 * a]; sy += p[3 * a + 1]; sz += p[3 * a + 2];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o); }
  const double cx = sx / m, cy = sy / m, cz = sz / m;
  float* o = out + (size_t)c * (3 * m + 1);
  for (int k = lane; k < m; k += 64) {
    const int a = idx[k];
    o[3 * k] = (float)(p[3 * a] - cx); o[3 * k + 1] = (float)(p[3 * a + 1] - cy); o[3 * k + 2] = (float)(p[3 * a + 2] - cz);
  }
  if (lane == 0) o[3 * m] = 0.0f;
}

// largest eigenvalue of the symmetric 4x4 matrix with upper triangle k[0..9] = (00 01 02 03 11 12 13 22 23 33)
__device__ double ag_lambda_max4(const double (&k)[10]) {
  double A[4][4] = {{k[0], k[1], k[2], k[3]}, {k[1], k[4], k[5], k[6]}, {k[2], k[5], k[7], k[8]}, {k[3], k[6], k[8], k[9]}};
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
    if (off < 1e-30) break;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        const double apq = A[p][q];
        if (apq == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {           // columns p, q
          const double arp = A[r][p], arq = A[r][q];
          A[r][p] = c * arp - s * arq;
          A[r][q] = s * arp + c * arq;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {           // rows p, q
          const double apr = A[p][r], aqr = A[q][r];
          A[p][r] = c * apr - s * aqr;
          A[q][r] = s * apr + c * aqr;
        }
      }
    }
  }
  return fmax(fmax(A[0][0], A[1][1]), fmax(A[2][2], A[3][3]));
}

extern __shared__ float ag_eval_smem[];

// 16 x 16 pairs per workgroup: thread (ty, tx) = (reference ty, generated tx) of the tile
__global__ void __launch_bounds__(256) k_rmsd_matrix(const float* __restrict__ cref, const float* __restrict__ cgen,
                                                     const int32_t* __restrict__ perms, int R, int G, int m, int P,
                                                     float* __restrict__ out) {
  const int stride = 3 * m + 1;
  float* sref = ag_eval_smem;
  float* sgen = ag_eval_smem + 16 * stride;
  const int j0 = blockIdx.y * 16, i0 = blockIdx.x * 16;
  for (int t = threadIdx.x; t < 16 * stride; t += 256) {
    const int c = t / stride, o = t % stride;
    sref[t] = (j0 + c < R) ? cref[(size_t)(j0 + c) * stride + o] : 0.0f;
    sgen[t] = (i0 + c < G) ? cgen[(size_t)(i0 + c) * stride + o] : 0.0f;
  }
  __syncthreads();
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  if (j0 + ty >= R || i0 + tx >= G) return;
  const float* y = sref + ty * stride;      // reference
  const float* x = sgen + tx * stride;      // generated (probe)
  // squared norms from the SAME rounded coordinates the cross-covariance uses, in fp64: near RMSD = 0 the difference
  // |X|^2 + |Y|^2 - 2 lambda cancels to ~1e-16 relative only if both sides see identical inputs
  double gsum = 0.0;
  for (int k = 0; k < m; ++k) {
    const double x0 = x[3 * k], x1 = x[3 * k + 1], x2 = x[3 * k + 2];
    const double y0 = y[3 * k], y1 = y[3 * k + 1], y2 = y[3 * k + 2];
    gsum += (x0 * x0 + x1 * x1 + x2 * x2) + (y0 * y0 + y1 * y1 + y2 * y2);
  }
  double best = 1e300;
  for (int p = 0; p < (perms ? P : 1); ++p) {
    const int32_t* pm = perms ? perms + (size_t)p * m : nullptr;
    double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < m; ++k) {
      const int kr = pm ? pm[k] : k;
      const double x0 = x[3 * k], x1 = x[3 * k + 1], x2 = x[3 * k + 2];
      const double y0 = y[3 * kr], y1 = y[3 * kr + 1], y2 = y[3 * kr + 2];
      S[0] += x0 * y0; S[1] += x0 * y1; S[2] += x0 * y2;
      S[3] += x1 * y0; S[4] += x1 * y1; S[5] += x1 * y2;
      S[6] += x2 * y0; S[7] += x2 * y1; S[8] += x2 * y2;
    }
    const double Sxx = S[0], Sxy = S[1], Sxz = S[2], Syx = S[3], Syy = S[4], Syz = S[5], Szx = S[6], Szy = S[7], Szz = S[8];
    const double K[10] = {Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx,
                          Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz,
                          -Sxx + Syy - Szz, Syz + Szy,
                          -Sxx - Syy + Szz};
    const double msd = (gsum - 2.0 * ag_lambda_max4(K)) / m;
    best = fmin(best, msd);
  }
  out[(size_t)(j0 + ty) * G + i0 + tx] = (float)sqrt(fmax(best, 0.0));
}

// one wave per row (blockIdx.y == 0) or per column (== 1)
__global__ void __launch_bounds__(64) k_matrix_minima(const float* __restrict__ mat, int R, int G, float* __restrict__ row_min,
                                                      float* __restrict__ col_min) {
  const int lane = threadIdx.x, i = blockIdx.x;
  float v = INFINITY;
  if (blockIdx.y == 0) {
    if (i >= R) return;
    for (int c = lane; c < G; c += 64) v = fminf(v, mat[(size_t)i * G + c]);
  } else {
    if (i >= G) return;
    for (int r = lane; r < R; r += 64) v = fminf(v, mat[(size_t)r * G + i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  if (lane == 0) (blockIdx.y == 0 ? row_min : col_min)[i] = v;
}

}  // namespace

extern "C" int agdiff_rmsd_matrix(const float* pos_ref, const float* pos_gen, const int32_t* atom_idx, const int32_t* perms,
                                  int32_t R, int32_t G, int32_t n, int32_t m, int32_t P, float* scratch, float* out,
                                  void* stream) {
  if (!pos_ref || !pos_gen || !atom_idx || !scratch || !out || R < 0 || G < 0 || n <= 0 || m <= 0 || m > n || (perms && P <= 0))
    return AGDIFF_ERR_ARG;
  if (m > AGDIFF_RMSD_MAX_ATOMS) return AGDIFF_ERR_LIMIT;
  if (R == 0 || G == 0) return AGDIFF_OK;
  hipStream_t st = (hipStream_t)stream;
  float* cref = scratch;
  float* cgen = scratch + (size_t)R * (3 * m + 1);
  k_center_selected<<<dim3((unsigned)R), dim3(64), 0, st>>>(pos_ref, atom_idx, n, m, cref);
  AG_CHECK_LAUNCH();
  k_center_selected<<<dim3((unsigned)G), dim3(64), 0, st>>>(pos_gen, atom_idx, n, m, cgen);
  AG_CHECK_LAUNCH();
  const size_t smem = (size_t)2 * 16 * (3 * m + 1) * sizeof(float);
  static std::atomic<uint64_t> attr_done{0};
  if (smem > 48 * 1024 && !ag_allow_big_lds(attr_done, (size_t)2 * 16 * (3 * AGDIFF_RMSD_MAX_ATOMS + 1) * sizeof(float), k_rmsd_matrix))
    return AGDIFF_ERR_LAUNCH;
  k_rmsd_matrix<<<dim3((unsigned)((G + 15) / 16), (unsigned)((R + 15) / 16)), dim3(256), smem, st>>>(cref, cgen, perms, R, G, m, P, out);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}

extern "C" int agdiff_matrix_minima(const float* mat, int32_t R, int32_t G, float* row_min, float* col_min, void* stream) {
  if (!mat || !row_min || !col_min || R <= 0 || G <= 0) return AGDIFF_ERR_ARG;
  const int mx = R > G ? R : G;
  k_matrix_minima<<<dim3((unsigned)mx, 2), dim3(64), 0, (hipStream_t)stream>>>(mat, R, G, row_min, col_min);
  AG_CHECK_LAUNCH();
  return AGDIFF_OK;
}
