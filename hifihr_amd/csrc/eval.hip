// Evaluation path: Procrustes-with-scale alignment of a predicted point set to its ground truth and
// the aligned per-point error, batched -- one workgroup per sample, no host round trip.
//
// Replaces the per-sample numpy loop of reference train_hrnet.py:227-243 around utils/train_utils.py:267-290 (align_w_scale:
// centre, Frobenius-normalise, scipy.linalg.orthogonal_procrustes, apply).  With A = (gt - mean)/s1 and B = (pred - mean)/s2
// (s = Frobenius norm + 1e-8), scipy takes  u w v^T = svd(A^T B),  R = u v^T,  scale = sum(w)  -- no determinant
// correction, a reflection is allowed -- and the aligned prediction is  (B R^T) scale s1 + mean(gt).
// R is the orthogonal polar factor of M = A^T B:  M = R P,  P = (M^T M)^(1/2) = V diag(w) V^T,  so
//   R = M V diag(1/w) V^T   and   scale = trace(P) = sum(w),
// with V, w^2 from a Jacobi eigen-decomposition of the symmetric 3x3 M^T M.  All of it in fp64 (numpy does the same on
// fp64 arrays); the points are read as fp32, 12 bytes per point per set: the kernel is HBM / latency bound and tiny.
#include <hip/hip_runtime.h>

#include "hifihr_internal.h"

namespace hifihr {

namespace {

constexpr int kThreads = 256;

// sum of `v` over the workgroup, result to every thread
__device__ double block_sum(double v, double* red) {
  __syncthreads();
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  return red[0];
}

// eigen-decomposition of a symmetric 3x3 (cyclic Jacobi): S = V diag(l) V^T
__device__ void jacobi3(double S[3][3], double V[3][3], double l[3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 32; ++sweep) {
    const double off = S[0][1] * S[0][1] + S[0][2] * S[0][2] + S[1][2] * S[1][2];
    const double diag = S[0][0] * S[0][0] + S[1][1] * S[1][1] + S[2][2] * S[2][2];
    if (off <= 1e-60 + 1e-34 * diag) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (S[p][q] == 0.0) continue;
        const double theta = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {              // S <- S J (columns p, q)
          const double a = S[k][p], b = S[k][q];
          S[k][p] = c * a - s * b; S[k][q] = s * a + c * b;
        }
        for (int k = 0; k < 3; ++k) {              // S <- J^T S (rows p, q)
          const double a = S[p][k], b = S[q][k];
          S[p][k] = c * a - s * b; S[q][k] = s * a + c * b;
        }
        for (int k = 0; k < 3; ++k) {              // V <- V J
          const double a = V[k][p], b = V[k][q];
          V[k][p] = c * a - s * b; V[k][q] = s * a + c * b;
        }
      }
  }
  l[0] = S[0][0]; l[1] = S[1][1]; l[2] = S[2][2];
}

}  // namespace

__global__ __launch_bounds__(kThreads) void procrustes_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int N,
                                                             float* __restrict__ aligned, float* __restrict__ err_sum) {
  __shared__ double red[kThreads];
  __shared__ double sR[9], sScale;
  const int b = blockIdx.x;
  const float* P = pred + (size_t)b * N * 3;
  const float* G = gt + (size_t)b * N * 3;
  // pass 1: means
  double a[6] = {0, 0, 0, 0, 0, 0};
  for (int n = threadIdx.x; n < N; n += kThreads)
    for (int k = 0; k < 3; ++k) { a[k] += (double)G[n * 3 + k]; a[3 + k] += (double)P[n * 3 + k]; }
  double t1[3], t2[3];
  for (int k = 0; k < 3; ++k) { t1[k] = block_sum(a[k], red) / N; t2[k] = block_sum(a[3 + k], red) / N; }
  // pass 2: Frobenius norms and the cross-covariance of the centred sets
  double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, n1 = 0.0, n2 = 0.0;
  for (int n = threadIdx.x; n < N; n += kThreads) {
    double g[3], p[3];
    for (int k = 0; k < 3; ++k) { g[k] = (double)G[n * 3 + k] - t1[k]; p[k] = (double)P[n * 3 + k] - t2[k]; }
    for (int i = 0; i < 3; ++i) {
      n1 += g[i] * g[i]; n2 += p[i] * p[i];
      for (int j = 0; j < 3; ++j) m[i * 3 + j] += g[i] * p[j];
    }
  }
  n1 = block_sum(n1, red); n2 = block_sum(n2, red);
  for (int k = 0; k < 9; ++k) m[k] = block_sum(m[k], red);
  const double s1 = sqrt(n1) + 1e-8, s2 = sqrt(n2) + 1e-8;
  if (threadIdx.x == 0) {
    double M[3][3], S[3][3], V[3][3], l[3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) M[i][j] = m[i * 3 + j] / (s1 * s2);          // A^T B
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) S[i][j] = M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j];   // M^T M
    jacobi3(S, V, l);
    double w[3], scale = 0.0;
    for (int k = 0; k < 3; ++k) { w[k] = sqrt(l[k] > 0.0 ? l[k] : 0.0); scale += w[k]; }
    // R = M V diag(1/w) V^T; a vanishing singular value (coplanar input) leaves that direction out instead of dividing by 0
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double r = 0.0;
        for (int k = 0; k < 3; ++k) {
          if (w[k] <= 1e-150) continue;
          const double mv = M[i][0] * V[0][k] + M[i][1] * V[1][k] + M[i][2] * V[2][k];
          r += mv / w[k] * V[j][k];
        }
        sR[i * 3 + j] = r;
      }
    sScale = scale;
  }
  __syncthreads();
  // pass 3: aligned_n = R ((pred_n - t2) / s2) * scale * s1 + t1, error against gt_n
  const double f = sScale * s1 / s2;
  double e = 0.0;
  for (int n = threadIdx.x; n < N; n += kThreads) {
    double p[3], d2 = 0.0;
    for (int k = 0; k < 3; ++k) p[k] = (double)P[n * 3 + k] - t2[k];
    for (int i = 0; i < 3; ++i) {
      const double v = (sR[i * 3] * p[0] + sR[i * 3 + 1] * p[1] + sR[i * 3 + 2] * p[2]) * f + t1[i];
      if (aligned != nullptr) aligned[((size_t)b * N + n) * 3 + i] = (float)v;
      const double d = v - (double)G[n * 3 + i];
      d2 += d * d;
    }
    e += sqrt(d2);
  }
  e = block_sum(e, red);
  if (threadIdx.x == 0) err_sum[b] = (float)e;
}

hipError_t launch_procrustes(const float* pred, const float* gt, int B, int N, float* aligned, float* err_sum, hipStream_t st) {
  if (B <= 0 || N <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(procrustes_kernel, dim3(B), dim3(kThreads), 0, st, pred, gt, N, aligned, err_sum);
  return hipGetLastError();
}

}  // namespace hifihr
