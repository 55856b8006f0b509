// chamfer.hip -- bidirectional nearest-neighbour squared distance (metrics/chamfer3D/chamfer3D.cu:12,155).
// xyz tensors are point-major [b,n,3]. One thread per query point; the target cloud is staged through
// LDS in 1024-point tiles and read back as wave-wide broadcasts (every lane reads the same address).
#include "common.h"

#define CH_TILE 1024
__global__ __launch_bounds__(256) void nm_distance_kernel(int n, int m, const float *__restrict__ xyz,
                                                          const float *__restrict__ xyz2, float *__restrict__ result,
                                                          int *__restrict__ result_i) {
  __shared__ float buf[CH_TILE * 3];
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool ok = j < n;
  const float *q = xyz + ((size_t)b * n + (ok ? j : 0)) * 3;
  const float x1 = q[0], y1 = q[1], z1 = q[2];
  const float *tg = xyz2 + (size_t)b * m * 3;
  float best = INFINITY;  // strict '<' from +inf == "first element, then strictly smaller": lowest index wins ties
  int best_i = 0;
  for (int k2 = 0; k2 < m; k2 += CH_TILE) {
    const int kn = min(CH_TILE, m - k2);
    __syncthreads();
    for (int e = threadIdx.x; e < kn * 3; e += 256) buf[e] = tg[(size_t)k2 * 3 + e];
    __syncthreads();
    for (int k = 0; k < kn; ++k) {
      const float d = sqdist3(buf[k * 3 + 0] - x1, buf[k * 3 + 1] - y1, buf[k * 3 + 2] - z1);
      if (d < best) {
        best = d;
        best_i = k2 + k;
      }
    }
  }
  if (ok) {
    result[(size_t)b * n + j] = best;
    result_i[(size_t)b * n + j] = best_i;
  }
}

extern "C" int p2pb_chamfer_forward(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1,
                                    float *dist2, int *idx1, int *idx2, void *stream) {
  if (b <= 0 || n <= 0 || m <= 0) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(nm_distance_kernel, dim3(cdiv(n, 256), b), dim3(256), 0, s, n, m, xyz1, xyz2, dist1, idx1);
  hipLaunchKernelGGL(nm_distance_kernel, dim3(cdiv(m, 256), b), dim3(256), 0, s, m, n, xyz2, xyz1, dist2, idx2);
  return p2pb_launch_status();
}

__global__ __launch_bounds__(256) void nm_distance_grad_kernel(int n, int m, const float *__restrict__ xyz1,
                                                               const float *__restrict__ xyz2,
                                                               const float *__restrict__ grad_dist1,
                                                               const int *__restrict__ idx1, float *grad_xyz1,
                                                               float *grad_xyz2) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const float *p = xyz1 + ((size_t)b * n + j) * 3;
  const int j2 = idx1[(size_t)b * n + j];
  const float *q = xyz2 + ((size_t)b * m + j2) * 3;
  const float g = grad_dist1[(size_t)b * n + j] * 2;
  float *g1 = grad_xyz1 + ((size_t)b * n + j) * 3;
  float *g2 = grad_xyz2 + ((size_t)b * m + j2) * 3;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float v = g * (p[a] - q[a]);
    atomicAdd(g1 + a, v);
    atomicAdd(g2 + a, -v);
  }
}

extern "C" int p2pb_chamfer_backward(int b, int n, int m, const float *xyz1, const float *xyz2, float *gradxyz1,
                                     float *gradxyz2, const float *graddist1, const float *graddist2,
                                     const int *idx1, const int *idx2, void *stream) {
  if (b <= 0 || n <= 0 || m <= 0) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(nm_distance_grad_kernel, dim3(cdiv(n, 256), b), dim3(256), 0, s, n, m, xyz1, xyz2, graddist1,
                     idx1, gradxyz1, gradxyz2);
  hipLaunchKernelGGL(nm_distance_grad_kernel, dim3(cdiv(m, 256), b), dim3(256), 0, s, m, n, xyz2, xyz1, graddist2,
                     idx2, gradxyz2, gradxyz1);
  return p2pb_launch_status();
}
