// chamfer.hip -- bidirectional nearest-neighbour squared distance (metrics/chamfer3D/chamfer3D.cu:12,155).
// xyz tensors are point-major [b,n,3]. The kernel is VALU-bound by construction (8 instructions per point pair
// against 12 bytes of LDS broadcast traffic: the one-query-per-thread form measured 4.4 T pairs/s at B = 32,
// N = M = 8192 = the 39.3 T lane-op/s VALU rate), so the levers are instructions per pair and occupancy:
//   * TWO queries per thread held as a packed pair: the three differences, the square and the two fmas are v_pk_*
//     instructions on (query A, query B) -- 6 packed + 2 x 3 compare / select = 12 instructions per 2 pairs instead of
//     16 -- and every LDS read (one 16-byte (x, y, z, pad) record per target) serves both queries;
//   * a small batch leaves the chip empty (B = 4: 64 workgroups): the target cloud is then split over blockIdx.y and
//     the partial results meet in a 64-bit atomicMin on (distance bits, index) -- distances are non-negative, so their
//     bit patterns order like the values, and the low word makes the LOWEST index win a tie, which is exactly the
//     reference's "first strict minimum" (chamfer3D.cu:60-100). Same bits as the oracle in both forms.
#include "common.h"

#define CH_TILE 512
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64c;

template <bool SPLIT>
__global__ __launch_bounds__(256) void nm_distance_kernel(int n, int m, int mchunk, const float *__restrict__ xyz,
                                                          const float *__restrict__ xyz2, float *__restrict__ result,
                                                          int *__restrict__ result_i, u64c *__restrict__ keys) {
  __shared__ float4 buf[CH_TILE];
  const int b = blockIdx.z;
  const int j0 = blockIdx.x * 512 + threadIdx.x, j1 = j0 + 256;
  const bool ok0 = j0 < n, ok1 = j1 < n;
  const float *q0 = xyz + ((size_t)b * n + (ok0 ? j0 : 0)) * 3, *q1 = xyz + ((size_t)b * n + (ok1 ? j1 : 0)) * 3;
  const f2 x1 = {q0[0], q1[0]}, y1 = {q0[1], q1[1]}, z1 = {q0[2], q1[2]};
  const float *tg = xyz2 + (size_t)b * m * 3;
  float best0 = INFINITY, best1 = INFINITY;  // strict '<' from +inf: the lowest index wins ties
  int bi0 = 0, bi1 = 0;
  const int k_lo = blockIdx.y * mchunk, k_hi = min(m, k_lo + mchunk);
  for (int k2 = k_lo; k2 < k_hi; k2 += CH_TILE) {
    const int kn = min(CH_TILE, k_hi - k2);
    __syncthreads();
    for (int e = threadIdx.x; e < kn; e += 256) {
      const float *t = tg + (size_t)(k2 + e) * 3;
      buf[e] = make_float4(t[0], t[1], t[2], 0.0f);
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < kn; ++k) {
      const float4 t = buf[k];
      const f2 dx = f2{t.x, t.x} - x1, dy = f2{t.y, t.y} - y1, dz = f2{t.z, t.z} - z1;
      // fma(dz, dz, fma(dy, dy, dx * dx)) per half: the arithmetic contract of sqdist3, two queries per instruction
      const f2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
      if (d[0] < best0) {
        best0 = d[0];
        bi0 = k2 + k;
      }
      if (d[1] < best1) {
        best1 = d[1];
        bi1 = k2 + k;
      }
    }
  }
  if (SPLIT) {
    if (ok0) atomicMin(keys + (size_t)b * n + j0, ((u64c)__float_as_uint(best0) << 32) | (unsigned)bi0);
    if (ok1) atomicMin(keys + (size_t)b * n + j1, ((u64c)__float_as_uint(best1) << 32) | (unsigned)bi1);
  } else {
    if (ok0) {
      result[(size_t)b * n + j0] = best0;
      result_i[(size_t)b * n + j0] = bi0;
    }
    if (ok1) {
      result[(size_t)b * n + j1] = best1;
      result_i[(size_t)b * n + j1] = bi1;
    }
  }
}

__global__ __launch_bounds__(256) void nm_keys_init_kernel(size_t total, u64c *__restrict__ keys) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < total) keys[i] = ~(u64c)0;
}

__global__ __launch_bounds__(256) void nm_keys_unpack_kernel(size_t total, const u64c *__restrict__ keys,
                                                             float *__restrict__ result, int *__restrict__ result_i) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  result[i] = __uint_as_float((unsigned)(keys[i] >> 32));
  result_i[i] = (int)(unsigned)keys[i];
}

// one direction: for every point of xyz its nearest point of xyz2. The target cloud is split over `chunks` workgroup
// rows when the query side alone cannot fill the chip; the 64-bit keys then live in the caller-provided dist / idx
// arrays' place: keys_ws (b*n u64) is scratch carved from ws
static void nm_direction(int b, int n, int m, const float *xyz, const float *xyz2, float *dist, int *idx, u64c *keys_ws,
                         hipStream_t s) {
  const int qblocks = (n + 511) / 512;
  int chunks = 1;
  if (keys_ws) {
    while ((long)qblocks * b * chunks < 1024 && (m + chunks - 1) / chunks > 2 * CH_TILE) chunks *= 2;
  }
  if (chunks == 1) {
    hipLaunchKernelGGL(nm_distance_kernel<false>, dim3(qblocks, 1, b), dim3(256), 0, s, n, m, m, xyz, xyz2, dist, idx,
                       (u64c *)nullptr);
    return;
  }
  const size_t total = (size_t)b * n;
  const int mchunk = ((m + chunks - 1) / chunks + CH_TILE - 1) / CH_TILE * CH_TILE;
  hipLaunchKernelGGL(nm_keys_init_kernel, dim3(cdiv((long)total, 256)), dim3(256), 0, s, total, keys_ws);
  hipLaunchKernelGGL(nm_distance_kernel<true>, dim3(qblocks, (m + mchunk - 1) / mchunk, b), dim3(256), 0, s, n, m, mchunk,
                     xyz, xyz2, dist, idx, keys_ws);
  hipLaunchKernelGGL(nm_keys_unpack_kernel, dim3(cdiv((long)total, 256)), dim3(256), 0, s, total, keys_ws, dist, idx);
}

extern "C" size_t p2pb_chamfer_ws_bytes(int b, int n, int m) { return (size_t)b * (size_t)(n > m ? n : m) * sizeof(u64c); }

// ws: p2pb_chamfer_ws_bytes(b, n, m) bytes of scratch, or NULL (then a small batch simply uses fewer workgroups)
extern "C" int p2pb_chamfer_forward_ws(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1,
                                       float *dist2, int *idx1, int *idx2, void *ws, void *stream) {
  if (b <= 0 || n <= 0 || m <= 0) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  nm_direction(b, n, m, xyz1, xyz2, dist1, idx1, (u64c *)ws, s);
  nm_direction(b, m, n, xyz2, xyz1, dist2, idx2, (u64c *)ws, s);
  return p2pb_launch_status();
}

extern "C" int p2pb_chamfer_forward(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1,
                                    float *dist2, int *idx1, int *idx2, void *stream) {
  return p2pb_chamfer_forward_ws(b, n, m, xyz1, xyz2, dist1, dist2, idx1, idx2, nullptr, stream);
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
