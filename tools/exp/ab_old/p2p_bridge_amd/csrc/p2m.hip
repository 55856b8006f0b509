// p2m.hip -- point <-> triangle-mesh squared distances (evaluation metric P2M).
//
// Replaces pytorch3d._C.point_face_dist_forward / face_point_dist_forward as called by metrics/p2m.py:66,131
// (point_mesh_face_distance_custom :307-375) and models/evaluation.py:329-353. pytorch3d is a pip dependency of
// the reference (not vendored): the distance below restates its published geometry (pytorch3d/csrc/utils/
// geometry_utils.cuh: PointTriangle3DistanceForward, IsInsideTriangle, BarycentricCoords3Forward,
// PointLine3DistanceForward, kEpsilon = 1e-8):
//   unit normal n = (v2-v0) x (v1-v0) / (|.| + eps); t = (v0 - p).n; p0 = p + t n; if the triangle's area is at
//   least min_triangle_area and p0's barycentric coordinates all lie in [0,1] the distance is t^2, else the minimum
//   of the three point-segment distances (a segment shorter than eps counts as its end point).
// One object at a time (the reference's metric is unbatched: "Batch is not supported", metrics/metrics.py:211).
//   point_face : one thread per point, triangles streamed through LDS in tiles of 256 (9 floats each, read as
//                broadcasts), running (min, first index);
//   face_point : one thread per triangle (its 9 floats in registers), points streamed through LDS.
// Bound: VALU (~60 FLOP per pair); 50 k points x 20 k faces = 1e9 pairs.
#include "common.h"

#define P2M_EPS 1e-8f
#define P2M_TILE 256

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

__device__ __forceinline__ float point_segment_d2(V3 p, V3 v0, V3 v1) {
  const V3 d = sub3(v1, v0);
  const float l2 = dot3(d, d);
  if (l2 <= P2M_EPS) {
    const V3 q = sub3(p, v1);
    return dot3(q, q);
  }
  float t = dot3(d, sub3(p, v0)) / l2;
  t = fminf(fmaxf(t, 0.0f), 1.0f);
  const V3 q = {p.x - (v0.x + t * d.x), p.y - (v0.y + t * d.y), p.z - (v0.z + t * d.z)};
  return dot3(q, q);
}

__device__ __forceinline__ float point_triangle_d2(V3 p, V3 v0, V3 v1, V3 v2, float min_area) {
  V3 n = cross3(sub3(v2, v0), sub3(v1, v0));
  const float nn = sqrtf(dot3(n, n));
  const float inv = 1.0f / (nn + P2M_EPS);
  n = {n.x * inv, n.y * inv, n.z * inv};
  const float t = dot3(sub3(v0, p), n);
  const V3 p0 = {p.x + t * n.x, p.y + t * n.y, p.z + t * n.z};
  bool inside = false;
  if (0.5f * nn >= min_area) {  // AreaOfTriangle = |cross| / 2
    const V3 e0 = sub3(v1, v0), e1 = sub3(v2, v0), e2 = sub3(p0, v0);
    const float d00 = dot3(e0, e0), d01 = dot3(e0, e1), d11 = dot3(e1, e1), d20 = dot3(e2, e0), d21 = dot3(e2, e1);
    const float denom = d00 * d11 - d01 * d01 + P2M_EPS;
    const float w1 = (d11 * d20 - d01 * d21) / denom, w2 = (d00 * d21 - d01 * d20) / denom;
    const float w0 = 1.0f - w1 - w2;
    inside = (0.0f <= w0 && w0 <= 1.0f) && (0.0f <= w1 && w1 <= 1.0f) && (0.0f <= w2 && w2 <= 1.0f);
  }
  if (inside) return t * t;
  return fminf(fminf(point_segment_d2(p, v0, v1), point_segment_d2(p, v0, v2)), point_segment_d2(p, v1, v2));
}

__global__ __launch_bounds__(256) void point_face_kernel(int np, int nt, const float *__restrict__ pts,
                                                         const float *__restrict__ tris, float min_area,
                                                         float *__restrict__ dist, int *__restrict__ idx) {
  __shared__ float tile[P2M_TILE * 9];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool ok = i < np;
  const V3 p = ok ? V3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]} : V3{0.0f, 0.0f, 0.0f};
  float best = 3.4e38f;
  int bi = 0;
  for (int t0 = 0; t0 < nt; t0 += P2M_TILE) {
    const int cnt = min(P2M_TILE, nt - t0);
    __syncthreads();
    for (int e = threadIdx.x; e < cnt * 9; e += 256) tile[e] = tris[(size_t)t0 * 9 + e];
    __syncthreads();
    for (int k = 0; k < cnt; ++k) {
      const float *q = tile + 9 * k;
      const float d = point_triangle_d2(p, V3{q[0], q[1], q[2]}, V3{q[3], q[4], q[5]}, V3{q[6], q[7], q[8]}, min_area);
      if (d < best) {  // first minimum in index order
        best = d;
        bi = t0 + k;
      }
    }
  }
  if (ok) {
    dist[i] = best;
    idx[i] = bi;
  }
}

__global__ __launch_bounds__(256) void face_point_kernel(int np, int nt, const float *__restrict__ pts,
                                                         const float *__restrict__ tris, float min_area,
                                                         float *__restrict__ dist, int *__restrict__ idx) {
  __shared__ float tile[P2M_TILE * 3];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool ok = i < nt;
  const float *q = tris + (size_t)(ok ? i : 0) * 9;
  const V3 v0 = {q[0], q[1], q[2]}, v1 = {q[3], q[4], q[5]}, v2 = {q[6], q[7], q[8]};
  float best = 3.4e38f;
  int bi = 0;
  for (int p0 = 0; p0 < np; p0 += P2M_TILE) {
    const int cnt = min(P2M_TILE, np - p0);
    __syncthreads();
    for (int e = threadIdx.x; e < cnt * 3; e += 256) tile[e] = pts[(size_t)p0 * 3 + e];
    __syncthreads();
    for (int k = 0; k < cnt; ++k) {
      const float d = point_triangle_d2(V3{tile[3 * k], tile[3 * k + 1], tile[3 * k + 2]}, v0, v1, v2, min_area);
      if (d < best) {
        best = d;
        bi = p0 + k;
      }
    }
  }
  if (ok) {
    dist[i] = best;
    idx[i] = bi;
  }
}

// points f32[np,3], tris f32[nt,3,3] (the mesh's faces as vertex triples) -> dist f32[np] squared distance of every
// point to its closest triangle, idx i32[np] that triangle (first minimum)
extern "C" int p2pb_point_face_dist(int np, int nt, const float *points, const float *tris, float min_triangle_area,
                                    float *dist, int *idx, void *stream) {
  if (np <= 0 || nt <= 0) return P2PB_EINVAL;
  hipLaunchKernelGGL(point_face_kernel, dim3(cdiv(np, 256)), dim3(256), 0, (hipStream_t)stream, np, nt, points, tris,
                     min_triangle_area, dist, idx);
  return p2pb_launch_status();
}

// -> dist f32[nt] squared distance of every triangle to its closest point, idx i32[nt] that point
extern "C" int p2pb_face_point_dist(int np, int nt, const float *points, const float *tris, float min_triangle_area,
                                    float *dist, int *idx, void *stream) {
  if (np <= 0 || nt <= 0) return P2PB_EINVAL;
  hipLaunchKernelGGL(face_point_kernel, dim3(cdiv(nt, 256)), dim3(256), 0, (hipStream_t)stream, np, nt, points, tris,
                     min_triangle_area, dist, idx);
  return p2pb_launch_status();
}
