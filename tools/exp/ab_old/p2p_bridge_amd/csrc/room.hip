// room.hip -- the two device pieces of the room pipeline (denoise_room.py):
//   * exact radius query: all points within r of each patch centre, in ascending point index -- replaces
//     sklearn.neighbors.KDTree.query_radius (denoise_room.py:459-464);
//   * running-mean merge of overlapping patch predictions back onto the room's points -- replaces the numba loops
//     update_prediction_noisy_batches (denoise_room.py:263-289).
// A room has 10^5..10^6 points and a few hundred patch centres: brute force is ~10^9 distance evaluations (about a
// millisecond); one WAVE per centre walks the cloud 64 points per step and compacts the hits in index order with
// ballot + mbcnt, the same idiom as the ball query of neighbors.hip -- two passes (count, fill) because the lists
// are ragged.
#include "common.h"

// counts[s] = #{ i : |p_i - c_s|^2 <= r2 }       points f32[n,3] (point-major, as the room is stored), centers f32[s,3]
__global__ __launch_bounds__(256) void radius_count_kernel(int s, int n, const float *__restrict__ centers,
                                                           const float *__restrict__ points, float r2,
                                                           int *__restrict__ counts) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= s) return;
  const int lane = lane_id();
  const float cx = centers[c * 3], cy = centers[c * 3 + 1], cz = centers[c * 3 + 2];
  int cnt = 0;
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    bool hit = false;
    if (i < n) hit = sqdist3(points[(size_t)i * 3] - cx, points[(size_t)i * 3 + 1] - cy, points[(size_t)i * 3 + 2] - cz) <= r2;
    cnt += __popcll(__ballot(hit));
  }
  if (lane == 0) counts[c] = cnt;
}

// out[offsets[s] ..] = the hit indices of centre s, ascending
__global__ __launch_bounds__(256) void radius_fill_kernel(int s, int n, const float *__restrict__ centers,
                                                          const float *__restrict__ points, float r2,
                                                          const long long *__restrict__ offsets,
                                                          int *__restrict__ out) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= s) return;
  const int lane = lane_id();
  const float cx = centers[c * 3], cy = centers[c * 3 + 1], cz = centers[c * 3 + 2];
  long long pos = offsets[c];
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    bool hit = false;
    if (i < n) hit = sqdist3(points[(size_t)i * 3] - cx, points[(size_t)i * 3 + 1] - cy, points[(size_t)i * 3 + 2] - cz) <= r2;
    const unsigned long long mask = __ballot(hit);
    if (hit) out[pos + mbcnt(mask)] = i;
    pos += __popcll(mask);
  }
}

extern "C" int p2pb_radius_count(int s, int n, const float *centers, const float *points, float radius, int *counts,
                                 void *stream) {
  if (s <= 0 || n <= 0 || !centers || !points || !counts || !(radius >= 0.0f)) return P2PB_EINVAL;
  hipLaunchKernelGGL(radius_count_kernel, dim3(cdiv(s, 4)), dim3(256), 0, (hipStream_t)stream, s, n, centers, points,
                     radius * radius, counts);
  return p2pb_launch_status();
}

extern "C" int p2pb_radius_fill(int s, int n, const float *centers, const float *points, float radius,
                                const long long *offsets, int *out, void *stream) {
  if (s <= 0 || n <= 0 || !centers || !points || !offsets || !out || !(radius >= 0.0f)) return P2PB_EINVAL;
  hipLaunchKernelGGL(radius_fill_kernel, dim3(cdiv(s, 4)), dim3(256), 0, (hipStream_t)stream, s, n, centers, points,
                     radius * radius, offsets, out);
  return p2pb_launch_status();
}

// ---- merge: every room point ends as the mean of all patch predictions that map to it (the reference's sequential
// running mean, (mean * (k - 1) + x) / k, is that mean; it runs in float64 there, so do the sums here). Only the first
// cuts[p] entries of patch p count (the rest are the random duplicates that padded a small radius patch).
__global__ __launch_bounds__(256) void merge_accumulate_kernel(int npatch, int k, const float *__restrict__ pred,
                                                               const int *__restrict__ idx,
                                                               const int *__restrict__ cuts, double *__restrict__ sums,
                                                               int *__restrict__ counts) {
  const int p = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (p >= npatch || j >= k || j >= cuts[p]) return;
  const int i = idx[(size_t)p * k + j];
  const float *v = pred + ((size_t)p * k + j) * 3;
  atomicAdd(sums + (size_t)i * 3, (double)v[0]);
  atomicAdd(sums + (size_t)i * 3 + 1, (double)v[1]);
  atomicAdd(sums + (size_t)i * 3 + 2, (double)v[2]);
  atomicAdd(counts + i, 1);
}

__global__ __launch_bounds__(256) void merge_finish_kernel(int n, const double *__restrict__ sums,
                                                           const int *__restrict__ counts,
                                                           const float *__restrict__ original, float *__restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = counts[i];
#pragma unroll
  for (int a = 0; a < 3; ++a)
    out[(size_t)i * 3 + a] = c > 0 ? (float)(sums[(size_t)i * 3 + a] / (double)c) : original[(size_t)i * 3 + a];
}

// pred f32[npatch,k,3] (de-normalised predictions), idx i32[npatch,k] (room point of every patch point), cuts i32[npatch];
// sums f64[n,3] and counts i32[n] are accumulators the CALLER zeroes once per room (several batches add into them)
extern "C" int p2pb_merge_accumulate(int npatch, int k, const float *pred, const int *idx, const int *cuts,
                                     double *sums, int *counts, void *stream) {
  if (npatch <= 0 || k <= 0 || !pred || !idx || !cuts || !sums || !counts) return P2PB_EINVAL;
  hipLaunchKernelGGL(merge_accumulate_kernel, dim3(cdiv(k, 256), npatch), dim3(256), 0, (hipStream_t)stream, npatch, k,
                     pred, idx, cuts, sums, counts);
  return p2pb_launch_status();
}

// out f32[n,3] = sums / counts where counts > 0, else `original` (points no patch reached)
extern "C" int p2pb_merge_finish(int n, const double *sums, const int *counts, const float *original, float *out,
                                 void *stream) {
  if (n <= 0 || !sums || !counts || !original || !out) return P2PB_EINVAL;
  hipLaunchKernelGGL(merge_finish_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, sums, counts, original,
                     out);
  return p2pb_launch_status();
}
