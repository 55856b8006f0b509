// knn.hip -- exact K nearest neighbours with large K (object patch extraction).
//
// Replaces pytorch3d.ops.knn_points(seeds, cloud, K=patch_size, return_nn=True) as called by
// denoise_object.py:91 (pytorch3d is a pip dependency of the reference, not vendored: its published contract
// is "the K smallest squared distances per query, ascending, with their indices"). K is a whole patch (2048 .. 4096
// points of a 10^4 .. 10^5-point cloud), so this is a selection problem, not the small-K heap of a usual kNN:
//   one workgroup (1024 threads) per query point:
//   1. squared distances to all n points -> keys in a global scratch row (distances are >= 0, so their bit
//      patterns order like unsigned integers); the cloud row is read once, coalesced;
//   2. radix select: four 8-bit passes of a 256-bin LDS histogram find the K-th smallest key T and the number
//      of keys below it (reads the n keys from L2 four times: 4n x 4 B per query);
//   3. ordered compaction (wave ballots + one LDS scan per 1024 keys): every key < T and the first
//      K - #(< T) keys == T in index order -> (key << 32 | index) pairs in LDS;
//   4. bitonic sort of the <= 4096 pairs in LDS (ascending distance, ties by ascending index);
//   5. dist2 / idx / the neighbours' coordinates are written from the sorted pairs.
// Bound: HBM/L2 traffic (20 n bytes per query); the 73 queries of a 50 k-point object are 73 workgroups.
#include "common.h"

typedef unsigned long long u64;
#define KNN_THREADS 1024
#define KNN_MAXK 4096

// exclusive position of this thread's flagged element among the workgroup's flagged elements (thread order),
// total in *total; wcnt = 16 ints of LDS scratch. Two barriers.
__device__ __forceinline__ int block_rank(bool flag, int *wcnt, int *total) {
  const unsigned long long bal = __ballot(flag);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wcnt[wave] = __popcll(bal);
  __syncthreads();
  int before = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < KNN_THREADS / 64; ++w) {
    const int c = wcnt[w];
    if (w < wave) before += c;
    tot += c;
  }
  __syncthreads();
  *total = tot;
  return before + mbcnt(bal);
}

__global__ __launch_bounds__(KNN_THREADS) void knn_select_kernel(int s, int n, int k, int kp,
                                                                const float *__restrict__ query,
                                                                const float *__restrict__ points,
                                                                unsigned *__restrict__ keys_ws,
                                                                float *__restrict__ dist2, int *__restrict__ idx,
                                                                float *__restrict__ nn) {
  __shared__ u64 pairs[KNN_MAXK];
  __shared__ int hist[256];
  __shared__ int wcnt[KNN_THREADS / 64];
  __shared__ unsigned sel_prefix;
  __shared__ int sel_need;
  const int t = threadIdx.x;
  const int q = blockIdx.x, b = blockIdx.y;
  const float *p = points + (size_t)b * n * 3;
  const float *qp = query + ((size_t)b * s + q) * 3;
  unsigned *keys = keys_ws + ((size_t)b * s + q) * n;
  const float qx = qp[0], qy = qp[1], qz = qp[2];

  // 1. keys
  for (int j = t; j < n; j += KNN_THREADS) {
    const float d = sqdist3(p[3 * j] - qx, p[3 * j + 1] - qy, p[3 * j + 2] - qz);
    keys[j] = __float_as_uint(d);
  }
  if (t == 0) {
    sel_prefix = 0u;
    sel_need = k;  // rank (1-based) of the wanted key among the keys matching the prefix so far
  }
  __syncthreads();

  // 2. radix select, most significant byte first
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    if (t < 256) hist[t] = 0;
    __syncthreads();
    const unsigned prefix = sel_prefix;
    for (int j = t; j < n; j += KNN_THREADS) {
      const unsigned key = keys[j];
      if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
    }
    __syncthreads();
    if (t < 64) {  // one wave scans the 256 bins: lane l owns bins 4l .. 4l+3
      int c[4], tot = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        c[i] = hist[4 * t + i];
        tot += c[i];
      }
      int incl = tot;  // inclusive scan over the lanes
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (t >= off) incl += o;
      }
      int before = incl - tot;
      const int need = sel_need;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (need > before && need <= before + c[i]) {  // exactly one (lane, i) satisfies this
          sel_prefix = prefix | ((unsigned)(4 * t + i) << shift);
          sel_need = need - before;
        }
        before += c[i];
      }
    }
    __syncthreads();
  }
  const unsigned T = sel_prefix;  // the K-th smallest key
  const int need_eq = sel_need;   // how many keys == T belong to the result (>= 1)
  __syncthreads();

  // 3. ordered compaction
  int base_less = 0, base_eq = 0;
  const int nless = k - need_eq;
  for (int j0 = 0; j0 < n; j0 += KNN_THREADS) {
    const int j = j0 + t;
    const unsigned key = j < n ? keys[j] : 0xFFFFFFFFu;
    const bool less = j < n && key < T, eq = j < n && key == T;
    int tot_less, tot_eq;
    const int rl = block_rank(less, wcnt, &tot_less);
    const int re = block_rank(eq, wcnt, &tot_eq);
    if (less) pairs[base_less + rl] = ((u64)key << 32) | (unsigned)j;
    if (eq && base_eq + re < need_eq) pairs[nless + base_eq + re] = ((u64)key << 32) | (unsigned)j;
    base_less += tot_less;
    base_eq += tot_eq;
  }
  for (int i = k + t; i < kp; i += KNN_THREADS) pairs[i] = ~0ull;  // padding sorts last
  __syncthreads();

  // 4. bitonic sort of kp pairs (kp = power of two >= k)
  for (int size = 2; size <= kp; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = t; i < (kp >> 1); i += KNN_THREADS) {
        const int lo = ((i / stride) * stride << 1) + (i % stride);
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const u64 a = pairs[lo], c = pairs[hi];
        if ((a > c) == up) {
          pairs[lo] = c;
          pairs[hi] = a;
        }
      }
      __syncthreads();
    }
  }

  // 5. outputs
  const size_t ob = ((size_t)b * s + q) * k;
  for (int i = t; i < k; i += KNN_THREADS) {
    const u64 pr = pairs[i];
    const int j = (int)(unsigned)pr;
    if (dist2) dist2[ob + i] = __uint_as_float((unsigned)(pr >> 32));
    if (idx) idx[ob + i] = j;
    if (nn) {
      nn[(ob + i) * 3] = p[3 * j];
      nn[(ob + i) * 3 + 1] = p[3 * j + 1];
      nn[(ob + i) * 3 + 2] = p[3 * j + 2];
    }
  }
}

extern "C" size_t p2pb_knn_points_ws_bytes(int b, int s, int n) { return (size_t)b * s * n * sizeof(unsigned); }

// query f32[b,s,3], points f32[b,n,3] (point-major, as pytorch3d) -> dist2 f32[b,s,k] ascending, idx i32[b,s,k],
// nn f32[b,s,k,3] (each may be NULL); 1 <= k <= min(n, 4096); ws: p2pb_knn_points_ws_bytes(b,s,n) bytes
extern "C" int p2pb_knn_points(int b, int s, int n, int k, const float *query, const float *points, float *dist2,
                               int *idx, float *nn, void *ws, void *stream) {
  if (b <= 0 || s <= 0 || n <= 0 || k <= 0 || k > n || k > KNN_MAXK || !ws) return P2PB_EINVAL;
  int kp = 1;
  while (kp < k) kp <<= 1;
  hipLaunchKernelGGL(knn_select_kernel, dim3(s, b), dim3(KNN_THREADS), 0, (hipStream_t)stream, s, n, k, kp, query, points,
                     (unsigned *)ws, dist2, idx, nn);
  return p2pb_launch_status();
}
