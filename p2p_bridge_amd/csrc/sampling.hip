// sampling.hip -- furthest point sampling for gfx950 (PN2/pvcnn_sampling_gpu.cu:92-184).
//
// FPS is M-1 strictly dependent rounds; only the B clouds are independent, so the kernel is
// latency-bound by construction: one workgroup per cloud, and everything a round touches stays on
// the CU. Each thread owns PPT points: coordinates AND running min-distances live in VGPRs for the
// whole kernel (the reference re-reads a global `distances` array and a 3072-point shared cache
// every round). A round is: PPT fused distance updates per lane, a 64-bit (distance, tie-key)
// max-reduction with DPP row shifts/broadcasts inside each wave, one LDS slot per wave, ONE
// s_barrier, and a 16-entry row reduction that every wave repeats redundantly (cheaper than a
// second barrier). The winner's coordinates come from an LDS copy of the cloud when it fits.
//
// Tie-break parity: the reference's block is 512 threads, thread t scans k = t, t+512, ... and keeps
// the first strict maximum, then a shared-memory tree keeps the LEFT operand on ties (:170). The
// net order is (d desc, k mod 512 asc, k asc); it is encoded in the low word of the key so any
// thread layout reproduces it.
#include "common.h"
#include <cstdlib>

typedef unsigned long long u64;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u64 dpp_max_step(u64 v) {
  const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, ROW_MASK, 0xf, false);
  const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, ROW_MASK, 0xf, false);
  const u64 o = ((u64)ohi << 32) | olo;
  return o > v ? o : v;
}

// max over the 64 lanes; result valid in lane 63, returned broadcast to all lanes
__device__ __forceinline__ u64 wave_max_u64(u64 v) {
  v = dpp_max_step<0x111, 0xf>(v);  // row_shr:1
  v = dpp_max_step<0x112, 0xf>(v);  // row_shr:2
  v = dpp_max_step<0x114, 0xf>(v);  // row_shr:4
  v = dpp_max_step<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row max
  v = dpp_max_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
  v = dpp_max_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3 -> lane 63 holds the wave max
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
  return ((u64)hi << 32) | lo;
}

// max over each 16-lane row; valid in lane 15 of the row
__device__ __forceinline__ u64 row_max_u64(u64 v) {
  v = dpp_max_step<0x111, 0xf>(v);
  v = dpp_max_step<0x112, 0xf>(v);
  v = dpp_max_step<0x114, 0xf>(v);
  v = dpp_max_step<0x118, 0xf>(v);
  return v;
}

__device__ __forceinline__ u64 fps_key(float best, int k) {
  // best >= 0 for any real point; threads without points carry best = -1 -> lowest key
  const unsigned hi = best >= 0.0f ? (__float_as_uint(best) + 1u) : 0u;
  const unsigned sec = ((unsigned)(k & 511) << 20) | (unsigned)(k >> 9);  // lower is better
  return ((u64)hi << 32) | (u64)(~sec);
}

__device__ __forceinline__ int fps_key_index(u64 key) {
  const unsigned sec = ~(unsigned)key;
  return (int)(((sec & 0xFFFFFu) << 9) | (sec >> 20));
}

// Point index owned by (thread t, slot i). A thread must visit its points in ascending
// (k mod 512, k) so that "first strict maximum" inside the thread agrees with the global tie order.
// With >= 512 threads (or <= 512 points in total) the natural strided order already does; with 256
// threads and more than 512 points a thread owns 2 residue classes and walks them class-major.
template <int THREADS, int PPT>
__device__ __forceinline__ int fps_point(int t, int i) {
  if (THREADS >= 512 || THREADS * PPT <= 512) return t + i * THREADS;
  constexpr int CPT = THREADS < 512 ? 512 / THREADS : 1;  // residue classes per thread
  constexpr int PPC = PPT >= CPT ? PPT / CPT : 1;         // points per class
  return t + (i / PPC) * THREADS + 512 * (i % PPC);
}

template <int THREADS, int PPT, bool LDS_XYZ>
__global__ __launch_bounds__(THREADS) void fps_kernel(int n, int m, const float *__restrict__ coords,
                                                      int *__restrict__ indices) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = THREADS / 64;
  u64 *slots = (u64 *)smem;                       // [2][16]
  float *sxyz = (float *)(smem + 2 * 16 * sizeof(u64));  // [3][n] when LDS_XYZ
  const int t = threadIdx.x;
  const float *c = coords + (size_t)blockIdx.x * 3 * n;
  int *out = indices + (size_t)blockIdx.x * m;

  float x[PPT], y[PPT], z[PPT], dist[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = fps_point<THREADS, PPT>(t, i);
    const bool ok = k < n;
    x[i] = ok ? c[k] : 0.0f;
    y[i] = ok ? c[k + n] : 0.0f;
    z[i] = ok ? c[k + 2 * n] : 0.0f;
    dist[i] = ok ? 1e38f : -1.0f;  // PN2/pvcnn_sampling.cpp:56 ; -1 = "no point here", never selected
    if (LDS_XYZ && ok) {
      sxyz[k] = x[i];
      sxyz[n + k] = y[i];
      sxyz[2 * n + k] = z[i];
    }
  }
  if (t < 32) slots[t] = 0;
  if (t == 0) out[0] = 0;
  __syncthreads();

  int old = 0;
  for (int j = 1; j < m; ++j) {
    float x1, y1, z1;
    if (LDS_XYZ) {
      x1 = sxyz[old];
      y1 = sxyz[n + old];
      z1 = sxyz[2 * n + old];
    } else {
      x1 = c[old];
      y1 = c[old + n];
      z1 = c[old + 2 * n];
    }
    float best = -1.0f;
    int bi = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float d = sqdist3(x[i] - x1, y[i] - y1, z[i] - z1);
      const float d2 = fminf(d, dist[i]);
      dist[i] = d2;
      if (d2 > best) {
        best = d2;
        bi = i;
      }
    }
    u64 key = wave_max_u64(fps_key(best, fps_point<THREADS, PPT>(t, bi)));
    if (NW > 1) {
      u64 *sl = slots + (j & 1) * 16;
      if ((t & 63) == 0) sl[t >> 6] = key;
      __syncthreads();
      u64 v = sl[t & 15];  // entries >= NW stay 0 (identity)
      v = row_max_u64(v);
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 15);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 15);
      key = ((u64)hi << 32) | lo;
    }
    old = fps_key_index(key);
    if (t == 0) out[j] = old;
  }
}

// n > 16384: running distances in global scratch, coordinates streamed from L2 each round.
// only_if != nullptr: the kernel is the on-device fallback of fps_coop_kernel and runs only when that flag was raised.
__global__ __launch_bounds__(1024) void fps_big_kernel(int n, int m, const float *__restrict__ coords,
                                                       float *__restrict__ dist_ws, int *__restrict__ indices,
                                                       const int *__restrict__ only_if) {
  __shared__ u64 slots[2][16];
  const int t = threadIdx.x;
  if (only_if && __hip_atomic_load(only_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
  const float *c = coords + (size_t)blockIdx.x * 3 * n;
  float *dist = dist_ws + (size_t)blockIdx.x * n;
  int *out = indices + (size_t)blockIdx.x * m;
  for (int k = t; k < n; k += 1024) dist[k] = 1e38f;
  if (t < 32) slots[t >> 4][t & 15] = 0;
  if (t == 0) out[0] = 0;
  __syncthreads();
  int old = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = c[old], y1 = c[old + n], z1 = c[old + 2 * n];
    float best = -1.0f;
    int bk = 0;
    for (int k = t; k < n; k += 1024) {
      const float d = sqdist3(c[k] - x1, c[k + n] - y1, c[k + 2 * n] - z1);
      const float d2 = fminf(d, dist[k]);
      dist[k] = d2;
      if (d2 > best) {
        best = d2;
        bk = k;
      }
    }
    u64 key = wave_max_u64(fps_key(best, bk));
    if ((t & 63) == 0) slots[j & 1][t >> 6] = key;
    __syncthreads();
    u64 v = row_max_u64(slots[j & 1][t & 15]);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 15);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 15);
    old = fps_key_index(((u64)hi << 32) | lo);
    if (t == 0) out[j] = old;
  }
}

// n > 16384, the object-merge case (denoise_object.py:112: 3N patch points -> N, N = 10^4 .. 10^5) and the
// 50000-point room patches of BASELINE configs 4-5: one workgroup would stream the whole cloud from L2 every round
// (28 us per round at 150 k points). Here FPS_G workgroups share one cloud: every thread keeps its PPT points and
// their running distances in registers, a round is a local (wave + LDS) argmax, ONE 8-byte word per workgroup
// published to global memory, and an all-gather by polling: lane i of wave 0 spins on word i, reduces, and
// broadcasts the winner through LDS. The word carries the round number (mod 8) in the three spare bits of the
// tie-key (n <= 2^19 leaves them free; they are equal in all words of a round, so the maximum is unaffected): a
// reader recognises a fresh word from the word itself, a 64-bit store is single-copy atomic, and no ordering
// between different locations is needed -- hence no release / acquire fences (which would write back / invalidate
// the XCD's whole L2 every round). Words are double-buffered by round parity (a workgroup can only be one round
// ahead of the slowest reader; a slot is rewritten every second round, so a stale word's tag differs by 2 mod 8).
// All FPS_G workgroups must be resident together: 64 x 1024 threads is a quarter of the chip; the host checks the
// occupancy, and a bounded spin turns a lost peer into an error flag (never a hang) that triggers the
// single-workgroup kernel as an on-device fallback (fps_big_kernel(only_if = flag)) -- the indices are always valid.
#define FPS_G 64
__device__ __forceinline__ u64 fps_tagged(u64 key, int j) {  // key's low word is ~sec, sec < 2^29
  return key ^ ((u64)(unsigned)(j & 7) << 29);
}
__device__ __forceinline__ bool fps_tag_is(u64 word, int j) {
  return ((~(unsigned)word) >> 29) == (unsigned)(j & 7);
}
__device__ __forceinline__ u64 fps_untag(u64 word) { return word | ((u64)7u << 29); }

template <int PPT>
__global__ __launch_bounds__(1024) void fps_coop_kernel(int n, int m, const float *__restrict__ coords,
                                                        u64 *__restrict__ keys, int *__restrict__ indices,
                                                        int *__restrict__ err) {
  __shared__ u64 slots[2][16];
  __shared__ int winner[2];
  const int t = threadIdx.x, g = blockIdx.x, bi = blockIdx.y;
  const float *c = coords + (size_t)bi * 3 * n;
  int *out = indices + (size_t)bi * m;
  u64 *kslot = keys + (size_t)bi * 2 * FPS_G;
  int *flag = err + bi;

  float x[PPT], y[PPT], z[PPT], dist[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = g * 1024 + t + i * (FPS_G * 1024);
    const bool ok = k < n;
    x[i] = ok ? c[k] : 0.0f;
    y[i] = ok ? c[k + n] : 0.0f;
    z[i] = ok ? c[k + 2 * n] : 0.0f;
    dist[i] = ok ? 1e38f : -1.0f;
  }
  if (t < 32) slots[t >> 4][t & 15] = 0;
  if (g == 0 && t == 0) out[0] = 0;
  __syncthreads();

  int old = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = c[old], y1 = c[old + n], z1 = c[old + 2 * n];
    float best = -1.0f;
    int bk = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float d = sqdist3(x[i] - x1, y[i] - y1, z[i] - z1);
      const float d2 = fminf(d, dist[i]);
      dist[i] = d2;
      if (d2 > best) {
        best = d2;
        bk = g * 1024 + t + i * (FPS_G * 1024);
      }
    }
    u64 key = wave_max_u64(fps_key(best, bk));
    if ((t & 63) == 0) slots[j & 1][t >> 6] = key;
    __syncthreads();
    if (t < 64) {  // wave 0: the workgroup's maximum -> its global word, then gather everybody's
      u64 v = row_max_u64(slots[j & 1][t & 15]);
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 15);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 15);
      const int par = (j & 1) * FPS_G;
      // device-scope write-through store / L2-bypassing loads of ONE self-describing word (see above)
      if (t == 0)
        __hip_atomic_store(&kslot[par + g], fps_tagged(((u64)hi << 32) | lo, j), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      u64 w;
      while (!fps_tag_is(w = __hip_atomic_load(&kslot[par + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), j)) {
        if (++spins > (1 << 22)) {  // a peer never arrived (not co-resident?): raise the flag, the fallback kernel runs
          __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          spins = -1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      const u64 all = wave_max_u64(fps_untag(w));
      const bool lost = __ballot(spins < 0) != 0;
      if (t == 0) winner[j & 1] = lost ? -1 : fps_key_index(all);
    }
    __syncthreads();
    old = winner[j & 1];
    if (old < 0) return;  // (the peers run into their own spin bound)
    if (g == 0 && t == 0) out[j] = old;
  }
}

template <int THREADS, int PPT>
static void fps_launch(int b, int n, int m, const float *coords, int *idx, hipStream_t s) {
  const size_t base = 2 * 16 * sizeof(u64);
  const size_t xyz = (size_t)3 * n * sizeof(float);
  if (base + xyz <= 144 * 1024) {
    hipLaunchKernelGGL((fps_kernel<THREADS, PPT, true>), dim3(b), dim3(THREADS), base + xyz, s, n, m, coords, idx);
  } else {
    hipLaunchKernelGGL((fps_kernel<THREADS, PPT, false>), dim3(b), dim3(THREADS), base, s, n, m, coords, idx);
  }
}

extern "C" int p2pb_furthest_point_sampling(int b, int n, int m, const float *coords, float *dist_ws, int *idx,
                                            void *stream) {
  if (b <= 0 || n <= 0 || m < 0) return P2PB_EINVAL;
  if (m == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set = false;
  if (!attr_set) {  // allow > 64 KiB of dynamic LDS for the coordinate cache
    const int big = 160 * 1024;
    (void)hipFuncSetAttribute((const void *)fps_kernel<1024, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void *)fps_kernel<1024, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void *)fps_kernel<1024, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void *)fps_kernel<512, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    attr_set = true;
  }
  if (n <= 64) fps_launch<64, 1>(b, n, m, coords, idx, s);
  else if (n <= 128) fps_launch<64, 2>(b, n, m, coords, idx, s);
  else if (n <= 256) fps_launch<64, 4>(b, n, m, coords, idx, s);
  else if (n <= 512) fps_launch<256, 2>(b, n, m, coords, idx, s);
  else if (n <= 1024) fps_launch<256, 4>(b, n, m, coords, idx, s);
  else if (n <= 2048) fps_launch<256, 8>(b, n, m, coords, idx, s);
  else if (n <= 4096) fps_launch<1024, 4>(b, n, m, coords, idx, s);
  else if (n <= 8192) fps_launch<512, 16>(b, n, m, coords, idx, s);  // 8 waves: 1.71 ms vs 2.10 with 16 x 8 points
  else if (n <= 16384) fps_launch<1024, 16>(b, n, m, coords, idx, s);
  else {
    if (!dist_ws) return P2PB_EINVAL;
    hipLaunchKernelGGL(fps_big_kernel, dim3(b), dim3(1024), 0, s, n, m, coords, dist_ws, idx, (const int *)nullptr);
  }
  return p2pb_launch_status();
}

__global__ void fps_set_flags_kernel(int b, int *__restrict__ flags) {
  for (int i = threadIdx.x; i < b; i += 64) flags[i] = 1;
}

// ws = [b][2][FPS_G] tagged words | [b] error flags (+ pad) | [b][n] fallback distances
static size_t fps_coop_head_bytes(int b) { return ((size_t)b * 2 * FPS_G * sizeof(u64) + (size_t)b * sizeof(int) + 15) & ~(size_t)15; }
extern "C" size_t p2pb_fps_coop_ws_bytes(int b, int n) {
  return fps_coop_head_bytes(b) + (size_t)b * (size_t)n * sizeof(float);
}

// Large clouds (16384 < n <= 524288), any b (launched four clouds at a time: b * 64 workgroups of 16 waves must be
// resident together): same result as p2pb_furthest_point_sampling. ws: p2pb_fps_coop_ws_bytes(b, n) bytes, head zeroed
// by the callee; after the call the ints at ws + 2*FPS_G*8*b are per-cloud flags (1 = the cooperative kernel lost a
// peer and the single-workgroup fallback produced that cloud's indices: slower, same result). Returns P2PB_EINVAL
// outside the range or when the device cannot hold one launch's workgroups at once (callers use
// p2pb_furthest_point_sampling then).
extern "C" int p2pb_furthest_point_sampling_coop(int b, int n, int m, const float *coords, void *ws, int *idx,
                                                 void *stream) {
  if (b <= 0 || n <= 16384 || n > FPS_G * 1024 * 8 || m < 0 || !ws) return P2PB_EINVAL;
  if (m == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int ppt = (n + FPS_G * 1024 - 1) / (FPS_G * 1024);
  static int resident = -1;  // workgroups of the widest variant the device holds at once
  if (resident < 0) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fps_coop_kernel<8>, 1024, 0) != hipSuccess)
      return P2PB_EINVAL;
    resident = cus * per_cu;
  }
  if (resident < FPS_G) return P2PB_EINVAL;
  const int per_launch = resident / FPS_G < 4 ? resident / FPS_G : 4;
  const size_t head = fps_coop_head_bytes(b);
  int e = p2pb_zero_async(ws, head, s);
  if (e != 0) return e;
  u64 *keys = (u64 *)ws;
  int *err = (int *)(keys + (size_t)b * 2 * FPS_G);
  float *dist = (float *)((char *)ws + head);
  // test hook: raise every cloud's flag up front, so the on-device fallback recomputes everything (tests/ check that
  // the indices are the same and that the flags report it)
  static const bool force_fallback = getenv("P2PB_FPS_COOP_TEST_FALLBACK") != nullptr;
  if (force_fallback) hipLaunchKernelGGL(fps_set_flags_kernel, dim3(1), dim3(64), 0, s, b, err);
  for (int b0 = 0; b0 < b; b0 += per_launch) {
    const int nb = b - b0 < per_launch ? b - b0 : per_launch;
    dim3 grid(FPS_G, nb);
    const float *c0 = coords + (size_t)b0 * 3 * n;
    u64 *k0 = keys + (size_t)b0 * 2 * FPS_G;
    int *i0 = idx + (size_t)b0 * m, *e0 = err + b0;
    if (ppt <= 1) hipLaunchKernelGGL(fps_coop_kernel<1>, grid, dim3(1024), 0, s, n, m, c0, k0, i0, e0);
    else if (ppt <= 2) hipLaunchKernelGGL(fps_coop_kernel<2>, grid, dim3(1024), 0, s, n, m, c0, k0, i0, e0);
    else if (ppt <= 4) hipLaunchKernelGGL(fps_coop_kernel<4>, grid, dim3(1024), 0, s, n, m, c0, k0, i0, e0);
    else hipLaunchKernelGGL(fps_coop_kernel<8>, grid, dim3(1024), 0, s, n, m, c0, k0, i0, e0);
  }
  // on-device fallback, one workgroup per cloud, returns at once unless that cloud's flag was raised
  for (int bi = 0; bi < b; ++bi)
    hipLaunchKernelGGL(fps_big_kernel, dim3(1), dim3(1024), 0, s, n, m, coords + (size_t)bi * 3 * n,
                       dist + (size_t)bi * n, idx + (size_t)bi * m, (const int *)(err + bi));
  return p2pb_launch_status();
}
