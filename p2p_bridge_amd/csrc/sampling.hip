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
__device__ __forceinline__ unsigned dpp_umax_step(unsigned v) {
  const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);  // (0 = identity of max)
  return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = dpp_umax_step<0x111, 0xf>(v);  // row_shr:1
  v = dpp_umax_step<0x112, 0xf>(v);  // row_shr:2
  v = dpp_umax_step<0x114, 0xf>(v);  // row_shr:4
  v = dpp_umax_step<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row max
  v = dpp_umax_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
  v = dpp_umax_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3 -> lane 63 holds the wave max
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// max of 64-bit keys over the 64 lanes, broadcast to all lanes: the lexicographic maximum as two 32-bit reductions (the high
// words, then the low words of the lanes that hold the winning high word) -- v_max_u32 steps instead of 64-bit compare +
// two selects per step (the round of the small-cloud FPS is bound by its VALU instruction count)
__device__ __forceinline__ u64 wave_max_u64(u64 v) {
  const unsigned hi = (unsigned)(v >> 32);
  const unsigned H = wave_max_u32(hi);
  const unsigned L = wave_max_u32(hi == H ? (unsigned)v : 0u);
  return ((u64)H << 32) | L;
}

// max over the first 16-lane row of the wave, returned in every lane (the callers' rows all hold the same 16 values): the same
// two 32-bit reductions
__device__ __forceinline__ unsigned row_max_u32(unsigned v) {
  v = dpp_umax_step<0x111, 0xf>(v);
  v = dpp_umax_step<0x112, 0xf>(v);
  v = dpp_umax_step<0x114, 0xf>(v);
  v = dpp_umax_step<0x118, 0xf>(v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 15);
}
__device__ __forceinline__ u64 row_max_u64(u64 v) {
  const unsigned hi = (unsigned)(v >> 32);
  const unsigned H = row_max_u32(hi);
  const unsigned L = row_max_u32(hi == H ? (unsigned)v : 0u);
  return ((u64)H << 32) | L;
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
      float d2;  // fminf(d, dist[i]) as the bare instruction (fminf canonicalises an operand with a v_max_f32 x, x first: 17 of the
      asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d), "v"(dist[i]));  // round's 258 VALU instructions; same value, NaN included)
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
      float d2;  // (bare v_min_f32: see fps_kernel)
      asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d), "v"(dist[i]));
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
  if (base + xyz <= 160 * 1024) {  // (the whole LDS of a CU: 12500-point clouds, level 1 of BASELINE configs 4-5, fit)
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
    (void)hipFuncSetAttribute((const void *)fps_kernel<512, 32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    attr_set = true;
  }
  static const bool fps_mid_wide = p2pb_experiment_long("fps_mid", 512) != 1024;
  if (n <= 64) fps_launch<64, 1>(b, n, m, coords, idx, s);
  else if (n <= 128) fps_launch<64, 2>(b, n, m, coords, idx, s);
  else if (n <= 256) fps_launch<64, 4>(b, n, m, coords, idx, s);
  else if (n <= 512) fps_launch<256, 2>(b, n, m, coords, idx, s);
  else if (n <= 1024) fps_launch<256, 4>(b, n, m, coords, idx, s);
  else if (n <= 2048) fps_launch<256, 8>(b, n, m, coords, idx, s);
  else if (n <= 4096) fps_launch<1024, 4>(b, n, m, coords, idx, s);
  else if (n <= 8192) fps_launch<512, 16>(b, n, m, coords, idx, s);  // 8 waves: 1.71 ms vs 2.10 with 16 x 8 points
  else if (n <= 16384 && fps_mid_wide) fps_launch<512, 32>(b, n, m, coords, idx, s);  // 8 waves x 32 points per lane: the round is
  else if (n <= 16384) fps_launch<1024, 16>(b, n, m, coords, idx, s);                  // bound by the reduction ACROSS waves (A/B: P2PB_FPS_MID=1024)
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
  static const bool force_fallback = p2pb_experiment_long("fps_coop_test_fallback", 0) != 0;
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

// ------------------------------------------------------------------------------------------------
// Large clouds, pruned (exact): a round of FPS only changes the running distance of points closer to the new sample
// than their current value, and after j samples that is a neighbourhood of ~n/j points -- 1.4 M updates in all for
// 50000 -> 12500 instead of 625 M. The cloud is binned into a 16^3 grid over its bounding box (fps_grid_build_kernel:
// (x, y, z, id) records contiguous per cell + every cell's tight bounding box); ONE workgroup per cloud then keeps,
// per cell, the 64-bit key of its farthest point (distance bits | tie-key: the reference's total order, so any
// reduction order gives the reference's winner) and that point's coordinates, in the registers of the lane that owns
// the cell. A round: every lane tests its four cells -- the squared distance from the new sample to the cell's box is a
// LOWER bound of sqdist3 for every point inside, in fp32 too, because subtraction, multiplication and fma are monotone
// under rounding, so a cell whose bound is >= its current maximum cannot change and is skipped; the wave recomputes
// the others (all 64 lanes on one cell, exactly fps_kernel's arithmetic: sqdist3, fminf); then the argmax over the
// cell keys, lane -> wave -> 16 slots in LDS, ONE barrier, the winner's coordinates riding along so that no dependent
// global load sits on the round's critical path. (First version: a shared list of cells built with LDS atomics and
// three barriers per round -- 2.4 us per round, of which 0.64 the list and 1.4 the barriers + reductions.)
// Same indices as fps_kernel / the oracle for any cloud (duplicates, lattices, degenerate boxes included).
// ------------------------------------------------------------------------------------------------
#define FG_G 16
#define FG_CELLS (FG_G * FG_G * FG_G)

__global__ __launch_bounds__(1024) void fps_grid_build_kernel(int n, const float *__restrict__ coords,
                                                              int *__restrict__ cell_start, float4 *__restrict__ rec,
                                                              float *__restrict__ cbox) {
  __shared__ int cnt[FG_CELLS];
  __shared__ int part[1024];
  __shared__ float red[6][16];
  __shared__ float sbox[6];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float *c = coords + (size_t)b * 3 * n;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = t; k < n; k += 1024)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = c[k + (size_t)a * n];
      lo[a] = fminf(lo[a], v);
      hi[a] = fmaxf(hi[a], v);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
    }
    if (lane == 0) {
      red[a][wave] = lo[a];
      red[3 + a][wave] = hi[a];
    }
  }
  for (int i = t; i < FG_CELLS; i += 1024) cnt[i] = 0;
  __syncthreads();
  if (t == 0) {
    // 16 layers along EACH axis of the cloud's own bounding box (round 4; it was a cube of the largest extent): a surface
    // patch is a thin slab of its bounding cube -- 450-550 of the 4096 cells occupied, 100 points apiece at 50000, and a
    // cell update costs by the point -- while its own box spreads it over 3-4 x as many cells
    for (int a = 0; a < 3; ++a) {
      float l = INFINITY, h = -INFINITY;
      for (int w = 0; w < 16; ++w) {
        l = fminf(l, red[a][w]);
        h = fmaxf(h, red[3 + a][w]);
      }
      sbox[a] = l;
      sbox[3 + a] = (float)FG_G / fmaxf(h - l, 1e-12f);
    }
  }
  __syncthreads();
  auto cell_of = [&](int k) {  // (any deterministic binning will do: the boxes below are the points' own)
    int q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float f = (c[k + (size_t)a * n] - sbox[a]) * sbox[3 + a];
      q[a] = f >= 0.0f ? min((int)f, FG_G - 1) : 0;  // (NaN coordinates land in cell 0)
    }
    return (q[2] * FG_G + q[1]) * FG_G + q[0];
  };
  for (int k = t; k < n; k += 1024) atomicAdd(&cnt[cell_of(k)], 1);
  __syncthreads();
  int c4[4], tot = 0;  // exclusive scan of the 4096 counts: thread t owns cells 4t .. 4t+3
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    c4[i] = cnt[4 * t + i];
    tot += c4[i];
  }
  part[t] = tot;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = t >= o ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - tot;
  int *cs = cell_start + (size_t)b * (FG_CELLS + 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    cs[4 * t + i] = run;
    cnt[4 * t + i] = run;  // becomes the fill cursor
    run += c4[i];
  }
  if (t == 1023) cs[FG_CELLS] = run;
  __syncthreads();
  float4 *rc = rec + (size_t)b * n;
  for (int k = t; k < n; k += 1024)
    rc[atomicAdd(&cnt[cell_of(k)], 1)] = make_float4(c[k], c[k + (size_t)n], c[k + (size_t)2 * n], __int_as_float(k));
  __syncthreads();
  float *bx = cbox + (size_t)b * FG_CELLS * 6;  // the tight box of every cell's points
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cell = 4 * t + i, s0 = cs[cell], s1 = s0 + c4[i];
    float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int x = s0; x < s1; ++x) {
      const float4 r = rc[x];
      l[0] = fminf(l[0], r.x), h[0] = fmaxf(h[0], r.x);
      l[1] = fminf(l[1], r.y), h[1] = fmaxf(h[1], r.y);
      l[2] = fminf(l[2], r.z), h[2] = fmaxf(h[2], r.z);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      bx[(size_t)cell * 6 + a] = l[a];
      bx[(size_t)cell * 6 + 3 + a] = h[a];
    }
  }
}

// Cell ownership: wave w owns, in every (cy, cz) row of the grid, the cell with cx = (w - 3 cy - 9 cz) mod 16, lane l its
// rows l, l + 64, l + 128, l + 192. The 27 cells of a 3x3x3 neighbourhood differ by dx + 3 dy + 9 dz, all distinct
// in [-13, 13], so they fall on the 16 waves at most two apiece: the cells a sample can change are spread over the
// waves by construction, every wave deals with its own without a list, an atomic or a barrier, and the only
// workgroup-wide step of a round is the final argmax over 16 per-wave maxima.
__device__ __forceinline__ int fg_cell(int wave, int row) {
  const int cy = row & 15, cz = row >> 4;
  return ((wave - 3 * cy - 9 * cz) & 15) + 16 * cy + 256 * cz;
}

typedef float fg_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void fps_grid_kernel(int n, int m, const float *__restrict__ coords,
                                                        const int *__restrict__ cell_start,
                                                        const float4 *__restrict__ rec, const float *__restrict__ cbox,
                                                        float *__restrict__ mind, int *__restrict__ indices) {
  __shared__ u64 slots[2][16];      // per-wave maxima (double-buffered by round parity: one barrier per round)
  __shared__ __attribute__((aligned(16))) float sxyz[2][16][4];  // ... and the coordinates of those points
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x;
  const float *c = coords + (size_t)b * 3 * n;
  const int *cs = cell_start + (size_t)b * (FG_CELLS + 1);
  const float4 *rc = rec + (size_t)b * n;
  const float *bxp = cbox + (size_t)b * FG_CELLS * 6;
  float *md = mind + (size_t)b * n;
  int *out = indices + (size_t)b * m;

  // this lane's four cells: record range, tight box, key of the farthest point (distance bits | tie-key: the
  // reference's total order) and that point's coordinates -- all in registers
  int s0[4], cn[4];
  float blo[4][3], bhi[4][3], cmax[4], cx[4], cy[4], cz[4];
  u64 ckey[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cell = fg_cell(wave, lane + 64 * i);
    s0[i] = cs[cell];
    cn[i] = cs[cell + 1] - s0[i];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      blo[i][a] = bxp[(size_t)cell * 6 + a];
      bhi[i][a] = bxp[(size_t)cell * 6 + 3 + a];
    }
    cmax[i] = cn[i] > 0 ? 1e38f : -1.0f;  // every point starts at 1e38 (PN2/pvcnn_sampling.cpp:56)
    ckey[i] = 0;
    cx[i] = cy[i] = cz[i] = 0.0f;
  }
  for (int k = t; k < n; k += 1024) md[k] = 1e38f;
  if (t < 32) slots[t >> 4][t & 15] = 0;
  if (t == 0) out[0] = 0;
  float sx = c[0], sy = c[n], sz = c[(size_t)2 * n];  // sample 0 = point 0
  __syncthreads();

  for (int j = 1; j < m; ++j) {
    // ---- the wave's cells this sample can change: the squared distance to the cell's box bounds sqdist3 of every
    // point inside from below (also in fp32: the operations are monotone under rounding), so bound >= current
    // maximum means nothing in the cell changes. Round 1 visits every cell: that is what initialises the keys.
    // (Handing several cells of a wave to 16-lane rows through LDS so that their load latencies overlap measured
    //  20 % SLOWER than taking them one after the other with all 64 lanes: a wave rarely has more than two.)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dx = fmaxf(fmaxf(blo[i][0] - sx, sx - bhi[i][0]), 0.0f);
      const float dy = fmaxf(fmaxf(blo[i][1] - sy, sy - bhi[i][1]), 0.0f);
      const float dz = fmaxf(fmaxf(blo[i][2] - sz, sz - bhi[i][2]), 0.0f);
      const bool hit = cn[i] > 0 && (j == 1 || !(sqdist3(dx, dy, dz) >= cmax[i]));
      unsigned long long todo = __ballot(hit);
      while (todo) {  // (wave-uniform) all 64 lanes recompute one cell with exactly fps_kernel's arithmetic
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int p0 = __builtin_amdgcn_readlane(s0[i], src), pn = __builtin_amdgcn_readlane(cn[i], src);
        u64 best = 0;
        float bxv = 0.0f, byv = 0.0f, bzv = 0.0f;
        for (int i0 = 0; i0 < pn; i0 += 256) {  // four points per lane in flight
          float4 r[4];
          float dold[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int k = i0 + 64 * q + lane;
            if (k < pn) {
              r[q] = rc[p0 + k];
              dold[q] = md[p0 + k];
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int k = i0 + 64 * q + lane;
            if (k < pn) {
              const float d = sqdist3(r[q].x - sx, r[q].y - sy, r[q].z - sz);
              float d2;  // (bare v_min_f32: see fps_kernel)
              asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d), "v"(dold[q]));
              if (d2 != dold[q]) md[p0 + k] = d2;
              const u64 key = fps_key(d2, __float_as_int(r[q].w));
              if (key > best) {
                best = key;
                bxv = r[q].x, byv = r[q].y, bzv = r[q].z;
              }
            }
          }
        }
        const u64 wbest = wave_max_u64(best);
        const int from = __builtin_ctzll(__ballot(best == wbest));  // (keys are unique: the point index is part of them)
        const float wx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bxv), from));
        const float wy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, byv), from));
        const float wz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bzv), from));
        if (lane == src) {
          ckey[i] = wbest;
          cmax[i] = __uint_as_float((unsigned)(wbest >> 32) - 1u);
          cx[i] = wx, cy[i] = wy, cz[i] = wz;
        }
      }
    }
    // ---- argmax: lane -> wave -> 16 slots
    u64 key = ckey[0];
    float kx = cx[0], ky = cy[0], kz = cz[0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (ckey[i] > key) {
        key = ckey[i];
        kx = cx[i], ky = cy[i], kz = cz[i];
      }
    const u64 wkey = wave_max_u64(key);
    if (key == wkey && (wkey != 0 ? true : lane == 0)) {  // (an all-empty wave: lane 0 writes the zero key)
      slots[j & 1][wave] = wkey;
      // (one 16-byte store: three adjacent floats become a ds_write_b96, which misbehaved beside another stream's matrix
      //  kernels -- voxelize.hip devox_cl_kernel, round 4)
      *(volatile fg_f32x4 *)&sxyz[j & 1][wave][0] = fg_f32x4{kx, ky, kz, 0.0f};
    }
    __syncthreads();
    const u64 mine = slots[j & 1][t & 15];
    const u64 v = row_max_u64(mine);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 15);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 15);
    const u64 fin = ((u64)hi << 32) | lo;
    const int ws = __builtin_ctzll(__ballot(lane < 16 && mine == fin));
    const fg_f32x4 sq = *(volatile fg_f32x4 *)&sxyz[j & 1][ws][0];
    sx = sq[0];
    sy = sq[1];
    sz = sq[2];
    if (t == 0) out[j] = fps_key_index(fin);
  }
}

static size_t fps_grid_head_bytes(int b) { return (((size_t)b * (FG_CELLS + 1) * 4) + 15) & ~(size_t)15; }
// cell_start i32[b][G^3+1] | cell boxes f32[b][G^3][6] | records float4[b][n] | running distances f32[b][n]
extern "C" size_t p2pb_fps_grid_ws_bytes(int b, int n) {
  return fps_grid_head_bytes(b) + (size_t)b * ((size_t)FG_CELLS * 6 * 4 + (size_t)n * 20);
}

// coords f32[b,3,n] -> idx i32[b,m], the indices of p2pb_furthest_point_sampling, for any n >= 1 (meant for n > 16384);
// ws: p2pb_fps_grid_ws_bytes(b, n) bytes, 16-byte aligned. One workgroup per cloud.
extern "C" int p2pb_furthest_point_sampling_grid(int b, int n, int m, const float *coords, void *ws, int *idx,
                                                 void *stream) {
  if (b <= 0 || n <= 0 || n >= (1 << 29) || m < 0 || !ws) return P2PB_EINVAL;
  if (m == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  // per-cloud regions are laid out array by array so that one launch serves the batch
  char *w = (char *)ws;
  int *cell_start = (int *)w;
  w += fps_grid_head_bytes(b);
  float *cbox = (float *)w;
  w += (size_t)b * FG_CELLS * 6 * 4;
  float4 *rec = (float4 *)w;
  w += (size_t)b * n * 16;
  float *mind = (float *)w;
  hipLaunchKernelGGL(fps_grid_build_kernel, dim3(b), dim3(1024), 0, s, n, coords, cell_start, rec, cbox);
  hipLaunchKernelGGL(fps_grid_kernel, dim3(b), dim3(1024), 0, s, n, m, coords, cell_start, rec, cbox, mind, idx);
  return p2pb_launch_status();
}
