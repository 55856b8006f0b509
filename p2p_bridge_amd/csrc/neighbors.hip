// neighbors.hip -- neighbourhood search / gather kernels for gfx950.
//   p2pb_ball_query                  (PN2/pvcnn_ball_query_gpu.cu:19)
//   p2pb_grouping_forward/backward   (PN2/pvcnn_grouping_gpu.cu:18,62)
//   p2pb_gather_features_*           (PN2/pvcnn_sampling_gpu.cu:17,55)
//   p2pb_three_nn_interpolate_*      (PN2/pvcnn_neighbor_interpolate_gpu.cu:20,96,154)
// The reference runs ONE thread block per cloud for each of these; here every kernel is spread over
// (points or centres) x channels x batch so a B=32 launch covers all 256 CUs.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// ball query: one 64-lane wave per centre. The wave streams the cloud 64 points at a time
// (lane-consecutive = coalesced reads of the [3,N] coordinate rows), a wave ballot marks the
// in-radius lanes and the popcount of the lower lanes is the output slot, which reproduces the
// reference's "first u hits in ascending point index" order exactly; the scan stops as soon as
// u hits are found.
// ------------------------------------------------------------------------------------------------
template <int UNROLL>
__global__ __launch_bounds__(256) void ball_query_kernel(int n, int m, float r2, int u,
                                                         const float *__restrict__ centers,
                                                         const float *__restrict__ points, int *__restrict__ idx) {
  const int b = blockIdx.y;
  const int lane = lane_id();
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= m) return;
  const float *p = points + (size_t)b * 3 * n;
  const float *ce = centers + (size_t)b * 3 * m;
  int *o = idx + ((size_t)b * m + j) * u;
  const float cx = ce[j], cy = ce[j + m], cz = ce[j + 2 * m];
  int cnt = 0, first = 0;
  for (int base = 0; base < n && cnt < u; base += 64 * UNROLL) {
    float px[UNROLL], py[UNROLL], pz[UNROLL];
#pragma unroll
    for (int q = 0; q < UNROLL; ++q) {
      const int k = base + q * 64 + lane;
      const bool ok = k < n;
      px[q] = ok ? p[k] : 0.0f;
      py[q] = ok ? p[k + n] : 0.0f;
      pz[q] = ok ? p[k + 2 * n] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < UNROLL; ++q) {
      const int k = base + q * 64 + lane;
      const float d2 = sqdist3(cx - px[q], cy - py[q], cz - pz[q]);
      const bool in = (k < n) && (d2 < r2);
      const unsigned long long mask = __ballot(in);
      if (mask) {
        const int slot = cnt + mbcnt(mask);
        if (in && slot < u) o[slot] = k;
        if (cnt == 0) first = base + q * 64 + (int)__builtin_ctzll(mask);
        cnt += (int)__builtin_popcountll(mask);
      }
    }
  }
  // PN2/pvcnn_ball_query_gpu.cu:44-50: on the first hit every slot is set to it; later hits
  // overwrite slots 0..cnt-1. No hit at all leaves the zero initialisation (pvcnn_ball_query.cpp:21).
  for (int v = lane; v < u; v += 64)
    if (v >= cnt) o[v] = cnt > 0 ? first : 0;
}

// The same scan with the cloud in LDS. One wave per centre straight from global memory makes every wave re-read the
// cloud through L1/L2: 2048 centres x ~48 % of 8192 points x 12 B x 32 samples = 3 GB of cache traffic per call at the
// first level. Alone the kernel is bound by instruction issue either way (190 us: ~25 instructions per 64-point step,
// mostly the ballot / slot bookkeeping -- packed fp32 for the distances changes nothing), but it runs on the geometry
// stream BESIDE the GEMMs and convolutions of the main stream, which need that cache bandwidth: with a workgroup of
// 16 waves copying the cloud into LDS once (3 x n floats, 96 KB at n = 8192) and every wave walking `cpw` centres over
// it, the sampler gains 0.9 % end to end. Same hit order, same outputs.
template <int UNROLL>
__global__ __launch_bounds__(1024) void ball_query_lds_kernel(int n, int m, float r2, int u, int cpw,
                                                              const float *__restrict__ centers,
                                                              const float *__restrict__ points,
                                                              int *__restrict__ idx) {
  extern __shared__ float bq_pts[];  // [3][n]
  const int b = blockIdx.y;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const float *p = points + (size_t)b * 3 * n;
  const float *ce = centers + (size_t)b * 3 * m;
  for (int k = threadIdx.x; k < 3 * n; k += 1024) bq_pts[k] = p[k];
  __syncthreads();
  const float *sx = bq_pts, *sy = bq_pts + n, *sz = bq_pts + 2 * n;
  const int j0 = (blockIdx.x * 16 + wave) * cpw;
  for (int j = j0; j < min(j0 + cpw, m); ++j) {
    int *o = idx + ((size_t)b * m + j) * u;
    const float cx = ce[j], cy = ce[j + m], cz = ce[j + 2 * m];
    int cnt = 0, first = 0;
    for (int base = 0; base < n && cnt < u; base += 64 * UNROLL) {
      float px[UNROLL], py[UNROLL], pz[UNROLL];
#pragma unroll
      for (int q = 0; q < UNROLL; ++q) {
        const int k = min(base + q * 64 + lane, n - 1);
        px[q] = sx[k];
        py[q] = sy[k];
        pz[q] = sz[k];
      }
#pragma unroll
      for (int q = 0; q < UNROLL; ++q) {
        const int k = base + q * 64 + lane;
        const float d2 = sqdist3(cx - px[q], cy - py[q], cz - pz[q]);
        const bool in = (k < n) && (d2 < r2);
        const unsigned long long mask = __ballot(in);
        if (mask) {
          const int slot = cnt + mbcnt(mask);
          if (in && slot < u) o[slot] = k;
          if (cnt == 0) first = base + q * 64 + (int)__builtin_ctzll(mask);
          cnt += (int)__builtin_popcountll(mask);
        }
      }
    }
    for (int v = lane; v < u; v += 64)
      if (v >= cnt) o[v] = cnt > 0 ? first : 0;
  }
}

extern "C" int p2pb_ball_query(int b, int n, int m, float r2, int u, const float *centers, const float *points,
                               int *idx, void *stream) {
  if (b <= 0 || n <= 0 || m <= 0 || u <= 0) return P2PB_EINVAL;
  const size_t lds = (size_t)3 * n * sizeof(float);
  if (lds <= 144 * 1024 && (long)m * b >= 2048) {  // the cloud fits in LDS and there are enough centres to share it
    static bool once = false;
    if (!once) {
      (void)hipFuncSetAttribute((const void *)ball_query_lds_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
      once = true;
    }
    // centres per wave: about two workgroups per CU over the whole launch, at least 1
    int cpw = (int)(((long)m * b + 16L * 512 - 1) / (16L * 512));
    if (cpw < 1) cpw = 1;
    hipLaunchKernelGGL(ball_query_lds_kernel<4>, dim3(cdiv(m, 16 * cpw), b), dim3(1024), lds, (hipStream_t)stream, n, m,
                       r2, u, cpw, centers, points, idx);
    return p2pb_launch_status();
  }
  hipLaunchKernelGGL(ball_query_kernel<4>, dim3(cdiv(m, 4), b), dim3(256), 0, (hipStream_t)stream, n, m, r2, u,
                     centers, points, idx);
  return p2pb_launch_status();
}

// ------------------------------------------------------------------------------------------------
// grouping: out[b,c,j,k] = feat[b,c,idx[b,j,k]] . One thread per (j,k) slot, looping a chunk of
// channels: the index is read once, the 32x write amplification goes out as lane-consecutive stores.
// ------------------------------------------------------------------------------------------------
template <int CC>
__global__ __launch_bounds__(256) void grouping_kernel(int c, int n, int mu, const float *__restrict__ feat,
                                                       const int *__restrict__ idx, float *__restrict__ out) {
  const int b = blockIdx.z;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= mu) return;
  const int id = idx[(size_t)b * mu + q];
  const int c0 = blockIdx.y * CC, c1 = min(c0 + CC, c);
  for (int l = c0; l < c1; ++l) out[((size_t)b * c + l) * mu + q] = feat[((size_t)b * c + l) * n + id];
}

extern "C" int p2pb_grouping_forward(int b, int c, int n, int m, int u, const float *feat, const int *idx, float *out,
                                     void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || m <= 0 || u <= 0) return P2PB_EINVAL;
  constexpr int CC = 8;
  hipLaunchKernelGGL(grouping_kernel<CC>, dim3(cdiv((long)m * u, 256), cdiv(c, CC), b), dim3(256), 0,
                     (hipStream_t)stream, c, n, m * u, feat, idx, out);
  return p2pb_launch_status();
}

// set-abstraction operand in one pass: out[b, 0:3, j, k] = coords[b, :, idx] - centers[b, :, j] (relative
// neighbour coordinates, models/pvcnn.py:117-118) and out[b, 3:, j, k] = feat[b, :, idx] (:124-126), i.e.
// grouping x2 + subtract + concat of the unfused graph without the two intermediate tensors
template <int CC>
__global__ __launch_bounds__(256) void group_concat_kernel(int c, int n, int m, int u,
                                                           const float *__restrict__ coords,
                                                           const float *__restrict__ centers,
                                                           const float *__restrict__ feat, const int *__restrict__ idx,
                                                           float *__restrict__ out) {
  const int b = blockIdx.z;
  const int mu = m * u;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= mu) return;
  const int id = idx[(size_t)b * mu + q];
  const int j = q / u;
  const int ct = 3 + c;
  const int c0 = blockIdx.y * CC, c1 = min(c0 + CC, ct);
  for (int l = c0; l < c1; ++l) {
    float v;
    if (l < 3) v = coords[((size_t)b * 3 + l) * n + id] - centers[((size_t)b * 3 + l) * m + j];
    else v = feat[((size_t)b * c + (l - 3)) * n + id];
    out[((size_t)b * ct + l) * mu + q] = v;
  }
}

extern "C" int p2pb_group_concat(int b, int c, int n, int m, int u, const float *coords, const float *centers,
                                 const float *feat, const int *idx, float *out, void *stream) {
  if (b <= 0 || c < 0 || n <= 0 || m <= 0 || u <= 0) return P2PB_EINVAL;
  constexpr int CC = 8;
  hipLaunchKernelGGL(group_concat_kernel<CC>, dim3(cdiv((long)m * u, 256), cdiv(3 + c, CC), b), dim3(256), 0,
                     (hipStream_t)stream, c, n, m, u, coords, centers, feat, idx, out);
  return p2pb_launch_status();
}

// ------------------------------------------------------------------------------------------------
// First layer of a set-abstraction MLP applied BEFORE the grouping (inference): a 1x1 convolution is linear, so
//   W [xyz[idx] - centre ; f[idx]] + bias  =  Z[:, idx] - Cx[:, centre],   Z = W [xyz ; f] + bias (N points),
//                                                                          Cx = W_xyz centre (M centres),
// i.e. the GEMM runs on the N points instead of the M*U grouped positions (4x .. 32x fewer) and the
// (3+C)-channel grouped tensor of models/pvcnn.py:117-126 is never built. This kernel gathers the C1-channel
// result, subtracts the centre term and emits the {sum, sum of squares} partials of the GroupNorm that follows
// (one slot per half-wave = 32 positions; reduced in fixed order by gn_affine_kernel).
//   z f32[b,c,n], cx f32[b,c,m] (or NULL), idx i32[b,m,u] -> out f32[b,c,m*u], stats f32[b, nslots, c, 2]
// ------------------------------------------------------------------------------------------------
// [b, c, n] -> [b, n, c] (32 x 32 LDS tiles)
__global__ __launch_bounds__(256) void nb_transpose_kernel(int c, int n, const float *__restrict__ in,
                                                           float *__restrict__ out) {
  __shared__ float t[32][33];
  const int b = blockIdx.z, n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float *src = in + (size_t)b * c * n;
  float *dst = out + (size_t)b * c * n;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cc = c0 + ty + 8 * k, nn = n0 + tx;
    t[ty + 8 * k][tx] = (cc < c && nn < n) ? src[(size_t)cc * n + nn] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int nn = n0 + ty + 8 * k, cc = c0 + tx;
    if (cc < c && nn < n) dst[(size_t)nn * c + cc] = t[tx][ty + 8 * k];
  }
}

// Gathers run on POINT-MAJOR copies (zt f32[b,n,c], cxt f32[b,m,c]): lane = channel reads one contiguous row per
// neighbour (a channel-major gather touches 4 bytes per 64-byte line), 64 positions per workgroup go through an LDS
// transpose so the channel-major output is written in 256-byte runs; a wave then owns whole channel rows of the
// tile, and the statistics are two half-wave sums per row (slot = 32 positions).
__global__ __launch_bounds__(256) void group_sub_kernel(int c, int n, int m, int u, int nslots,
                                                        const float *__restrict__ zt, const float *__restrict__ cxt,
                                                        const int *__restrict__ idx, float *__restrict__ out,
                                                        float *__restrict__ stats) {
  __shared__ float tile[64][65];  // [channel][position]
  const int b = blockIdx.z, p0 = blockIdx.x * 64, t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int mu = m * u;
  // the wave's 16 neighbour indices (and centre rows) are fetched once, by lanes 0..15
  const int qmine = p0 + wave * 16 + (lane & 15);
  const int id_mine = qmine < mu ? idx[(size_t)b * mu + qmine] : -1;
  // narrow layers (c <= 32) gather two positions per step, one per half-wave
  const int cpl = c <= 32 ? 32 : 64, pps = 64 / cpl;
  const int sub = lane / cpl, chl = lane % cpl;
  {
    const int c0 = blockIdx.y * cpl;  // one channel chunk per workgroup (more workgroups on the small levels)
    const int ch = c0 + chl;
#pragma unroll 4
    for (int it = 0; it < 16 / pps; ++it) {
      const int pl = wave * 16 + it * pps + sub;
      const int id = __shfl(id_mine, it * pps + sub);
      float v = 0.0f;
      if (id >= 0 && ch < c) {
        v = zt[((size_t)b * n + id) * c + ch];
        if (cxt) v -= cxt[((size_t)b * m + (p0 + pl) / u) * c + ch];
      }
      tile[chl][pl] = v;
    }
    __syncthreads();
    const int pt = lane;
    // a wave owns rows wave, wave + 4, ... (four waves write four adjacent output rows at a time: measured 15 % faster
    // than consecutive rows per wave)
    const int per = cpl / 4;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int cr = wave + 4 * k;
      const float v = tile[cr & 63][pt];  // zero outside the tensor: statistics unaffected
      if (k < per && c0 + cr < c) {
        if (out && p0 + pt < mu) out[((size_t)b * c + c0 + cr) * mu + p0 + pt] = v;  // (out == NULL: statistics only)
        const float s1 = halfwave_sum_to_last(v), s2 = halfwave_sum_to_last(v * v);
        if ((lane & 31) == 31) {
          float *p = stats + (((size_t)b * nslots + blockIdx.x * 2 + (lane >> 5)) * c + c0 + cr) * 2;
          p[0] = s1;
          p[1] = s2;
        }
      }
    }
    __syncthreads();
  }
}

// STATISTICS ONLY (the two-layer set abstractions: the consumer, pw_wide_kernel<GATHER>, gathers the grouped operand itself and
// only the GroupNorm partials of the grouped tensor are needed first). Round 5: this used to be group_sub_kernel with out == NULL --
// the LDS transpose that exists for the output's sake and ten DPP steps per channel row, 114 us for the first level of the bench;
// here a lane IS a channel: a wave walks its 128 positions, every lane adds its own {v, v * v} (v = z[idx] - cx, group_sub's
// arithmetic) in position order, and the partial of (sample, slot = wave, channel) is one store -- no LDS, no cross-lane step
// (two positions per step for c <= 32: the half-waves' sums are added once at the end). Slots of 128 positions: a quarter of the
// partials gn_affine has to reduce afterwards.
#define GST_PW 128
template <int CPL>
__global__ __launch_bounds__(256) void group_stats_kernel(int c, int n, int m, int u, int nslots, const float *__restrict__ zt,
                                                          const float *__restrict__ cxt, const int *__restrict__ idx,
                                                          float *__restrict__ stats) {
  constexpr int PPS = 64 / CPL;  // positions per step
  const int b = blockIdx.z, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = blockIdx.x * 4 + wave;
  if (slot >= nslots) return;
  const int mu = m * u, p0 = slot * GST_PW;
  const int sub = lane / CPL, ch = blockIdx.y * CPL + lane % CPL;
  // the wave's 128 neighbour indices: two per lane, handed round by lane broadcasts
  const int ia = p0 + lane < mu ? idx[(size_t)b * mu + p0 + lane] : -1;
  const int ib = p0 + 64 + lane < mu ? idx[(size_t)b * mu + p0 + 64 + lane] : -1;
  const float *zb = zt + (size_t)b * n * c + ch;
  const float *cb = cxt ? cxt + (size_t)b * m * c + ch : nullptr;
  float s1 = 0.0f, s2 = 0.0f;
  const bool chok = ch < c;
#pragma unroll 8
  for (int it = 0; it < GST_PW / PPS; ++it) {
    const int pl = it * PPS + sub;
    const int id = pl < 64 ? __shfl(ia, pl) : __shfl(ib, pl - 64);
    if (id >= 0 && chok) {
      float v = zb[(size_t)id * c];
      if (cb) v -= cb[(size_t)((p0 + pl) / u) * c];
      s1 += v;
      s2 += v * v;
    }
  }
  if (PPS == 2) {
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
  }
  if (chok && sub == 0) {
    float *q = stats + (((size_t)b * nslots + slot) * c + ch) * 2;
    q[0] = s1;
    q[1] = s2;
  }
}

extern "C" int p2pb_group_sub_stats_slots(int m, int u) { return (int)(((long)m * u + GST_PW - 1) / GST_PW); }
// zt f32[b,n,c], cxt f32[b,m,c] | NULL (point-major), idx i32[b,m,u] -> stats_part f32[b, p2pb_group_sub_stats_slots(m,u), c, 2]
extern "C" int p2pb_group_sub_stats(int b, int c, int n, int m, int u, const float *zt, const float *cxt, const int *idx,
                                    float *stats_part, void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || m <= 0 || u <= 0 || !zt || !idx || !stats_part || (long)m * u > 0x7fffffffL) return P2PB_EINVAL;
  const int nslots = p2pb_group_sub_stats_slots(m, u);
  hipStream_t s = (hipStream_t)stream;
  if (c <= 32)
    hipLaunchKernelGGL(group_stats_kernel<32>, dim3(cdiv(nslots, 4), cdiv(c, 32), b), dim3(256), 0, s, c, n, m, u, nslots, zt, cxt, idx,
                       stats_part);
  else
    hipLaunchKernelGGL(group_stats_kernel<64>, dim3(cdiv(nslots, 4), cdiv(c, 64), b), dim3(256), 0, s, c, n, m, u, nslots, zt, cxt, idx,
                       stats_part);
  return p2pb_launch_status();
}

extern "C" size_t p2pb_group_sub_stats_floats(int b, int c, int m, int u) {
  return (size_t)b * (((size_t)m * u + 63) / 64 * 2) * c * 2;
}

// ws: f32[b*(n+m)*c] scratch for the point-major copies
extern "C" int p2pb_group_sub(int b, int c, int n, int m, int u, const float *z, const float *cx, const int *idx,
                              float *out, float *stats_part, float *ws, void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || m <= 0 || u <= 0 || !stats_part) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const float *zt = z, *cxt = cx;  // ws == NULL: z f32[b,n,c] and cx f32[b,m,c] are point-major already
  if (ws) {
    float *zw = ws, *cw = ws + (size_t)b * n * c;
    hipLaunchKernelGGL(nb_transpose_kernel, dim3(cdiv(n, 32), cdiv(c, 32), b), dim3(256), 0, s, c, n, z, zw);
    if (cx) hipLaunchKernelGGL(nb_transpose_kernel, dim3(cdiv(m, 32), cdiv(c, 32), b), dim3(256), 0, s, c, m, cx, cw);
    zt = zw;
    cxt = cx ? cw : nullptr;
  }
  const int nblk = (int)(((long)m * u + 63) / 64);
  hipLaunchKernelGGL(group_sub_kernel, dim3(nblk, cdiv(c, c <= 32 ? 32 : 64), b), dim3(256), 0, s, c, n, m, u, nblk * 2,
                     zt, cxt, idx, out, stats_part);
  return p2pb_launch_status();
}

template <int CC>
__global__ __launch_bounds__(256) void grouping_grad_kernel(int c, int n, int mu, const float *__restrict__ gy, size_t gyp,
                                                            const int *__restrict__ idx, float *__restrict__ gx) {
  const int b = blockIdx.z;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= mu) return;
  const int id = idx[(size_t)b * mu + q];
  const int c0 = blockIdx.y * CC, c1 = min(c0 + CC, c);
  for (int l = c0; l < c1; ++l) atomicAdd(gx + ((size_t)b * c + l) * n + id, gy[(size_t)b * gyp + (size_t)l * mu + q]);
}

// rows of CH channels in LDS (common.h "scatter-add backward passes")
template <int CH>
__global__ __launch_bounds__(SCAT_THREADS) void grouping_grad_lds_kernel(int c, int n, int Lp, int mu, const float *__restrict__ gy,
                                                                        size_t gyp, const int *__restrict__ idx, float *__restrict__ gx) {
  extern __shared__ float rows[];
  const int b = blockIdx.y, c0 = blockIdx.x * CH, nch = min(CH, c - c0);
  scat_zero(rows, CH * Lp);
  const int *ib = idx + (size_t)b * mu;
  const float *g0 = gy + (size_t)b * gyp + (size_t)c0 * mu;  // (gyp: floats between two samples of gy, >= c * mu)
  for (int q = threadIdx.x; q < mu; q += blockDim.x) {
    const int id = ib[q];
#pragma unroll
    for (int j = 0; j < CH; ++j)
      if (j < nch) atomicAdd(rows + j * Lp + id, g0[(size_t)j * mu + q]);
  }
  scat_store(rows, n, Lp, nch, gx + ((size_t)b * c + c0) * n);
}

template <int CH>
static int grouping_grad_lds_launch(int b, int c, int n, int mu, const float *gy, size_t gyp, const int *idx, float *gx, hipStream_t s) {
  const int Lp = (n + 3) & ~3;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void *)grouping_grad_lds_kernel<CH>, hipFuncAttributeMaxDynamicSharedMemorySize, SCAT_LDS_MAX);
    once = true;
  }
  hipLaunchKernelGGL(grouping_grad_lds_kernel<CH>, dim3(cdiv(c, CH), b), dim3(scat_threads()), sizeof(float) * (size_t)CH * Lp, s, c, n,
                     Lp, mu, gy, gyp, idx, gx);
  return p2pb_launch_status();
}

// grad_y: sample b starts at grad_y + b * gy_pitch floats (gy_pitch >= c*m*u; a channel slice of a wider tensor, e.g. one part of
// a concatenation's gradient, is read in place instead of through a contiguous copy)
extern "C" int p2pb_grouping_backward_pitched(int b, int c, int n, int m, int u, const float *grad_y, long gy_pitch, const int *idx,
                                              float *grad_x, void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || m <= 0 || u <= 0 || gy_pitch < (long)c * m * u) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const size_t gyp = (size_t)gy_pitch;
  switch (scat_rows(n, c, 8)) {
    case 0: break;
    case 1: return grouping_grad_lds_launch<1>(b, c, n, m * u, grad_y, gyp, idx, grad_x, s);
    case 2: case 3: return grouping_grad_lds_launch<2>(b, c, n, m * u, grad_y, gyp, idx, grad_x, s);
    case 8: return grouping_grad_lds_launch<8>(b, c, n, m * u, grad_y, gyp, idx, grad_x, s);
    default: return grouping_grad_lds_launch<4>(b, c, n, m * u, grad_y, gyp, idx, grad_x, s);
  }
  if (p2pb_deterministic()) return P2PB_EINVAL;  // (rows beyond the LDS: only the global-atomic kernel is left)
  int e = p2pb_zero_async(grad_x, sizeof(float) * (size_t)b * c * n, s);
  if (e != 0) return e;
  constexpr int CC = 8;
  hipLaunchKernelGGL(grouping_grad_kernel<CC>, dim3(cdiv((long)m * u, 256), cdiv(c, CC), b), dim3(256), 0, s, c, n,
                     m * u, grad_y, gyp, idx, grad_x);
  return p2pb_launch_status();
}
extern "C" int p2pb_grouping_backward(int b, int c, int n, int m, int u, const float *grad_y, const int *idx,
                                      float *grad_x, void *stream) {
  return p2pb_grouping_backward_pitched(b, c, n, m, u, grad_y, (long)c * m * u, idx, grad_x, stream);
}

// ------------------------------------------------------------------------------------------------
// gather: out[b,c,j] = feat[b,c,idx[b,j]]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_kernel(int c, int n, int m, const float *__restrict__ feat,
                                                     const int *__restrict__ idx, float *__restrict__ out) {
  const int b = blockIdx.z, l = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  out[((size_t)b * c + l) * m + j] = feat[((size_t)b * c + l) * n + idx[(size_t)b * m + j]];
}

extern "C" int p2pb_gather_features_forward(int b, int c, int n, int m, const float *feat, const int *idx, float *out,
                                            void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || m <= 0) return P2PB_EINVAL;
  hipLaunchKernelGGL(gather_kernel, dim3(cdiv(m, 256), c, b), dim3(256), 0, (hipStream_t)stream, c, n, m, feat, idx,
                     out);
  return p2pb_launch_status();
}

__global__ __launch_bounds__(256) void gather_grad_kernel(int c, int n, int m, const float *__restrict__ gy,
                                                          const int *__restrict__ idx, float *__restrict__ gx) {
  const int b = blockIdx.z, l = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  atomicAdd(gx + ((size_t)b * c + l) * n + idx[(size_t)b * m + j], gy[((size_t)b * c + l) * m + j]);
}

extern "C" int p2pb_gather_features_backward(int b, int c, int n, int m, const float *grad_y, const int *idx,
                                             float *grad_x, void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || m <= 0) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int e = p2pb_zero_async(grad_x, sizeof(float) * (size_t)b * c * n, s);
  if (e != 0) return e;
  hipLaunchKernelGGL(gather_grad_kernel, dim3(cdiv(m, 256), c, b), dim3(256), 0, s, c, n, m, grad_y, idx, grad_x);
  return p2pb_launch_status();
}

// ------------------------------------------------------------------------------------------------
// three nearest centres + inverse-squared-distance weights: one thread per point, the centre
// coordinates are staged through LDS in tiles and read back as wave-wide broadcasts.
// The reference keeps its three bests as doubles initialised to 1e40 (never reached by an fp32
// distance); +inf in fp32 takes exactly the same branches and clamps to the same 1e10f.
// ------------------------------------------------------------------------------------------------
#define NN_TILE 2048
__global__ __launch_bounds__(256) void three_nn_kernel(int n, int m, const float *__restrict__ points,
                                                       const float *__restrict__ centers, float *__restrict__ weights,
                                                       int *__restrict__ indices) {
  __shared__ float sc[3][NN_TILE];
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const float *p = points + (size_t)b * 3 * n;
  const float *ce = centers + (size_t)b * 3 * m;
  const bool ok = j < n;
  const float ux = ok ? p[j] : 0.0f, uy = ok ? p[j + n] : 0.0f, uz = ok ? p[j + 2 * n] : 0.0f;
  float best0 = INFINITY, best1 = INFINITY, best2 = INFINITY;
  int i0 = 0, i1 = 0, i2 = 0;
  for (int k0 = 0; k0 < m; k0 += NN_TILE) {
    const int kn = min(NN_TILE, m - k0);
    __syncthreads();
    for (int k = threadIdx.x; k < kn; k += 256) {
      sc[0][k] = ce[k0 + k];
      sc[1][k] = ce[k0 + k + m];
      sc[2][k] = ce[k0 + k + 2 * m];
    }
    __syncthreads();
    auto insert = [&](float d, int k) {
      if (d < best2) {
        best2 = d;
        i2 = k;
        if (d < best1) {
          best2 = best1;
          i2 = i1;
          best1 = d;
          i1 = k;
          if (d < best0) {
            best1 = best0;
            i1 = i0;
            best0 = d;
            i0 = k;
          }
        }
      }
    };
    for (int k = 0; k < kn; ++k) insert(sqdist3(ux - sc[0][k], uy - sc[1][k], uz - sc[2][k]), k0 + k);
  }
  if (!ok) return;
  best0 = fmaxf(fminf(1e10f, best0), 1e-10f);
  best1 = fmaxf(fminf(1e10f, best1), 1e-10f);
  best2 = fmaxf(fminf(1e10f, best2), 1e-10f);
  const float d0d1 = best0 * best1, d0d2 = best0 * best2, d1d2 = best1 * best2;
  const float inv = __fdiv_rn(1.0f, d0d1 + d0d2 + d1d2);
  float *w = weights + (size_t)b * 3 * n;
  int *id = indices + (size_t)b * 3 * n;
  w[j] = d1d2 * inv;
  id[j] = i0;
  w[j + n] = d0d2 * inv;
  id[j + n] = i1;
  w[j + 2 * n] = d0d1 * inv;
  id[j + 2 * n] = i2;
}

// ------------------------------------------------------------------------------------------------
// The same search through a uniform grid over the centres (exact): brute force evaluates n x m pairs (537 M per call
// at the first level; ~190 us of pure instruction issue, more beside the main stream), the grid ~50 per point.
//   nn_cells_build : one workgroup per cloud. Cubic cells of edge h = (largest bounding-box extent of the centres) /
//                    NNC_G; count (LDS atomics) -> exclusive scan -> fill -> every cell's short id list sorted
//                    ascending. cell_start i32[b][G^3 + 1], cell_ids i32[b][m], box f32[b][4] = (min x, y, z, h).
//   three_nn_cells : one thread per point, the cloud's records + cell table in LDS: visit the cells within Chebyshev
//                    radius rho of the point's (clamped) cell, rho = 1 (the 3x3x3 block), 2, ... (one more shell
//                    each); every centre outside is farther than rho * h, so the search stops as
//                    soon as three are held and the third distance is <= (rho h)^2. Candidates arrive out of index
//                    order, so ties are broken explicitly (smaller index first) -- what "first strict minimum in
//                    ascending index" gives the brute-force kernel. Same distances (sqdist3), same weights.
// ------------------------------------------------------------------------------------------------
#define NNC_G 16
#define NNC_CELLS (NNC_G * NNC_G * NNC_G)

__global__ __launch_bounds__(1024) void nn_cells_build_kernel(int m, const float *__restrict__ centers,
                                                              int *__restrict__ cell_start, int *__restrict__ cell_ids,
                                                              float4 *__restrict__ cell_rec, float *__restrict__ box) {
  __shared__ int cnt[NNC_CELLS];
  __shared__ int part[1024];
  __shared__ float red[6][16];
  __shared__ float sbox[4];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float *ce = centers + (size_t)b * 3 * m;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = t; k < m; k += 1024)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = ce[k + a * m];
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
  for (int i = t; i < NNC_CELLS; i += 1024) cnt[i] = 0;
  __syncthreads();
  if (t == 0) {
    float mn[3], ext = 0.0f;
    for (int a = 0; a < 3; ++a) {
      float l = INFINITY, h = -INFINITY;
      for (int w = 0; w < 16; ++w) {
        l = fminf(l, red[a][w]);
        h = fmaxf(h, red[3 + a][w]);
      }
      mn[a] = l;
      ext = fmaxf(ext, h - l);
    }
    const float hcell = fmaxf(ext, 1e-12f) / NNC_G;
    for (int a = 0; a < 3; ++a) {
      sbox[a] = mn[a];
      box[(size_t)b * 4 + a] = mn[a];
    }
    sbox[3] = hcell;
    box[(size_t)b * 4 + 3] = hcell;
  }
  __syncthreads();
  const float inv = 1.0f / sbox[3];
  auto cell_of = [&](int k) {
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = min(max((int)floorf((ce[k + a * m] - sbox[a]) * inv), 0), NNC_G - 1);
    return (c[2] * NNC_G + c[1]) * NNC_G + c[0];
  };
  for (int k = t; k < m; k += 1024) atomicAdd(&cnt[cell_of(k)], 1);
  __syncthreads();
  // exclusive scan of the 4096 counts: thread t owns cells 4t .. 4t+3
  int c4[4], tot = 0;
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
  int *cs = cell_start + (size_t)b * (NNC_CELLS + 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    cs[4 * t + i] = run;
    cnt[4 * t + i] = run;  // becomes the fill cursor
    run += c4[i];
  }
  if (t == 1023) cs[NNC_CELLS] = run;
  __syncthreads();
  int *ids = cell_ids + (size_t)b * m;
  for (int k = t; k < m; k += 1024) ids[atomicAdd(&cnt[cell_of(k)], 1)] = k;
  __syncthreads();
  // ascending ids inside every cell (insertion sort of a short list by the cell's owner thread)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int s0 = cs[4 * t + i], s1 = s0 + c4[i];
    for (int x = s0 + 1; x < s1; ++x) {
      const int v = ids[x];
      int y = x - 1;
      while (y >= s0 && ids[y] > v) {
        ids[y + 1] = ids[y];
        --y;
      }
      ids[y + 1] = v;
    }
    // the cell's centres as (x, y, z, id) records, contiguous: what the search streams through LDS
    float4 *rec = cell_rec + (size_t)b * m;
    for (int x = s0; x < s1; ++x) {
      const int k = ids[x];
      rec[x] = make_float4(ce[k], ce[k + m], ce[k + 2 * m], __int_as_float(k));
    }
  }
}

template <bool LDS_REC>  // false (m > NNC_MAX_M, PVDL's 12500-centre level): the records stay in global memory (L2-resident:
                         // 16 m bytes per cloud), only the cell table goes to LDS -- many workgroups per CU hide the L2 trips
__global__ __launch_bounds__(256) void three_nn_cells_kernel(int n, int m, const float *__restrict__ points,
                                                             const int *__restrict__ cell_start,
                                                             const float4 *__restrict__ cell_rec,
                                                             const float *__restrict__ box, float *__restrict__ weights,
                                                             int *__restrict__ indices) {
  // the cloud's cell table and sorted centre records, shared by the workgroup's 256 points (all of one cloud):
  // every candidate is then one 16-byte LDS read at an address known up front (no dependent global loads)
  extern __shared__ float4 nnc_lds[];
  const int b = blockIdx.y;
  const float4 *rec = LDS_REC ? (const float4 *)nnc_lds : cell_rec + (size_t)b * m;
  int *cs = (int *)(nnc_lds + (LDS_REC ? m : 0));
  if (LDS_REC)
    for (int i = threadIdx.x; i < m; i += 256) nnc_lds[i] = cell_rec[(size_t)b * m + i];
  for (int i = threadIdx.x; i <= NNC_CELLS; i += 256) cs[i] = cell_start[(size_t)b * (NNC_CELLS + 1) + i];
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const float *p = points + (size_t)b * 3 * n;
  const float ux = p[j], uy = p[j + n], uz = p[j + 2 * n];
  const float h = box[(size_t)b * 4 + 3], inv = 1.0f / h;
  const int cx = min(max((int)floorf((ux - box[(size_t)b * 4]) * inv), 0), NNC_G - 1);
  const int cy = min(max((int)floorf((uy - box[(size_t)b * 4 + 1]) * inv), 0), NNC_G - 1);
  const int cz = min(max((int)floorf((uz - box[(size_t)b * 4 + 2]) * inv), 0), NNC_G - 1);
  float best0 = INFINITY, best1 = INFINITY, best2 = INFINITY;
  int i0 = 0, i1 = 0, i2 = 0;
  // (distance, index) lexicographic: what ascending-index brute force with strict '<' selects
  auto before = [](float d, int k, float bd, int bi) { return d < bd || (d == bd && k < bi); };
  // (always_inline: as an out-of-line call the by-reference captures -- the three bests -- would live in scratch)
  auto consider = [&](const float4 r) __attribute__((always_inline)) {
    const int k = __float_as_int(r.w);
    const float d = sqdist3(ux - r.x, uy - r.y, uz - r.z);
    if (d <= best2) {  // (one compare rejects almost every candidate)
      // sorted insertion as selects: written with nested ifs + shifts the compiler turned the three bests into a
      // dynamically indexed scratch array (8x slower)
      const bool c2 = before(d, k, best2, i2), c1 = before(d, k, best1, i1), c0 = before(d, k, best0, i0);
      const float n2 = c1 ? best1 : (c2 ? d : best2), n1 = c0 ? best0 : (c1 ? d : best1), n0 = c0 ? d : best0;
      const int m2 = c1 ? i1 : (c2 ? k : i2), m1 = c0 ? i0 : (c1 ? k : i1), m0 = c0 ? k : i0;
      best2 = n2;
      best1 = n1;
      best0 = n0;
      i2 = m2;
      i1 = m1;
      i0 = m0;
    }
  };
  auto visit_run = [&](int q0, int q1) __attribute__((always_inline)) {  // consecutive cells = consecutive records
    for (int q = q0; q < q1; q += 4) {  // four independent LDS reads in flight, candidates taken in order
      float4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = rec[min(q + u, q1 - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (q + u < q1) consider(r[u]);
    }
  };
  for (int rho = 1; rho < NNC_G; ++rho) {
    const int z0 = max(cz - rho, 0), z1 = min(cz + rho, NNC_G - 1);
    const int y0 = max(cy - rho, 0), y1 = min(cy + rho, NNC_G - 1);
    const int x0 = max(cx - rho, 0), x1 = min(cx + rho, NNC_G - 1);
    for (int z = z0; z <= z1; ++z)
      for (int y = y0; y <= y1; ++y) {
        const int row = (z * NNC_G + y) * NNC_G;  // the cells of one x-row are consecutive
        // rho = 1: the whole 3x3x3 block (ring 0 alone can never end the search); later: the shell of Chebyshev
        // radius rho -- full rows on its z / y faces, the two end cells on the interior rows
        const bool face = rho == 1 || z == cz - rho || z == cz + rho || y == cy - rho || y == cy + rho;
        if (face) {
          visit_run(cs[row + x0], cs[row + x1 + 1]);
        } else {
          if (cx - rho >= 0) visit_run(cs[row + cx - rho], cs[row + cx - rho + 1]);
          if (cx + rho <= NNC_G - 1) visit_run(cs[row + cx + rho], cs[row + cx + rho + 1]);
        }
      }
    // everything not visited yet is farther than rho * h (in at least one axis the cell index differs by > rho)
    const float bound = (float)rho * h;
    if (best2 <= bound * bound * 0.9999f && best2 < INFINITY) break;  // (margin: cell assignment rounds)
    if (x0 == 0 && y0 == 0 && z0 == 0 && x1 == NNC_G - 1 && y1 == NNC_G - 1 && z1 == NNC_G - 1) break;  // all cells seen
  }
  best0 = fmaxf(fminf(1e10f, best0), 1e-10f);
  best1 = fmaxf(fminf(1e10f, best1), 1e-10f);
  best2 = fmaxf(fminf(1e10f, best2), 1e-10f);
  const float d0d1 = best0 * best1, d0d2 = best0 * best2, d1d2 = best1 * best2;
  const float invw = __fdiv_rn(1.0f, d0d1 + d0d2 + d1d2);
  float *w = weights + (size_t)b * 3 * n;
  int *id = indices + (size_t)b * 3 * n;
  w[j] = d1d2 * invw;
  id[j] = i0;
  w[j + n] = d0d2 * invw;
  id[j + n] = i1;
  w[j + 2 * n] = d0d1 * invw;
  id[j + 2 * n] = i2;
}

#define NNC_MAX_M 8192  // records + cell table in LDS: 16 m + 16.4 KB <= 148 KB

extern "C" size_t p2pb_three_nn_cells_ws_bytes(int b, int m) {
  return (size_t)b * m * 16 + ((size_t)b * (NNC_CELLS + 1) + (size_t)b * m) * sizeof(int) + (size_t)b * 4 * sizeof(float);
}

// p2pb_three_nn through a uniform grid over the centres: same idx / w. m >= 3 (records in LDS up to 8192 centres, in L2 above);
// ws: p2pb_three_nn_cells_ws_bytes(b, m) bytes, 16-byte aligned
extern "C" int p2pb_three_nn_cells(int b, int m, int n, const float *points, const float *centers, int *idx, float *w,
                                   void *ws, void *stream) {
  if (b <= 0 || n <= 0 || m < 3 || m > (1 << 24) || !ws || ((uintptr_t)ws & 15)) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float4 *cell_rec = (float4 *)ws;
  int *cell_start = (int *)(cell_rec + (size_t)b * m);
  int *cell_ids = cell_start + (size_t)b * (NNC_CELLS + 1);
  float *box = (float *)(cell_ids + (size_t)b * m);
  const bool in_lds = m <= NNC_MAX_M;
  const size_t lds = (in_lds ? (size_t)m * 16 : 0) + (NNC_CELLS + 1) * sizeof(int) + 16;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void *)three_nn_cells_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  hipLaunchKernelGGL(nn_cells_build_kernel, dim3(b), dim3(1024), 0, s, m, centers, cell_start, cell_ids, cell_rec, box);
  if (in_lds)
    hipLaunchKernelGGL(three_nn_cells_kernel<true>, dim3(cdiv(n, 256), b), dim3(256), lds, s, n, m, points, cell_start, cell_rec,
                       box, w, idx);
  else
    hipLaunchKernelGGL(three_nn_cells_kernel<false>, dim3(cdiv(n, 256), b), dim3(256), lds, s, n, m, points, cell_start, cell_rec,
                       box, w, idx);
  return p2pb_launch_status();
}

template <int CC>
__global__ __launch_bounds__(256) void three_interp_kernel(int c, int m, int n, const float *__restrict__ cfeat,
                                                           const int *__restrict__ indices,
                                                           const float *__restrict__ weights,
                                                           float *__restrict__ out) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int *id = indices + (size_t)b * 3 * n;
  const float *w = weights + (size_t)b * 3 * n;
  const int a0 = id[j], a1 = id[j + n], a2 = id[j + 2 * n];
  const float w0 = w[j], w1 = w[j + n], w2 = w[j + 2 * n];
  const int c0 = blockIdx.y * CC, c1 = min(c0 + CC, c);
  for (int l = c0; l < c1; ++l) {
    const float *f = cfeat + ((size_t)b * c + l) * m;
    out[((size_t)b * c + l) * n + j] = __fmaf_rn(f[a2], w2, __fmaf_rn(f[a1], w1, f[a0] * w0));
  }
}

extern "C" int p2pb_three_nn_interpolate_forward(int b, int c, int m, int n, const float *points,
                                                 const float *centers, const float *cfeat, int *idx, float *w,
                                                 float *out, void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || m <= 0) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(three_nn_kernel, dim3(cdiv(n, 256), b), dim3(256), 0, s, n, m, points, centers, w, idx);
  constexpr int CC = 16;
  hipLaunchKernelGGL(three_interp_kernel<CC>, dim3(cdiv(n, 256), cdiv(c, CC), b), dim3(256), 0, s, c, m, n, cfeat,
                     idx, w, out);
  return p2pb_launch_status();
}

// the two halves of the op as separate entry points: the search depends on coordinates only, so the
// sampler runs it on a side stream (geometry pipeline) while the feature path is still busy
extern "C" int p2pb_three_nn(int b, int m, int n, const float *points, const float *centers, int *idx, float *w,
                             void *stream) {
  if (b <= 0 || n <= 0 || m <= 0) return P2PB_EINVAL;
  hipLaunchKernelGGL(three_nn_kernel, dim3(cdiv(n, 256), b), dim3(256), 0, (hipStream_t)stream, n, m, points, centers,
                     w, idx);
  return p2pb_launch_status();
}

extern "C" int p2pb_three_interpolate(int b, int c, int m, int n, const float *cfeat, const int *idx, const float *w,
                                      float *out, void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || m <= 0) return P2PB_EINVAL;
  constexpr int CC = 16;
  hipLaunchKernelGGL(three_interp_kernel<CC>, dim3(cdiv(n, 256), cdiv(c, CC), b), dim3(256), 0, (hipStream_t)stream, c,
                     m, n, cfeat, idx, w, out);
  return p2pb_launch_status();
}

// Feature propagation with the first 1x1 convolution applied before the interpolation (inference): interpolation and
// convolution are both linear, so W [interp(g) ; skip] + bias = interp(W_g g) + (W_s skip + bias): the GEMM on the
// interpolated channels runs on the m coarse points instead of the n fine ones and the concatenated tensor of
// models/pvcnn.py:457-461 is never built.  out[b,c,j] = sum_k w_k * cz[b,c,idx_k] + add[b,c,j] (+ bias[c]), plus the
// {sum, sum of squares} partials of the GroupNorm that follows (one slot per half-wave, as group_sub_kernel).
__global__ __launch_bounds__(256) void three_interp_add_kernel(int c, int m, int n, int nslots,
                                                               const float *__restrict__ czt,
                                                               const int *__restrict__ indices,
                                                               const float *__restrict__ weights,
                                                               const float *__restrict__ add,
                                                               const float *__restrict__ bias, float *__restrict__ out,
                                                               float *__restrict__ stats) {
  __shared__ float tile[64][65];  // [channel][position]
  __shared__ int sid[64][3];
  __shared__ float sw[64][3];
  const int b = blockIdx.z, p0 = blockIdx.x * 64, t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  if (t < 64) {
    const int j = min(p0 + t, n - 1);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      sid[t][k] = indices[((size_t)b * 3 + k) * n + j];
      sw[t][k] = weights[((size_t)b * 3 + k) * n + j];
    }
  }
  __syncthreads();
  {
    const int c0 = blockIdx.y * 64;  // one 64-channel chunk per workgroup
    const int ch = c0 + lane;
    if (ch < c) {
      const float *f = czt + (size_t)b * m * c + ch;  // point-major coarse features: one contiguous row per neighbour
#pragma unroll 4
      for (int pl = wave * 16; pl < wave * 16 + 16; ++pl)
        tile[lane][pl] = __fmaf_rn(f[(size_t)sid[pl][2] * c], sw[pl][2],
                                   __fmaf_rn(f[(size_t)sid[pl][1] * c], sw[pl][1], f[(size_t)sid[pl][0] * c] * sw[pl][0]));
    }
    __syncthreads();
    const int pt = lane;
    const bool pok = p0 + pt < n;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int cr = wave + 4 * k;  // (see group_sub_kernel)
      if (c0 + cr < c) {
        float v = 0.0f;
        if (pok) {
          const size_t o = ((size_t)b * c + c0 + cr) * n + p0 + pt;
          v = tile[cr][pt];
          if (add) v += add[o];
          if (bias) v += bias[c0 + cr];
          out[o] = v;
        }
        const float s1 = halfwave_sum_to_last(v), s2 = halfwave_sum_to_last(v * v);
        if ((lane & 31) == 31) {
          float *p = stats + (((size_t)b * nslots + blockIdx.x * 2 + (lane >> 5)) * c + c0 + cr) * 2;
          p[0] = s1;
          p[1] = s2;
        }
      }
    }
    __syncthreads();
  }
}

// stats_part: f32[p2pb_group_sub_stats_floats(b, c, n, 1)]; ws: f32[b*m*c] scratch (point-major copy of cz)
extern "C" int p2pb_three_interpolate_add(int b, int c, int m, int n, const float *cz, const int *idx, const float *w,
                                          const float *add, const float *bias, float *out, float *stats_part,
                                          float *ws, void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || m <= 0 || !stats_part) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const float *czt = cz;  // ws == NULL: cz f32[b,m,c] is point-major already
  if (ws) {
    hipLaunchKernelGGL(nb_transpose_kernel, dim3(cdiv(m, 32), cdiv(c, 32), b), dim3(256), 0, s, c, m, cz, ws);
    czt = ws;
  }
  const int nblk = (n + 63) / 64;
  hipLaunchKernelGGL(three_interp_add_kernel, dim3(nblk, cdiv(c, 64), b), dim3(256), 0, s, c, m, n, nblk * 2, czt, idx, w,
                     add, bias, out, stats_part);
  return p2pb_launch_status();
}

template <int CC>
__global__ __launch_bounds__(256) void three_interp_grad_kernel(int c, int n, int m, const float *__restrict__ gy, size_t gyp,
                                                                const int *__restrict__ indices,
                                                                const float *__restrict__ weights,
                                                                float *__restrict__ gx) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int *id = indices + (size_t)b * 3 * n;
  const float *w = weights + (size_t)b * 3 * n;
  const int a0 = id[j], a1 = id[j + n], a2 = id[j + 2 * n];
  const float w0 = w[j], w1 = w[j + n], w2 = w[j + 2 * n];
  const int c0 = blockIdx.y * CC, c1 = min(c0 + CC, c);
  for (int l = c0; l < c1; ++l) {
    const float g = gy[(size_t)b * gyp + (size_t)l * n + j];
    float *o = gx + ((size_t)b * c + l) * m;
    atomicAdd(o + a0, g * w0);
    atomicAdd(o + a1, g * w1);
    atomicAdd(o + a2, g * w2);
  }
}

template <int CH>
__global__ __launch_bounds__(SCAT_THREADS) void three_interp_grad_lds_kernel(int c, int n, int m, int Lp, const float *__restrict__ gy,
                                                                            size_t gyp, const int *__restrict__ indices,
                                                                            const float *__restrict__ weights, float *__restrict__ gx) {
  extern __shared__ float rows[];
  const int b = blockIdx.y, c0 = blockIdx.x * CH, nch = min(CH, c - c0);
  scat_zero(rows, CH * Lp);
  const int *id = indices + (size_t)b * 3 * n;
  const float *w = weights + (size_t)b * 3 * n;
  const float *g0 = gy + (size_t)b * gyp + (size_t)c0 * n;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const int a0 = id[j], a1 = id[j + n], a2 = id[j + 2 * n];
    const float w0 = w[j], w1 = w[j + n], w2 = w[j + 2 * n];
#pragma unroll
    for (int l = 0; l < CH; ++l) {
      if (l < nch) {
        const float g = g0[(size_t)l * n + j];
        float *o = rows + l * Lp;
        atomicAdd(o + a0, g * w0);
        atomicAdd(o + a1, g * w1);
        atomicAdd(o + a2, g * w2);
      }
    }
  }
  scat_store(rows, m, Lp, nch, gx + ((size_t)b * c + c0) * m);
}

template <int CH>
static int three_interp_grad_lds_launch(int b, int c, int n, int m, const float *gy, size_t gyp, const int *idx, const float *w,
                                        float *gx, hipStream_t s) {
  const int Lp = (m + 3) & ~3;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void *)three_interp_grad_lds_kernel<CH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              SCAT_LDS_MAX);
    once = true;
  }
  hipLaunchKernelGGL(three_interp_grad_lds_kernel<CH>, dim3(cdiv(c, CH), b), dim3(scat_threads()), sizeof(float) * (size_t)CH * Lp, s, c,
                     n, m, Lp, gy, gyp, idx, w, gx);
  return p2pb_launch_status();
}

// (gy_pitch: as p2pb_grouping_backward_pitched, >= c*n)
extern "C" int p2pb_three_nn_interpolate_backward_pitched(int b, int c, int n, int m, const float *grad_y, long gy_pitch,
                                                          const int *idx, const float *w, float *grad_x, void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || m <= 0 || gy_pitch < (long)c * n) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const size_t gyp = (size_t)gy_pitch;
  switch (scat_rows(m, c, 8)) {
    case 0: break;
    case 1: return three_interp_grad_lds_launch<1>(b, c, n, m, grad_y, gyp, idx, w, grad_x, s);
    case 2: case 3: return three_interp_grad_lds_launch<2>(b, c, n, m, grad_y, gyp, idx, w, grad_x, s);
    case 8: return three_interp_grad_lds_launch<8>(b, c, n, m, grad_y, gyp, idx, w, grad_x, s);
    default: return three_interp_grad_lds_launch<4>(b, c, n, m, grad_y, gyp, idx, w, grad_x, s);
  }
  if (p2pb_deterministic()) return P2PB_EINVAL;  // (rows beyond the LDS: only the global-atomic kernel is left)
  int e = p2pb_zero_async(grad_x, sizeof(float) * (size_t)b * c * m, s);
  if (e != 0) return e;
  constexpr int CC = 16;
  hipLaunchKernelGGL(three_interp_grad_kernel<CC>, dim3(cdiv(n, 256), cdiv(c, CC), b), dim3(256), 0, s, c, n, m,
                     grad_y, gyp, idx, w, grad_x);
  return p2pb_launch_status();
}
extern "C" int p2pb_three_nn_interpolate_backward(int b, int c, int n, int m, const float *grad_y, const int *idx,
                                                  const float *w, float *grad_x, void *stream) {
  return p2pb_three_nn_interpolate_backward_pitched(b, c, n, m, grad_y, (long)c * n, idx, w, grad_x, stream);
}
