// voxelize.hip -- point <-> voxel kernels for gfx950.
//   p2pb_voxel_coords                 (models/pvcnn.py:215-228)
//   p2pb_avg_voxelize_forward/backward (PN2/vox_gpu.cu:18,50,92)
//   p2pb_trilinear_devoxelize_*        (PN2/trilinear_devox_gpu.cu:21,123)
// All of these are HBM-bound: the dominant traffic is the dense [C, r^3] grid (written once by
// voxelize including its zeros, read once by devoxelize); the N x (3+C) point tensor is read with
// lane-consecutive (coalesced) accesses, the grid is written lane-consecutive over the voxel index.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// voxel_coords: one workgroup per cloud. Summation order is part of the contract (see oracle):
// 256 lane-strided double partials, then a fixed binary tree.
// ------------------------------------------------------------------------------------------------
// (round 5: since the level-0 preparation runs on the sampler's main stream this kernel is on the critical path. The first form --
//  256 threads, 32 dependent load + add steps per axis, then two more strided passes -- took 44-63 us for 98 KB per cloud. Now 1024
//  threads; the 256 partials are summed by threads 0..255 in the SAME order, their loads issued eight at a time; the maximum and
//  the normalisation pass, which have no order, use every thread.)
__global__ __launch_bounds__(1024) void voxel_coords_kernel(int n, int r, int normalize, float eps,
                                                            const float *__restrict__ coords,
                                                            float *__restrict__ norm, int *__restrict__ vox) {
  __shared__ double part[3][256];
  __shared__ float smax[1024];
  const int t = threadIdx.x;
  const float *c = coords + (size_t)blockIdx.x * 3 * n;
  float *o = norm + (size_t)blockIdx.x * 3 * n;
  int *v = vox + (size_t)blockIdx.x * 3 * n;
  if (t < 256) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int k0 = t; k0 < n; k0 += 8 * 256) {
      float x0[8], x1[8], x2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u * 256;
        const bool ok = k < n;
        x0[u] = ok ? c[k] : 0.0f;
        x1[u] = ok ? c[n + k] : 0.0f;
        x2[u] = ok ? c[2 * n + k] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + u * 256 < n) {
          s0 += (double)x0[u];
          s1 += (double)x1[u];
          s2 += (double)x2[u];
        }
    }
    part[0][t] = s0;
    part[1][t] = s1;
    part[2][t] = s2;
  }
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) {
      part[0][t] += part[0][t + s];
      part[1][t] += part[1][t + s];
      part[2][t] += part[2][t + s];
    }
    __syncthreads();
  }
  const float m0 = (float)(part[0][0] / (double)n);
  const float m1 = (float)(part[1][0] / (double)n);
  const float m2 = (float)(part[2][0] / (double)n);
  float mx = 0.0f;
#pragma unroll 4
  for (int k = t; k < n; k += 1024) {
    float s = sqdist3(c[k] - m0, c[n + k] - m1, c[2 * n + k] - m2);
    mx = s > mx ? s : mx;
  }
  smax[t] = mx;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {  // (a maximum: exact in any order)
    if (t < s) smax[t] = smax[t + s] > smax[t] ? smax[t + s] : smax[t];
    __syncthreads();
  }
  const float denom = sqrtf(smax[0]) * 2.0f + eps;  // sqrtf: correctly rounded (IEEE) under hipcc's default; __fsqrt_rn maps to the 1-ulp native v_sqrt_f32 (found at N = 12500: 40 % of the voxel coordinates one ulp off)
  const float mean[3] = {m0, m1, m2};
  const float rf = (float)r, hi = (float)(r - 1);
  for (int a = 0; a < 3; ++a)
#pragma unroll 4
    for (int k = t; k < n; k += 1024) {
      float x = c[a * n + k] - mean[a];
      if (normalize)
        x = __fdiv_rn(x, denom) + 0.5f;
      else
        x = __fdiv_rn(x + 1.0f, 2.0f);
      x = x * rf;
      x = fminf(fmaxf(x, 0.0f), hi);
      o[a * n + k] = x;
      v[a * n + k] = (int)rintf(x);
    }
}

extern "C" int p2pb_voxel_coords(int b, int n, int r, int normalize, float eps, const float *coords, float *norm,
                                 int *vox, void *stream) {
  if (b <= 0 || n <= 0 || r <= 0) return P2PB_EINVAL;
  hipLaunchKernelGGL(voxel_coords_kernel, dim3(b), dim3(1024), 0, (hipStream_t)stream, n, r, normalize, eps, coords,
                     norm, vox);
  return p2pb_launch_status();
}

// ------------------------------------------------------------------------------------------------
// avg_voxelize forward, deterministic (each voxel sums its points in ascending point index):
//   1 count  : ind[i] = x*r2 + y*r + z ; cnt[ind]++                    (int atomics: order independent)
//   2 scan   : cur[v] = exclusive prefix of cnt; occ[] = compacted list of non-empty voxels, nocc
//   3 fill   : list[cur[v]++] = i                                     (order inside a voxel arbitrary ...)
//   4 sort   : ... one WAVE per non-empty voxel ranks its ids by counting and writes slist[] ascending
//   5 zero   : the dense [C, r^3] grid is streamed out as zeros (16-byte stores; the dominant HBM traffic:
//              a PU-Net patch occupies only ~2-3 % of a 32^3 grid)
//   6 gather : one wave per non-empty voxel, lane = channel: acc += feat[c, p] * (1/cnt) over the voxel's
//              points in ascending order, then out[c, v] = acc
// The point cloud of a patch is a 2-manifold, so a voxel that is occupied holds ~10 points (8192 points
// over ~800 voxels at r = 32): work is organised per occupied voxel, not per voxel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vox_count_kernel(int n, int r, const int *__restrict__ coords,
                                                        int *__restrict__ ind, int *__restrict__ cnt) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int *c = coords + (size_t)b * 3 * n;
  const int v = c[i] * r * r + c[i + n] * r + c[i + 2 * n];
  ind[(size_t)b * n + i] = v;
  atomicAdd(cnt + (size_t)b * r * r * r + v, 1);
}

// block-wide exclusive scan of one int per thread (1024 threads); returns the exclusive prefix, total in *tot
__device__ __forceinline__ int block_exscan_1024(int x, int *wsum, int *tot) {
  const int t = threadIdx.x;
  int inc = x;
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(inc, d);
    if ((t & 63) >= d) inc += y;
  }
  __syncthreads();  // wsum may still be read from a previous call
  if ((t & 63) == 63) wsum[t >> 6] = inc;
  __syncthreads();
  int base = 0, all = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < (t >> 6)) base += wsum[w];
    all += wsum[w];
  }
  *tot = all;
  return base + inc - x;
}

__global__ __launch_bounds__(1024) void vox_scan_kernel(int n, int r3, const int *__restrict__ cnt,
                                                        int *__restrict__ cur, int *__restrict__ occ,
                                                        int *__restrict__ nocc) {
  __shared__ int wsum[16];
  const int t = threadIdx.x;
  const int *c = cnt + (size_t)blockIdx.x * r3;
  int *o = cur + (size_t)blockIdx.x * r3;
  int *oc = occ + (size_t)blockIdx.x * n;
  const int per = (r3 + 1023) / 1024;
  const int beg = t * per, end = min(beg + per, r3);
  int tot;
  if (per == 32 && r3 == 32768) {  // r = 32: a thread's 32 counts in registers (eight 16-byte loads, issued together) -- the two
    // scalar passes of the general form were 64 dependent 4-byte loads at a 128-byte lane stride (55 us per launch)
    typedef int i32x4v __attribute__((ext_vector_type(4)));
    i32x4v x[8];
    const i32x4v *c4 = (const i32x4v *)(c + beg);
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = c4[q];
    int s = 0, k = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s += x[q][i];
        k += x[q][i] > 0;
      }
    int run = block_exscan_1024(s, wsum, &tot);
    int kpos = block_exscan_1024(k, wsum, &tot);
    if (t == 0) nocc[blockIdx.x] = tot;
    i32x4v *o4 = (i32x4v *)(o + beg);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      i32x4v w;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        w[i] = run;
        run += x[q][i];
        if (x[q][i] > 0) oc[kpos++] = beg + 4 * q + i;
      }
      o4[q] = w;
    }
    return;
  }
  int s = 0, k = 0;
  for (int v = beg; v < end; ++v) {
    const int x = c[v];
    s += x;
    k += x > 0;
  }
  int run = block_exscan_1024(s, wsum, &tot);
  int kpos = block_exscan_1024(k, wsum, &tot);
  if (t == 0) nocc[blockIdx.x] = tot;
  for (int v = beg; v < end; ++v) {
    const int x = c[v];
    o[v] = run;
    run += x;
    if (x > 0) oc[kpos++] = v;
  }
}

__global__ __launch_bounds__(256) void vox_fill_kernel(int n, int r3, const int *__restrict__ ind,
                                                       int *__restrict__ cur, int *__restrict__ list) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int v = ind[(size_t)b * n + i];
  const int pos = atomicAdd(cur + (size_t)b * r3 + v, 1);
  list[(size_t)b * n + pos] = i;
}

// one wave per non-empty voxel: slist[start + rank(id)] = id, rank by counting (ids are distinct)
__global__ __launch_bounds__(256) void vox_sort_kernel(int n, int r3, const int *__restrict__ cnt,
                                                       const int *__restrict__ cur, const int *__restrict__ occ,
                                                       const int *__restrict__ nocc, const int *__restrict__ list,
                                                       int *__restrict__ slist) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= nocc[b]) return;
  const int lane = lane_id();
  const int v = occ[(size_t)b * n + k];
  const int cn = cnt[(size_t)b * r3 + v];
  const int start = cur[(size_t)b * r3 + v] - cn;  // cur points at the segment end after the fill
  const int *seg = list + (size_t)b * n + start;
  int *dst = slist + (size_t)b * n + start;
  if (cn <= 64) {  // the usual case: the ids sit in the lanes, ranks through lane broadcasts (no load inside the loop)
    const int mine = lane < cn ? seg[lane] : 0x7fffffff;
    int rank = 0;
    for (int j = 0; j < cn; ++j) rank += __shfl(mine, j) < mine;
    if (lane < cn) dst[rank] = mine;
    return;
  }
  for (int l0 = 0; l0 < cn; l0 += 64) {
    const int l = l0 + lane;
    const int mine = l < cn ? seg[l] : 0x7fffffff;
    int rank = 0;
    for (int j = 0; j < cn; ++j) rank += seg[j] < mine;  // wave-uniform address: one broadcast load
    if (l < cn) dst[rank] = mine;
  }
}

__global__ __launch_bounds__(256) void vox_gather_kernel(int c, int n, int r3, const int *__restrict__ cnt,
                                                         const int *__restrict__ cur, const int *__restrict__ occ,
                                                         const int *__restrict__ nocc, const int *__restrict__ slist,
                                                         const float *__restrict__ feat, float *__restrict__ out) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= nocc[b]) return;
  const int lane = lane_id();
  const int v = occ[(size_t)b * n + k];
  const int cn = cnt[(size_t)b * r3 + v];
  const int *seg = slist + (size_t)b * n + (cur[(size_t)b * r3 + v] - cn);
  const float div = (float)(1.0 / (double)(float)cn);  // PN2/vox_gpu.cu:70 divides a double literal
  const float *f = feat + (size_t)b * c * n;
  float *o = out + (size_t)b * c * r3 + v;
  for (int c0 = 0; c0 < c; c0 += 64) {
    const int ch = c0 + lane;
    if (ch < c) {
      const float *fj = f + (size_t)ch * n;
      float acc = 0.0f;
      for (int q = 0; q < cn; ++q) acc += fj[seg[q]] * div;
      o[(size_t)ch * r3] = acc;
    }
  }
}

extern "C" size_t p2pb_avg_voxelize_ws_bytes(int b, int n, int r) {
  return sizeof(int) * ((size_t)b * r * r * r + 3 * (size_t)b * n + (size_t)b);
}

extern "C" int p2pb_avg_voxelize_forward(int b, int c, int n, int r, const int *coords, const float *feat, int *ind,
                                         int *cnt, float *out, void *ws, void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || r <= 0 || !ws) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int r3 = r * r * r;
  int *cur = (int *)ws;
  int *list = cur + (size_t)b * r3;
  int *slist = list + (size_t)b * n;
  int *occ = slist + (size_t)b * n;
  int *nocc = occ + (size_t)b * n;
  int e = p2pb_zero_async(cnt, sizeof(int) * (size_t)b * r3, s);
  if (e != 0) return e;
  hipLaunchKernelGGL(vox_count_kernel, dim3(cdiv(n, 256), b), dim3(256), 0, s, n, r, coords, ind, cnt);
  hipLaunchKernelGGL(vox_scan_kernel, dim3(b), dim3(1024), 0, s, n, r3, cnt, cur, occ, nocc);
  hipLaunchKernelGGL(vox_fill_kernel, dim3(cdiv(n, 256), b), dim3(256), 0, s, n, r3, ind, cur, list);
  const int maxocc = n < r3 ? n : r3;
  hipLaunchKernelGGL(vox_sort_kernel, dim3(cdiv(maxocc, 4), b), dim3(256), 0, s, n, r3, cnt, cur, occ, nocc, list,
                     slist);
  e = p2pb_zero_async(out, sizeof(float) * (size_t)b * c * r3, s);
  if (e != 0) return e;
  hipLaunchKernelGGL(vox_gather_kernel, dim3(cdiv(maxocc, 4), b), dim3(256), 0, s, c, n, r3, cnt, cur, occ, nocc, slist,
                     feat, out);
  return p2pb_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Voxel-major variants for the fused inference branch (grid f32[b, r^3, c], a voxel's channels contiguous):
// the convolutions stage and store contiguous channel runs, and both ends of the branch become coalesced:
//   voxelize : features are first transposed to point-major [b, n, c] (LDS tile transpose), then one wave per
//              occupied voxel reads whole 4c-byte point rows (lane = channel) and writes one contiguous row --
//              the channel-major form reads and writes 4 bytes per 32-byte sector on both sides;
//   devoxelize: lane = channel reads the 8 corner rows, results go through an LDS transpose so the
//              channel-major output [b, c, n] is written in 256-byte runs.
// Same arithmetic, same summation order as the reference-layout kernels (bit-identical values).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_cn_kernel(int c, int n, const float *__restrict__ in,
                                                           float *__restrict__ out) {
  __shared__ float t[32][33];
  const int b = blockIdx.z, n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
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

// the whole grid in one pass: a thread owns V consecutive channels of one voxel -- zeros for an empty voxel, the
// ascending-order mean (same arithmetic as above) otherwise; replaces zero-fill + gather over the occupied list
typedef float vox_f32x4 __attribute__((ext_vector_type(4)));
// ALIGNED: c % 4 == 0, 16-byte loads / stores; else (e.g. the 3 + 32 channels of the first PVConv) a thread still owns
// four consecutive channels -- one occupancy lookup per quad -- but moves them as scalars and clips the last quad
template <bool ALIGNED>
__global__ __launch_bounds__(256) void vox_gather_cl_all_kernel(int c, int n, int r3, const int *__restrict__ cnt,
                                                                const int *__restrict__ cur,
                                                                const int *__restrict__ slist,
                                                                const float *__restrict__ feat_t,
                                                                float *__restrict__ out) {
  const int b = blockIdx.y, cv = (c + 3) / 4;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)r3 * cv) return;
  const int v = (int)(e / cv), ch = (int)(e % cv) * 4;
  const int nch = min(4, c - ch);
  const int cn = cnt[(size_t)b * r3 + v];
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (cn > 0) {
    const int *seg = slist + (size_t)b * n + (cur[(size_t)b * r3 + v] - cn);
    const float div = (float)(1.0 / (double)(float)cn);  // PN2/vox_gpu.cu:70 divides a double literal
    const float *f = feat_t + (size_t)b * n * c + ch;
    for (int q = 0; q < cn; ++q) {
      const float *fq = f + (size_t)seg[q] * c;
      if (ALIGNED) {
        const vox_f32x4 x = *(const vox_f32x4 *)fq;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += x[i] * div;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < nch) acc[i] += fq[i] * div;
      }
    }
  }
  float *o = out + ((size_t)b * r3 + v) * c + ch;
  if (ALIGNED) {
    *(vox_f32x4 *)o = vox_f32x4{acc[0], acc[1], acc[2], acc[3]};
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < nch) o[i] = acc[i];
  }
}

// Occupied voxels only, after a streaming zero-fill of the grid (p2pb_zero_async: 16 B per lane, 7 TB/s into the
// memory-side cache for the 134 MB level-0 grid of the bench; the one-pass kernels above visit every voxel and write the
// same bytes at 1 TB/s: 145 us against 18 + ~10). A thread owns one float of one occupied voxel's row; same arithmetic
// and summation order as the one-pass kernels.
__global__ __launch_bounds__(256) void vox_gather_cl_occ_kernel(int c, int n, int r3, const int *__restrict__ cnt,
                                                                const int *__restrict__ cur,
                                                                const int *__restrict__ occ,
                                                                const int *__restrict__ nocc,
                                                                const int *__restrict__ slist,
                                                                const float *__restrict__ feat_t,
                                                                float *__restrict__ out) {
  const int b = blockIdx.y;
  const size_t total = (size_t)nocc[b] * c;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int k = (int)(e / c), ch = (int)(e % c);
    const int v = occ[(size_t)b * n + k];
    const int cn = cnt[(size_t)b * r3 + v];
    const int *seg = slist + (size_t)b * n + (cur[(size_t)b * r3 + v] - cn);
    const float div = (float)(1.0 / (double)(float)cn);  // PN2/vox_gpu.cu:70 divides a double literal
    const float *f = feat_t + (size_t)b * n * c + ch;
    float acc = 0.0f;
    for (int q = 0; q < cn; ++q) acc += f[(size_t)seg[q] * c] * div;
    out[((size_t)b * r3 + v) * c + ch] = acc;
  }
}

// The grid straight in the pre-split operand format of the voxel convolutions (conv3d.hip "S format"): out
// u32x4[b][r^3][ceil(c/16)][2 planes][2 khalf], the fp16 pair (h0 | h1) of 4 x mean for channels chunk*16 + khalf*8 + i --
// 4 bytes per (voxel, channel) like the fp32 grid, channels padded with zeros to a multiple of 16. A thread owns 8
// consecutive channels of one voxel; same means (ascending point order) as the kernels above, then the split the
// convolution's staging phase would apply -- the first convolution of a PVConv then stages with LDS-DMA alone.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void vox_gather_cl_split_kernel(int c, int nchunk, int n, int r3,
                                                                  const int *__restrict__ cnt, const int *__restrict__ cur,
                                                                  const int *__restrict__ slist,
                                                                  const float *__restrict__ feat_t, u32x4 *__restrict__ out) {
  const int b = blockIdx.y, ng = nchunk * 2;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)r3 * ng) return;
  const int v = (int)(e / ng), g = (int)(e % ng), ch = g * 8;
  const int cn = cnt[(size_t)b * r3 + v];
  float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if (cn > 0 && ch < c) {
    const int *seg = slist + (size_t)b * n + (cur[(size_t)b * r3 + v] - cn);
    const float div = (float)(1.0 / (double)(float)cn);  // PN2/vox_gpu.cu:70 divides a double literal
    const float *f = feat_t + (size_t)b * n * c + ch;
    for (int q = 0; q < cn; ++q) {
      const float *fq = f + (size_t)seg[q] * c;
      if (ALIGNED) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (ch + 4 * h < c) {
            const vox_f32x4 x = *(const vox_f32x4 *)(fq + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[4 * h + i] += x[i] * div;
          }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (ch + i < c) acc[i] += fq[i] * div;
      }
    }
  }
  u32x4 p0, p1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned a0, a1, a2;
    split_pair<SPLIT_F16X3>(acc[2 * i], acc[2 * i + 1], a0, a1, a2);
    p0[i] = a0;
    p1[i] = a1;
  }
  u32x4 *dst = out + (((size_t)b * r3 + v) * nchunk + (g >> 1)) * 4 + (g & 1);
  dst[0] = p0;
  dst[2] = p1;
}

// Occupied voxels only, after a streaming zero-fill of the split grid (the fp16 pair of 0 is all-zero bits): the form for
// channel counts that are not a multiple of 4 (the 3 + 32 channels of the first PVConv: scalar point-row loads) and, since round 5,
// for every r = 32 grid (at most a quarter of the voxels is occupied: the one-pass kernel above wrote the 268 MB of the 64-channel
// level-0 grid at 1.2 TB/s, 226 us per 32 patches; the zero-fill streams at ~7 TB/s). The kernel is LATENCY-bound by its fullest
// voxel -- a thread adds its voxel's points one after the other (ascending index: the reference's mean, bit for bit), and each
// point used to be two dependent L2 round trips (list entry -> row): 150-170 us whatever the batch. Now the list entries and rows
// of FOUR points are in flight before the first is added; the additions keep their order.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void vox_gather_cl_occ_split_kernel(int c, int nchunk, int n, int r3,
                                                                      const int *__restrict__ cnt, const int *__restrict__ cur,
                                                                      const int *__restrict__ occ, const int *__restrict__ nocc,
                                                                      const int *__restrict__ slist,
                                                                      const float *__restrict__ feat_t, u32x4 *__restrict__ out) {
  const int b = blockIdx.y, ng = nchunk * 2;
  const size_t total = (size_t)nocc[b] * ng;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int k = (int)(e / ng), g = (int)(e % ng), ch = g * 8;
    const int v = occ[(size_t)b * n + k];
    const int cn = cnt[(size_t)b * r3 + v];
    const int *seg = slist + (size_t)b * n + (cur[(size_t)b * r3 + v] - cn);
    const float div = (float)(1.0 / (double)(float)cn);  // PN2/vox_gpu.cu:70 divides a double literal
    const float *f = feat_t + (size_t)b * n * c + ch;
    float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int q0 = 0; q0 < cn; q0 += 4) {
      int id[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) id[u] = seg[min(q0 + u, cn - 1)];
      float x[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float *fq = f + (size_t)id[u] * c;
        if (ALIGNED) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            vox_f32x4 y = vox_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (ch + 4 * h < c) y = *(const vox_f32x4 *)(fq + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) x[u][4 * h + i] = y[i];
          }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) x[u][i] = ch + i < c ? fq[i] : 0.0f;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (q0 + u < cn) {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] += x[u][i] * div;
        }
    }
    u32x4 p0, p1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned a0, a1, a2;
      split_pair<SPLIT_F16X3>(acc[2 * i], acc[2 * i + 1], a0, a1, a2);
      p0[i] = a0;
      p1[i] = a1;
    }
    u32x4 *dst = out + (((size_t)b * r3 + v) * nchunk + (g >> 1)) * 4 + (g & 1);
    dst[0] = p0;
    dst[2] = p1;
  }
}

// The coordinate-only half of the voxelisation (occupancy counts + per-voxel sorted point lists): it depends on
// the voxel coordinates alone, so the sampler runs it once per (level, resolution) on the geometry stream and every
// PVConv of that level reuses it. ws: p2pb_avg_voxelize_ws_bytes(b,n,r) bytes, consumed by ..._cl_gather.
extern "C" int p2pb_voxel_sort(int b, int n, int r, const int *coords, int *ind, int *cnt, void *ws, void *stream) {
  if (b <= 0 || n <= 0 || r <= 0 || !ws) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int r3 = r * r * r;
  int *cur = (int *)ws;
  int *list = cur + (size_t)b * r3;
  int *slist = list + (size_t)b * n;
  int *occ = slist + (size_t)b * n;
  int *nocc = occ + (size_t)b * n;
  int e = p2pb_zero_async(cnt, sizeof(int) * (size_t)b * r3, s);
  if (e != 0) return e;
  hipLaunchKernelGGL(vox_count_kernel, dim3(cdiv(n, 256), b), dim3(256), 0, s, n, r, coords, ind, cnt);
  hipLaunchKernelGGL(vox_scan_kernel, dim3(b), dim3(1024), 0, s, n, r3, cnt, cur, occ, nocc);
  hipLaunchKernelGGL(vox_fill_kernel, dim3(cdiv(n, 256), b), dim3(256), 0, s, n, r3, ind, cur, list);
  const int maxocc = n < r3 ? n : r3;
  hipLaunchKernelGGL(vox_sort_kernel, dim3(cdiv(maxocc, 4), b), dim3(256), 0, s, n, r3, cnt, cur, occ, nocc, list,
                     slist);
  return p2pb_launch_status();
}

// The feature half: out f32[b, r^3, c] (voxel-major) from feat f32[b,c,n] and the sort of p2pb_voxel_sort
// (cnt, ws); feat_t f32[b, n, c] scratch
extern "C" int p2pb_avg_voxelize_cl_gather(int b, int c, int n, int r, const float *feat, const int *cnt, const void *ws,
                                           float *out, float *feat_t, void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || r <= 0 || !ws || !feat_t) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int r3 = r * r * r;
  const int *cur = (const int *)ws;
  const int *slist = cur + (size_t)b * r3 + (size_t)b * n;
  const int *occ = slist + (size_t)b * n;
  const int *nocc = occ + (size_t)b * n;
  const int maxocc = n < r3 ? n : r3;
  hipLaunchKernelGGL(transpose_cn_kernel, dim3(cdiv(n, 32), cdiv(c, 32), b), dim3(256), 0, s, c, n, feat, feat_t);
  // rows of whole 16-byte quads: one pass over every voxel (zeros for the empty ones). Other channel counts (the 3 + 32
  // channels of the first PVConv): zero-fill + the occupied voxels only -- measured 181 vs 234 us for the whole
  // voxelisation at the bench's level-0 shape; for aligned rows the one-pass form is as fast or faster (tools/exp_voxelize.py)
  static const int onepass = (int)p2pb_experiment_long("vox_onepass", -1);  // (A/B switch: 0 / 1 force)
  if (onepass == 0 || (onepass < 0 && (c & 3) != 0)) {
    const int e = p2pb_zero_async(out, (size_t)b * r3 * c * sizeof(float), s);
    if (e != 0) return e;
    const size_t nwg = cdiv((size_t)maxocc * c, 256);
    hipLaunchKernelGGL(vox_gather_cl_occ_kernel, dim3((unsigned)(nwg > 65536 ? 65536 : nwg), b), dim3(256), 0, s, c, n, r3,
                       cnt, cur, occ, nocc, slist, feat_t, out);
    return p2pb_launch_status();
  }
  const dim3 grid((unsigned)cdiv((size_t)r3 * ((c + 3) / 4), 256), b);
  if ((c & 3) == 0)
    hipLaunchKernelGGL(vox_gather_cl_all_kernel<true>, grid, dim3(256), 0, s, c, n, r3, cnt, cur, slist, feat_t, out);
  else
    hipLaunchKernelGGL(vox_gather_cl_all_kernel<false>, grid, dim3(256), 0, s, c, n, r3, cnt, cur, slist, feat_t, out);
  return p2pb_launch_status();
}

// The feature half, straight into the pre-split operand format (S format) of the voxel convolutions:
// out_split b * r^3 * ceil(c/16) * 64 bytes; feat_t f32[b, n, c] scratch
extern "C" int p2pb_avg_voxelize_cl_gather_split(int b, int c, int n, int r, const float *feat, const int *cnt, const void *ws,
                                                 void *out_split, float *feat_t, void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || r <= 0 || !ws || !feat_t || !out_split) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int r3 = r * r * r, nchunk = (c + 15) / 16;
  const int *cur = (const int *)ws;
  const int *slist = cur + (size_t)b * r3 + (size_t)b * n;
  hipLaunchKernelGGL(transpose_cn_kernel, dim3(cdiv(n, 32), cdiv(c, 32), b), dim3(256), 0, s, c, n, feat, feat_t);
  static const int onepass = (int)p2pb_experiment_long("vox_onepass", -1);  // (A/B switch: 0 / 1 force)
  // zero-fill + occupied voxels only: ragged rows (the rule of p2pb_avg_voxelize_cl_gather) and every grid with at least four
  // voxels per point (r = 32 at 8192 points: <= 25 % occupied)
  if (onepass == 0 || (onepass < 0 && ((c & 3) != 0 || (size_t)r3 >= 4 * (size_t)n))) {
    const int *occ = slist + (size_t)b * n;
    const int *nocc = occ + (size_t)b * n;
    const int maxocc = n < r3 ? n : r3;
    const int e = p2pb_zero_async(out_split, (size_t)b * r3 * nchunk * 64, s);
    if (e != 0) return e;
    const size_t nwg = cdiv((size_t)maxocc * nchunk * 2, 256);
    const dim3 g((unsigned)(nwg > 65536 ? 65536 : nwg), b);
    if ((c & 3) == 0)
      hipLaunchKernelGGL(vox_gather_cl_occ_split_kernel<true>, g, dim3(256), 0, s, c, nchunk, n, r3, cnt, cur, occ, nocc, slist, feat_t,
                         (u32x4 *)out_split);
    else
      hipLaunchKernelGGL(vox_gather_cl_occ_split_kernel<false>, g, dim3(256), 0, s, c, nchunk, n, r3, cnt, cur, occ, nocc, slist, feat_t,
                         (u32x4 *)out_split);
    return p2pb_launch_status();
  }
  const dim3 grid((unsigned)cdiv((size_t)r3 * nchunk * 2, 256), b);
  if ((c & 3) == 0)
    hipLaunchKernelGGL(vox_gather_cl_split_kernel<true>, grid, dim3(256), 0, s, c, nchunk, n, r3, cnt, cur, slist, feat_t,
                       (u32x4 *)out_split);
  else
    hipLaunchKernelGGL(vox_gather_cl_split_kernel<false>, grid, dim3(256), 0, s, c, nchunk, n, r3, cnt, cur, slist, feat_t,
                       (u32x4 *)out_split);
  return p2pb_launch_status();
}

// both halves: out f32[b, r^3, c] (voxel-major); feat_t f32[b, n, c] scratch; everything else as p2pb_avg_voxelize_forward
extern "C" int p2pb_avg_voxelize_cl_forward(int b, int c, int n, int r, const int *coords, const float *feat, int *ind,
                                            int *cnt, float *out, float *feat_t, void *ws, void *stream) {
  const int e = p2pb_voxel_sort(b, n, r, coords, ind, cnt, ws, stream);
  if (e != 0) return e;
  return p2pb_avg_voxelize_cl_gather(b, c, n, r, feat, cnt, ws, out, feat_t, stream);
}

template <int CC>
__global__ __launch_bounds__(256) void vox_grad_kernel(int c, int n, int r3, const int *__restrict__ ind,
                                                       const int *__restrict__ cnt, const float *__restrict__ gy,
                                                       float *__restrict__ gx) {
  const int b = blockIdx.z;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c0 = blockIdx.y * CC, c1 = min(c0 + CC, c);
  const int pos = ind[(size_t)b * n + i];
  const int cn = cnt[(size_t)b * r3 + pos];
  const float div = cn > 0 ? (float)(1.0 / (double)(float)cn) : 0.0f;
  const float *g = gy + (size_t)b * c * r3 + pos;
  float *o = gx + (size_t)b * c * n + i;
  for (int j = c0; j < c1; ++j) o[(size_t)j * n] = cn > 0 ? g[(size_t)j * r3] * div : 0.0f;
}

extern "C" int p2pb_avg_voxelize_backward(int b, int c, int n, int r3, const int *ind, const int *cnt,
                                          const float *grad_y, float *grad_x, void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || r3 <= 0) return P2PB_EINVAL;
  constexpr int CC = 16;
  hipLaunchKernelGGL(vox_grad_kernel<CC>, dim3(cdiv(n, 256), cdiv(c, CC), b), dim3(256), 0, (hipStream_t)stream, c, n,
                     r3, ind, cnt, grad_y, grad_x);
  return p2pb_launch_status();
}

// ------------------------------------------------------------------------------------------------
// trilinear devoxelize
// ------------------------------------------------------------------------------------------------
struct Corners {
  int idx[8];
  float w[8];
};

__device__ __forceinline__ Corners devox_corners(float x, float y, float z, int r) {
  Corners k;
  const int r2 = r * r;
  const float xl = floorf(x), yl = floorf(y), zl = floorf(z);
  const float xd1 = x - xl, yd1 = y - yl, zd1 = z - zl;
  const float xd0 = 1.0f - xd1, yd0 = 1.0f - yd1, zd0 = 1.0f - zd1;
  k.w[0] = xd0 * yd0 * zd0;
  k.w[1] = xd0 * yd0 * zd1;
  k.w[2] = xd0 * yd1 * zd0;
  k.w[3] = xd0 * yd1 * zd1;
  k.w[4] = xd1 * yd0 * zd0;
  k.w[5] = xd1 * yd0 * zd1;
  k.w[6] = xd1 * yd1 * zd0;
  k.w[7] = xd1 * yd1 * zd1;
  const int xlo = (int)xl, ylo = (int)yl, zlo = (int)zl;
  const int xhi = (xd1 > 0) ? -1 : 0, yhi = (yd1 > 0) ? -1 : 0, zhi = (zd1 > 0) ? 1 : 0;
  k.idx[0] = xlo * r2 + ylo * r + zlo;
  k.idx[1] = k.idx[0] + zhi;
  k.idx[2] = k.idx[0] + (yhi & r);
  k.idx[3] = k.idx[2] + zhi;
  k.idx[4] = k.idx[0] + (xhi & r2);
  k.idx[5] = k.idx[4] + zhi;
  k.idx[6] = k.idx[4] + (yhi & r);
  k.idx[7] = k.idx[6] + zhi;
  return k;
}

// AFF: the grid holds the RAW second-convolution output; the per-(sample, channel) affine that stands for
// AdaGN + SE gating (feat*A + B) is applied to each corner value before interpolating, i.e. the same
// "transform, then interpolate" order as the unfused graph, without a pass over the grid in between.
template <int CC, bool AFF>
__global__ __launch_bounds__(256) void devox_kernel(int c, int n, int r, int training,
                                                    const float *__restrict__ coords, const float *__restrict__ feat,
                                                    const float *__restrict__ aff_a, const float *__restrict__ aff_b,
                                                    int *__restrict__ inds, float *__restrict__ wgts,
                                                    float *__restrict__ outs) {
  const int b = blockIdx.z;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int r3 = r * r * r;
  const float *co = coords + (size_t)b * 3 * n;
  const Corners k = devox_corners(co[i], co[i + n], co[i + 2 * n], r);
  if (training && blockIdx.y == 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      wgts[(size_t)b * 8 * n + (size_t)q * n + i] = k.w[q];
      inds[(size_t)b * 8 * n + (size_t)q * n + i] = k.idx[q];
    }
  }
  const int c0 = blockIdx.y * CC, c1 = min(c0 + CC, c);
  const float *f = feat + (size_t)b * c * r3;
  float *o = outs + (size_t)b * c * n + i;
  for (int j = c0; j < c1; ++j) {
    const float *fj = f + (size_t)j * r3;
    float fv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) fv[q] = fj[k.idx[q]];
    if (AFF) {
      const float a = aff_a[(size_t)b * c + j], bb = aff_b[(size_t)b * c + j];
#pragma unroll
      for (int q = 0; q < 8; ++q) fv[q] = fv[q] * a + bb;
    }
    float acc = k.w[0] * fv[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) acc = __fmaf_rn(k.w[q], fv[q], acc);
    o[(size_t)j * n] = acc;
  }
}

extern "C" int p2pb_trilinear_devoxelize_forward(int b, int c, int n, int r, int is_training, const float *coords,
                                                 const float *feat, int *inds, float *wgts, float *outs,
                                                 void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || r <= 0) return P2PB_EINVAL;
  if (is_training && (!inds || !wgts)) return P2PB_EINVAL;
  constexpr int CC = 16;
  hipLaunchKernelGGL((devox_kernel<CC, false>), dim3(cdiv(n, 256), cdiv(c, CC), b), dim3(256), 0, (hipStream_t)stream,
                     c, n, r, is_training, coords, feat, (const float *)nullptr, (const float *)nullptr, inds, wgts,
                     outs);
  return p2pb_launch_status();
}

// inference-only fused form: outs[b,c,i] = sum_k w_k * (feat[b,c,idx_k] * aff_a[b,c] + aff_b[b,c])
extern "C" int p2pb_trilinear_devoxelize_affine(int b, int c, int n, int r, const float *coords, const float *feat,
                                                const float *aff_a, const float *aff_b, float *outs, void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || r <= 0 || !aff_a || !aff_b) return P2PB_EINVAL;
  constexpr int CC = 16;
  hipLaunchKernelGGL((devox_kernel<CC, true>), dim3(cdiv(n, 256), cdiv(c, CC), b), dim3(256), 0, (hipStream_t)stream,
                     c, n, r, 0, coords, feat, aff_a, aff_b, (int *)nullptr, (float *)nullptr, outs);
  return p2pb_launch_status();
}

// voxel-major grid f32[b, r^3, c] -> outs f32[b, c, n] = sum_k w_k * (grid[idx_k]*aff_a + aff_b)
// add (optional, f32[b,c,n]) with add_scale/add_shift f32[b,c]: outs += swish(add*add_scale + add_shift) -- PVConv's
// point branch (conv -> norm -> Swish, models/pvcnn.py:286,325) joined to the voxel branch in the same pass
__global__ __launch_bounds__(256) void devox_cl_kernel(int c, int n, int r, const float *__restrict__ coords,
                                                       const float *__restrict__ grid, const float *__restrict__ aff_a,
                                                       const float *__restrict__ aff_b, const float *__restrict__ add,
                                                       const float *__restrict__ add_scale,
                                                       const float *__restrict__ add_shift, float *__restrict__ outs) {
  __shared__ float tile[64][65];  // [channel][point]
  __shared__ __attribute__((aligned(16))) int sidx[64][8];
  __shared__ __attribute__((aligned(16))) float sw[64][8];
  const int b = blockIdx.z, p0 = blockIdx.x * 64, t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int r3 = r * r * r;
  // Every wave stages the trilinear corners of ITS OWN 16 points (lanes 0..15), with 16-byte LDS stores only.
  // Round 4: the first form -- wave 0 staged all 64 points and the compiler merged a row's eight stores into ds_write_b96 +
  // ds_write2_b32 + ds_write_b32 -- returned wrong values for wave 3's points when a matrix kernel of ANOTHER stream shared the CU
  // (the two-chain sampler: 10-15 % of such launches had one wave's 16 points x 64 channels off by O(1); alone, or beside
  // element-wise kernels, never; tools/dbg/devox_conc.py). Either change alone removes it (own points: 0 of 90 launches
  // wrong; 16-byte stores: 0 of 90 as flat_store_dwordx4 -- what a volatile store through a generic pointer compiles to -- and
  // 0 of 270 as the ds_write_b128 below); ds_write_b96 is not used anywhere in this library any more.
  if (lane < 16) {
    const int tt = wave * 16 + lane;
    const int i = min(p0 + tt, n - 1);
    const float *co = coords + (size_t)b * 3 * n;
    const Corners k = devox_corners(co[i], co[i + n], co[i + 2 * n], r);
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    // (address_space(3): a volatile store through a generic pointer is a flat_store_dwordx4)
#define DV_LDS __attribute__((address_space(3)))
    *(volatile DV_LDS i32x4 *)&sidx[tt][0] = i32x4{k.idx[0], k.idx[1], k.idx[2], k.idx[3]};
    *(volatile DV_LDS i32x4 *)&sidx[tt][4] = i32x4{k.idx[4], k.idx[5], k.idx[6], k.idx[7]};
    *(volatile DV_LDS f32x4 *)&sw[tt][0] = f32x4{k.w[0], k.w[1], k.w[2], k.w[3]};
    *(volatile DV_LDS f32x4 *)&sw[tt][4] = f32x4{k.w[4], k.w[5], k.w[6], k.w[7]};
#undef DV_LDS
  }
  __syncthreads();
  const float *g = grid + (size_t)b * r3 * c;
  {
    const int c0 = blockIdx.y * 64;  // one 64-channel chunk per workgroup
    const int ch = c0 + lane;
    if (ch < c) {
      const float a = aff_a ? aff_a[(size_t)b * c + ch] : 1.0f, bb = aff_b ? aff_b[(size_t)b * c + ch] : 0.0f;
#pragma unroll 2
      for (int pl = wave * 16; pl < wave * 16 + 16; ++pl) {
        float fv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) fv[q] = g[(size_t)sidx[pl][q] * c + ch];
        if (aff_a) {
#pragma unroll
          for (int q = 0; q < 8; ++q) fv[q] = fv[q] * a + bb;
        }
        float acc = sw[pl][0] * fv[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) acc = __fmaf_rn(sw[pl][q], fv[q], acc);
        tile[lane][pl] = acc;
      }
    }
    __syncthreads();
    const int pt = t & 63;
    if (p0 + pt < n) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int cr = (t >> 6) + 4 * k;
        if (c0 + cr < c) {
          const size_t o = ((size_t)b * c + c0 + cr) * n + p0 + pt;
          float v = tile[cr][pt];
          if (add) {
            const float z = add[o] * add_scale[(size_t)b * c + c0 + cr] + add_shift[(size_t)b * c + c0 + cr];
            v = z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.44269504088896340736f)) + v;
          }
          outs[o] = v;
        }
      }
    }
    __syncthreads();
  }
}

// the same pass with 16-byte gathers (round 5, last session): a lane owns FOUR channels of one point -- 16 lanes cover the 64-channel
// chunk, a wave instruction fetches the corner rows of four points -- a quarter of the load instructions for the same bytes; every
// (point, channel) value goes through the same multiplies and fused adds in the same order: bit-identical. c % 64 == 0.
__global__ __launch_bounds__(256) void devox_cl4_kernel(int c, int n, int r, const float *__restrict__ coords,
                                                        const float *__restrict__ grid, const float *__restrict__ aff_a,
                                                        const float *__restrict__ aff_b, const float *__restrict__ add,
                                                        const float *__restrict__ add_scale,
                                                        const float *__restrict__ add_shift, float *__restrict__ outs) {
  __shared__ float tile[64][65];  // [channel][point]
  __shared__ __attribute__((aligned(16))) int sidx[64][8];
  __shared__ __attribute__((aligned(16))) float sw[64][8];
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int b = blockIdx.z, p0 = blockIdx.x * 64, t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int r3 = r * r * r;
  if (lane < 16) {  // (every wave stages its own 16 points with 16-byte LDS stores: see devox_cl_kernel)
    const int tt = wave * 16 + lane;
    const int i = min(p0 + tt, n - 1);
    const float *co = coords + (size_t)b * 3 * n;
    const Corners k = devox_corners(co[i], co[i + n], co[i + 2 * n], r);
#define DV_LDS __attribute__((address_space(3)))
    *(volatile DV_LDS i32x4 *)&sidx[tt][0] = i32x4{k.idx[0], k.idx[1], k.idx[2], k.idx[3]};
    *(volatile DV_LDS i32x4 *)&sidx[tt][4] = i32x4{k.idx[4], k.idx[5], k.idx[6], k.idx[7]};
    *(volatile DV_LDS f32x4 *)&sw[tt][0] = f32x4{k.w[0], k.w[1], k.w[2], k.w[3]};
    *(volatile DV_LDS f32x4 *)&sw[tt][4] = f32x4{k.w[4], k.w[5], k.w[6], k.w[7]};
#undef DV_LDS
  }
  __syncthreads();
  const float *g = grid + (size_t)b * r3 * c;
  const int c0 = blockIdx.y * 64;
  {
    const int cq = lane & 15, ps = lane >> 4;
    const int ch4 = c0 + 4 * cq;
    f32x4 a4 = {1.0f, 1.0f, 1.0f, 1.0f}, b4 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (aff_a) {
      a4 = *(const f32x4 *)(aff_a + (size_t)b * c + ch4);
      b4 = *(const f32x4 *)(aff_b + (size_t)b * c + ch4);
    }
#pragma unroll 2
    for (int it = 0; it < 4; ++it) {
      const int pl = wave * 16 + it * 4 + ps;
      f32x4 fv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) fv[q] = *(const f32x4 *)(g + (size_t)sidx[pl][q] * c + ch4);
      if (aff_a) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) fv[q][i] = fv[q][i] * a4[i] + b4[i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float acc = sw[pl][0] * fv[0][i];
#pragma unroll
        for (int q = 1; q < 8; ++q) acc = __fmaf_rn(sw[pl][q], fv[q][i], acc);
        tile[4 * cq + i][pl] = acc;
      }
    }
  }
  __syncthreads();
  const int pt = t & 63;
  if (p0 + pt < n) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int cr = (t >> 6) + 4 * k;
      const size_t o = ((size_t)b * c + c0 + cr) * n + p0 + pt;
      float v = tile[cr][pt];
      if (add) {
        const float z = add[o] * add_scale[(size_t)b * c + c0 + cr] + add_shift[(size_t)b * c + c0 + cr];
        v = z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.44269504088896340736f)) + v;
      }
      outs[o] = v;
    }
  }
}

extern "C" int p2pb_trilinear_devoxelize_cl_affine(int b, int c, int n, int r, const float *coords, const float *grid,
                                                   const float *aff_a, const float *aff_b, const float *add,
                                                   const float *add_scale, const float *add_shift, float *outs,
                                                   void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || r <= 0 || ((aff_a == nullptr) != (aff_b == nullptr))) return P2PB_EINVAL;
  if (add && (!add_scale || !add_shift)) return P2PB_EINVAL;
  static const long wide = p2pb_experiment_long("devox_cl4", 1);  // (A/B switch: 0 = the 4-byte gathers everywhere)
  if (wide && c % 64 == 0 && (((size_t)grid | (size_t)aff_a | (size_t)aff_b) & 15) == 0)  // (16-byte loads: rows and tables aligned)
    hipLaunchKernelGGL(devox_cl4_kernel, dim3(cdiv(n, 64), c / 64, b), dim3(256), 0, (hipStream_t)stream, c, n, r, coords, grid,
                       aff_a, aff_b, add, add_scale, add_shift, outs);
  else
    hipLaunchKernelGGL(devox_cl_kernel, dim3(cdiv(n, 64), cdiv(c, 64), b), dim3(256), 0, (hipStream_t)stream, c, n, r, coords, grid,
                       aff_a, aff_b, add, add_scale, add_shift, outs);
  return p2pb_launch_status();
}

template <int CC>
__global__ __launch_bounds__(256) void devox_grad_kernel(int c, int n, int r3, const int *__restrict__ inds,
                                                         const float *__restrict__ wgts,
                                                         const float *__restrict__ gy, float *__restrict__ gx) {
  const int b = blockIdx.z;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int idx[8];
  float w[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    idx[q] = inds[(size_t)b * 8 * n + (size_t)q * n + i];
    w[q] = wgts[(size_t)b * 8 * n + (size_t)q * n + i];
  }
  const int c0 = blockIdx.y * CC, c1 = min(c0 + CC, c);
  for (int j = c0; j < c1; ++j) {
    const float g = gy[((size_t)b * c + j) * n + i];
    float *o = gx + ((size_t)b * c + j) * r3;
#pragma unroll
    for (int q = 0; q < 8; ++q) atomicAdd(o + idx[q], w[q] * g);
  }
}

// the same with the CH grids of a workgroup in LDS (common.h "scatter-add backward passes"); Lp = r3 rounded up to 4
template <int CH>
__global__ __launch_bounds__(SCAT_THREADS) void devox_grad_lds_kernel(int c, int n, int r3, int Lp, const int *__restrict__ inds,
                                                                     const float *__restrict__ wgts, const float *__restrict__ gy,
                                                                     float *__restrict__ gx) {
  extern __shared__ float rows[];
  const int b = blockIdx.y, c0 = blockIdx.x * CH, nch = min(CH, c - c0);
  scat_zero(rows, CH * Lp);
  const int *ib = inds + (size_t)b * 8 * n;
  const float *wb = wgts + (size_t)b * 8 * n;
  const float *g0 = gy + ((size_t)b * c + c0) * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int idx[8];
    float w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      idx[q] = ib[(size_t)q * n + i];
      w[q] = wb[(size_t)q * n + i];
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      if (j < nch) {
        const float g = g0[(size_t)j * n + i];
#pragma unroll
        for (int q = 0; q < 8; ++q) atomicAdd(rows + j * Lp + idx[q], w[q] * g);
      }
    }
  }
  scat_store(rows, r3, Lp, nch, gx + ((size_t)b * c + c0) * r3);
}

template <int CH>
static int devox_grad_lds_launch(int b, int c, int n, int r3, const int *inds, const float *wgts, const float *gy, float *gx,
                                 hipStream_t s) {
  const int Lp = (r3 + 3) & ~3;
  const size_t lds = sizeof(float) * (size_t)CH * Lp;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void *)devox_grad_lds_kernel<CH>, hipFuncAttributeMaxDynamicSharedMemorySize, SCAT_LDS_MAX);
    once = true;
  }
  hipLaunchKernelGGL(devox_grad_lds_kernel<CH>, dim3(cdiv(c, CH), b), dim3(scat_threads()), lds, s, c, n, r3, Lp, inds, wgts, gy, gx);
  return p2pb_launch_status();
}

extern "C" int p2pb_trilinear_devoxelize_backward(int b, int c, int n, int r3, const int *inds, const float *wgts,
                                                  const float *grad_y, float *grad_x, void *stream) {
  if (b <= 0 || c <= 0 || n <= 0 || r3 <= 0) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  switch (scat_rows(r3, c, 16)) {
    case 0: break;
    case 1: return devox_grad_lds_launch<1>(b, c, n, r3, inds, wgts, grad_y, grad_x, s);
    case 2: case 3: return devox_grad_lds_launch<2>(b, c, n, r3, inds, wgts, grad_y, grad_x, s);
    case 4: case 5: case 6: case 7: return devox_grad_lds_launch<4>(b, c, n, r3, inds, wgts, grad_y, grad_x, s);
    case 16: return devox_grad_lds_launch<16>(b, c, n, r3, inds, wgts, grad_y, grad_x, s);
    default: return devox_grad_lds_launch<8>(b, c, n, r3, inds, wgts, grad_y, grad_x, s);
  }
  if (p2pb_deterministic()) return P2PB_EINVAL;  // (rows beyond the LDS: only the global-atomic kernel is left)
  int e = p2pb_zero_async(grad_x, sizeof(float) * (size_t)b * c * r3, s);
  if (e != 0) return e;
  constexpr int CC = 16;
  hipLaunchKernelGGL(devox_grad_kernel<CC>, dim3(cdiv(n, 256), cdiv(c, CC), b), dim3(256), 0, s, c, n, r3, inds, wgts,
                     grad_y, grad_x);
  return p2pb_launch_status();
}
