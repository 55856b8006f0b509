// attention.hip -- the core of LinearAttention (models/modules.py:165-194) for gfx950, forward and backward.
//
//   qkv f32[b, 3*heads*32, n]  (the output of to_qkv, channel order (q | k | v) x heads x 32)
//   ks   = softmax over n of every k row
//   ctx  = ks v^T                   [32 x 32] per (sample, head):  ctx[d][e] = sum_n ks[d,n] v[e,n]
//   out  = ctx^T q                  out[e,n] = sum_d ctx[d][e] q[d,n]      -> f32[b, heads*32, n]
//
// The two 1x1 convolutions around it (to_qkv, to_out) are the pointwise GEMM kernels of pointwise.hip; this file
// is what sits between them. Tokens are the last level's centres (8 .. 195 in the BASELINE configs), so the work
// is tiny and latency-bound: ONE workgroup per (sample, head) keeps the whole head in LDS / registers and runs
// the three phases back to back -- the reference spends 2 einsum launches + a softmax + 3 rearranges on it.
// Any n is accepted (the rows are streamed in 64-token tiles), so a PVConv-level attention (cfg attentions[i] = 1
// on a set-abstraction stage) takes the same kernel.
#include "common.h"

#define LA_D 32    // dim_head (models/modules.py:170: dim_head=32 is never overridden)
#define LA_T 256   // threads: 4 waves
#define LA_TILE 64 // tokens per tile

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_maxf(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// row statistics of the k block: rmax[d] = max_n k[d,n], rinv[d] = 1 / sum_n exp(k[d,n] - rmax[d])
__device__ __forceinline__ void la_row_stats(const float *__restrict__ k, int n, float *rmax, float *rinv) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  for (int d = w; d < LA_D; d += LA_T / 64) {
    const float *row = k + (size_t)d * n;
    float m = -INFINITY;
    for (int i = l; i < n; i += 64) m = fmaxf(m, row[i]);
    m = wave_maxf(m);
    float s = 0.0f;
    for (int i = l; i < n; i += 64) s += expf(row[i] - m);
    s = wave_sum(s);
    if (l == 0) {
      rmax[d] = m;
      rinv[d] = 1.0f / s;
    }
  }
}

__global__ __launch_bounds__(LA_T) void linear_attention_fwd_kernel(int heads, int n, const float *__restrict__ qkv,
                                                                    float *__restrict__ out,
                                                                    float *__restrict__ ctx_out) {
  __shared__ float rmax[LA_D], rinv[LA_D];
  __shared__ float ks[LA_D][LA_TILE + 1], vs[LA_D][LA_TILE + 1];
  __shared__ float ctx[LA_D][LA_D + 1];
  const int h = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const size_t hid = (size_t)heads * LA_D;
  const float *q = qkv + ((size_t)b * 3 * hid + (size_t)h * LA_D) * n;
  const float *k = q + hid * n;
  const float *v = k + hid * n;
  la_row_stats(k, n, rmax, rinv);
  __syncthreads();
  // ctx[d][e]: thread owns d = t / 8, e = (t % 8) * 4 .. +3
  const int cd = t >> 3, ce = (t & 7) * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int n0 = 0; n0 < n; n0 += LA_TILE) {
    const int tn = min(LA_TILE, n - n0);
    for (int e = t; e < LA_D * LA_TILE; e += LA_T) {
      const int r = e / LA_TILE, c = e % LA_TILE;
      const bool ok = c < tn;
      ks[r][c] = ok ? expf(k[(size_t)r * n + n0 + c] - rmax[r]) * rinv[r] : 0.0f;
      vs[r][c] = ok ? v[(size_t)r * n + n0 + c] : 0.0f;
    }
    __syncthreads();
    for (int c = 0; c < tn; ++c) {
      const float kk = ks[cd][c];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __fmaf_rn(kk, vs[ce + i][c], acc[i]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) ctx[cd][ce + i] = acc[i];
  __syncthreads();
  if (ctx_out) {  // saved for the backward pass
    float *co = ctx_out + ((size_t)b * heads + h) * LA_D * LA_D;
    for (int e = t; e < LA_D * LA_D; e += LA_T) co[e] = ctx[e / LA_D][e % LA_D];
  }
  // out[e][n]: lane = token (coalesced), wave w owns e = w*8 .. w*8+7
  float *o = out + ((size_t)b * hid + (size_t)h * LA_D) * n;
  const int w = t >> 6, l = t & 63;
  for (int n0 = 0; n0 < n; n0 += 64) {
    const int c = n0 + l;
    if (c < n) {
      float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int d = 0; d < LA_D; ++d) {
        const float qq = q[(size_t)d * n + c];
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = __fmaf_rn(ctx[d][w * 8 + i], qq, r[i]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) o[(size_t)(w * 8 + i) * n + c] = r[i];
    }
  }
}

// backward: given g = dL/dout f32[b, heads*32, n] (+ qkv and the saved ctx) -> dqkv f32[b, 3*heads*32, n]
//   dq[d,n]  = sum_e ctx[d][e] g[e,n]
//   dctx[d][e] = sum_n q[d,n] g[e,n]
//   dv[e,n]  = sum_d dctx[d][e] ks[d,n]
//   dks[d,n] = sum_e dctx[d][e] v[e,n];   dk[d,n] = ks[d,n] (dks[d,n] - sum_m ks[d,m] dks[d,m])
__global__ __launch_bounds__(LA_T) void linear_attention_bwd_kernel(int heads, int n, const float *__restrict__ qkv,
                                                                    const float *__restrict__ ctx_in,
                                                                    const float *__restrict__ g,
                                                                    float *__restrict__ dqkv) {
  __shared__ float rmax[LA_D], rinv[LA_D], rdot[LA_D];
  __shared__ float as[LA_D][LA_TILE + 1], bs[LA_D][LA_TILE + 1];
  __shared__ float ctx[LA_D][LA_D + 1], dctx[LA_D][LA_D + 1];
  const int h = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const size_t hid = (size_t)heads * LA_D;
  const float *q = qkv + ((size_t)b * 3 * hid + (size_t)h * LA_D) * n;
  const float *k = q + hid * n;
  const float *v = k + hid * n;
  const float *go = g + ((size_t)b * hid + (size_t)h * LA_D) * n;
  float *dq = dqkv + ((size_t)b * 3 * hid + (size_t)h * LA_D) * n;
  float *dk = dq + hid * n;
  float *dv = dk + hid * n;
  la_row_stats(k, n, rmax, rinv);
  {
    const float *ci = ctx_in + ((size_t)b * heads + h) * LA_D * LA_D;
    for (int e = t; e < LA_D * LA_D; e += LA_T) ctx[e / LA_D][e % LA_D] = ci[e];
  }
  __syncthreads();
  const int cd = t >> 3, ce = (t & 7) * 4;
  // dctx[d][e] = sum_n q[d,n] g[e,n]
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int n0 = 0; n0 < n; n0 += LA_TILE) {
    const int tn = min(LA_TILE, n - n0);
    for (int e = t; e < LA_D * LA_TILE; e += LA_T) {
      const int r = e / LA_TILE, c = e % LA_TILE;
      const bool ok = c < tn;
      as[r][c] = ok ? q[(size_t)r * n + n0 + c] : 0.0f;
      bs[r][c] = ok ? go[(size_t)r * n + n0 + c] : 0.0f;
    }
    __syncthreads();
    for (int c = 0; c < tn; ++c) {
      const float qq = as[cd][c];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __fmaf_rn(qq, bs[ce + i][c], acc[i]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) dctx[cd][ce + i] = acc[i];
  __syncthreads();
  const int w = t >> 6, l = t & 63;
  // rdot[d] = sum_n ks[d,n] dks[d,n] with dks[d,n] = sum_e dctx[d][e] v[e,n]; wave w owns d = w*8 .. +7
  {
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = l; c < n; c += 64) {
      float dks[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int e = 0; e < LA_D; ++e) {
        const float vv = v[(size_t)e * n + c];
#pragma unroll
        for (int i = 0; i < 8; ++i) dks[i] = __fmaf_rn(dctx[w * 8 + i][e], vv, dks[i]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int d = w * 8 + i;
        part[i] = __fmaf_rn(expf(k[(size_t)d * n + c] - rmax[d]) * rinv[d], dks[i], part[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float s = wave_sum(part[i]);
      if (l == 0) rdot[w * 8 + i] = s;
    }
  }
  __syncthreads();
  for (int n0 = 0; n0 < n; n0 += 64) {
    const int c = n0 + l;
    if (c >= n) continue;
    float rq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // dq[d = w*8+i]
    float rk[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // dks[d = w*8+i]
    for (int e = 0; e < LA_D; ++e) {
      const float gg = go[(size_t)e * n + c], vv = v[(size_t)e * n + c];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        rq[i] = __fmaf_rn(ctx[w * 8 + i][e], gg, rq[i]);
        rk[i] = __fmaf_rn(dctx[w * 8 + i][e], vv, rk[i]);
      }
    }
    float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // dv[e = w*8+i]
    for (int d = 0; d < LA_D; ++d) {
      const float kk = expf(k[(size_t)d * n + c] - rmax[d]) * rinv[d];
#pragma unroll
      for (int i = 0; i < 8; ++i) rv[i] = __fmaf_rn(dctx[d][w * 8 + i], kk, rv[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int d = w * 8 + i;
      const float kk = expf(k[(size_t)d * n + c] - rmax[d]) * rinv[d];
      dq[(size_t)d * n + c] = rq[i];
      dk[(size_t)d * n + c] = kk * (rk[i] - rdot[d]);
      dv[(size_t)d * n + c] = rv[i];
    }
  }
}

extern "C" int p2pb_linear_attention_forward(int b, int heads, int dim_head, int n, const float *qkv, float *out,
                                             float *ctx, void *stream) {
  if (b <= 0 || heads <= 0 || n <= 0 || dim_head != LA_D || !qkv || !out) return P2PB_EINVAL;
  hipLaunchKernelGGL(linear_attention_fwd_kernel, dim3(heads, b), dim3(LA_T), 0, (hipStream_t)stream, heads, n, qkv,
                     out, ctx);
  return p2pb_launch_status();
}

extern "C" int p2pb_linear_attention_backward(int b, int heads, int dim_head, int n, const float *qkv,
                                              const float *ctx, const float *grad_out, float *grad_qkv,
                                              void *stream) {
  if (b <= 0 || heads <= 0 || n <= 0 || dim_head != LA_D || !qkv || !ctx || !grad_out || !grad_qkv)
    return P2PB_EINVAL;
  hipLaunchKernelGGL(linear_attention_bwd_kernel, dim3(heads, b), dim3(LA_T), 0, (hipStream_t)stream, heads, n, qkv,
                     ctx, grad_out, grad_qkv);
  return p2pb_launch_status();
}
