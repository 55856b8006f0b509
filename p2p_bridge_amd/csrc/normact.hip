// normact.hip -- GroupNorm (+ AdaGN style) (+ Swish) for TRAINING: the backward pass of
//     y = act( GN_groups(x) * gamma_c + beta_c ) * factor_bc + bias_bc )           act = Swish or identity
// (models/modules.py:341-358 AdaGN.forward, torch.nn.GroupNorm, Swish :14-19; SharedMLP's conv -> norm -> Swish triple,
// models/pvcnn.py:162-205) in two launches instead of the ~20 elementwise / reduction kernels the eager autograd graph
// runs per layer. The forward pass is the inference machinery: the producing convolution emits {sum, sum of squares}
// partials, gn_affine_kernel folds the norm to a per-(sample, channel) affine  u = A x + B  (p2pb_gn_affine_params_ex
// additionally returns the group mean / rstd this file needs), p2pb_affine_act applies it.
//
// Backward, with n = (x - mu_g) rstd_g the normalised value and gu = gy * act'(u):
//     per row (b,c):   S1 = sum_p gu ,  S2 = sum_p gu x      -> T1 = S1 , T2 = sum_p gu n = rstd (S2 - mu S1)
//     d bias_bc = T1 ,  d factor_bc = gamma_c T2 + beta_c T1 ,  d gamma_c = sum_b factor T2 ,  d beta_c = sum_b factor T1
//     m1_bg = sum_{c in g} gamma factor T1 / (cg P) ,  m2_bg = sum_{c in g} gamma factor T2 / (cg P)
//     dx = rstd (gu gamma factor - m1 - n m2)  =  gu A + x c2_bg + c3_bg ,  c2 = -rstd^2 m2 ,  c3 = -rstd m1 + mu rstd^2 m2
// All reductions run in a fixed order (deterministic), row sums in fp32 per thread + fp64 across threads.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoid_exact(float v) { return 1.0f / (1.0f + expf(-v)); }
// the FORWARD Swish of the library (pointwise.hip swishf: hardware exp2 / reciprocal; the same bits as p2pb_affine_act)
__device__ __forceinline__ float na_swishf(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}

// gu = gy * d act(u) / du
__device__ __forceinline__ float act_grad(float u, float gy, int swish) {
  if (!swish) return gy;
  const float s = sigmoid_exact(u);
  return gy * (s * (1.0f + u * (1.0f - s)));
}

// ---- what the training step folds into the norm's launches (round 6: every line below used to be 1-3 ATen launches per layer) ----
//   gmean   f32[b,c] | NULL : gradient of the row MEAN of the (activation-free) output, SE3d's squeeze input, which the forward pass
//                             takes from the statistics (p2pb_gn_affine_params_ex chmean): gy_eff = gy + gmean[b,c] / P
//   dropout (thresh != 0)   : y = act(u) * m, m = 1/(1-p) with probability 1-p, else 0 (nn.Dropout after the Swish,
//                             models/pvcnn.py:268-272). m is a counter-based hash of (element index, seed[0..1] from DEVICE memory,
//                             layer salt): the backward pass regenerates it, a captured step replays with fresh seeds (the seed
//                             tensor is drawn by torch's graph-safe generator once per forward pass)
//   residual (+ rgate)      : y += residual * rgate[b,c] (PVConv: point branch + devoxelised grid * SE gate, models/pvcnn.py:322-326
//                             with the gate moved behind the linear devoxelisation); backward: dres = gy * rgate, drgate = sum_p gy res
struct NaExtra {
  size_t gy_pitch;  // floats between two samples of gy (>= c * P: gy may be a channel slice of a wider tensor, read in place)
  int c;
  const float *gmean, *residual, *rgate;
  float *dres, *drgate;
  const unsigned *seed;
  unsigned salt, thresh;
  float keep_scale;
};
__device__ __forceinline__ unsigned na_hash(unsigned long long idx, unsigned k0, unsigned k1) {
  unsigned h = (unsigned)idx + (unsigned)(idx >> 32) * 0x85ebca6bu + k0;
  h ^= h >> 16;
  h *= 0x7feb352du;
  h ^= h >> 15;
  h *= 0x846ca68bu;
  h ^= h >> 16;
  h += k1;
  h ^= h >> 15;
  h *= 0x2c1b3c6du;
  h ^= h >> 12;
  return h;
}
// the dropout factor of element idx: keep_scale or 0
__device__ __forceinline__ float na_keep(unsigned long long idx, unsigned k0, unsigned k1, unsigned thresh, float keep_scale) {
  return na_hash(idx, k0, k1) >= thresh ? keep_scale : 0.0f;
}

__device__ __forceinline__ double block_sum_256(double v, double *sm) {
  const int t = threadIdx.x;
  sm[t] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) sm[t] += sm[t + s];
    __syncthreads();
  }
  const double r = sm[0];
  __syncthreads();
  return r;
}

// one workgroup per row (b,c): rows[bc] = {S1, S2}; ex.drgate[bc] = sum_p gy res
__global__ __launch_bounds__(256) void na_bwd_reduce_kernel(int P, const float *__restrict__ x,
                                                            const float *__restrict__ gy,
                                                            const float *__restrict__ scale,
                                                            const float *__restrict__ shift, int swish,
                                                            float *__restrict__ rows, NaExtra ex) {
  __shared__ double sm[256];
  const int bc = blockIdx.x, t = threadIdx.x;
  const float sc = scale[bc], sh = shift[bc];
  const float *xr = x + (size_t)bc * P, *gr = gy + (size_t)(bc / ex.c) * ex.gy_pitch + (size_t)(bc % ex.c) * P;
  const float *rr = ex.drgate ? ex.residual + (size_t)bc * P : nullptr;
  const float gm = ex.gmean ? ex.gmean[bc] / (float)P : 0.0f;
  const unsigned k0 = ex.thresh ? ex.seed[0] ^ (ex.salt * 0x9e3779b9u) : 0u, k1 = ex.thresh ? ex.seed[1] : 0u;
  const unsigned long long e0 = (unsigned long long)bc * (unsigned)P;
  float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  if ((P & 3) == 0) {
    const f32x4 *x4 = (const f32x4 *)xr, *g4 = (const f32x4 *)gr, *r4 = (const f32x4 *)rr;
    for (int p = t; p < P / 4; p += 256) {
      const f32x4 xv = x4[p], gv = g4[p];
      if (rr) {
        const f32x4 rv = r4[p];
#pragma unroll
        for (int i = 0; i < 4; ++i) s3 += gv[i] * rv[i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float g = gv[i];
        if (ex.gmean) g += gm;
        if (ex.thresh) g *= na_keep(e0 + 4 * p + i, k0, k1, ex.thresh, ex.keep_scale);
        const float gu = act_grad(xv[i] * sc + sh, g, swish);
        s1 += gu;
        s2 += gu * xv[i];
      }
    }
  } else {
    for (int p = t; p < P; p += 256) {
      const float xv = xr[p];
      float g = gr[p];
      if (rr) s3 += g * rr[p];
      if (ex.gmean) g += gm;
      if (ex.thresh) g *= na_keep(e0 + p, k0, k1, ex.thresh, ex.keep_scale);
      const float gu = act_grad(xv * sc + sh, g, swish);
      s1 += gu;
      s2 += gu * xv;
    }
  }
  const double a = block_sum_256((double)s1, sm), b2 = block_sum_256((double)s2, sm);
  if (t == 0) {
    rows[(size_t)bc * 2] = (float)a;
    rows[(size_t)bc * 2 + 1] = (float)b2;
  }
  if (ex.drgate) {
    const double c3 = block_sum_256((double)s3, sm);
    if (t == 0) ex.drgate[bc] = (float)c3;
  }
}

// dx = gu * A + x * c2 + c3, with the parameter pass folded into the prologue (it used to be a launch of its own: eight
// workgroups of serial fp64 arithmetic, 8 us of latency per layer on the step's critical path). Every workgroup of row
// (b, ch) forms the two group coefficients of (b, g) itself -- thread k < cg evaluates channel g*cg + k's terms in fp64,
// thread 0 adds them in ascending channel order (the order of the old kernel: same bits) -- and the first workgroup of the
// row also writes the row's parameter gradients: dstyle[b, ch], and for b == 0 dgamma / dbeta summed over the samples in
// ascending order.
__global__ __launch_bounds__(256) void na_bwd_apply_kernel(int nb, int c, int groups, int P, const float *__restrict__ x,
                                                           const float *__restrict__ gy,
                                                           const float *__restrict__ scale,
                                                           const float *__restrict__ shift, int swish,
                                                           const float *__restrict__ rows,
                                                           const float *__restrict__ mean_rstd,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ beta,
                                                           const float *__restrict__ style, int style_stride,
                                                           float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                           float *__restrict__ dstyle, float *__restrict__ dx, NaExtra ex) {
  __shared__ double w1[256], w2[256];
  __shared__ float cf[2];
  const int bc = blockIdx.y, b = bc / c, ch = bc % c, cg = c / groups, g = ch / cg, t = threadIdx.x;
  const double mu = (double)mean_rstd[((size_t)b * groups + g) * 2], rstd = (double)mean_rstd[((size_t)b * groups + g) * 2 + 1];
  if (t < cg) {
    const int k = g * cg + t;
    const double ga = gamma ? (double)gamma[k] : 1.0;
    const double s1 = (double)rows[((size_t)b * c + k) * 2], s2 = (double)rows[((size_t)b * c + k) * 2 + 1];
    const double f = style ? (double)style[(size_t)b * style_stride + k] : 1.0;
    w1[t] = ga * f * s1;
    w2[t] = ga * f * (rstd * (s2 - mu * s1));
  }
  __syncthreads();
  if (t == 0) {
    double m1 = 0.0, m2 = 0.0;
    for (int k = 0; k < cg; ++k) {
      m1 += w1[k];
      m2 += w2[k];
    }
    const double n = (double)P * cg;
    m1 /= n;
    m2 /= n;
    cf[0] = (float)(-rstd * rstd * m2);
    cf[1] = (float)(-rstd * m1 + mu * rstd * rstd * m2);
  }
  if (blockIdx.x == 0 && t == 64) {  // (a lane of the second wave: beside thread 0's serial sum)
    const double ga = gamma ? (double)gamma[ch] : 1.0, be = beta ? (double)beta[ch] : 0.0;
    if (style) {
      const double s1 = (double)rows[(size_t)bc * 2], s2 = (double)rows[(size_t)bc * 2 + 1];
      const double t2 = rstd * (s2 - mu * s1);
      dstyle[(size_t)b * 2 * c + ch] = (float)(ga * t2 + be * s1);  // d factor
      dstyle[(size_t)b * 2 * c + c + ch] = (float)s1;               // d bias
    }
    if (b == 0 && (dgamma || dbeta)) {
      double dga = 0.0, dbe = 0.0;
      for (int bb = 0; bb < nb; ++bb) {
        const double m = (double)mean_rstd[((size_t)bb * groups + g) * 2], r = (double)mean_rstd[((size_t)bb * groups + g) * 2 + 1];
        const double s1 = (double)rows[((size_t)bb * c + ch) * 2], s2 = (double)rows[((size_t)bb * c + ch) * 2 + 1];
        const double f = style ? (double)style[(size_t)bb * style_stride + ch] : 1.0;
        dga += f * (r * (s2 - m * s1));
        dbe += f * s1;
      }
      if (dgamma) dgamma[ch] = (float)dga;
      if (dbeta) dbeta[ch] = (float)dbe;
    }
  }
  __syncthreads();
  const float sc = scale[bc], sh = shift[bc];
  const float c2 = cf[0], c3 = cf[1];
  const float *xr = x + (size_t)bc * P, *gr = gy + (size_t)b * ex.gy_pitch + (size_t)ch * P;
  float *dr = dx + (size_t)bc * P;
  float *er = ex.dres ? ex.dres + (size_t)bc * P : nullptr;
  const float rg = (ex.dres && ex.rgate) ? ex.rgate[bc] : 1.0f;
  const float gm = ex.gmean ? ex.gmean[bc] / (float)P : 0.0f;
  const unsigned k0 = ex.thresh ? ex.seed[0] ^ (ex.salt * 0x9e3779b9u) : 0u, k1 = ex.thresh ? ex.seed[1] : 0u;
  const unsigned long long e0 = (unsigned long long)bc * (unsigned)P;
  if ((P & 3) == 0) {
    const f32x4 *x4 = (const f32x4 *)xr, *g4 = (const f32x4 *)gr;
    f32x4 *d4 = (f32x4 *)dr, *e4 = (f32x4 *)er;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P / 4; p += gridDim.x * 256) {
      const f32x4 xv = x4[p], gv = g4[p];
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float g = gv[i];
        if (ex.gmean) g += gm;
        if (ex.thresh) g *= na_keep(e0 + 4 * p + i, k0, k1, ex.thresh, ex.keep_scale);
        o[i] = act_grad(xv[i] * sc + sh, g, swish) * sc + xv[i] * c2 + c3;
      }
      d4[p] = o;
      if (er) e4[p] = gv * rg;
    }
  } else {
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) {
      const float xv = xr[p];
      float g = gr[p];
      if (er) er[p] = g * rg;
      if (ex.gmean) g += gm;
      if (ex.thresh) g *= na_keep(e0 + p, k0, k1, ex.thresh, ex.keep_scale);
      dr[p] = act_grad(xv * sc + sh, g, swish) * sc + xv * c2 + c3;
    }
  }
}

static unsigned na_thresh(float drop_p) {
  const double t = (double)drop_p * 4294967296.0;
  return t <= 0.0 ? 0u : t >= 4294967295.0 ? 0xffffffffu : (unsigned)t;
}

// The same with the step's neighbours folded in (struct NaExtra above): gy_pitch = floats between two samples of gy (0 = c * npos;
// a channel slice of a concatenation's gradient is read in place), gmean f32[b,c] | NULL (requires swish == 0 and no
// dropout: it is the gradient of the mean of the activation-free output); residual f32[b,c,npos] + rgate f32[b,c] | NULL with
// outputs dres f32[b,c,npos] (NULL without a gate: the residual's gradient is gy itself) and drgate f32[b,c]; dropout
// 0 <= drop_p < 1 with seed = two 32-bit words in DEVICE memory and the layer's salt (the forward pass's values).
extern "C" int p2pb_norm_act_backward_ex(int b, int c, int groups, int npos, const float *x, const float *gy,
                                         const float *scale, const float *shift, const float *mean_rstd,
                                         const float *gamma, const float *beta, const float *style, int style_stride,
                                         int swish, long gy_pitch, const float *gmean, const float *residual, const float *rgate,
                                         float drop_p, const unsigned *seed, unsigned salt, float *dx, float *dgamma,
                                         float *dbeta, float *dstyle, float *dres, float *drgate, float *ws, void *stream) {
  if (b <= 0 || c <= 0 || groups <= 0 || c % groups != 0 || c / groups > 256 || npos <= 0 || !x || !gy || !scale ||
      !shift || !mean_rstd || !dx || !ws || (style && (!dstyle || style_stride < 2 * c)))
    return P2PB_EINVAL;
  if (gy_pitch == 0) gy_pitch = (long)c * npos;
  if (gy_pitch < (long)c * npos || ((npos & 3) == 0 && (gy_pitch & 3) != 0)) return P2PB_EINVAL;
  if (!(drop_p >= 0.0f && drop_p < 1.0f) || (drop_p > 0.0f && !seed) || (gmean && (swish || drop_p > 0.0f)) ||
      ((dres || drgate) && (!rgate || !residual || !dres || !drgate)))
    return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float *rows = ws;
  NaExtra ex = {};
  ex.gy_pitch = (size_t)gy_pitch, ex.c = c;
  ex.gmean = gmean, ex.residual = residual, ex.rgate = rgate, ex.dres = dres, ex.drgate = drgate, ex.seed = seed, ex.salt = salt;
  ex.thresh = na_thresh(drop_p), ex.keep_scale = 1.0f / (1.0f - drop_p);
  hipLaunchKernelGGL(na_bwd_reduce_kernel, dim3(b * c), dim3(256), 0, s, npos, x, gy, scale, shift, swish, rows, ex);
  const int per = (npos & 3) == 0 ? npos / 4 : npos;
  const unsigned gx = (unsigned)((per + 255) / 256 > 32 ? 32 : (per + 255) / 256);
  hipLaunchKernelGGL(na_bwd_apply_kernel, dim3(gx, b * c), dim3(256), 0, s, b, c, groups, npos, x, gy, scale, shift, swish,
                     rows, mean_rstd, gamma, beta, style, style_stride, dgamma, dbeta, dstyle, dx, ex);
  return p2pb_launch_status();
}

// x, gy f32[b,c,npos]; scale, shift f32[b,c] and mean_rstd f32[b,groups,2] from p2pb_gn_affine_params_ex; gamma, beta
// f32[c] or NULL; style rows (factor[c] | bias[c]) with pitch style_stride, or NULL. Outputs: dx f32[b,c,npos],
// dgamma / dbeta f32[c] (may be NULL), dstyle f32[b,2c] (required iff style). ws: 2*b*c + 2*b*groups floats.
extern "C" int p2pb_norm_act_backward(int b, int c, int groups, int npos, const float *x, const float *gy,
                                      const float *scale, const float *shift, const float *mean_rstd,
                                      const float *gamma, const float *beta, const float *style, int style_stride,
                                      int swish, float *dx, float *dgamma, float *dbeta, float *dstyle, float *ws,
                                      void *stream) {
  return p2pb_norm_act_backward_ex(b, c, groups, npos, x, gy, scale, shift, mean_rstd, gamma, beta, style, style_stride, swish, 0,
                                   nullptr, nullptr, nullptr, 0.0f, nullptr, 0u, dx, dgamma, dbeta, dstyle, nullptr, nullptr, ws,
                                   stream);
}

// y = drop(act(x * scale[b,c] + shift[b,c])) + residual * rgate[b,c]: the forward pass of the folded norm in train() with the same
// neighbours folded in (p2pb_affine_act is the plain form). residual / rgate may be NULL (rgate alone is ignored).
template <int V>
static __global__ __launch_bounds__(256) void affine_act_train_kernel(int P, const float *__restrict__ x,
                                                                      const float *__restrict__ scale,
                                                                      const float *__restrict__ shift, int swish,
                                                                      float *__restrict__ y, NaExtra ex) {
  typedef float vec __attribute__((ext_vector_type(V)));
  const int bc = blockIdx.y;
  const float sc = scale[bc], sh = shift[bc];
  const vec *xr = (const vec *)(x + (size_t)bc * P);
  const vec *rr = ex.residual ? (const vec *)(ex.residual + (size_t)bc * P) : nullptr;
  vec *yr = (vec *)(y + (size_t)bc * P);
  const float rg = ex.rgate ? ex.rgate[bc] : 1.0f;
  const unsigned k0 = ex.thresh ? ex.seed[0] ^ (ex.salt * 0x9e3779b9u) : 0u, k1 = ex.thresh ? ex.seed[1] : 0u;
  const unsigned long long e0 = (unsigned long long)bc * (unsigned)P;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < P / V; p += gridDim.x * 256) {
    vec v = xr[p];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      float t = v[i] * sc + sh;
      if (swish) t = na_swishf(t);
      if (ex.thresh) t *= na_keep(e0 + (unsigned long long)V * p + i, k0, k1, ex.thresh, ex.keep_scale);
      v[i] = t;
    }
    if (rr) v += rr[p] * rg;
    yr[p] = v;
  }
}

extern "C" int p2pb_affine_act_train(int b, int c, int npos, const float *x, const float *scale, const float *shift, int swish,
                                     const float *residual, const float *rgate, float drop_p, const unsigned *seed,
                                     unsigned salt, float *y, void *stream) {
  if (b <= 0 || c <= 0 || npos <= 0 || !x || !scale || !shift || !y || !(drop_p >= 0.0f && drop_p < 1.0f) ||
      (drop_p > 0.0f && !seed))
    return P2PB_EINVAL;
  NaExtra ex = {};
  ex.residual = residual, ex.rgate = residual ? rgate : nullptr, ex.seed = seed, ex.salt = salt;
  ex.thresh = na_thresh(drop_p), ex.keep_scale = 1.0f / (1.0f - drop_p);
  hipStream_t s = (hipStream_t)stream;
  if (npos % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15) == 0) {
    const int p4 = npos / 4;
    const unsigned gx = (unsigned)((p4 + 255) / 256 > 64 ? 64 : (p4 + 255) / 256);
    hipLaunchKernelGGL(affine_act_train_kernel<4>, dim3(gx, b * c), dim3(256), 0, s, npos, x, scale, shift, swish, y, ex);
  } else {
    const unsigned gx = (unsigned)((npos + 255) / 256 > 64 ? 64 : (npos + 255) / 256);
    hipLaunchKernelGGL(affine_act_train_kernel<1>, dim3(gx, b * c), dim3(256), 0, s, npos, x, scale, shift, swish, y, ex);
  }
  return p2pb_launch_status();
}

// ---- squeeze-excite gate for TRAINING (round 5) ------------------------------------------------------------------------------
// SE3d (models/modules.py:362-378): gate = sigmoid(W2 relu(W1 mean)), mean f32[b,c] the per-channel mean of the voxel grid, W1
// [hidden, c], W2 [c, hidden], no biases. Eager autograd runs two hipBLASLt GEMMs of a few hundred MACs, a ReLU and a sigmoid
// forward and about ten launches backward, per PVConv (eight per step); here: one launch forward, two backward. Fixed summation
// orders (deterministic): hidden rows by a wave with an xor tree, everything else ascending.
__global__ __launch_bounds__(256) void se_gate_fwd_kernel(int c, int hidden, const float *__restrict__ mean,
                                                          const float *__restrict__ w1, const float *__restrict__ w2,
                                                          float *__restrict__ hid, float *__restrict__ gate) {
  __shared__ float s[1024], h[128];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int k = t; k < c; k += 256) s[k] = mean[(size_t)b * c + k];
  __syncthreads();
  for (int j = wave; j < hidden; j += 4) {
    float a = 0.0f;
    for (int k = lane; k < c; k += 64) a = __fmaf_rn(w1[(size_t)j * c + k], s[k], a);
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m);
    if (lane == 0) {
      h[j] = a > 0.0f ? a : 0.0f;
      hid[(size_t)b * hidden + j] = h[j];
    }
  }
  __syncthreads();
  for (int k = t; k < c; k += 256) {
    float a = 0.0f;
    for (int j = 0; j < hidden; ++j) a = __fmaf_rn(w2[(size_t)k * hidden + j], h[j], a);
    gate[(size_t)b * c + k] = sigmoid_exact(a);
  }
}

// per sample: dz2 = dgate g (1 - g), dh = W2^T dz2, dz1 = dh [hid > 0], dmean = W1^T dz1; dz2 / dz1 also go to ws for the weights
__global__ __launch_bounds__(256) void se_gate_bwd_sample_kernel(int c, int hidden, const float *__restrict__ w1,
                                                                 const float *__restrict__ w2, const float *__restrict__ hid,
                                                                 const float *__restrict__ gate, const float *__restrict__ dgate,
                                                                 float *__restrict__ dmean, float *__restrict__ dz2o,
                                                                 float *__restrict__ dz1o) {
  __shared__ float z2[1024], z1[128];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int k = t; k < c; k += 256) {
    const float g = gate[(size_t)b * c + k];
    const float v = dgate[(size_t)b * c + k] * (g * (1.0f - g));
    z2[k] = v;
    dz2o[(size_t)b * c + k] = v;
  }
  __syncthreads();
  for (int j = wave; j < hidden; j += 4) {
    float a = 0.0f;
    for (int k = lane; k < c; k += 64) a = __fmaf_rn(w2[(size_t)k * hidden + j], z2[k], a);
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m);
    if (lane == 0) {
      const float v = hid[(size_t)b * hidden + j] > 0.0f ? a : 0.0f;
      z1[j] = v;
      dz1o[(size_t)b * hidden + j] = v;
    }
  }
  __syncthreads();
  for (int k = t; k < c; k += 256) {
    float a = 0.0f;
    for (int j = 0; j < hidden; ++j) a = __fmaf_rn(w1[(size_t)j * c + k], z1[j], a);
    dmean[(size_t)b * c + k] = a;
  }
}

// dW2[k][j] = sum_b dz2[b,k] hid[b,j] ,  dW1[j][k] = sum_b dz1[b,j] mean[b,k]   (b ascending)
__global__ __launch_bounds__(256) void se_gate_bwd_weights_kernel(int nb, int c, int hidden, const float *__restrict__ mean,
                                                                  const float *__restrict__ hid, const float *__restrict__ dz2,
                                                                  const float *__restrict__ dz1, float *__restrict__ dw1,
                                                                  float *__restrict__ dw2) {
  const int e = blockIdx.x * 256 + threadIdx.x, n = c * hidden;
  if (e >= 2 * n) return;
  float a = 0.0f;
  if (e < n) {  // dW2, row-major [c][hidden]
    const int k = e / hidden, j = e % hidden;
    for (int b = 0; b < nb; ++b) a = __fmaf_rn(dz2[(size_t)b * c + k], hid[(size_t)b * hidden + j], a);
    dw2[e] = a;
  } else {  // dW1, row-major [hidden][c]
    const int j = (e - n) / c, k = (e - n) % c;
    for (int b = 0; b < nb; ++b) a = __fmaf_rn(dz1[(size_t)b * hidden + j], mean[(size_t)b * c + k], a);
    dw1[e - n] = a;
  }
}

// mean f32[b,c], w1 f32[hidden,c], w2 f32[c,hidden] -> hid f32[b,hidden] (post-ReLU), gate f32[b,c]
extern "C" int p2pb_se_gate_forward(int b, int c, int hidden, const float *mean, const float *w1, const float *w2, float *hid,
                                    float *gate, void *stream) {
  if (b <= 0 || c <= 0 || c > 1024 || hidden <= 0 || hidden > 128 || !mean || !w1 || !w2 || !hid || !gate) return P2PB_EINVAL;
  hipLaunchKernelGGL(se_gate_fwd_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, c, hidden, mean, w1, w2, hid, gate);
  return p2pb_launch_status();
}
// + dgate f32[b,c] -> dmean f32[b,c], dw1 f32[hidden,c], dw2 f32[c,hidden]; ws: b * (c + hidden) floats
extern "C" int p2pb_se_gate_backward(int b, int c, int hidden, const float *mean, const float *w1, const float *w2,
                                     const float *hid, const float *gate, const float *dgate, float *dmean, float *dw1,
                                     float *dw2, float *ws, void *stream) {
  if (b <= 0 || c <= 0 || c > 1024 || hidden <= 0 || hidden > 128 || !mean || !w1 || !w2 || !hid || !gate || !dgate || !dmean ||
      !dw1 || !dw2 || !ws)
    return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float *dz2 = ws, *dz1 = ws + (size_t)b * c;
  hipLaunchKernelGGL(se_gate_bwd_sample_kernel, dim3(b), dim3(256), 0, s, c, hidden, w1, w2, hid, gate, dgate, dmean, dz2, dz1);
  hipLaunchKernelGGL(se_gate_bwd_weights_kernel, dim3((2 * c * hidden + 255) / 256), dim3(256), 0, s, b, c, hidden, mean, hid, dz2,
                     dz1, dw1, dw2);
  return p2pb_launch_status();
}

// ---- max over the last axis, forward and backward (training, round 6) ---------------------------------------------------------
// The neighbour max of a set abstraction (models/pvcnn.py:414: x[B,C,M,U].max(-1)) and Pnet2Stage's two global max-pools
// (:923,930: amax over the N points) were ATen reductions with an eq / mul / div / sum / copy backward each (0.3 ms of the
// config-3 step: profiles/r05b_train_aten_ops.txt). Here: rows of u contiguous floats, L lanes per row (L = 8 .. 64 by u), 16-byte
// loads where the rows allow, a shuffle reduction on (value, index) -- first index on ties, a NaN wins (it must reach the loss
// like torch's max lets it) --, and a backward that WRITES the whole gradient (gy at the arg-max, zero elsewhere: no fill launch).
__device__ __forceinline__ bool rm_better(float a, int ia, float b, int ib) {  // is (a, ia) the pick over (b, ib)?
  const bool na = a != a, nb = b != b;
  if (na || nb) return na && (!nb || ia < ib);
  return a > b || (a == b && ia < ib);
}
template <int L>
static __global__ __launch_bounds__(256) void row_max_fwd_kernel(long rows, int u, const float *__restrict__ x,
                                                                 float *__restrict__ y, int *__restrict__ idx) {
  const long row = ((long)blockIdx.x * 256 + threadIdx.x) / L;
  const int l = threadIdx.x % L;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  if (row < rows) {
    const float *p = x + row * (long)u;
    if ((u & 3) == 0 && (((size_t)x) & 15) == 0) {
      for (int k = 4 * l; k < u; k += 4 * L) {
        const float4 v = *(const float4 *)(p + k);
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (rm_better(e[i], k + i, best, bi)) best = e[i], bi = k + i;
      }
    } else {
      for (int k = l; k < u; k += L)
        if (rm_better(p[k], k, best, bi)) best = p[k], bi = k;
    }
  }
#pragma unroll
  for (int o = L / 2; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (rm_better(ob, oi, best, bi)) best = ob, bi = oi;
  }
  if (row < rows && l == 0) {
    y[row] = best;
    idx[row] = bi;
  }
}
template <int L>
static __global__ __launch_bounds__(256) void row_max_bwd_kernel(long rows, int u, const float *__restrict__ gy,
                                                                 const int *__restrict__ idx, float *__restrict__ gx) {
  const long row = ((long)blockIdx.x * 256 + threadIdx.x) / L;
  const int l = threadIdx.x % L;
  if (row >= rows) return;
  const float g = gy[row];
  const int at = idx[row];
  float *p = gx + row * (long)u;
  if ((u & 3) == 0 && (((size_t)gx) & 15) == 0) {
    for (int k = 4 * l; k < u; k += 4 * L)
      *(float4 *)(p + k) = make_float4(k == at ? g : 0.0f, k + 1 == at ? g : 0.0f, k + 2 == at ? g : 0.0f, k + 3 == at ? g : 0.0f);
  } else {
    for (int k = l; k < u; k += L) p[k] = k == at ? g : 0.0f;
  }
}
static int rm_lanes(int u) { return u <= 32 ? 8 : u <= 64 ? 16 : u <= 128 ? 32 : 64; }
// x f32[rows, u] -> y f32[rows] = max over the row, idx i32[rows] = its first position
extern "C" int p2pb_row_max_forward(long rows, int u, const float *x, float *y, int *idx, void *stream) {
  if (rows <= 0 || u <= 0 || !x || !y || !idx) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int L = rm_lanes(u);
  const unsigned grid = (unsigned)((rows * L + 255) / 256);
  switch (L) {
    case 8: hipLaunchKernelGGL(row_max_fwd_kernel<8>, dim3(grid), dim3(256), 0, s, rows, u, x, y, idx); break;
    case 16: hipLaunchKernelGGL(row_max_fwd_kernel<16>, dim3(grid), dim3(256), 0, s, rows, u, x, y, idx); break;
    case 32: hipLaunchKernelGGL(row_max_fwd_kernel<32>, dim3(grid), dim3(256), 0, s, rows, u, x, y, idx); break;
    default: hipLaunchKernelGGL(row_max_fwd_kernel<64>, dim3(grid), dim3(256), 0, s, rows, u, x, y, idx); break;
  }
  return p2pb_launch_status();
}
// gy f32[rows], idx i32[rows] -> gx f32[rows, u], every element written
extern "C" int p2pb_row_max_backward(long rows, int u, const float *gy, const int *idx, float *gx, void *stream) {
  if (rows <= 0 || u <= 0 || !gy || !idx || !gx) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int L = rm_lanes(u);
  const unsigned grid = (unsigned)((rows * L + 255) / 256);
  switch (L) {
    case 8: hipLaunchKernelGGL(row_max_bwd_kernel<8>, dim3(grid), dim3(256), 0, s, rows, u, gy, idx, gx); break;
    case 16: hipLaunchKernelGGL(row_max_bwd_kernel<16>, dim3(grid), dim3(256), 0, s, rows, u, gy, idx, gx); break;
    case 32: hipLaunchKernelGGL(row_max_bwd_kernel<32>, dim3(grid), dim3(256), 0, s, rows, u, gy, idx, gx); break;
    default: hipLaunchKernelGGL(row_max_bwd_kernel<64>, dim3(grid), dim3(256), 0, s, rows, u, gy, idx, gx); break;
  }
  return p2pb_launch_status();
}
