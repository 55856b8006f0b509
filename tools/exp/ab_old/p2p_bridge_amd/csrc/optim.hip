// optim.hip -- the tail of a training step as three launches: clip_grad_norm_ + Adam / AdamW over every parameter tensor.
//
// The reference's step is  clip_grad_norm_(model.parameters(), 1.0); optimizer.step()  (train.py:127-133) with
// torch.optim.AdamW(lr 3e-4, betas 0.9 / 0.999, weight_decay 1e-5) from models/model_loader.py:13-33; in PyTorch that is
// one norm per tensor, a stack, a norm of norms, a multiply per tensor and a dozen multi-tensor launches of AdamW --
// 3.3 ms of the 20.6 ms captured step at BASELINE config 3's shape for 26 M parameters whose update moves 0.85 GB
// (0.15 ms at the HBM rate). Here: a table of (param, grad, exp_avg, exp_avg_sq, numel) per tensor and a table of
// (tensor, start) per 8192-element chunk, both in HBM; one workgroup per chunk.
//   1. optim_sumsq_kernel     partial[chunk] = sum g^2 (fp32 per lane, fixed tree, double across the workgroup)
//   2. optim_finish_kernel    one workgroup: total norm (double, fixed order), clip coefficient min(1, max_norm / (norm + 1e-6)),
//                             finite flag, step += 1, bias corrections 1 - beta^step -> ctl[]
//   3. optim_update_kernel    g *= coef (written back: p.grad after the step is the clipped gradient, as in the reference),
//                             then torch's single-tensor AdamW arithmetic in its order (_single_tensor_adam):
//                               p *= 1 - lr * wd                         (AdamW; Adam: g += wd * p)
//                               m += (g - m) * (1 - beta1)               (lerp_)
//                               v = v * beta2 + ((1 - beta2) * g) * g    (mul_ / addcmul_)
//                               p += -(lr / bc1) * (m / (sqrt(v) / sqrt(bc2) + eps))     (addcdiv_)
//                             skipped entirely (gradients still scaled) when the norm is not finite and skip_nonfinite is
//                             set: the GradScaler of the reference's loop skips optimizer.step() then (train.py:124-131).
// Everything the host changes between steps (lr) is read from ctl[], so a captured hipGraph of the step stays valid.
#include "common.h"

#define OPT_CHUNK 8192
#define OPT_THREADS 256
// ctl layout (floats / doubles share the buffer as 16 doubles): see p2pb_hip.h p2pb_optim_*
#define CTL_STEP 0   // number of updates applied so far
#define CTL_LR 1     // learning rate of the next update (host-written)
#define CTL_NORM 2   // total gradient norm of the last step (before clipping)
#define CTL_COEF 3   // clip coefficient applied
#define CTL_SKIP 4   // 1.0 when the last update was skipped (non-finite norm)
#define CTL_BC1 5    // 1 - beta1^step
#define CTL_BC2S 6   // sqrt(1 - beta2^step)

struct OptEntry {
  float *p, *g, *m, *v;
  long n;
};

__device__ __forceinline__ double opt_block_sum(double x, double *red) {
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = x;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < OPT_THREADS / 64; ++i) s += red[i];
  return s;  // thread 0 only
}

__global__ __launch_bounds__(OPT_THREADS) void optim_sumsq_kernel(const OptEntry *__restrict__ tab, const int2 *__restrict__ chunks,
                                                                 double *__restrict__ partial) {
  __shared__ double red[OPT_THREADS / 64];
  const int2 c = chunks[blockIdx.x];
  const OptEntry e = tab[c.x];
  const long start = (long)c.y * OPT_CHUNK;
  const long len = min((long)OPT_CHUNK, e.n - start);
  const float *g = e.g + start;
  float s = 0.f;
  if ((((uintptr_t)g) & 15) == 0) {
    const long nq = len >> 2;
    for (long i = threadIdx.x; i < nq; i += OPT_THREADS) {
      const float4 q = ((const float4 *)g)[i];
      s += q.x * q.x;
      s += q.y * q.y;
      s += q.z * q.z;
      s += q.w * q.w;
    }
    for (long i = (nq << 2) + threadIdx.x; i < len; i += OPT_THREADS) s += g[i] * g[i];
  } else {
    for (long i = threadIdx.x; i < len; i += OPT_THREADS) s += g[i] * g[i];
  }
  const double t = opt_block_sum((double)s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(OPT_THREADS) void optim_finish_kernel(int nchunks, const double *__restrict__ partial, double *__restrict__ ctl,
                                                                  double max_norm, double beta1, double beta2, int skip_nonfinite,
                                                                  unsigned *__restrict__ amax, int ntensors) {
  __shared__ double red[OPT_THREADS / 64];
  if (amax)  // the update kernel's workgroups add the updated tensors' max |w| to these (atomicMax on the bits)
    for (int i = threadIdx.x; i < ntensors; i += OPT_THREADS) amax[i] = 0u;
  double s = 0.0;
  for (int i = threadIdx.x; i < nchunks; i += OPT_THREADS) s += partial[i];
  const double t = opt_block_sum(s, red);
  if (threadIdx.x == 0) {
    const double norm = sqrt(t);
    const bool finite = isfinite(norm);
    double coef = 1.0;
    if (max_norm > 0.0) {  // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1 (fp32 there)
      coef = (double)((float)max_norm / ((float)norm + 1e-6f));
      if (coef > 1.0) coef = 1.0;
    }
    const bool skip = skip_nonfinite && !finite;
    const double step = ctl[CTL_STEP] + (skip ? 0.0 : 1.0);
    ctl[CTL_STEP] = step;
    ctl[CTL_NORM] = norm;
    ctl[CTL_COEF] = coef;
    ctl[CTL_SKIP] = skip ? 1.0 : 0.0;
    ctl[CTL_BC1] = 1.0 - pow(beta1, step);
    ctl[CTL_BC2S] = sqrt(1.0 - pow(beta2, step));
  }
}

template <bool DECOUPLED>
__device__ __forceinline__ void opt_update(float &p, float &g, float &m, float &v, float coef, float decay, float wd, float omb1, float b2,
                                           float omb2, float eps, float step_size, float bc2s, bool skip, bool clip) {
  if (clip) g = g * coef;
  if (skip) return;
  float gg = g;
  if (DECOUPLED)
    p = p * decay;  // decay = 1 - lr * wd, evaluated in double as Python does
  else if (wd != 0.f)
    gg = __fmaf_rn(wd, p, gg);
  m = __fmaf_rn(omb1, gg - m, m);        // lerp_: self + weight * (end - self), contracted as ATen's kernel is
  v = __fmaf_rn(omb2 * gg, gg, v * b2);  // mul_(beta2) then addcmul_: self + (value * t1) * t2
  const float denom = sqrtf(v) / bc2s + eps;
  p = __fmaf_rn(-step_size, m / denom, p);  // addcdiv_: self + value * (t1 / t2)
}

template <bool DECOUPLED>
__global__ __launch_bounds__(OPT_THREADS) void optim_update_kernel(const OptEntry *__restrict__ tab, const int2 *__restrict__ chunks,
                                                                  const double *__restrict__ ctl, double wd_d, float omb1, float b2, float omb2,
                                                                  float eps, int clip, unsigned *__restrict__ amax) {
  __shared__ float wmax[OPT_THREADS / 64];
  float pmax = 0.0f;  // max |p| of this chunk after the update: the weight packs of the next forward take their
                      // per-tensor fp16 scale from it (p2pb_*_pack_weights_split_amax) instead of a reduction launch each
  const int2 c = chunks[blockIdx.x];
  const OptEntry e = tab[c.x];
  const long start = (long)c.y * OPT_CHUNK;
  const long len = min((long)OPT_CHUNK, e.n - start);
  float *p = e.p + start, *g = e.g + start, *m = e.m + start, *v = e.v + start;
  const float coef = (float)ctl[CTL_COEF], wd = (float)wd_d;
  const bool skip = ctl[CTL_SKIP] != 0.0;
  const float decay = (float)(1.0 - ctl[CTL_LR] * (double)wd_d);
  const float step_size = (float)(ctl[CTL_LR] / ctl[CTL_BC1]);  // torch: lr / bias_correction1 in double, then fp32 `value`
  const float bc2s = (float)ctl[CTL_BC2S];
  const bool al = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
  long done = 0;
  if (al) {
    const long nq = len >> 2;
    for (long i = threadIdx.x; i < nq; i += OPT_THREADS) {
      float4 P = ((float4 *)p)[i], G = ((float4 *)g)[i], M = ((float4 *)m)[i], V = ((float4 *)v)[i];
      opt_update<DECOUPLED>(P.x, G.x, M.x, V.x, coef, decay, wd, omb1, b2, omb2, eps, step_size, bc2s, skip, clip);
      opt_update<DECOUPLED>(P.y, G.y, M.y, V.y, coef, decay, wd, omb1, b2, omb2, eps, step_size, bc2s, skip, clip);
      opt_update<DECOUPLED>(P.z, G.z, M.z, V.z, coef, decay, wd, omb1, b2, omb2, eps, step_size, bc2s, skip, clip);
      opt_update<DECOUPLED>(P.w, G.w, M.w, V.w, coef, decay, wd, omb1, b2, omb2, eps, step_size, bc2s, skip, clip);
      if (clip) ((float4 *)g)[i] = G;
      if (!skip) {
        ((float4 *)p)[i] = P;
        ((float4 *)m)[i] = M;
        ((float4 *)v)[i] = V;
      }
      pmax = fmaxf(fmaxf(pmax, fmaxf(fabsf(P.x), fabsf(P.y))), fmaxf(fabsf(P.z), fabsf(P.w)));
    }
    done = nq << 2;
  }
  for (long i = done + threadIdx.x; i < len; i += OPT_THREADS) {
    float P = p[i], G = g[i], M = m[i], V = v[i];
    opt_update<DECOUPLED>(P, G, M, V, coef, decay, wd, omb1, b2, omb2, eps, step_size, bc2s, skip, clip);
    if (clip) g[i] = G;
    if (!skip) {
      p[i] = P;
      m[i] = M;
      v[i] = V;
    }
    pmax = fmaxf(pmax, fabsf(P));
  }
  if (amax) {  // (fmaxf drops a NaN operand, as the packs' own reduction does)
    for (int o = 32; o > 0; o >>= 1) pmax = fmaxf(pmax, __shfl_xor(pmax, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = pmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float w = wmax[0];
      for (int i = 1; i < OPT_THREADS / 64; ++i) w = fmaxf(w, wmax[i]);
      atomicMax(amax + c.x, __builtin_bit_cast(unsigned, w));
    }
  }
}

extern "C" {

size_t p2pb_optim_entry_bytes(void) { return sizeof(OptEntry); }
int p2pb_optim_chunk(void) { return OPT_CHUNK; }

int p2pb_optim_clip_adam_step(int nchunks, const void *table, const int *chunks, double *partial, double *ctl, double max_norm,
                              double beta1, double beta2, double eps, double weight_decay, int decoupled, int skip_nonfinite,
                              unsigned *amax, int ntensors, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (nchunks <= 0) return 0;
  if (!table || !chunks || !partial || !ctl || (amax && ntensors <= 0)) return (int)hipErrorInvalidValue;
  optim_sumsq_kernel<<<nchunks, OPT_THREADS, 0, stream>>>((const OptEntry *)table, (const int2 *)chunks, partial);
  optim_finish_kernel<<<1, OPT_THREADS, 0, stream>>>(nchunks, partial, ctl, max_norm, beta1, beta2, skip_nonfinite, amax, ntensors);
  const double wd = weight_decay;
  const float omb1 = (float)(1.0 - beta1), b2 = (float)beta2, omb2 = (float)(1.0 - beta2), e = (float)eps;
  if (decoupled)
    optim_update_kernel<true><<<nchunks, OPT_THREADS, 0, stream>>>((const OptEntry *)table, (const int2 *)chunks, ctl, wd, omb1, b2, omb2, e,
                                                                    max_norm > 0.0, amax);
  else
    optim_update_kernel<false><<<nchunks, OPT_THREADS, 0, stream>>>((const OptEntry *)table, (const int2 *)chunks, ctl, wd, omb1, b2, omb2, e,
                                                                     max_norm > 0.0, amax);
  return p2pb_launch_status();
}
}
