// common.h -- shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p2pb_hip.h"

#define P2PB_WAVE 64

static inline int p2pb_launch_status() { return (int)hipGetLastError(); }
// arithmetic of the split-operand kernels: SPLIT_F16X3 (default) or SPLIT_BF16X6; defined in abi.hip, p2pb_set_split_terms
int p2pb_split_terms_now();  // the calling thread's override, else the process default (abi.hip)
#define p2pb_g_split_terms (p2pb_split_terms_now())

static inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

// Squared distance with the arithmetic contract of DESIGN.md: nvcc contracts
// dx*dx + dy*dy + dz*dz into  fma(dz,dz, fma(dy,dy, dx*dx)); the sources are compiled with
// -ffp-contract=off and the sequence is spelled out (identical in oracle/p2pb_oracle.c).
__device__ __forceinline__ float sqdist3(float dx, float dy, float dz) {
  return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx));
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int mbcnt(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// sum over each 32-lane half of the wave with DPP row shifts (VALU-rate, no LDS traffic); the result is
// valid in lanes 31 and 63 only. Summation order is fixed by the instruction sequence (deterministic).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_step(float v) {
  const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, o);
}
__device__ __forceinline__ float halfwave_sum_to_last(float v) {
  v = dpp_add_step<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add_step<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add_step<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add_step<0x118, 0xf>(v);  // row_shr:8  -> lane 15 of each 16-lane row = row sum
  v = dpp_add_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3 -> lanes 31 / 63 = half-wave sums
  return v;
}

// ---- fp32 operands on the bf16 matrix pipe (conv3d.hip "split-operand form", pointwise.hip) ----
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// (a, b) -> the three bf16 terms of each, packed as pairs (a in the low half): x = x0 + x1 + x2 with
// x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1), round-to-nearest-even (v_cvt_pk_bf16_f32); both
// residuals are exact in fp32, so the three terms carry 24+ significand bits
__device__ __forceinline__ void split3(float a, float b, unsigned &p0, unsigned &p1, unsigned &p2) {
  f32x2 v = {a, b};
  const bf16x2 q0 = __builtin_convertvector(v, bf16x2);
  v = v - __builtin_convertvector(q0, f32x2);
  const bf16x2 q1 = __builtin_convertvector(v, bf16x2);
  v = v - __builtin_convertvector(q1, f32x2);
  const bf16x2 q2 = __builtin_convertvector(v, bf16x2);
  p0 = __builtin_bit_cast(unsigned, q0);
  p1 = __builtin_bit_cast(unsigned, q1);
  p2 = __builtin_bit_cast(unsigned, q2);
}

// ---- fp16-pair split (SPLIT_F16X3): x*S = h0 + h1 with h0 = fp16(x*S), h1 = fp16(x*S - h0), round-to-nearest-even ----
// fp16 carries 11 significand bits, so two terms carry 22 and a product evaluated as h1*g0 + h0*g1 + h0*g0 (fp16 x fp16
// is exact in fp32) misses the exact one by <= 3 * 2^-22 |x*y| -- three matrix products instead of the six of the bf16
// split, at the price of fp16's exponent range. Activations are scaled by SPLIT_F16_SX; |x| < 16380 is the exact range.
// OUT OF RANGE IS LOUD (round 3; it used to clamp): the conversion is IEEE, so |x*S| >= 65520, an infinity or a NaN
// becomes h0 = +-inf / NaN, the residual h1 = x*S - h0 is -+inf / NaN too, and every output the operand reaches is
// non-finite -- exactly how an fp32 overflow shows in the reference, only earlier. Every layer's output feeds a
// GroupNorm, so the non-finite value spreads to the whole sample and P2PB.sample() / the training loss see it
// (p2pb.py: re-run on bf16x6 or raise). Below |x*S| = 2^-3 the low term is subnormal and the representation error
// is an ABSOLUTE 2^-25 / S (3.7e-9) instead of a relative 2^-22. Weights get a per-tensor power-of-two scale chosen at
// pack time from max |w| (any finite weights are in range); 1 / (S_x * S_w) is stored behind the packed weights and
// applied to the accumulators (exact: a power of two).
#define SPLIT_BF16X6 6
#define SPLIT_BF16X3 3
#define SPLIT_F16X3 16
#define SPLIT_F16_SX 4.0f
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2h(float a, float b, unsigned &p0, unsigned &p1) {
  f32x2 v = {a, b};
  const f16x2 q0 = __builtin_convertvector(v, f16x2);
  v = v - __builtin_convertvector(q0, f32x2);
  // (round 4: the residual as one v_fma_mix_f32 per element -- 4 instead of 5 instructions per pair, same bits -- measured
  //  0.3 % SLOWER on the widest GEMM and 2 % slower together with the pinned-register fma of pw_pp512.h; not kept)
  const f16x2 q1 = __builtin_convertvector(v, f16x2);
  p0 = __builtin_bit_cast(unsigned, q0);
  p1 = __builtin_bit_cast(unsigned, q1);
}
// the split of one staged pair in the arithmetic MODE (6 / 3: bf16 terms; 16: fp16 pair of the scaled value)
template <int MODE>
__device__ __forceinline__ void split_pair(float a, float b, unsigned &p0, unsigned &p1, unsigned &p2) {
  if constexpr (MODE == SPLIT_F16X3) {
    split2h(a * SPLIT_F16_SX, b * SPLIT_F16_SX, p0, p1);
    p2 = 0u;
  } else {
    split3(a, b, p0, p1, p2);
  }
}
// "f16x2w" PRICING EXPERIMENT (profiles/r06_f16x2w_ab.txt): activations stay an fp16 pair, a weight becomes ONE
// fp16 term (11 significand bits) -- two products instead of three. Numerics: P2PB_EXPERIMENT="x2w=1" makes the weight packs
// write a zero low plane (SPLIT_X2W_FLAG on the pack kernels' mode argument), so the shipped kernels compute exactly that
// arithmetic (the third product adds exact zeros) and every parity test can be run on it. Time: -DP2PB_X2W_TIMING builds drop
// the product with the weights' low plane from every split kernel (timing only: with ordinary packs the result is wrong).
// Verdict of the experiment: 1.0e-3-class network error against the 1e-4 gates -- not shipped, not selectable as a conv_math.
#define SPLIT_X2W_FLAG 0x100
#ifdef P2PB_X2W_TIMING
#define X2W_KEEP_LOW_WEIGHT_PRODUCT 0
#else
#define X2W_KEEP_LOW_WEIGHT_PRODUCT 1
#endif
constexpr __host__ __device__ int split_planes(int mode) { return mode == SPLIT_BF16X6 ? 3 : 2; }
template <int MODE>
__device__ __forceinline__ f32x16 split_mfma(const u32x4 &a, const u32x4 &b, const f32x16 &c) {
  if constexpr (MODE == SPLIT_F16X3)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// pack-time weight scale of the fp16 mode: the power of two that brings max |w| into [2^13, 2^14)
__device__ __forceinline__ float f16_weight_scale(float wmax) {
  if (!(wmax > 0.0f) || wmax > 3.0e38f) return 1.0f;  // all-zero, or an infinite weight (then the pack carries the inf)
  int e;
  (void)frexpf(wmax, &e);  // wmax = m * 2^e, m in [0.5, 1)
  return ldexpf(1.0f, 14 - e);
}
// max |w| of a tensor into *slot (uint bits of a non-negative float order like the float), slot zeroed before
static __global__ void absmax_bits_kernel(const float *__restrict__ w, size_t n, unsigned *__restrict__ slot) {
  __shared__ float part[4];
  float m = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  // one atomic per workgroup (they serialise on the one line: 256 of them were most of this kernel's 10 us)
  if (threadIdx.x == 0) atomicMax(slot, __builtin_bit_cast(unsigned, fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));
}
// workgroups for a tensor of n elements: 16 elements per thread, at most 64
static inline unsigned absmax_blocks(size_t n) {
  const size_t b = (n + 4095) / 4096;
  return (unsigned)(b < 1 ? 1 : b > 64 ? 64 : b);
}

// ---- GroupNorm(+AdaGN) finisher: the {sum, sum of squares} partials of a producing kernel -> per-(sample, channel) scale / shift ----
// (the arithmetic of gn_affine_kernel, conv3d.hip, which calls this too: one workgroup, >= 256 threads of it, per (sample, group);
//  fixed summation order, double accumulation: the same bits whichever kernel runs it.) Callers: gn_affine_kernel -- also the launch
// that pointwise.hip puts behind a producer whose caller armed a finisher (p2pb_gn_finisher_arm) -- and the small kernels that fold
// the norm into their own prologue (far_field_kernel, pvconv_tail_kernel, minmax_act_pool_kernel). Round 4's forms that ran it in
// the LAST workgroup of the producing GEMM (tickets, device-scope partials) measured slower than the launch they replaced
// (248.5 -> 255-260 ms per sample call) and left the library in round 5.
struct GnFinish {
  const float *gamma, *beta, *style;  // [c] | NULL, [c] | NULL, rows of (factor[c] | bias[c]) | NULL
  float *scale, *shift, *chmean;      // f32[b, c] outputs (chmean may be NULL); scale == NULL: no finisher
  double count_per_channel;
  int style_stride, groups;
  float eps;
};
typedef float gnf_f32x2 __attribute__((ext_vector_type(2)));
// lds: 4 x 256 doubles per 256-thread SLICE. Every thread of the workgroup calls it (same barriers); a slice is 256 consecutive
// threads with vt = the thread's index inside it, `live` = the slice has a group to finish (a 1024-thread workgroup finishes four
// groups at once: far_field_kernel, pvconv_tail_kernel). Outputs: f.scale / f.shift / f.chmean (global, each may be NULL) and the
// optional tables tab_* (f32[c] of THIS sample, e.g. in LDS, for a kernel that goes on to use the values itself).
__device__ __forceinline__ void gn_finish_group_v(int c, int nslots, const float *__restrict__ part, const GnFinish &f, int b, int g,
                                                  double *lds, int vt, bool live, float *mean_rstd = nullptr,
                                                  float *tab_scale = nullptr, float *tab_shift = nullptr, float *tab_mean = nullptr) {
  double *rs = lds, *rq = lds + 256, *chs = lds + 512, *chq = lds + 768;
  const int t = vt;
  const int cg = c / f.groups, g0 = g * cg;
  const int nt = 256 / cg;  // partial accumulators per channel
  const int k = t % cg, j = t / cg;
  // the channel's affine parameters are requested NOW (threads t < cg use them after two barriers: their latency hides behind the
  // partials' reduction instead of following it)
  float p_ga = 1.0f, p_be = 0.0f, p_fa = 1.0f, p_bi = 0.0f;
  if (live && t < cg) {
    const int ch = g0 + t;
    if (f.gamma) p_ga = f.gamma[ch];
    if (f.beta) p_be = f.beta[ch];
    if (f.style) {
      p_fa = f.style[(size_t)b * f.style_stride + ch];
      p_bi = f.style[(size_t)b * f.style_stride + c + ch];
    }
  }
  double s = 0.0, q = 0.0;
  if (live && j < nt) {  // (same order, four slots' loads in flight: the plain loop was one L2 round trip per slot)
    const float *p0 = part + ((size_t)b * nslots * c + g0 + k) * 2;
    const size_t pitch = (size_t)c * 2;
    int sl = j;
    for (; sl + 7 * nt < nslots; sl += 8 * nt) {  // (eight in flight where there are that many: round 5)
      gnf_f32x2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float *pp = p0 + (size_t)(sl + u * nt) * pitch;
        v[u] = *(const gnf_f32x2 *)pp;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s += (double)v[u][0];
        q += (double)v[u][1];
      }
    }
    for (; sl + 3 * nt < nslots; sl += 4 * nt) {
      gnf_f32x2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float *pp = p0 + (size_t)(sl + u * nt) * pitch;
        v[u] = *(const gnf_f32x2 *)pp;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s += (double)v[u][0];
        q += (double)v[u][1];
      }
    }
    for (; sl < nslots; sl += nt) {
      const float *p = p0 + (size_t)sl * pitch;
      s += (double)p[0];
      q += (double)p[1];
    }
  }
  if (live) {
    rs[t] = s;
    rq[t] = q;
  }
  __syncthreads();
  if (live && t < cg) {
    double ts = 0.0, tq = 0.0;
    for (int jj = 0; jj < nt; ++jj) {
      ts += rs[jj * cg + t];
      tq += rq[jj * cg + t];
    }
    chs[t] = ts;
    chq[t] = tq;
  }
  __syncthreads();
  if (live && t < cg) {
    double gs = 0.0, gq = 0.0;
    for (int kk = 0; kk < cg; ++kk) {
      gs += chs[kk];
      gq += chq[kk];
    }
    const int ch = g0 + t;
    const double n = f.count_per_channel * cg;
    const double mean = gs / n;
    double var = gq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)f.eps);
    const double ga = (double)p_ga, be = (double)p_be, fa = (double)p_fa, bi = (double)p_bi;
    const double sc = rstd * ga * fa;
    const double sh = (be - mean * rstd * ga) * fa + bi;
    const float cm = (float)(sc * (chs[t] / f.count_per_channel) + sh);
    if (f.scale) f.scale[(size_t)b * c + ch] = (float)sc;
    if (f.shift) f.shift[(size_t)b * c + ch] = (float)sh;
    if (f.chmean) f.chmean[(size_t)b * c + ch] = cm;
    if (tab_scale) tab_scale[ch] = (float)sc;
    if (tab_shift) tab_shift[ch] = (float)sh;
    if (tab_mean) tab_mean[ch] = cm;
    if (mean_rstd && t == 0) {  // training: the backward pass of the norm needs the group moments (normact.hip)
      mean_rstd[((size_t)b * f.groups + g) * 2] = (float)mean;
      mean_rstd[((size_t)b * f.groups + g) * 2 + 1] = (float)rstd;
    }
  }
  __syncthreads();  // (lds may be reused by the caller, or by the next group)
}
// one group by the first 256 threads of the workgroup (threads >= 256 idle through the barriers); lds: 4 x 256 doubles
__device__ __forceinline__ void gn_finish_group(int c, int nslots, const float *__restrict__ part, const GnFinish &f, int b, int g,
                                                double *lds, float *mean_rstd = nullptr) {
  gn_finish_group_v(c, nslots, part, f, b, g, lds, (int)threadIdx.x, threadIdx.x < 256, mean_rstd);
}
// ALL groups of sample b by a workgroup of 256 * NS threads, NS groups at a time (lds: NS x 1024 doubles); the same bits as one
// gn_affine launch. The tables (if given) are complete after it returns (it ends on a barrier).
template <int NS>
__device__ __forceinline__ void gn_finish_sample(int c, int nslots, const float *__restrict__ part, const GnFinish &f, int b, double *lds,
                                                 float *tab_scale = nullptr, float *tab_shift = nullptr, float *tab_mean = nullptr) {
  const int slice = (int)threadIdx.x >> 8, vt = (int)threadIdx.x & 255;
  for (int g0 = 0; g0 < f.groups; g0 += NS) {
    const int g = g0 + slice;
    gn_finish_group_v(c, nslots, part, f, b, g < f.groups ? g : 0, lds + slice * 1024, vt, slice < NS && g < f.groups, nullptr,
                      tab_scale, tab_shift, tab_mean);
  }
}
// ---- 32 rows x 32 lanes -> one row total per lane ("reduce-scatter" over the half-wave) ----
// A GEMM epilogue holds, per lane, one value of each of 32 output-channel rows and needs every row's reduction
// over the 32 lanes of its half-wave. Reducing the rows one by one costs 5 DPP steps per row (160 per statistic);
// this network halves the register count at every level instead -- v_permlane16_swap_b32 pairs lanes L, L^16 and
// merges two registers in 2 instructions, the four in-row levels (row_ror:8, half mirror, quad perms) take 3 --
// 77 instructions per statistic, and lane l ends with the total of row (l & 31). Fixed order: deterministic.
template <int CTRL>
__device__ __forceinline__ float dpp_full(float v) {
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ void permlane16_swap(float &a, float &b) {
  // inline asm: the builtin's second result is mis-assigned by this compiler (both results alias one register)
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
struct RowAdd {
  __device__ static float f(float a, float b) { return a + b; }
};
// (the min / max steps as the bare instructions: behind a DPP / permlane move the compiler no longer knows its operand is a
//  canonical float and puts a canonicalising v_max_f32 x, x in front of every fminf / fmaxf -- 266 of them in the pooling
//  epilogue of the ping-pong GEMM; v_min_f32 / v_max_f32 return the same value, a quiet NaN only if both operands are NaN)
__device__ __forceinline__ float vmin_raw(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
struct RowMin {
  __device__ static float f(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  }
};
struct RowMax {
  __device__ static float f(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  }
};
template <class Op>
__device__ __forceinline__ float rowreduce32(float (&v)[32]) {
  const int lane = (int)(threadIdx.x & 63);
#pragma unroll
  for (int i = 0; i < 16; ++i) {  // lanes L, L^16: the odd 16-lane rows keep v[i + 16]
    float a = v[i], b = v[i + 16];
    permlane16_swap(a, b);
    v[i] = Op::f(a, b);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {  // L, L^8 (row_ror:8)
    const float x = Op::f(v[i], dpp_full<0x128>(v[i])), y = Op::f(v[i + 8], dpp_full<0x128>(v[i + 8]));
    v[i] = (lane & 8) ? y : x;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // L, 7-L inside each group of 8 (row_half_mirror)
    const float x = Op::f(v[i], dpp_full<0x141>(v[i])), y = Op::f(v[i + 4], dpp_full<0x141>(v[i + 4]));
    v[i] = (lane & 4) ? y : x;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {  // L, L^2 (quad_perm [2,3,0,1])
    const float x = Op::f(v[i], dpp_full<0x4E>(v[i])), y = Op::f(v[i + 2], dpp_full<0x4E>(v[i + 2]));
    v[i] = (lane & 2) ? y : x;
  }
  const float x = Op::f(v[0], dpp_full<0xB1>(v[0])), y = Op::f(v[1], dpp_full<0xB1>(v[1]));  // L, L^1
  return (lane & 1) ? y : x;
}

// min / max (any associative, commutative Op) over ALIGNED GROUPS OF 8 LANES for 32 rows at once: the last three levels of the
// network above started from 32 values -- lanes L, 7 - L (row_half_mirror), L ^ 2, L ^ 1 -- leave rows i + 4 j (i = 0..3,
// j = lane & 7) of the lane's group in v[0..3]: 84 instructions instead of 6 per row and statistic (a set-abstraction
// neighbourhood of 32 positions at four positions per lane).
template <class Op>
__device__ __forceinline__ void groupreduce8(float (&v)[32]) {
  const int lane = (int)(threadIdx.x & 63);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float x = Op::f(v[i], dpp_full<0x141>(v[i])), y = Op::f(v[i + 16], dpp_full<0x141>(v[i + 16]));
    v[i] = (lane & 4) ? y : x;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x = Op::f(v[i], dpp_full<0x4E>(v[i])), y = Op::f(v[i + 8], dpp_full<0x4E>(v[i + 8]));
    v[i] = (lane & 2) ? y : x;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x = Op::f(v[i], dpp_full<0xB1>(v[i])), y = Op::f(v[i + 4], dpp_full<0xB1>(v[i + 4]));
    v[i] = (lane & 1) ? y : x;
  }
}

// the same over aligned groups of 16 lanes (first lanes L, 15 - L: row_mirror): rows i + 2 j (i = 0, 1; j = lane & 15) end in v[0..1]
template <class Op>
__device__ __forceinline__ void groupreduce16(float (&v)[32]) {
  const int lane = (int)(threadIdx.x & 63);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float x = Op::f(v[i], dpp_full<0x140>(v[i])), y = Op::f(v[i + 16], dpp_full<0x140>(v[i + 16]));
    v[i] = (lane & 8) ? y : x;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x = Op::f(v[i], dpp_full<0x141>(v[i])), y = Op::f(v[i + 8], dpp_full<0x141>(v[i + 8]));
    v[i] = (lane & 4) ? y : x;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x = Op::f(v[i], dpp_full<0x4E>(v[i])), y = Op::f(v[i + 4], dpp_full<0x4E>(v[i + 4]));
    v[i] = (lane & 2) ? y : x;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float x = Op::f(v[i], dpp_full<0xB1>(v[i])), y = Op::f(v[i + 2], dpp_full<0xB1>(v[i + 2]));
    v[i] = (lane & 1) ? y : x;
  }
}

// ---- which kernel form a pointwise launch took (debug / test hook: include/p2pb_hip.h p2pb_debug_pointwise_form) ----
enum {
  P2PB_FORM_PW_FP32 = 0,      // pw_conv_kernel: exact-fp32 MFMA, unaligned rows
  P2PB_FORM_PW_WIDE_FP32 = 1, // pw_wide_kernel, exact-fp32 MFMA
  P2PB_FORM_PW_WIDE_F16 = 2,  // pw_wide_kernel on the split pack (f16x3 products)
  P2PB_FORM_PW_SPLIT128 = 3,  // pw_split_kernel, 128-channel workgroups
  P2PB_FORM_PW_SPLIT256 = 4,  // pw_split_kernel, 256-channel workgroups
  P2PB_FORM_PW_PINGPONG = 5,  // pw_pp512_kernel
  P2PB_FORM_PW_GATHER = 6,    // pw_wide_kernel<GATHER> (grouped operand built on the fly)
};
void p2pb_note_pointwise_form(int cin, int cout, int npos, int form);
// the GroupNorm finisher armed for this thread's next statistics-producing launch (abi.hip) and the launch that runs it behind
// the producer (conv3d.hip)
bool p2pb_gn_finisher_take(GnFinish *out);
int p2pb_gn_affine_launch(int b, int c, int nslots, const float *part, const GnFinish &f, hipStream_t s);

// value of `key` in P2PB_EXPERIMENT="key=value;key=value" (the one variable behind every A/B switch: p2p_bridge_amd/_experiment.py), or
// dflt; abi.hip
long p2pb_experiment_long(const char *key, long dflt);

// zero-fill `nbytes` (multiple of 4) on stream `s` with a kernel node (see abi.hip)
int p2pb_zero_async(void *p, size_t nbytes, hipStream_t s);

// ---- scatter-add backward passes (devoxelise, grouping, three-NN interpolation): rows accumulated in LDS ----
// Each of these gradients is a scatter into rows gx[b, channel, 0..L) with L a grid (r^3) or a point count; with global
// fp32 atomics the chip retires ~14 G adds/s (profiles/r02_atomic_contention.txt), 0.55 ms for the 8.4 M adds of the
// r = 32 devoxelisation. A workgroup instead owns CH rows of one sample in LDS (CH * L floats <= 128 KB): it zeroes them,
// adds every contribution with ds_add_f32, and writes the rows out once, coalesced -- no zero-fill launch, no HBM atomics,
// and the output is written exactly once. Rows longer than the LDS take the global-atomic kernels.
#define SCAT_THREADS 512
#define SCAT_LDS_MAX (128 * 1024)
bool p2pb_deterministic();  // abi.hip: p2pb_set_deterministic
// threads per workgroup of the LDS-row kernels: one wave in deterministic mode (adds in program order), 8 waves otherwise
static inline int scat_threads() { return p2pb_deterministic() ? 64 : SCAT_THREADS; }
// channels per workgroup: as many rows as fit 64 KB (two workgroups per CU), at most `cap`; one row up to 128 KB; 0 = no fit
static inline int scat_rows(long L, int c, int cap) {
  if (L * 4 > SCAT_LDS_MAX) return 0;
  int ch = (int)((64 * 1024) / (L * 4));
  if (ch < 1) ch = 1;
  if (ch > cap) ch = cap;
  if (ch > c) ch = c;
  return ch;
}
__device__ __forceinline__ void scat_zero(float *rows, int count) {
  for (int i = threadIdx.x * 4; i < count; i += blockDim.x * 4) *(float4 *)(rows + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
}
// rows[j][0..L) -> gx[(b * c + c0 + j)][0..L) for the nch rows of this workgroup; L * 4 bytes need not be 16-aligned
__device__ __forceinline__ void scat_store(const float *rows, int L, int Lp, int nch, float *gx_rows) {
  __syncthreads();
  for (int j = 0; j < nch; ++j)
    for (int i = threadIdx.x; i < L; i += blockDim.x) gx_rows[(size_t)j * L + i] = rows[j * Lp + i];
}
