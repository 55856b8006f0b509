// Experiment: 3x3x3 convolution with fp32 operands split into three bf16 terms (x = x0 + x1 + x2, RNE),
// six bf16 MFMA products per fp32 product (x0y0 + x0y1 + x1y0 + x0y2 + x1y1 + x2y0), fp32 accumulate.
#include "../../p2p_bridge_amd/csrc/conv3d.hip"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define BCK 16  // input channels per LDS stage = K of one bf16 MFMA

// (a, b) -> packed bf16 pairs of the three split terms
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

// packed weights: wt[tap][chunk16][split 3][khalf 2][cout_pad][8 bf16]; element idx = channel chunk*16 + khalf*8 + idx
__global__ void conv3d_pack_bf16_kernel(int cout, int cin, int nchunk, int cout_pad, const float *__restrict__ w,
                                        unsigned short *__restrict__ wt) {
  const size_t total = (size_t)27 * nchunk * 2 * cout_pad * 8;  // one thread per (tap, chunk, khalf, co, idx)
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int idx = (int)(e & 7);
    size_t q = e >> 3;
    const int co = (int)(q % cout_pad);
    q /= cout_pad;
    const int kh = (int)(q & 1);
    q >>= 1;
    const int chunk = (int)(q % nchunk), tap = (int)(q / nchunk);
    const int ci = chunk * BCK + kh * 8 + idx;
    const float x = (co < cout && ci < cin) ? w[((size_t)co * cin + ci) * 27 + tap] : 0.0f;
    unsigned p0, p1, p2;
    split3(x, 0.0f, p0, p1, p2);
    const unsigned p[3] = {p0, p1, p2};
    for (int s = 0; s < 3; ++s)
      wt[((((size_t)(tap * nchunk + chunk) * 3 + s) * 2 + kh) * cout_pad + co) * 8 + idx] = (unsigned short)(p[s] & 0xffff);
  }
}

template <int R, bool COMPACT, int MT, bool XF>
__global__ __launch_bounds__(256, 2) void conv3d_k3_bf16_kernel(int cin, int cout, int nchunk, int cout_pad,
                                                             const float *__restrict__ in,
                                                             const unsigned short *__restrict__ wt,
                                                             const float *__restrict__ bias,
                                                             const float *__restrict__ out_class,
                                                             const float *__restrict__ in_scale,
                                                             const float *__restrict__ in_shift, int in_swish,
                                                             const float *__restrict__ in_sub, int skip_zero,
                                                             const int *__restrict__ brick_list,
                                                             const int *__restrict__ brick_count,
                                                             float *__restrict__ out, float *__restrict__ stats_part) {
  using G = ConvGeom<R, COMPACT>;
  constexpr int HD = G::TD + 2, HH = G::TH + 2, HW = G::TW + 2;
  constexpr int PLANE = HD * HH * HW;
  constexpr int NTILES = (G::TD * G::TH * G::TW) / 32;
  constexpr int BH = R / G::TH, BW = R / G::TW;
  constexpr int R3 = R * R * R;
  // tile[split][khalf][voxel] : 8 bf16 (16 bytes) = channels khalf*8 .. khalf*8+7 of the staged chunk
  __shared__ u32x4 tile[3 * 2 * PLANE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  constexpr int BD = R / G::TD;
  constexpr int NBRICK = BD * BH * BW;
  int bd, bh, bw, b = blockIdx.z;
  if (brick_list) {
    if ((int)blockIdx.x >= *brick_count) return;
    const int entry = brick_list[blockIdx.x];
    b = entry / NBRICK;
    const int bk = entry % NBRICK;
    bd = bk / (BH * BW);
    bh = (bk / BW) % BH;
    bw = bk % BW;
  } else if (COMPACT) {
    const int hi = blockIdx.x / BD, lo = blockIdx.x % BD;
    bh = hi / BW;
    bw = hi % BW;
    bd = (lo + 8 * BD - (3 * bh + 5 * bw)) % BD;
  } else {
    bd = blockIdx.x / (BH * BW);
    bh = (blockIdx.x / BW) % BH;
    bw = blockIdx.x % BW;
  }
  const int brick = (bd * BH + bh) * BW + bw;
  const int d0 = bd * G::TD, h0 = bh * G::TH, w0 = bw * G::TW;
  const int co0 = blockIdx.y * (32 * MT);

  int nbase[2];
  bool nact[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int t = 2 * wave + s;
    nact[s] = t < NTILES;
    constexpr int HB = G::TH / G::NH;
    const int td = (t / HB) * G::ND, th = (t % HB) * G::NH;
    const int jw = l31 % G::TW, jr = l31 / G::TW;
    const int jh = jr % G::NH, jd = jr / G::NH;
    nbase[s] = ((td + jd) * HH + (th + jh)) * HW + jw;
  }

  constexpr int NP = (PLANE + 255) / 256;
  int soff[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int e = tid + j * 256;
    const int dz = e / (HH * HW), hy = (e / HW) % HH, wx = e % HW;
    const int d = d0 - 1 + dz, h = h0 - 1 + hy, w = w0 - 1 + wx;
    const bool ok = e < PLANE && (unsigned)d < (unsigned)R && (unsigned)h < (unsigned)R && (unsigned)w < (unsigned)R;
    soff[j] = ok ? (d * R + h) * R + w : -1;
  }

  f32x16 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][s][r] = 0.0f;

  const float *inb = in + (size_t)b * cin * R3;
  float stg[BCK][NP];
  auto stage_load = [&](int ci0) {
#pragma unroll
    for (int c = 0; c < BCK; ++c)
#pragma unroll
      for (int j = 0; j < NP; ++j)
        stg[c][j] = (soff[j] >= 0 && ci0 + c < cin) ? inb[(size_t)(ci0 + c) * R3 + soff[j]] : 0.0f;
  };
  stage_load(0);

  for (int ci0 = 0; ci0 < cin; ci0 += BCK) {
    __syncthreads();
    int nonzero = 0;
#pragma unroll
    for (int c = 0; c < BCK; ++c) {
      float sc = 1.0f, sh = 0.0f, sub = 0.0f;
      const bool cok = ci0 + c < cin;
      if (XF && cok) {
        sc = in_scale[b * cin + ci0 + c];
        sh = in_shift[b * cin + ci0 + c];
        if (in_sub) sub = in_sub[b * cin + ci0 + c];
      }
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        float v = stg[c][j];
        if (XF && cok && soff[j] >= 0) v = xf_apply(v, sc, sh, in_swish) - sub;
        nonzero |= (v != 0.0f);
        stg[c][j] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int e = tid + j * 256;
      if (e < PLANE) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          u32x4 q[3];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            unsigned p0, p1, p2;
            split3(stg[h * 8 + 2 * i][j], stg[h * 8 + 2 * i + 1][j], p0, p1, p2);
            q[0][i] = p0;
            q[1][i] = p1;
            q[2][i] = p2;
          }
#pragma unroll
          for (int s = 0; s < 3; ++s) tile[(s * 2 + h) * PLANE + e] = q[s];
        }
      }
    }
    const int any = skip_zero ? __syncthreads_or(nonzero) : (__syncthreads(), 1);
    if (ci0 + BCK < cin) stage_load(ci0 + BCK);
    if (!any) continue;

    const u32x4 *wchunk = (const u32x4 *)wt + (((size_t)(ci0 / BCK) * 3) * 2 + khalf) * cout_pad + co0 + l31;
    const size_t wsplit_stride = (size_t)2 * cout_pad, wtap_stride = (size_t)nchunk * 3 * 2 * cout_pad;
    u32x4 a_cur[3][MT], bf[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int m = 0; m < MT; ++m) a_cur[s][m] = wchunk[s * wsplit_stride + m * 32];
#pragma unroll
      for (int n = 0; n < 2; ++n) bf[s][n] = tile[(s * 2 + khalf) * PLANE + nbase[n]];
    }
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      u32x4 a_nxt[3][MT], bf_nxt[3][2];
      if (tap + 1 < 27) {  // both operands of the next tap are requested before this tap is multiplied
        const int ntap = tap + 1;
        const int toff = ((ntap / 9) * HH + (ntap / 3) % 3) * HW + ntap % 3;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
          for (int m = 0; m < MT; ++m) a_nxt[s][m] = wchunk[(size_t)ntap * wtap_stride + s * wsplit_stride + m * 32];
#pragma unroll
          for (int n = 0; n < 2; ++n) bf_nxt[s][n] = tile[(s * 2 + khalf) * PLANE + nbase[n] + toff];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // small terms first
      constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_cur[PA[t]][m]),
                                                                __builtin_bit_cast(bf16x8, bf[PB[t]][n]), acc[m][n], 0, 0, 0);
      if (tap + 1 < 27) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
          for (int m = 0; m < MT; ++m) a_cur[s][m] = a_nxt[s][m];
#pragma unroll
          for (int n = 0; n < 2; ++n) bf[s][n] = bf_nxt[s][n];
        }
      }
    }
  }

  float *outb = out + (size_t)b * cout * R3;
  int vox[2], cls[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int t = 2 * wave + s;
    constexpr int HB = G::TH / G::NH;
    const int td = (t / HB) * G::ND, th = (t % HB) * G::NH;
    const int jw = l31 % G::TW, jr = l31 / G::TW;
    const int d = d0 + td + jr / G::NH, h = h0 + th + jr % G::NH, w = w0 + jw;
    vox[s] = (d * R + h) * R + w;
    const int cd = d == 0 ? 0 : (d == R - 1 ? 2 : 1), ch = h == 0 ? 0 : (h == R - 1 ? 2 : 1),
              cw = w == 0 ? 0 : (w == R - 1 ? 2 : 1);
    cls[s] = (cd * 3 + ch) * 3 + cw;
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      const bool cok = co < cout;
      const float bv = (cok && !out_class) ? bias[co] : 0.0f;
      float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (!nact[s]) continue;
        float v = acc[m][s][r] + bv;
        if (out_class && cok) v += out_class[((size_t)b * 27 + cls[s]) * cout + co];
        if (cok) outb[(size_t)co * R3 + vox[s]] = v;
        s1 += v;
        s2 += v * v;
      }
      if (stats_part) {
        s1 = halfwave_sum_to_last(s1);
        s2 = halfwave_sum_to_last(s2);
        if (l31 == 31 && cok) {
          float *p = stats_part + ((((size_t)b * NBRICK + brick) * 4 + wave) * cout + co) * 2;
          p[0] = s1;
          p[1] = s2;
        }
      }
    }
  }
}

extern "C" size_t exp_packed_halfs(int cout, int cin) {
  const int nchunk = (cin + BCK - 1) / BCK, cout_pad = (cout + 63) / 64 * 64;
  return (size_t)27 * nchunk * 3 * 2 * cout_pad * 8;
}
extern "C" int exp_pack(int cout, int cin, const float *w, unsigned short *wt, void *stream) {
  const int nchunk = (cin + BCK - 1) / BCK, cout_pad = (cout + 63) / 64 * 64;
  hipLaunchKernelGGL(conv3d_pack_bf16_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, cout, cin, nchunk, cout_pad, w, wt);
  return (int)hipGetLastError();
}
// r = 16 compact, MT = 2
extern "C" int exp_conv_r16(int b, int cin, int cout, const float *in, const unsigned short *wt, const float *bias,
                            const float *in_scale, const float *in_shift, int in_swish, int skip_zero, float *out,
                            float *stats, void *stream) {
  const int nchunk = (cin + BCK - 1) / BCK, cout_pad = (cout + 63) / 64 * 64;
  dim3 grid(conv_bricks(16), (cout + 63) / 64, b);
  if (in_scale)
    hipLaunchKernelGGL((conv3d_k3_bf16_kernel<16, true, 2, true>), grid, dim3(256), 0, (hipStream_t)stream, cin, cout, nchunk,
                       cout_pad, in, wt, bias, nullptr, in_scale, in_shift, in_swish, nullptr, skip_zero, nullptr, nullptr, out, stats);
  else
    hipLaunchKernelGGL((conv3d_k3_bf16_kernel<16, true, 2, false>), grid, dim3(256), 0, (hipStream_t)stream, cin, cout, nchunk,
                       cout_pad, in, wt, bias, nullptr, in_scale, in_shift, in_swish, nullptr, skip_zero, nullptr, nullptr, out, stats);
  return (int)hipGetLastError();
}

// calibration: back-to-back bf16 MFMAs on 4 accumulators, `occ` waves per SIMD
__global__ __launch_bounds__(256) void mfma_only_bf16(int iters, const float *__restrict__ in, float *__restrict__ out) {
  u32x4 a[3], b[3];
  for (int s = 0; s < 3; ++s) {
    a[s] = *(const u32x4 *)(in + (threadIdx.x * 3 + s) * 4);
    b[s] = *(const u32x4 *)(in + 4096 + (threadIdx.x * 3 + s) * 4);
  }
  f32x16 acc[4] = {};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t % 3]), __builtin_bit_cast(bf16x8, b[(t + q) % 3]), acc[q], 0, 0, 0);
  }
  float s = 0;
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
extern "C" int exp_mfma_only(int blocks, int iters, const float *in, float *out, void *stream) {
  hipLaunchKernelGGL(mfma_only_bf16, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, in, out);
  return (int)hipGetLastError();
}
