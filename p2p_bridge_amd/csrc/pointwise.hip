// pointwise.hip -- the shared point MLPs (1x1 convolutions + AdaGN/GroupNorm + Swish (+ neighbour max))
// of models/pvcnn.py:162-205 (SharedMLP), :388-424 (set abstraction), :446-467 (feature propagation),
// :905-932 (Pnet2Stage) as fused gfx950 kernels.
//
//   pw_conv      out[b,co,p] = bias[co] (+ bias_b[b,co]) + sum_ci W[co,ci] * xf(in[b,ci,p])
//                xf = identity, or the PREVIOUS layer's norm+activation folded to a per-(b,ci) affine
//                + Swish applied while the operand is staged into LDS; epilogue emits the {sum, sumsq}
//                partials the NEXT GroupNorm needs. So a chain conv-norm-act-conv-norm-act touches
//                each activation tensor exactly twice (one write, one read) instead of ~8 times.
//   affine_act   y = swish(x*scale[b,c] + shift[b,c]) (+ residual)           (last layer of a chain)
//   affine_act_max  y[b,c,m] = max_u swish(x[b,c,m,u]*scale + shift)         (set-abstraction pooling)
//
// The GEMM runs on the exact-fp32 MFMA (32x32x2): M = output channels, N = 32 consecutive positions
// (so stores are lane-consecutive), K = input channels. Small-C layers are HBM-bound (8 FLOP/B at
// C=32..64), the wide ones (Pnet2Stage 512->1024) MFMA-bound.
#include "common.h"
#include <atomic>
#include <stdint.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PW_CK 16  // input channels per register stage (2 sub-chunks of 8): 3 waves/SIMD stay resident (32 -> 2)

// Swish on the hardware exp2 / reciprocal units (see conv3d.hip fast_swish for the error budget)
__device__ __forceinline__ float swishf(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}

// packed weights: wp[cin_pad/8][2][cout_pad][4], element (chunk, khalf, co, kk) = W[co][chunk*8 + 2*kk + khalf]
//
// No LDS, no barriers: in a 1x1 convolution the B operand (activations) is not shared between waves --
// each wave owns 64 distinct positions -- so every lane loads its own MFMA B fragments straight from HBM
// (lanes 0..31 = 32 consecutive positions of channel 2kk, lanes 32..63 of channel 2kk+1: two 128-byte
// segments per load instruction) and the four waves of a workgroup run fully decoupled. The loads of
// chunk c+1 are issued before chunk c is multiplied; A fragments (weights) are 16-byte L1/L2 loads issued
// first, so the in-order vmcnt wait in front of the MFMAs never covers the HBM prefetch.
template <int MT, bool XF, bool STATS>
__global__ __launch_bounds__(256, 3) void pw_conv_kernel(int cin, int cout, int cout_pad, int P,
                                                      const float *__restrict__ in, const float *__restrict__ wp,
                                                      const float *__restrict__ bias,
                                                      const float *__restrict__ bias_b,
                                                      const float *__restrict__ in_scale,
                                                      const float *__restrict__ in_shift, int in_swish,
                                                      float *__restrict__ out, float *__restrict__ stats_part) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int p0 = blockIdx.x * 256, co0 = blockIdx.y * (32 * MT), b = blockIdx.z;
  const int pl[2] = {p0 + wave * 64 + l31, p0 + wave * 64 + 32 + l31};
  const bool pok[2] = {pl[0] < P, pl[1] < P};
  const float *inb = in + (size_t)b * cin * P;
  const int nchunk8 = (cin + 7) >> 3;
  __shared__ float pwc_bias[32 * MT];  // bias (+ per-sample bias) through LDS: see pw_wide_kernel
  if (tid < 32 * MT) {
    const int co = co0 + tid;
    float v = 0.0f;
    if (co < cout) {
      v = bias ? bias[co] : 0.0f;
      if (bias_b) v += bias_b[(size_t)b * cout + co];
    }
    pwc_bias[tid] = v;
  }
  __syncthreads();

  f32x16 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][s][r] = 0.0f;

  float bcur[PW_CK / 2][2], bnxt[PW_CK / 2][2];
  auto load_b = [&](int ci0, float(&dst)[PW_CK / 2][2]) {
#pragma unroll
    for (int kk = 0; kk < PW_CK / 2; ++kk) {
      const int ci = ci0 + 2 * kk + khalf;
#pragma unroll
      for (int s = 0; s < 2; ++s) dst[kk][s] = (ci < cin && pok[s]) ? inb[(size_t)ci * P + pl[s]] : 0.0f;
    }
  };
  load_b(0, bcur);
  const float *wbase = wp + ((size_t)khalf * cout_pad + co0 + l31) * 4;
  const size_t wchunk_stride = (size_t)2 * cout_pad * 4;
  f32x4 a_cur[PW_CK / 8][MT], a_nxt[PW_CK / 8][MT];
  auto load_a = [&](int chunk0, f32x4(&dst)[PW_CK / 8][MT]) {
#pragma unroll
    for (int sub = 0; sub < PW_CK / 8; ++sub) {
      const int ch = chunk0 + sub < nchunk8 ? chunk0 + sub : nchunk8 - 1;  // clamp: stays inside the buffer
#pragma unroll
      for (int m = 0; m < MT; ++m) dst[sub][m] = *(const f32x4 *)(wbase + (size_t)ch * wchunk_stride + (size_t)m * 32 * 4);
    }
  };
  load_a(0, a_cur);

  for (int ci0 = 0; ci0 < cin; ci0 += PW_CK) {
    const int chunk0 = ci0 >> 3;
    const bool more = ci0 + PW_CK < cin;
    if (more) {  // both operands of the NEXT chunk are requested before this chunk is multiplied
      load_a(chunk0 + PW_CK / 8, a_nxt);
      load_b(ci0 + PW_CK, bnxt);
    }
    if (XF) {
#pragma unroll
      for (int kk = 0; kk < PW_CK / 2; ++kk) {
        const int ci = ci0 + 2 * kk + khalf;
        if (ci < cin) {
          const float sc = in_scale[b * cin + ci], sh = in_shift[b * cin + ci];
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            float v = bcur[kk][s] * sc + sh;
            if (in_swish) v = swishf(v);
            bcur[kk][s] = pok[s] ? v : 0.0f;
          }
        }
      }
    }
#pragma unroll
    for (int sub = 0; sub < PW_CK / 8; ++sub) {
      if (chunk0 + sub >= nchunk8) break;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            acc[m][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[sub][m][kk], bcur[sub * 4 + kk][s], acc[m][s], 0, 0, 0);
      }
    }
    if (more) {
#pragma unroll
      for (int kk = 0; kk < PW_CK / 2; ++kk)
#pragma unroll
        for (int s = 0; s < 2; ++s) bcur[kk][s] = bnxt[kk][s];
#pragma unroll
      for (int sub = 0; sub < PW_CK / 8; ++sub)
#pragma unroll
        for (int m = 0; m < MT; ++m) a_cur[sub][m] = a_nxt[sub][m];
    }
  }

  float *outb = out + (size_t)b * cout * P;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      const float bv = pwc_bias[co - co0];
      float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int p = pl[s];
        const float v = acc[m][s][r] + bv;
        if (co < cout && pok[s]) {
          outb[(size_t)co * P + p] = v;
          if (STATS) {
            s1 += v;
            s2 += v * v;
          }
        }
      }
      if (STATS) {
        s1 = halfwave_sum_to_last(s1);
        s2 = halfwave_sum_to_last(s2);
        if (l31 == 31 && co < cout) {
          float *q = stats_part + ((((size_t)b * gridDim.x + blockIdx.x) * 4 + wave) * cout + co) * 2;
          q[0] = s1;
          q[1] = s2;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Wide tile (the default whenever rows are 16-byte aligned: npos % 4 == 0): a wave owns 32*MT output
// channels x 128 positions. Lane j of a half-wave holds positions 4j..4j+3 of one input channel in ONE
// 16-byte buffer load; MFMA column tile s is the position set {4j+s}, so the four tiles of a lane are the
// four components of that load and the epilogue stores 16 bytes per lane as well: 4x fewer memory
// instructions per MFMA than the one-position-per-lane kernel above, and the channel rows are addressed
// through scalar descriptors (no per-lane 64-bit address arithmetic). Chunks of 8 input channels are
// double-buffered in registers: 184 VGPRs, 2 waves/SIMD. Measured on 512->1024 x 262144 positions:
// 133 TFLOP/s without / 125 with the statistics epilogue (the MFMA-only loop of the same shape: 133),
// vs 86 for the narrow kernel.
// Ragged channel counts need no predicates: a row pair starting at ci >= cin is clamped to the last row
// (the packed weights are zero there, so the finite garbage contributes exactly 0), and the descriptor's
// num_records ends at the sample's last row, so the odd half of a half-valid pair reads hardware zeros.
// ------------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define PWW_CK 8

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_perm(float v) {
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, ROW_MASK, 0xf, false));
}
// min and max over aligned groups of g lanes. g = 2..16: every lane of the group ends with the result
// (xor-1, xor-2 inside quads, then the half-row and row mirrors); g = 32: lanes 31 / 63 hold their half-wave's.
__device__ __forceinline__ void group_minmax(float &mn, float &mx, int g) {
  if (g > 1) { mn = vmin_raw(mn, dpp_perm<0xB1, 0xf>(mn)); mx = vmax_raw(mx, dpp_perm<0xB1, 0xf>(mx)); }    // quad_perm [1,0,3,2]
  if (g > 2) { mn = vmin_raw(mn, dpp_perm<0x4E, 0xf>(mn)); mx = vmax_raw(mx, dpp_perm<0x4E, 0xf>(mx)); }    // quad_perm [2,3,0,1]
  if (g > 4) { mn = vmin_raw(mn, dpp_perm<0x141, 0xf>(mn)); mx = vmax_raw(mx, dpp_perm<0x141, 0xf>(mx)); }  // row_half_mirror
  if (g > 8) { mn = vmin_raw(mn, dpp_perm<0x140, 0xf>(mn)); mx = vmax_raw(mx, dpp_perm<0x140, 0xf>(mx)); }  // row_mirror
  if (g > 16) { mn = vmin_raw(mn, dpp_perm<0x142, 0xa>(mn)); mx = vmax_raw(mx, dpp_perm<0x142, 0xa>(mx)); } // row_bcast:15
}

// POOL: additionally emit {min, max} of the raw output over groups of pool_g lanes (= 4*pool_g consecutive
// positions: a set-abstraction neighbourhood) or, pool_g == 32, over the wave's 128 positions (global max-pool
// partials); `out` may then be NULL. Swish (like every activation the network uses) is quasi-convex, so
//   max_p act(scale*x_p + shift) = max(act(scale*min_p x_p + shift), act(scale*max_p x_p + shift)),
// and the pooled tensor is produced by p2pb_minmax_act from 2/U-th of the data without the layer's
// output ever being written or re-read.
// TERMS == SPLIT_F16X3: the same tiling, operand path and epilogue with the products on the 16-bit matrix pipe (fp16-pair
// split, three MFMAs of K = 16 instead of eight exact-fp32 ones of K = 2: 5.3x fewer matrix cycles -- the exact-fp32
// MFMAs were HALF the time of the set-abstraction neighbourhood layers; round 2 measurement, docs/history). `wp` is then the split
// pack of pw_split_kernel (fragments read straight from L1 / L2, output scale in its trailer); 16 input channels per step:
// lane (l31, khalf) loads rows 8 khalf .. + 7 of the step for its four positions, transforms and splits them once.
// GATHER (f16x3 form only): the operand is the GROUPED tensor of a set abstraction without ever being built --
// operand[ci, p] = zt[idx[p]][ci] - cxt[p / gu][ci] from point-major rows zt f32[b, gn, cin] (`in`), cxt f32[b, P / gu, cin]
// and the neighbour lists idx i32[b, P] (csrc/neighbors.hip group_sub_kernel's arithmetic, bit for bit): a lane fetches the
// 8-channel piece of its four positions' rows (32 contiguous bytes each, L2-resident: the ungrouped tensor is 1 MB per
// sample) instead of four channel-major quads of a 268 MB tensor that group_sub wrote and this kernel read back.
struct PwGather {
  const float *cxt;  // f32[b, P / gu, cin] or NULL
  const int *idx;    // i32[b, P]
  int gn, gu;        // points per cloud, neighbours per centre
};
// PG (pooling form, compile time): 0 none, 32 global-pool partials, 8 neighbourhoods of 8 lanes, 1 any other width (pool_g)
template <int MT, bool XF, bool STATS, int PG, int TERMS = 0, bool GATHER = false>
__global__ __launch_bounds__(256, 2) void pw_wide_kernel(int cin, int cout, int cout_pad, int P, int nslots,
                                                      const float *__restrict__ in, const float *__restrict__ wp,
                                                      const float *__restrict__ bias,
                                                      const float *__restrict__ bias_b,
                                                      const float *__restrict__ in_scale,
                                                      const float *__restrict__ in_shift, int in_swish,
                                                      float *__restrict__ out, float *__restrict__ stats_part,
                                                      float *__restrict__ mm_out, int pool_g, int out_pm,
                                                      PwGather gat = PwGather()) {
  static_assert(!GATHER || TERMS == SPLIT_F16X3, "the gathered operand exists in the f16x3 form");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int co0 = blockIdx.y * (32 * MT), b = blockIdx.z;
  const int p = blockIdx.x * 512 + wave * 128 + l31 * 4;
  const bool pok = p < P;
  const int pc = pok ? p : P - 4;  // clamped lanes multiply garbage that is never stored
  const float *inb = in + (size_t)b * cin * P;
  // bias (+ per-sample bias) of the workgroup's channels through LDS: fetched from global memory inside the epilogue's row
  // loops they were one serialised L2 round trip per row (conv3d.hip, tools/exp_conv_timeline.py)
  __shared__ float pww_bias[32 * MT];
  if (tid < 32 * MT) {
    const int co = co0 + tid;
    float v = 0.0f;
    if (co < cout) {
      v = bias ? bias[co] : 0.0f;
      if (bias_b) v += bias_b[(size_t)b * cout + co];
    }
    pww_bias[tid] = v;
  }
  __syncthreads();

  f32x16 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][s][r] = 0.0f;

  if constexpr (TERMS == SPLIT_F16X3) {
    constexpr int PWS_TILE_ = 2 * 3 * 2 * 128;  // (PWS_TILE of the split pack, defined below)
    const u32x4 *wp4 = (const u32x4 *)wp;
    const int ncoblk128 = (cout + 127) / 128, nchunk32 = (cin + 31) / 32;
    const u32x4 *wtile = wp4 + (size_t)(co0 >> 7) * PWS_TILE_ + khalf * 128 + (co0 & 127) + l31;
    const unsigned voffh = (unsigned)(khalf * 8 * P + pc) * 4u, rowb = (unsigned)P * 4u;
    f32x4 braw[8];
    u32x4 a_nx[MT][2];
    // GATHER: this lane's four neighbour rows and its centre row (positions pc .. pc + 3 share a centre: gu % 4 == 0)
    int gid[4] = {0, 0, 0, 0};
    const float *grow[4] = {nullptr, nullptr, nullptr, nullptr}, *gcen = nullptr;
    if constexpr (GATHER) {
      const int *ip = gat.idx + (size_t)b * P + pc;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        gid[t] = ip[t];
        grow[t] = in + ((size_t)b * gat.gn + gid[t]) * cin;
      }
      if (gat.cxt) gcen = gat.cxt + ((size_t)b * (P / gat.gu) + pc / gat.gu) * cin;
    }
    auto load_bh = [&](int ci0) {
      if constexpr (GATHER) {
        // rows are point-major: channels ci0 + 8 khalf .. + 7 of position t are 32 contiguous bytes (cin % 8 == 0)
        const int cb = ci0 + 8 * khalf;
        f32x4 cen[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
        if (gcen && cb < cin) {
          cen[0] = *(const f32x4 *)(gcen + cb);
          cen[1] = *(const f32x4 *)(gcen + cb + 4);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          f32x4 r0 = {0.0f, 0.0f, 0.0f, 0.0f}, r1 = r0;
          if (cb < cin) {
            r0 = *(const f32x4 *)(grow[t] + cb);
            r1 = *(const f32x4 *)(grow[t] + cb + 4);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            braw[i][t] = gcen ? r0[i] - cen[0][i] : r0[i];          // (group_sub_kernel: v = z; v -= cx)
            braw[4 + i][t] = gcen ? r1[i] - cen[1][i] : r1[i];
          }
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = min(ci0 + i, cin - 1);  // rows at or beyond cin: zero records -> hardware zeros (x zero weights)
        const int rec = ci0 + i < cin ? (int)((unsigned)(cin - row) * rowb) : 0;
        auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)(inb + (size_t)row * P), 0, rec, 0x00020000);
        braw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voffh, 0, 0));
      }
    };
    auto load_ah = [&](int ci0) {
      const u32x4 *t = wtile + (size_t)(ci0 >> 5) * ncoblk128 * PWS_TILE_ + ((ci0 >> 4) & 1) * (3 * 2 * 128);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) a_nx[m][pl] = t[pl * 256 + m * 32];
    };
    load_bh(0);
    load_ah(0);
    for (int ci0 = 0; ci0 < cin; ci0 += 16) {
      if (XF) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int ca = b * cin + min(ci0 + i, cin - 1), cb = b * cin + min(ci0 + 8 + i, cin - 1);
          const float sca = in_scale[ca], scb = in_scale[cb], sha = in_shift[ca], shb = in_shift[cb];
          const float sc = khalf ? scb : sca, sh = khalf ? shb : sha;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            float v = braw[i][t] * sc + sh;
            if (in_swish) v = swishf(v);
            braw[i][t] = v;
          }
        }
      }
      u32x4 pl0[4], pl1[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned p0, p1, p2;
          split_pair<SPLIT_F16X3>(braw[2 * i][t], braw[2 * i + 1][t], p0, p1, p2);
          pl0[t][i] = p0;
          pl1[t][i] = p1;
        }
      u32x4 a_cu[MT][2];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) a_cu[m][pl] = a_nx[m][pl];
      if (ci0 + 16 < cin) {  // the next step's loads fly during the MFMAs
        load_ah(ci0 + 16);
        load_bh(ci0 + 16);
      }
      // term by term over all the accumulators (a1 b0, a0 b1, a0 b0 per accumulator as before: same bits): the three MFMAs
      // that update one accumulator are 4 MT instructions apart instead of back to back
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (X2W_KEEP_LOW_WEIGHT_PRODUCT) acc[m][t] = split_mfma<SPLIT_F16X3>(a_cu[m][1], pl0[t], acc[m][t]);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = split_mfma<SPLIT_F16X3>(a_cu[m][0], pl1[t], acc[m][t]);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = split_mfma<SPLIT_F16X3>(a_cu[m][0], pl0[t], acc[m][t]);
    }
    const float oscale = ((const float *)(wp4 + (size_t)nchunk32 * ncoblk128 * PWS_TILE_))[1];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][t][r] *= oscale;
  }
  const unsigned voff = (unsigned)(khalf * P + pc) * 4u;
  const unsigned rowbytes = (unsigned)P * 4u;
  f32x4 bcur[PWW_CK / 2], bnxt[PWW_CK / 2];
  auto load_b = [&](int ci0, f32x4(&dst)[PWW_CK / 2]) {
#pragma unroll
    for (int kk = 0; kk < PWW_CK / 2; ++kk) {
      const int row0 = min(ci0 + 2 * kk, cin - 1);
      auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)(inb + (size_t)row0 * P), 0,
                                                  (int)(min(cin - row0, 2) * rowbytes), 0x00020000);
      dst[kk] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
    }
  };
  const float *wbase = wp + ((size_t)khalf * cout_pad + co0 + l31) * 4;
  const size_t wchunk_stride = (size_t)2 * cout_pad * 4;
  f32x4 a_cur[MT], a_nxt[MT];
  auto load_a = [&](int chunk, f32x4(&dst)[MT]) {
#pragma unroll
    for (int m = 0; m < MT; ++m) dst[m] = *(const f32x4 *)(wbase + (size_t)chunk * wchunk_stride + (size_t)m * 32 * 4);
  };
  if constexpr (TERMS == 0) {
    load_b(0, bnxt);
    load_a(0, a_nxt);
  }

  for (int ci0 = 0; TERMS == 0 && ci0 < cin; ci0 += PWW_CK) {
    // rotate (the vmcnt wait lands here), request the next chunk, then multiply the current one
#pragma unroll
    for (int kk = 0; kk < PWW_CK / 2; ++kk) bcur[kk] = bnxt[kk];
#pragma unroll
    for (int m = 0; m < MT; ++m) a_cur[m] = a_nxt[m];
    if (ci0 + PWW_CK < cin) {
      load_a((ci0 >> 3) + 1, a_nxt);
      load_b(ci0 + PWW_CK, bnxt);
    }
    if (XF) {
#pragma unroll
      for (int kk = 0; kk < PWW_CK / 2; ++kk) {
        // wave-uniform indices: the folded norm parameters travel through the scalar cache
        const int ca = b * cin + min(ci0 + 2 * kk, cin - 1), cb = b * cin + min(ci0 + 2 * kk + 1, cin - 1);
        const float sca = in_scale[ca], scb = in_scale[cb], sha = in_shift[ca], shb = in_shift[cb];
        const float sc = khalf ? scb : sca, sh = khalf ? shb : sha;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          float v = bcur[kk][s] * sc + sh;
          if (in_swish) v = swishf(v);
          bcur[kk][s] = v;
        }
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          acc[m][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[m][kk], bcur[kk][s], acc[m][s], 0, 0, 0);
  }

  if (out_pm) {  // point-major output f32[b, P, cout] (the consumer gathers whole rows); no statistics in this form
    float *ob = out + (size_t)b * P * cout;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cq = co0 + m * 32 + 8 * g + 4 * khalf;
        float bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = pww_bias[cq + i - co0];
        if (pok && cq < cout) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            float *q = ob + (size_t)(p + s) * cout + cq;
            const f32x4 v = {acc[m][s][4 * g] + bv[0], acc[m][s][4 * g + 1] + bv[1], acc[m][s][4 * g + 2] + bv[2],
                             acc[m][s][4 * g + 3] + bv[3]};
            if (cq + 3 < cout && (cout & 3) == 0) *(f32x4 *)q = v;
            else
              for (int i = 0; i < 4; ++i)
                if (cq + i < cout) q[i] = v[i];
          }
        }
      }
    return;
  }
  // ---- epilogue (round 5). The first form reduced every one of the 32 MT rows on its own -- five DPP steps per statistic, a
  // runtime-width min / max ladder, and a predicated store with its own 64-bit address for every row: ~9000 of the POOL kernel's
  // 11900 instructions ran ONCE per wave, and with two 16-channel steps per wave (the 32 -> 64 set-abstraction layer) the launch
  // was bound by issuing them (271 us for 8.6 GFLOP; profiles/r05_overlap.txt). Now: bias in place + stores, then the rows'
  // reductions as reduce-scatter networks (common.h rowreduce32: lane l31 ends with the total of row l31; groupreduce8 below:
  // neighbourhoods of 8 lanes) and ONE store per lane. Sums: the lane's four positions (v0 + v1) + (v2 + v3) as before, then the
  // network's fixed tree instead of the 5-step ladder (the partials change in their last bit, deterministically).
  float *outb = out ? out + (size_t)b * cout * P : nullptr;
  const int slot = (blockIdx.x * 4 + wave) * 2;  // this wave fills slot `slot` with its 128-position sums and zeroes slot + 1
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      const float bv = pww_bias[co - co0];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[m][t][r] += bv;
      if (co < cout && pok && outb)
        *(f32x4 *)(outb + (size_t)co * P + p) = f32x4{acc[m][0][r], acc[m][1][r], acc[m][2][r], acc[m][3][r]};
    }
  // the row whose total this lane holds after rowreduce32 (MT == 1: rows 0..15 twice, two statistics packed into one network)
  const int rrow = MT == 2 ? l31 : (l31 & 15);
  const int rco = co0 + (rrow >> 4) * 32 + (rrow & 3) + 8 * ((rrow & 15) >> 2) + 4 * khalf;
  auto rowval = [&](int kind, int m, int r) -> float {  // 0: sum, 1: sum of squares, 2: min, 3: -max over the lane's 4 positions
    const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
    const float v0 = acc[m][0][r], v1 = acc[m][1][r], v2 = acc[m][2][r], v3 = acc[m][3][r];
    const bool ok = co < cout && pok;
    if (kind == 0) return ok ? (v0 + v1) + (v2 + v3) : 0.0f;
    if (kind == 1) return ok ? (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3) : 0.0f;
    if (kind == 2) return pok ? vmin_raw(vmin_raw(v0, v1), vmin_raw(v2, v3)) : INFINITY;
    return pok ? -vmax_raw(vmax_raw(v0, v1), vmax_raw(v2, v3)) : INFINITY;
  };
  if (STATS) {
    float s1, s2;
    float tv[32];
    if constexpr (MT == 2) {
#pragma unroll
      for (int i = 0; i < 32; ++i) tv[i] = rowval(0, i >> 4, i & 15);
      s1 = rowreduce32<RowAdd>(tv);
#pragma unroll
      for (int i = 0; i < 32; ++i) tv[i] = rowval(1, i >> 4, i & 15);
      s2 = rowreduce32<RowAdd>(tv);
    } else {
#pragma unroll
      for (int i = 0; i < 32; ++i) tv[i] = rowval(i >> 4, 0, i & 15);
      s1 = s2 = rowreduce32<RowAdd>(tv);  // lanes 0..15: sums, lanes 16..31: sums of squares, of rows l31 & 15
    }
    if (rco < cout) {
      if (slot < nslots) {
        float *q = stats_part + (((size_t)b * nslots + slot) * cout + rco) * 2;
        if constexpr (MT == 2) *(f32x2 *)q = f32x2{s1, s2};
        else q[l31 >> 4] = s1;
      }
      if (slot + 1 < nslots) {
        float *q = stats_part + (((size_t)b * nslots + slot + 1) * cout + rco) * 2;
        if constexpr (MT == 2) *(f32x2 *)q = f32x2{0.0f, 0.0f};
        else q[l31 >> 4] = 0.0f;
      }
    }
  }
  if constexpr (PG != 0) {
    float tv[32];
    if constexpr (PG == 32) {  // global max-pool partials: {min, max} over the wave's 128 positions
      float mn, mx;
      if constexpr (MT == 2) {
#pragma unroll
        for (int i = 0; i < 32; ++i) tv[i] = rowval(2, i >> 4, i & 15);
        mn = rowreduce32<RowMin>(tv);
#pragma unroll
        for (int i = 0; i < 32; ++i) tv[i] = rowval(3, i >> 4, i & 15);
        mx = -rowreduce32<RowMin>(tv);
      } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) tv[i] = rowval(2 + (i >> 4), 0, i & 15);
        mn = rowreduce32<RowMin>(tv);  // lanes 0..15: min, lanes 16..31: -max, of rows l31 & 15
        mx = -mn;
      }
      if (rco < cout) {
        float *q = mm_out + ((((size_t)b * gridDim.x + blockIdx.x) * 4 + wave) * cout + rco) * 2;
        if constexpr (MT == 2) *(f32x2 *)q = f32x2{mn, mx};
        else q[l31 >> 4] = (l31 >> 4) ? mx : mn;
      }
    } else if constexpr (PG == 8) {  // neighbourhoods of 32 positions = aligned groups of 8 lanes (the bench's set abstractions)
      // groupreduce8 leaves rows i + 4 j (i = 0..3, j = lane & 7) of the lane's group in v[i]
      const int j = l31 & 7;
      const size_t ngrp = (size_t)(P / 32);
      if constexpr (MT == 2) {
        float tx[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          tv[i] = rowval(2, i >> 4, i & 15);
          tx[i] = rowval(3, i >> 4, i & 15);
        }
        groupreduce8<RowMin>(tv);
        groupreduce8<RowMin>(tx);
        const int cb = co0 + 32 * (j >> 2) + 8 * (j & 3) + 4 * khalf;  // rows i + 4 j: channels cb + i
        if (pok) {
          float *q = mm_out + (((size_t)b * cout + cb) * ngrp + p / 32) * 2;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (cb + i < cout) *(f32x2 *)(q + (size_t)i * ngrp * 2) = f32x2{tv[i], -tx[i]};
        }
      } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) tv[i] = rowval(2 + (i >> 4), 0, i & 15);
        groupreduce8<RowMin>(tv);  // j < 4: min of rows i + 4 j, j >= 4: -max of rows i + 4 (j - 4)
        const int cb = co0 + 8 * (j & 3) + 4 * khalf;
        if (pok) {
          float *q = mm_out + (((size_t)b * cout + cb) * ngrp + p / 32) * 2 + (j >> 2);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (cb + i < cout) q[(size_t)i * ngrp * 2] = (j >> 2) ? -tv[i] : tv[i];
        }
      }
    } else {  // other neighbourhood sizes: the per-row ladder
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
          float mn = rowval(2, m, r), mx = -rowval(3, m, r);
          group_minmax(mn, mx, pool_g);
          if (co < cout && (l31 & (pool_g - 1)) == 0 && pok) {
            const int u = 4 * pool_g;
            float *q = mm_out + (((size_t)b * cout + co) * (P / u) + p / u) * 2;
            q[0] = mn;
            q[1] = mx;
          }
        }
    }
  }
}

// w element (co, ci) at w[co * s_co + ci * s_ci]: (cin, 1) for a layer's own weight [cout][cin], (1, cout) for the adjoint
// (data-gradient) operator of a layer whose weight is [cin][cout] -- no transposed copy of the weight is ever made
__global__ void pw_pack_kernel(int cout, int cin, int cin_pad, int cout_pad, const float *__restrict__ w,
                               float *__restrict__ wp, long s_co, long s_ci) {
  const size_t total = (size_t)cin_pad * cout_pad;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int kk = (int)(e & 3);
    const int co = (int)((e >> 2) % cout_pad);
    const int kh = (int)((e / ((size_t)4 * cout_pad)) & 1);
    const int chunk = (int)(e / ((size_t)8 * cout_pad));
    const int ci = chunk * 8 + 2 * kk + kh;
    wp[e] = (co < cout && ci < cin) ? w[(size_t)co * s_co + (size_t)ci * s_ci] : 0.0f;
  }
}

static inline int pw_cin_pad(int cin) { return (cin + 7) / 8 * 8; }
static inline int pw_cout_pad(int cout) { return (cout + 127) / 128 * 128; }

extern "C" size_t p2pb_pointwise_packed_floats(int cout, int cin) {
  return (size_t)pw_cin_pad(cin) * pw_cout_pad(cout);
}

static int pw_pack(int cout, int cin, const float *w, float *wp, bool adjoint, void *stream) {
  if (cout <= 0 || cin <= 0) return P2PB_EINVAL;
  const size_t total = p2pb_pointwise_packed_floats(cout, cin);
  hipLaunchKernelGGL(pw_pack_kernel, dim3((unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)), dim3(256),
                     0, (hipStream_t)stream, cout, cin, pw_cin_pad(cin), pw_cout_pad(cout), w, wp, adjoint ? 1L : (long)cin,
                     adjoint ? (long)cout : 1L);
  return p2pb_launch_status();
}
extern "C" int p2pb_pointwise_pack_weights(int cout, int cin, const float *w, float *wp, void *stream) {
  return pw_pack(cout, cin, w, wp, false, stream);
}
extern "C" int p2pb_pointwise_pack_weights_adjoint(int cout, int cin, const float *w_forward, float *wp, void *stream) {
  return pw_pack(cout, cin, w_forward, wp, true, stream);
}

extern "C" size_t p2pb_pointwise_stats_floats(int b, int cout, int npos) {
  return (size_t)b * ((npos + 255) / 256) * 4 * cout * 2;
}

// ------------------------------------------------------------------------------------------------
// Split-operand form of the same GEMM for the matrix-bound layers (wide channel counts): fp32 operands as
// three bf16 terms, six bf16 MFMA products per fp32 product, fp32 accumulate -- the arithmetic of
// conv3d.hip's split kernel (fp32-faithful: dropped terms < 2^-26 |x*w|), 2.67x fewer matrix cycles.
// At that rate the operands can no longer stream through per-lane global loads (the wide kernel above
// would need ~50 B/clk/CU of L1 bandwidth), so this one is the classic LDS-tiled GEMM:
//   workgroup = 4 waves as 2 (M) x 2 (N): 128 output channels x 128 positions, 32 input channels per stage;
//   A: pre-split packed weights, one contiguous 24 KB tile per (stage, 128-channel block), brought into a
//      double-buffered LDS tile by the LDS-DMA path (global_load_lds_dwordx4: no registers, no ds_write --
//      measured, the VGPR->LDS store path is what bounds this kernel: staging off = 141 -> 206 TFLOP/s);
//   B: each wave loads 8 channels x 128 positions (8-byte coalesced loads through scalar row descriptors),
//      applies the folded norm + Swish ONCE per element, splits, and writes 16-byte groups of 8 channels;
//   LDS[kstep][split][khalf][128 rows] x 16 B for both, so every MFMA fragment is one conflict-free
//   ds_read_b128 (positions are stored even/odd de-interleaved: lane j of N-tile n owns position 2j+n,
//   which also makes the epilogue's stores 8 bytes per lane).
// Global loads of the next stage fly during the MFMAs of the current one (register staged).
// ------------------------------------------------------------------------------------------------
#define PWS_CK 32

#define PWS_TILE (2 * 3 * 2 * 128)                     // 16-byte groups per operand tile (24 KB)
#define PWS_LDS_BYTES (2 * PWS_TILE * 16)               // A + B
#include "pw_pp512.h"  // the ping-pong form of the f16x3 arithmetic for >= 512-channel layers (rounds 3-4)

// Epilogue of the split-operand GEMM kernels for one wave's 64 channels x NB x 64 positions: bias, stores (channel- or
// point-major), GroupNorm partials per 64-position slot, optional {min, max} for the pooling that follows.
// PL (pooling form, compile time): 0 none, 1 global-pool partials (pool_u == 0), 32 neighbourhoods of 32 positions, 2 any other
// neighbourhood size (the per-row ladder)
template <int PL, int WM, int NB>
__device__ __forceinline__ void pws_epilogue(f32x16 (&acc)[2][2 * NB], int b, int bx, int gx, int pblk, int co0,
                                             int wm, int wn, int l31, int khalf, int cout, int P, int nslots,
                                             const float *__restrict__ bias, const float *__restrict__ bias_b,
                                             float *__restrict__ out, float *__restrict__ stats_part,
                                             float *__restrict__ mm_out, int pool_u, int out_pm, const float *sb) {
  // sb: the workgroup's bias (+ per-sample bias) values [64 WM], staged in LDS by the kernel's prologue. Fetched from
  // global memory inside the row loops below they were one L2 round trip each, serialised by the loops' branches (the
  // same finding as in the convolutions' epilogue, conv3d.hip / tools/exp_conv_timeline.py).
#pragma unroll
  for (int pb = 0; pb < NB; ++pb) {  // the wave's NB blocks of 64 positions (even / odd tiles 2 pb, 2 pb + 1)
  const int p = pblk + 128 * pb + 2 * (wn * 32 + l31);
  const bool pok = p < P;
  if (WM == 2 && out_pm) {  // point-major output f32[b, P, cout]; no statistics in this form (128-channel form only)
    float *ob = out + (size_t)b * P * cout;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cq = co0 + wm * 64 + m * 32 + 8 * g + 4 * khalf;
        float bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = sb[cq + i - co0];
        if (pok && cq < cout) {
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            float *q = ob + (size_t)(p + n) * cout + cq;
            const f32x4 v = {acc[m][2 * pb + n][4 * g] + bv[0], acc[m][2 * pb + n][4 * g + 1] + bv[1],
                             acc[m][2 * pb + n][4 * g + 2] + bv[2], acc[m][2 * pb + n][4 * g + 3] + bv[3]};
            if (cq + 3 < cout && (cout & 3) == 0) *(f32x4 *)q = v;
            else
              for (int i = 0; i < 4; ++i)
                if (cq + i < cout) q[i] = v[i];
          }
        }
      }
    continue;
  }
  // ---- epilogue: bias, 8-byte stores, GroupNorm partials per 64-position slot, optional {min, max}.
  // Row index of the reductions: idx = m*16 + r; rowreduce32 leaves row (l31) in lane l31.
  float *outb = out ? out + (size_t)b * cout * P : nullptr;
  const int slot = (bx * NB + pb) * 2 + wn;
  const int pool_g = pool_u ? pool_u / 2 : 32;
  // pass 1: bias (in place), stores, neighbourhood {min, max}
#pragma unroll
  for (int m = 0; m < 2; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      const bool cok = co < cout;
      const float bv = sb[co - co0];
      acc[m][2 * pb][r] += bv;
      acc[m][2 * pb + 1][r] += bv;
      const f32x2 v = {acc[m][2 * pb][r], acc[m][2 * pb + 1][r]};
      if (cok && pok && outb) *(f32x2 *)(outb + (size_t)co * P + p) = v;
      if constexpr (PL == 2) {
        float mn = pok ? fminf(v[0], v[1]) : INFINITY, mx = pok ? fmaxf(v[0], v[1]) : -INFINITY;
        group_minmax(mn, mx, pool_g);
        if (cok && pok && (pool_g == 32 ? l31 == 31 : (l31 & (pool_g - 1)) == 0)) {
          float *q = mm_out + (((size_t)b * cout + co) * (P / pool_u) + p / pool_u) * 2;
          q[0] = mn;
          q[1] = mx;
        }
      }
    }
  }
  if constexpr (PL == 32) {
    // 32 neighbours = 16 lanes of two positions: the rows' {min, max} through a reduce-scatter network (common.h groupreduce16:
    // rows i + 2 j end in lane j of the group; round 5 -- the per-row ladder above was 12 instructions per row and statistic)
    float tn[32], tx[32];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v0 = acc[m][2 * pb][r], v1 = acc[m][2 * pb + 1][r];
        tn[m * 16 + r] = pok ? vmin_raw(v0, v1) : INFINITY;
        tx[m * 16 + r] = pok ? -vmax_raw(v0, v1) : INFINITY;
      }
    groupreduce16<RowMin>(tn);
    groupreduce16<RowMin>(tx);
    const int j = l31 & 15;  // rows 2 j, 2 j + 1: m = j >> 3, r = 2 (j & 7) + i
    const int cb = co0 + wm * 64 + 32 * (j >> 3) + 2 * (j & 1) + 8 * ((j & 7) >> 1) + 4 * khalf;
    if (pok) {
      const size_t ngrp = (size_t)(P / 32);
      float *q = mm_out + (((size_t)b * cout + cb) * ngrp + p / 32) * 2;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (cb + i < cout) *(f32x2 *)(q + (size_t)i * ngrp * 2) = f32x2{tn[i], -tx[i]};
    }
  }
  // this lane's row after the reductions; one 32-value array live at a time (register pressure: the other position
  // block's accumulators are still waiting)
  const int rm = l31 >> 4, rr = l31 & 15;
  const int rco = co0 + wm * 64 + rm * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * khalf;
  auto rowvals = [&](int kind, float (&v)[32]) {  // 0: sum, 1: sum of squares, 2: min, 3: max over the lane's pair
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        const float v0 = acc[m][2 * pb][r], v1 = acc[m][2 * pb + 1][r];
        const bool ok = co < cout && pok;
        v[m * 16 + r] = kind == 0 ? (ok ? v0 + v1 : 0.0f)
                        : kind == 1 ? (ok ? v0 * v0 + v1 * v1 : 0.0f)
                        : kind == 2 ? (pok ? fminf(v0, v1) : INFINITY)
                                    : (pok ? fmaxf(v0, v1) : -INFINITY);
      }
  };
  if (stats_part) {
    float tv[32];
    rowvals(0, tv);
    const float s1 = rowreduce32<RowAdd>(tv);
    rowvals(1, tv);
    const float s2 = rowreduce32<RowAdd>(tv);
    if (rco < cout) {
      float *q = stats_part + (((size_t)b * nslots + slot) * cout + rco) * 2;
      q[0] = s1;
      q[1] = s2;
      if (bx == gx - 1 && pb == NB - 1 && wn == 1)  // slots past the last position block (nslots is a multiple of 4)
        for (int sl = slot + 1; sl < nslots; ++sl) {
          float *z = stats_part + (((size_t)b * nslots + sl) * cout + rco) * 2;
          z[0] = 0.0f;
          z[1] = 0.0f;
        }
    }
  }
  if constexpr (PL == 1) {
    float tv[32];
    rowvals(2, tv);
    const float mn = rowreduce32<RowMin>(tv);
    rowvals(3, tv);
    const float mx = rowreduce32<RowMax>(tv);
    if (rco < cout) {
      float *q = mm_out + ((((size_t)b * gx * NB + bx * NB + pb) * 2 + wn) * cout + rco) * 2;
      q[0] = mn;
      q[1] = mx;
    }
  }
  }  // pb
}

// WM = waves along M: 2 -> 128 output channels per workgroup (4 waves, 48 KB of LDS, three workgroups per CU);
// 4 -> 256 channels (8 waves, 72 KB, two per CU): the activation tile is transformed / split / staged once per 256
// instead of once per 128 channels -- half the VALU + LDS-write work per MFMA -- for the layers whose grid still fills
// the chip (the global embedding's 512 -> 1024 GEMM). A wave's tile, fragments and epilogue are the same in both.
// NB = 128-position blocks per workgroup (1 or 2): with 2 a wave owns 64 channels x 128 positions (2 x 4 accumulator
// tiles), every A fragment feeds four MFMAs instead of two and the weight tile is streamed from L2 once per 256
// positions -- the 128 x 128 tiling moves 10.7 GB through L2 for the 512 -> 1024 x 262144 GEMM (6.4 GB of it the
// pre-split weights, re-read by 2048 position blocks), 256 x 256 moves 5.3 GB.
// TERMS: the arithmetic (common.h, p2pb_set_split_terms) -- SPLIT_F16X3 (default: fp16-pair split, three products, two
// operand planes: the third is neither fetched, written nor read) or SPLIT_BF16X6 (three bf16 terms, six products)
template <bool XF, int PL, int WM, int NB, int TERMS>
#ifndef PWS_WM4_WAVES
#define PWS_WM4_WAVES 4
#endif
__global__ __launch_bounds__(128 * WM, NB == 2 ? 2 : (WM == 2 ? 3 : PWS_WM4_WAVES)) void pw_split_kernel(int cin, int cout, int P, int nslots,
                                                       const float *__restrict__ in, const u32x4 *__restrict__ wp,
                                                       const float *__restrict__ bias,
                                                       const float *__restrict__ bias_b,
                                                       const float *__restrict__ in_scale,
                                                       const float *__restrict__ in_shift, int in_swish,
                                                       float *__restrict__ out, float *__restrict__ stats_part,
                                                       float *__restrict__ mm_out, int pool_u, int out_pm) {
  extern __shared__ u32x4 pws_lds[];  // [A: WM/2 blocks of 128 channels][B: NB blocks of 128 positions][XF: 2 cin floats]
  constexpr int NT = 128 * WM;
  constexpr int BS = 128 * NB;  // 16-byte groups per (kstep, split, khalf) row of the B tile
  u32x4 *lds_b = pws_lds + (WM / 2) * PWS_TILE;
  const u32x4 *lds_a = pws_lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform for the scalar descriptors
  const int l31 = lane & 31, khalf = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware order: workgroup ids go round-robin over the 8 XCDs (each with its own L2), so XCD x takes the x-th
  // contiguous eighth of (sample, position block, channel block) with the channel block fastest: the 2..8 workgroups
  // that stage the SAME activation tile run side by side on one XCD and share it in its L2 (the dispatch order
  // x + gx*(y + gy*z) put them 64 workgroups apart: 2.6x the algorithmic bytes from HBM).
  const int ncoblk = gridDim.y;
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const unsigned nblk = gridDim.x * gridDim.y * gridDim.z;
  const unsigned vid = nblk % 8 == 0 ? (lin % 8) * (nblk / 8) + lin / 8 : lin;
  const int bx = (vid / ncoblk) % gridDim.x, by = vid % ncoblk;
  const int b = vid / (ncoblk * gridDim.x);
  const int pblk = bx * (128 * NB), co0 = by * (64 * WM);
  const float *inb = in + (size_t)b * cin * P;
  const bool mact = co0 + wm * 64 < cout;  // this wave's 64 channels exist (wave-uniform)
  __shared__ float pws_bias[64 * WM];  // bias (+ per-sample bias) of the workgroup's channels; published by the stage barriers
  if (tid < 64 * WM) {
    const int co = co0 + tid;
    float v = 0.0f;
    if (co < cout) {
      v = bias ? bias[co] : 0.0f;
      if (bias_b) v += bias_b[(size_t)b * cout + co];
    }
    pws_bias[tid] = v;
  }

  f32x16 acc[2][2 * NB];  // [M-tile][position block * 2 + even/odd tile]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2 * NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

  // B staging: the stage's 4 channel groups (8 channels each) x NB position blocks are dealt to the 2*WM waves.
  //   WM == 2: wave w owns channel group w for ALL position blocks; lane l positions 2l, 2l+1 of each block;
  //   WM == 4, NB == 2: wave w owns channel group w >> 1 of position block w & 1; lane l positions 2l, 2l+1;
  //   WM == 4, NB == 1: wave w owns channel group w >> 1 for the position half w & 1; lane l position 64 (w & 1) + l.
  constexpr bool ONE = WM == 4 && NB == 1;      // one position per lane (4-byte loads)
  constexpr int NBW = WM == 2 ? NB : 1;         // position blocks staged by one wave
  const int bgrp = WM == 2 ? wave : wave >> 1, bsel = WM == 2 ? 0 : (wave & 1);
  unsigned voff[NBW];
#pragma unroll
  for (int q = 0; q < NBW; ++q) {
    const int pl = ONE ? pblk + 64 * bsel + lane : pblk + 128 * (WM == 2 ? q : bsel) + 2 * lane;
    voff[q] = (unsigned)(pl < P ? pl : P - (ONE ? 1 : 2)) * 4u;  // clamped lanes stage garbage that is never stored
  }
  float braw[NBW][8][ONE ? 1 : 2];  // (plain floats: an f32x2 with a dead half cost the 256-channel form 37 spilled VGPRs)
  auto load_b = [&](int ci0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = min(ci0 + 8 * bgrp + i, cin - 1);  // beyond cin: finite garbage x zero weights
      auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)(inb + (size_t)row * P), 0, P * 4, 0x00020000);
#pragma unroll
      for (int q = 0; q < NBW; ++q) {
        if constexpr (!ONE) {
          const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff[q], 0, 0));
          braw[q][i][0] = v[0];
          braw[q][i][1] = v[1];
        } else {
          braw[q][i][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[q], 0, 0));
        }
      }
    }
  };
  // A tile of stage `chunk` -> LDS, asynchronously: lane i of a wave lands at base + 16*i
  auto dma_a = [&](int chunk) {
    // (the pack is in 128-channel blocks; a 256-channel workgroup takes two consecutive ones)
    const int nblk128 = WM == 2 ? ncoblk : (cout + 127) / 128;
    const u32x4 *src = wp + ((size_t)chunk * nblk128 + by * (WM / 2)) * PWS_TILE;
    u32x4 *dst = pws_lds;
    const bool second_ok = WM == 2 || by * 2 + 1 < nblk128;  // odd block count: the last workgroup has one block only
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if ((second_ok || i * NT + tid < PWS_TILE) && (TERMS == 6 || (((i * NT + wave * 64) % PWS_TILE) / 256) % 3 != 2))
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i * NT + tid),
                                         (__attribute__((address_space(3))) void *)(dst + i * NT + wave * 64), 16, 0, 0);
  };
  load_b(0);
  // (the folded norm parameters of the operand travel through the scalar cache: an LDS broadcast at the top of the transform
  //  phase costs this kernel 1.2 % -- measured)

  for (int ci0 = 0; ci0 < cin; ci0 += PWS_CK) {
    __syncthreads();  // everyone is done reading the previous stage
    dma_a(ci0 / PWS_CK);  // lands while B is transformed and split below
    // ---- stage B: transform + split
    {
      constexpr int NE = ONE ? 1 : 2;
      if (XF) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int c = min(ci0 + 8 * bgrp + i, cin - 1);
          const float sc = in_scale[b * cin + c], sh = in_shift[b * cin + c];
#pragma unroll
          for (int q = 0; q < NBW; ++q)
#pragma unroll
            for (int e = 0; e < NE; ++e) {
              float v = braw[q][i][e] * sc + sh;
              if (in_swish) v = swishf(v);
              braw[q][i][e] = v;
            }
        }
      }
      const int kstep = bgrp >> 1, kh = bgrp & 1;
#pragma unroll
      for (int q = 0; q < NBW; ++q)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          u32x4 qq[3];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            unsigned p0, p1, p2;
            split_pair<TERMS>(braw[q][2 * i][e], braw[q][2 * i + 1][e], p0, p1, p2);
            qq[0][i] = p0;
            qq[1][i] = p1;
            qq[2][i] = p2;
          }
          // slot of position p of a 128-block: (p & 1) * 64 + (p >> 1)   (even / odd de-interleaved)
          const int blk = WM == 2 ? q : (NB == 2 ? bsel : 0);
          const int slot = blk * 128 + (ONE ? (lane & 1) * 64 + 32 * bsel + (lane >> 1) : e * 64 + lane);
#pragma unroll
          for (int s = 0; s < split_planes(TERMS); ++s) lds_b[((kstep * 3 + s) * 2 + kh) * BS + slot] = qq[s];
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this stage's A tile has landed
    __syncthreads();
    if (ci0 + PWS_CK < cin) load_b(ci0 + PWS_CK);  // next stage's B loads fly during the MFMAs
    if (!mact) continue;
#pragma unroll
    for (int kstep = 0; kstep < 2; ++kstep) {
      u32x4 af[3][2];
#pragma unroll
      for (int s = 0; s < split_planes(TERMS); ++s)
#pragma unroll
        for (int m = 0; m < 2; ++m)
          af[s][m] = lds_a[(wm >> 1) * PWS_TILE + ((kstep * 3 + s) * 2 + khalf) * 128 + (wm & 1) * 64 + m * 32 + l31];
      constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};  // small terms first
      // the B fragments of one position block (2 tiles x 3 terms) at a time: 24 registers live instead of 24 NB
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        u32x4 bf[3][2];
#pragma unroll
        for (int s = 0; s < split_planes(TERMS); ++s)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            bf[s][n] = lds_b[((kstep * 3 + s) * 2 + khalf) * BS + nb * 128 + n * 64 + wn * 32 + l31];
#pragma unroll
        for (int t = (TERMS == 6 ? 0 : 3); t < 6; ++t)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
              if (X2W_KEEP_LOW_WEIGHT_PRODUCT || TERMS != SPLIT_F16X3 || PA[t] != 1)
                acc[m][2 * nb + n] = split_mfma<TERMS>(af[PA[t]][m], bf[PB[t]][n], acc[m][2 * nb + n]);
      }
    }
  }
  if (!mact) return;
  {
    if constexpr (TERMS == SPLIT_F16X3) {  // 1 / (S_x S_w), a power of two, stored behind the pack
      const int nblk128 = WM == 2 ? ncoblk : (cout + 127) / 128;
      const float oscale = ((const float *)(wp + (size_t)((cin + PWS_CK - 1) / PWS_CK) * nblk128 * PWS_TILE))[1];
  #pragma unroll
      for (int m = 0; m < 2; ++m)
  #pragma unroll
        for (int n = 0; n < 2 * NB; ++n)
  #pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] *= oscale;
    }
  pws_epilogue<PL, WM, NB>(acc, b, bx, (int)gridDim.x, pblk, co0, wm, wn, l31, khalf, cout, P, nslots, bias, bias_b, out,
                             stats_part, mm_out, pool_u, out_pm, pws_bias);
  }
}

// split pack: wp[chunk32][cout block of 128][kstep 2][split 3][khalf 2][128 co][8 bf16],
// channel = chunk*32 + kstep*16 + khalf*8 + idx
// mode SPLIT_F16X3: planes 0, 1 hold the fp16 pair of w * S_w (plane 2 unused); trailer = {max|w| bits, 1 / (S_x S_w)}
__global__ void pw_pack_split_kernel(int cout, int cin, int nchunk, int ncoblk, const float *__restrict__ w,
                                     unsigned short *__restrict__ wp, int mode, float *__restrict__ trailer, long s_co, long s_ci,
                                     const unsigned *__restrict__ amax) {
  const size_t total = (size_t)nchunk * ncoblk * 2 * 2 * 128 * 8;  // (chunk, coblk, kstep, khalf, co, idx)
  const bool x2w = (mode & SPLIT_X2W_FLAG) != 0;  // (pricing experiment, common.h: a zero low plane)
  mode &= 0xff;
  const float wmax = amax ? __builtin_bit_cast(float, *amax) : trailer[0];  // (conv3d_pack_split_kernel: same convention)
  const float sw = mode == SPLIT_F16X3 ? f16_weight_scale(wmax) : 1.0f;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    trailer[1] = mode == SPLIT_F16X3 ? 1.0f / (SPLIT_F16_SX * sw) : 1.0f;
    if (amax) trailer[0] = wmax, trailer[2] = trailer[3] = 0.0f;  // (the whole trailer, as the zero fill of the other path)
  }
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int idx = (int)(e & 7);
    size_t q = e >> 3;
    const int col = (int)(q & 127);
    q >>= 7;
    const int kh = (int)(q & 1);
    q >>= 1;
    const int ks = (int)(q & 1);
    q >>= 1;
    const int cb = (int)(q % ncoblk), chunk = (int)(q / ncoblk);
    const int co = cb * 128 + col, ci = chunk * PWS_CK + ks * 16 + kh * 8 + idx;
    const float x = (co < cout && ci < cin) ? w[(size_t)co * s_co + (size_t)ci * s_ci] : 0.0f;
    unsigned p0, p1, p2;
    if (mode == SPLIT_F16X3) {
      split2h(x * sw, 0.0f, p0, p1);
      if (x2w) p1 = 0u;
      p2 = 0u;
    } else {
      split3(x, 0.0f, p0, p1, p2);
    }
    const unsigned pp[3] = {p0, p1, p2};
    for (int s = 0; s < 3; ++s)
      wp[((((((size_t)chunk * ncoblk + cb) * 2 + ks) * 3 + s) * 2 + kh) * 128 + col) * 8 + idx] = (unsigned short)(pp[s] & 0xffff);
  }
}

extern "C" size_t p2pb_pointwise_split_packed_bytes(int cout, int cin) {
  const size_t nchunk = (cin + PWS_CK - 1) / PWS_CK, ncoblk = (cout + 127) / 128;
  return nchunk * ncoblk * (2 * 3 * 2 * 128) * 16 + 16;  // + trailer {max|w| bits, output scale, -, -} (fp16 mode)
}

static int pw_pack_split(int cout, int cin, const float *w, void *wp, bool adjoint, void *stream, const unsigned *amax = nullptr) {
  if (cout <= 0 || cin <= 0) return P2PB_EINVAL;
  const int nchunk = (cin + PWS_CK - 1) / PWS_CK, ncoblk = (cout + 127) / 128;
  const size_t total = (size_t)nchunk * ncoblk * 2 * 2 * 128 * 8;
  // the pack is made for the arithmetic selected NOW (p2pb_set_split_terms); callers re-pack after a switch
  float *trailer = (float *)((char *)wp + (size_t)nchunk * ncoblk * PWS_TILE * 16);
  static const long x2w = p2pb_experiment_long("x2w", 0);
  const int mode = p2pb_g_split_terms;
  if (mode == SPLIT_F16X3 && !amax) {
    const int rc = p2pb_zero_async(trailer, 16, (hipStream_t)stream);
    if (rc) return rc;
    hipLaunchKernelGGL(absmax_bits_kernel, dim3(absmax_blocks((size_t)cout * cin)), dim3(256), 0, (hipStream_t)stream, w, (size_t)cout * cin,
                       (unsigned *)trailer);
  }
  hipLaunchKernelGGL(pw_pack_split_kernel, dim3((unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)),
                     dim3(256), 0, (hipStream_t)stream, cout, cin, nchunk, ncoblk, w, (unsigned short *)wp,
                     mode | ((x2w && mode == SPLIT_F16X3 && !adjoint) ? SPLIT_X2W_FLAG : 0), trailer,
                     adjoint ? 1L : (long)cin, adjoint ? (long)cout : 1L, mode == SPLIT_F16X3 ? amax : nullptr);
  return p2pb_launch_status();
}
extern "C" int p2pb_pointwise_pack_weights_split(int cout, int cin, const float *w, void *wp, void *stream) {
  return pw_pack_split(cout, cin, w, wp, false, stream);
}
extern "C" int p2pb_pointwise_pack_weights_split_adjoint(int cout, int cin, const float *w_forward, void *wp, void *stream) {
  return pw_pack_split(cout, cin, w_forward, wp, true, stream);
}
extern "C" int p2pb_pointwise_pack_weights_split_amax(int cout, int cin, const float *w, void *wp, const unsigned *amax_bits,
                                                      void *stream) {
  return amax_bits ? pw_pack_split(cout, cin, w, wp, false, stream, amax_bits) : P2PB_EINVAL;
}

// the finisher the entry point took for this launch (abi.hip p2pb_gn_finisher_arm); a launcher that hands it to its kernel
// clears `pending`, otherwise the entry point launches gn_affine_kernel behind the producer
static thread_local GnFinish tl_pw_fin;
static thread_local bool tl_pw_fin_pending = false;
static int pw_finish_behind(int rc, int b, int cout, int P, const float *stats_part, hipStream_t s) {
  if (!tl_pw_fin_pending) return rc;
  tl_pw_fin_pending = false;
  if (rc != 0) return rc;
  if (!stats_part) return P2PB_EINVAL;  // (a finisher armed for a launch without statistics)
  // the checks of p2pb_gn_affine_params_ex (a MyGroupNorm(32, cout) with cout % 32 != 0, or cout / groups > 256, ran
  // gn_finish_group with a truncated group / past its 256-entry LDS table and read gamma out of bounds)
  const GnFinish &f = tl_pw_fin;
  if (f.groups <= 0 || cout % f.groups != 0 || cout / f.groups > 256 || (f.style && f.style_stride < 2 * cout)) return P2PB_EINVAL;
  return p2pb_gn_affine_launch(b, cout, (P + 255) / 256 * 4, stats_part, tl_pw_fin, s);
}

static int pw_launch_split(int b, int cin, int cout, int P, const float *in, const void *wp, const float *bias,
                           const float *bias_b, const float *in_scale, const float *in_shift, int in_swish,
                           float *out, float *stats_part, float *minmax, int pool_u, int out_pm, hipStream_t s) {
  const bool xf = in_scale != nullptr;
  const int mode = p2pb_g_split_terms;
  // 256-channel workgroups when the grid still holds >= 4 of them per CU (P2PB_EXPERIMENT pw_wm=2 / 4 overrides: A/B timing)
  static const int wm_env = (int)p2pb_experiment_long("pw_wm", 0);
  const bool wm4 = !out_pm && (wm_env ? wm_env == 4
                                       : (cout >= 512 && (long)((P + 127) / 128) * ((cout + 255) / 256) * b >= 1024));
  // (256-position workgroups of THIS kernel -- half the weight traffic through L2 at one wave per SIMD less -- measured
  //  4 % / 7 % slower and are not instantiated; the wide tile lives in pw_pp512.h, with the pipeline it needs)
  dim3 grid((P + 127) / 128, wm4 ? (cout + 255) / 256 : (cout + 127) / 128, b);
  const int nslots = (P + 255) / 256 * 4;
  const u32x4 *w = (const u32x4 *)wp;
  // 72 KB of dynamic LDS (above the 64 KB default): opt in once per instantiation
#define LAUNCHW(XF, PL, WM, NB, TM)                                                                                  \
  do {                                                                                                               \
    static bool once = false;                                                                                        \
    const int lds = (WM / 2 + NB) * PWS_TILE * 16;                                                                   \
    if (!once) {                                                                                                     \
      (void)hipFuncSetAttribute((const void *)pw_split_kernel<XF, PL, WM, NB, TM>,                                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);                                    \
      once = true;                                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL((pw_split_kernel<XF, PL, WM, NB, TM>), grid, dim3(128 * WM), lds, s, cin, cout, P, nslots, in, \
                       w, bias, bias_b, in_scale, in_shift, in_swish, out, stats_part, minmax, pool_u, out_pm);       \
  } while (0)
#define LAUNCHF(XF, PL, WM, NB)                                       \
  do {                                                                \
    if (mode == SPLIT_F16X3) LAUNCHW(XF, PL, WM, NB, SPLIT_F16X3);     \
    else LAUNCHW(XF, PL, WM, NB, 6);                                   \
  } while (0)
  if (mode == SPLIT_BF16X3) {  // the training data gradient's arithmetic: plain operand, no pooling, channel-major output
    if (xf || minmax || out_pm) return P2PB_EINVAL;
    p2pb_note_pointwise_form(cin, cout, P, wm4 ? P2PB_FORM_PW_SPLIT256 : P2PB_FORM_PW_SPLIT128);
    if (wm4) LAUNCHW(false, 0, 4, 1, SPLIT_BF16X3);
    else LAUNCHW(false, 0, 2, 1, SPLIT_BF16X3);
    return p2pb_launch_status();
  }
#define LAUNCH(XF, PL)                    \
  do {                                    \
    if (wm4) LAUNCHF(XF, PL, 4, 1);        \
    else LAUNCHF(XF, PL, 2, 1);            \
  } while (0)
  // The layers that qualify for 256-channel workgroups AND come in whole 512-channel blocks with an even number of
  // 32-channel stages run the ping-pong kernel (pw_pp512.h: one 8-wave workgroup per CU on 512 channels x 128 positions, all
  // 160 KB of LDS, weight tiles by LDS-DMA, the two waves of a SIMD in opposite phase; any position count).
  // P2PB_EXPERIMENT pw_pp=0 keeps pw_split_kernel (A/B timing).
  static const int pp_env = (int)p2pb_experiment_long("pw_pp", 1);
  if (pp_env && wm4 && mode == SPLIT_F16X3 && cin % 64 == 0 && cout % 512 == 0 && (!minmax || pool_u == 0) &&
      (size_t)cin * P * 4 < (1ull << 31)) {  // (its operand descriptor holds one sample: 32-bit byte offsets)
    dim3 pgrid((P + 127) / 128, cout / 512, b);
#define LAUNCHP5(XF, PL)                                                                                              \
  do {                                                                                                                \
    static bool once = false;                                                                                         \
    if (!once) {                                                                                                      \
      (void)hipFuncSetAttribute((const void *)pw_pp512_kernel<XF, PL>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                P5_LDS_BYTES);                                                                        \
      once = true;                                                                                                    \
    }                                                                                                                 \
    hipLaunchKernelGGL((pw_pp512_kernel<XF, PL>), pgrid, dim3(512), P5_LDS_BYTES, s, cin, cout, P, nslots, in, w, bias, \
                       bias_b, in_scale, in_shift, in_swish, out, stats_part, minmax, pool_u);                        \
  } while (0)
    if (xf && minmax) LAUNCHP5(true, true);
    else if (xf) LAUNCHP5(true, false);
    else if (minmax) LAUNCHP5(false, true);
    else LAUNCHP5(false, false);
#undef LAUNCHP5
    p2pb_note_pointwise_form(cin, cout, P, P2PB_FORM_PW_PINGPONG);
    return p2pb_launch_status();
  }
  p2pb_note_pointwise_form(cin, cout, P, wm4 ? P2PB_FORM_PW_SPLIT256 : P2PB_FORM_PW_SPLIT128);
  const int pl = !minmax ? 0 : pool_u == 0 ? 1 : pool_u == 32 ? 32 : 2;
  if (xf) {
    if (pl == 0) LAUNCH(true, 0);
    else if (pl == 1) LAUNCH(true, 1);
    else if (pl == 32) LAUNCH(true, 32);
    else LAUNCH(true, 2);
  } else {
    if (pl == 0) LAUNCH(false, 0);
    else if (pl == 1) LAUNCH(false, 1);
    else if (pl == 32) LAUNCH(false, 32);
    else LAUNCH(false, 2);
  }
#undef LAUNCH
#undef LAUNCHF
#undef LAUNCHW
  return p2pb_launch_status();
}

static bool pw_wide_ok(int P, const float *in, const float *out) {
  // 16-byte rows: every row of in/out starts on a 16-byte boundary and holds whole quads
  return P % 4 == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
}

template <int MT>
static int pw_launch(int b, int cin, int cout, int P, const float *in, const float *wp, const float *bias,
                     const float *bias_b, const float *in_scale, const float *in_shift, int in_swish, float *out,
                     float *stats_part, float *minmax, int pool_g, int out_pm, hipStream_t s, bool split_pack = false) {
  const bool xf = in_scale != nullptr, st = stats_part != nullptr;
  // split_pack: wp is the f16x3 split pack (flags bits 2 + 7): plain statistics form, 16-byte rows, f16x3 selected
  if (split_pack && (!pw_wide_ok(P, in, out) || p2pb_g_split_terms != SPLIT_F16X3)) return P2PB_EINVAL;
  if (pw_wide_ok(P, in, out)) {
    dim3 grid((P + 511) / 512, (cout + 32 * MT - 1) / (32 * MT), b);
    const int nslots = (P + 255) / 256 * 4;
#define LAUNCHX(XF, ST, PL, TM)                                                                                       \
  hipLaunchKernelGGL((pw_wide_kernel<MT, XF, ST, PL, TM>), grid, dim3(256), 0, s, cin, cout, pw_cout_pad(cout), P,       \
                     nslots, in, wp, bias, bias_b, in_scale, in_shift, in_swish, out, stats_part, minmax, pool_g, out_pm,  \
                     PwGather())
#define LAUNCH(XF, ST, PL)                                 \
  do {                                                     \
    if (split_pack) LAUNCHX(XF, ST, PL, SPLIT_F16X3);       \
    else LAUNCHX(XF, ST, PL, 0);                            \
  } while (0)
    p2pb_note_pointwise_form(cin, cout, P, split_pack ? P2PB_FORM_PW_WIDE_F16 : P2PB_FORM_PW_WIDE_FP32);
    if (minmax) {
      const int pg = pool_g == 8 || pool_g == 32 ? pool_g : 1;
      if (xf) {
        if (pg == 8) LAUNCH(true, true, 8);
        else if (pg == 32) LAUNCH(true, true, 32);
        else LAUNCH(true, true, 1);
      } else {
        if (pg == 8) LAUNCH(false, true, 8);
        else if (pg == 32) LAUNCH(false, true, 32);
        else LAUNCH(false, true, 1);
      }
    } else if (xf && st) LAUNCH(true, true, 0);
    else if (xf) LAUNCH(true, false, 0);
    else if (st) LAUNCH(false, true, 0);
    else LAUNCH(false, false, 0);
#undef LAUNCH
#undef LAUNCHX
    return p2pb_launch_status();
  }
  if (minmax || out_pm) return P2PB_EINVAL;  // (the unaligned fallback: plain form only)
  p2pb_note_pointwise_form(cin, cout, P, P2PB_FORM_PW_FP32);
  dim3 grid((P + 255) / 256, (cout + 32 * MT - 1) / (32 * MT), b);
#define LAUNCH(XF, ST)                                                                                            \
  hipLaunchKernelGGL((pw_conv_kernel<MT, XF, ST>), grid, dim3(256), 0, s, cin, cout, pw_cout_pad(cout), P, in, wp, bias, \
                     bias_b, in_scale, in_shift, in_swish, out, stats_part)
  if (xf && st) LAUNCH(true, true);
  else if (xf) LAUNCH(true, false);
  else if (st) LAUNCH(false, true);
  else LAUNCH(false, false);
#undef LAUNCH
  return p2pb_launch_status();
}

static int pw_conv_forward_impl(int b, int cin, int cout, int npos, const float *in, const void *wp_any,
                                           const float *bias, const float *bias_b, const float *in_scale,
                                           const float *in_shift, int in_swish, int flags, float *out,
                                           float *stats_part, void *stream) {
  if (b <= 0 || cin <= 0 || cout <= 0 || npos <= 0 || !out) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int out_pm = (flags & 32) != 0;  // point-major output f32[b, npos, cout]
  if (out_pm && stats_part) return P2PB_EINVAL;
  if ((flags & 4) && (flags & 128)) {  // narrow layer on the split pack: the wide tiling with f16x3 products
    const float *wsp = (const float *)wp_any;
    return cout > 32 ? pw_launch<2>(b, cin, cout, npos, in, wsp, bias, bias_b, in_scale, in_shift, in_swish, out,
                                    stats_part, nullptr, 0, out_pm, s, true)
                     : pw_launch<1>(b, cin, cout, npos, in, wsp, bias, bias_b, in_scale, in_shift, in_swish, out,
                                    stats_part, nullptr, 0, out_pm, s, true);
  }
  if (flags & 4) {  // wp is the split pack
    if (!pw_wide_ok(npos, in, out)) return P2PB_EINVAL;
    return pw_launch_split(b, cin, cout, npos, in, wp_any, bias, bias_b, in_scale, in_shift, in_swish, out, stats_part,
                           nullptr, 0, out_pm, s);
  }
  const float *wp = (const float *)wp_any;
  // 64 output channels per wave (128 measured slower: the accumulators alone would take 256 registers)
  return cout > 32 ? pw_launch<2>(b, cin, cout, npos, in, wp, bias, bias_b, in_scale, in_shift, in_swish, out,
                                  stats_part, nullptr, 0, out_pm, s)
                   : pw_launch<1>(b, cin, cout, npos, in, wp, bias, bias_b, in_scale, in_shift, in_swish, out,
                                  stats_part, nullptr, 0, out_pm, s);
}

extern "C" int p2pb_pointwise_conv_forward(int b, int cin, int cout, int npos, const float *in, const void *wp_any,
                                           const float *bias, const float *bias_b, const float *in_scale,
                                           const float *in_shift, int in_swish, int flags, float *out,
                                           float *stats_part, void *stream) {
  tl_pw_fin_pending = p2pb_gn_finisher_take(&tl_pw_fin);  // (a finisher armed for this launch: abi.hip)
  const int rc = pw_conv_forward_impl(b, cin, cout, npos, in, wp_any, bias, bias_b, in_scale, in_shift, in_swish, flags, out,
                                      stats_part, stream);
  return pw_finish_behind(rc, b, cout, npos, stats_part, (hipStream_t)stream);
}

// pool_u = neighbourhood size (4, 8, 16, 32 or 64 consecutive positions) or 0 for the global pool
static int pool_lanes(int pool_u) { return pool_u == 0 ? 32 : pool_u / 4; }

extern "C" int p2pb_pointwise_pool_supported(int npos, int pool_u) {
  const bool uok = pool_u == 0 || pool_u == 4 || pool_u == 8 || pool_u == 16 || pool_u == 32 || pool_u == 64;
  return npos > 0 && npos % 4 == 0 && uok && (pool_u == 0 || npos % pool_u == 0);
}

extern "C" size_t p2pb_pointwise_minmax_floats(int b, int cout, int npos, int pool_u, int flags) {
  const bool split_tiling = (flags & 4) && !(flags & 128);  // (bit 7: the wide tiling on the split pack)
  if (pool_u == 0) return (size_t)b * (split_tiling ? (npos + 127) / 128 * 2 : (npos + 511) / 512 * 4) * cout * 2;
  return (size_t)b * cout * (npos / pool_u) * 2;
}

static int pw_conv_pool_forward_impl(int b, int cin, int cout, int npos, const float *in,
                                                const void *wp_any, const float *bias, const float *bias_b,
                                                const float *in_scale, const float *in_shift, int in_swish, int flags,
                                                float *out, float *stats_part, int pool_u, float *minmax,
                                                void *stream) {
  if (b <= 0 || cin <= 0 || cout <= 0 || npos <= 0 || !minmax || !stats_part) return P2PB_EINVAL;
  if (!p2pb_pointwise_pool_supported(npos, pool_u) || !pw_wide_ok(npos, in, out)) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if ((flags & 4) && (flags & 128)) {
    const float *wsp = (const float *)wp_any;
    const int gl = pool_lanes(pool_u);
    return cout > 32 ? pw_launch<2>(b, cin, cout, npos, in, wsp, bias, bias_b, in_scale, in_shift, in_swish, out,
                                    stats_part, minmax, gl, 0, s, true)
                     : pw_launch<1>(b, cin, cout, npos, in, wsp, bias, bias_b, in_scale, in_shift, in_swish, out,
                                    stats_part, minmax, gl, 0, s, true);
  }
  if (flags & 4)
    return pw_launch_split(b, cin, cout, npos, in, wp_any, bias, bias_b, in_scale, in_shift, in_swish, out, stats_part,
                           minmax, pool_u, 0, s);
  const float *wp = (const float *)wp_any;
  const int g = pool_lanes(pool_u);
  return cout > 32 ? pw_launch<2>(b, cin, cout, npos, in, wp, bias, bias_b, in_scale, in_shift, in_swish, out,
                                  stats_part, minmax, g, 0, s)
                   : pw_launch<1>(b, cin, cout, npos, in, wp, bias, bias_b, in_scale, in_shift, in_swish, out,
                                  stats_part, minmax, g, 0, s);
}

extern "C" int p2pb_pointwise_conv_pool_forward(int b, int cin, int cout, int npos, const float *in,
                                                const void *wp_any, const float *bias, const float *bias_b,
                                                const float *in_scale, const float *in_shift, int in_swish, int flags,
                                                float *out, float *stats_part, int pool_u, float *minmax,
                                                void *stream) {
  tl_pw_fin_pending = p2pb_gn_finisher_take(&tl_pw_fin);
  const int rc = pw_conv_pool_forward_impl(b, cin, cout, npos, in, wp_any, bias, bias_b, in_scale, in_shift, in_swish, flags, out,
                                           stats_part, pool_u, minmax, stream);
  return pw_finish_behind(rc, b, cout, npos, stats_part, (hipStream_t)stream);
}

// The last layer of a set abstraction's MLP on the GROUPED tensor without building it (pw_wide_kernel<GATHER>):
//   operand[ci, (mi, ui)] = zt[b, idx[b, mi, ui], ci] - cxt[b, mi, ci]   (zt f32[b,n,cin], cxt f32[b,m,cin] point-major,
//   idx i32[b,m,u]: what p2pb_group_sub writes out as f32[b,cin,m*u]), folded norm + Swish applied on load, then the
//   1x1 convolution with the statistics + neighbourhood {min, max} epilogue of p2pb_pointwise_conv_pool_forward (output
//   never stored). wp_split = the split pack; f16x3 arithmetic; cin % 8 == 0, u in {4, 8, 16, 32, 64}.
static int pw_conv_pool_gather_impl(int b, int cin, int cout, int n, int m, int u, const float *zt,
                                               const float *cxt, const int *idx, const void *wp_split, const float *bias,
                                               const float *in_scale, const float *in_shift, int in_swish,
                                               float *stats_part, float *minmax, void *stream) {
  const long npos = (long)m * u;
  if (b <= 0 || cin <= 0 || cout <= 0 || n <= 0 || m <= 0 || !zt || !idx || !wp_split || !in_scale || !in_shift ||
      !stats_part || !minmax || (cin & 7) || npos > 0x7fffffffL || !p2pb_pointwise_pool_supported((int)npos, u) || u == 0 ||
      p2pb_g_split_terms != SPLIT_F16X3 || (((uintptr_t)zt | (uintptr_t)cxt) & 15))
    return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int P = (int)npos, nslots = (P + 255) / 256 * 4, gl = pool_lanes(u);
  const PwGather gat = {cxt, idx, n, u};
  const float *wsp = (const float *)wp_split;
#define LAUNCHG(MTV, PGV)                                                                                                 \
  hipLaunchKernelGGL((pw_wide_kernel<MTV, true, true, PGV, SPLIT_F16X3, true>),                                    \
                     dim3((P + 511) / 512, (cout + 32 * MTV - 1) / (32 * MTV), b), dim3(256), 0, s, cin, cout,             \
                     pw_cout_pad(cout), P, nslots, zt, wsp, bias, (const float *)nullptr, in_scale, in_shift, in_swish,    \
                     (float *)nullptr, stats_part, minmax, gl, 0, gat)
  if (gl == 8) {
    if (cout > 32) LAUNCHG(2, 8);
    else LAUNCHG(1, 8);
  } else {
    if (cout > 32) LAUNCHG(2, 1);
    else LAUNCHG(1, 1);
  }
#undef LAUNCHG
  p2pb_note_pointwise_form(cin, cout, P, P2PB_FORM_PW_GATHER);
  return p2pb_launch_status();
}

extern "C" int p2pb_pointwise_conv_pool_gather(int b, int cin, int cout, int n, int m, int u, const float *zt,
                                               const float *cxt, const int *idx, const void *wp_split, const float *bias,
                                               const float *in_scale, const float *in_shift, int in_swish,
                                               float *stats_part, float *minmax, void *stream) {
  tl_pw_fin_pending = p2pb_gn_finisher_take(&tl_pw_fin);
  const int rc = pw_conv_pool_gather_impl(b, cin, cout, n, m, u, zt, cxt, idx, wp_split, bias, in_scale, in_shift, in_swish,
                                          stats_part, minmax, stream);
  return pw_finish_behind(rc, b, cout, (int)((long)m * u), stats_part, (hipStream_t)stream);
}


// y = max(act(scale*min + shift), act(scale*max + shift)):
//   nslots == 0: minmax f32[b, c, m, 2] -> y f32[b, c, m]      (set-abstraction neighbour max)
//   nslots  > 0: minmax f32[b, nslots, c, 2] -> y f32[b, c]    (global max-pool; partials reduced first)
__global__ __launch_bounds__(256) void minmax_act_kernel(int c, int m, int nslots, const float *__restrict__ mm,
                                                         const float *__restrict__ scale,
                                                         const float *__restrict__ shift, int swish,
                                                         float *__restrict__ y, size_t total) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    float mn, mx;
    size_t bc;
    if (nslots == 0) {
      bc = e / m;
      const float2 v = *(const float2 *)(mm + e * 2);
      mn = v.x;
      mx = v.y;
    } else {
      bc = e;
      const size_t b = e / c, ch = e % c;
      mn = INFINITY;
      mx = -INFINITY;
      for (int sl = 0; sl < nslots; ++sl) {
        const float2 v = *(const float2 *)(mm + ((b * nslots + sl) * c + ch) * 2);
        mn = fminf(mn, v.x);
        mx = fmaxf(mx, v.y);
      }
    }
    const float sc = scale[bc], sh = shift[bc];
    float lo = mn * sc + sh, hi = mx * sc + sh;
    if (swish) {
      lo = swishf(lo);
      hi = swishf(hi);
    }
    y[e] = fmaxf(lo, hi);
  }
}

// global pool (nslots > 0) with the slot loop spread over 8 waves: 32 channels x 8 slot classes per workgroup, min / max
// combined through LDS (exact, order-free) -- one thread per (sample, channel) walked 128+ slots serially: 69 us for the
// 1024-channel embedding of the bench
// part != NULL (round 5): the GroupNorm that precedes the activation is folded here too -- the workgroup finishes the groups its
// 32 channels belong to from the producing GEMM's statistics partials (gn_finish_group: gn_affine_kernel's bits), WRITES
// scale / shift (fin.scale / fin.shift: the next GEMM applies them to the same tensor on load) and pools with them; the
// gn_affine launch between the GEMM and this kernel is gone.
__global__ __launch_bounds__(256) void minmax_act_pool_kernel(int c, int nslots, const float *__restrict__ mm,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ shift, int swish,
                                                              float *__restrict__ y, const float *__restrict__ part, int nslots_st,
                                                              GnFinish fin) {
  __shared__ float smn[8][32], smx[8][32];
  __shared__ double gl[4 * 256];
  extern __shared__ float mmp_tab[];  // folded form: scale[c] | shift[c] of this sample
  const int b = blockIdx.y, ch = blockIdx.x * 32 + (threadIdx.x & 31), part_i = threadIdx.x >> 5;
  float mn = INFINITY, mx = -INFINITY;
  if (ch < c)
    for (int sl = part_i; sl < nslots; sl += 8) {
      const float2 v = *(const float2 *)(mm + (((size_t)b * nslots + sl) * c + ch) * 2);
      mn = fminf(mn, v.x);
      mx = fmaxf(mx, v.y);
    }
  smn[part_i][threadIdx.x & 31] = mn;
  smx[part_i][threadIdx.x & 31] = mx;
  if (part != nullptr) {
    const int cg = c / fin.groups, c0 = blockIdx.x * 32, c1 = min(c0 + 32, c) - 1;
    for (int g = c0 / cg; g <= c1 / cg; ++g)
      gn_finish_group_v(c, nslots_st, part, fin, b, g, gl, (int)threadIdx.x, true, nullptr, mmp_tab, mmp_tab + c);
  }
  __syncthreads();
  if (part_i != 0 || ch >= c) return;
#pragma unroll
  for (int p = 1; p < 8; ++p) {
    mn = fminf(mn, smn[p][threadIdx.x]);
    mx = fmaxf(mx, smx[p][threadIdx.x]);
  }
  const float sc = part ? mmp_tab[ch] : scale[(size_t)b * c + ch], sh = part ? mmp_tab[c + ch] : shift[(size_t)b * c + ch];
  float lo = mn * sc + sh, hi = mx * sc + sh;
  if (swish) {
    lo = swishf(lo);
    hi = swishf(hi);
  }
  y[(size_t)b * c + ch] = fmaxf(lo, hi);
}

extern "C" int p2pb_minmax_act(int b, int c, int m, int nslots, const float *minmax, const float *scale,
                               const float *shift, int swish, float *y, void *stream) {
  if (b <= 0 || c <= 0 || m <= 0 || nslots < 0) return P2PB_EINVAL;
  if (nslots >= 16) {
    hipLaunchKernelGGL(minmax_act_pool_kernel, dim3((c + 31) / 32, b), dim3(256), 0, (hipStream_t)stream, c, nslots, minmax,
                       scale, shift, swish, y, (const float *)nullptr, 0, GnFinish());
    return p2pb_launch_status();
  }
  const size_t total = nslots == 0 ? (size_t)b * c * m : (size_t)b * c;
  const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(minmax_act_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, c, m, nslots, minmax, scale,
                     shift, swish, y, total);
  return p2pb_launch_status();
}
// the global max-pool (minmax f32[b, nslots, c, 2] -> y f32[b, c]) with the GroupNorm in front of the activation folded in: part
// f32[b, nslots_st, c, 2] = the producing GEMM's statistics partials; scale / shift f32[b, c] are OUTPUTS (p2pb_gn_affine_params'
// values, same bits). Replaces MyGroupNorm + Swish + the max-pool of models/pvcnn.py:905-932 behind a Pnet2Stage GEMM.
extern "C" int p2pb_minmax_act_pool_gn(int b, int c, int nslots, const float *minmax, const float *part, int nslots_st,
                                       double count_per_channel, int groups, const float *gamma, const float *beta,
                                       const float *style, int style_stride, float eps, int swish, float *scale, float *shift,
                                       float *y, void *stream) {
  if (b <= 0 || c <= 0 || nslots <= 0 || !minmax || !part || nslots_st <= 0 || !scale || !shift || !y || groups <= 0 ||
      c % groups != 0 || c / groups > 256 || (style && style_stride < 2 * c) || (size_t)c * 8 > 32 * 1024)
    return P2PB_EINVAL;
  GnFinish f = {};
  f.gamma = gamma, f.beta = beta, f.style = style, f.scale = scale, f.shift = shift, f.style_stride = style_stride, f.groups = groups;
  f.eps = eps, f.count_per_channel = count_per_channel;
  hipLaunchKernelGGL(minmax_act_pool_kernel, dim3((c + 31) / 32, b), dim3(256), (size_t)c * 8, (hipStream_t)stream, c, nslots, minmax,
                     (const float *)nullptr, (const float *)nullptr, swish, y, part, nslots_st, f);
  return p2pb_launch_status();
}

// ------------------------------------------------------------------------------------------------
// nn.Linear on a handful of rows: out[b, co] = bias[co] + sum_ci w[co, ci] x[b, ci]   (b <= a few dozen)
// The per-evaluation Linears of the network -- every AdaGN's style Linear on the global embedding (concatenated: 1024 ->
// 13184 for PVDS, models/modules.py:337-345), the time embedding's two (models/unet_pvc.py:108-112), the global embedding's
// per-sample bias (models/pvcnn.py:926) -- are weight-streaming GEMVs (54 MB of weights for the styles). They used to go
// through torch's BLAS, whose per-stream WORKSPACE a captured graph bakes in: two sampler chains replaying their graphs side
// by side then shared one workspace and corrupted each other's GEMMs (round 4, tests/test_full_size_parity_gpu.py::
// test_c2_bench_dispatch_two_chains_b32 -- the corruption found there turned out to be the devoxelisation's, voxelize.hip; the
// workspace sharing is real all the same).
// This kernel needs no scratch: the exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32: fp32 products, fp32 accumulate) with
// M = 32 weight rows, N = the batch rows (<= 16 per chunk, x staged in LDS with a 4-float row pad: conflict-free 16-byte reads),
// K split over the four waves of a workgroup and summed through LDS in a fixed order (deterministic). A lane's weight operand is
// one 16-byte load W[row][k0 + 4 h .. + 3] feeding four MFMAs (k pairs (4 h + i) of both half-waves), so the weights stream
// through once, 32 contiguous bytes per row and step, every 128-byte line consumed by the same wave within four steps.
#define LR_BC 16
__global__ __launch_bounds__(256) void linear_rows_kernel(int B, int cin, int cout, const float *__restrict__ x, long xs,
                                                          const float *__restrict__ w, long ws,
                                                          const float *__restrict__ bias, float *__restrict__ out, long os,
                                                          int bc) {
  extern __shared__ float lr_x[];  // [bc][cin + 4]; reused for the cross-wave sum [4][32][17]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int row0 = blockIdx.x * 32;
  const int xp = cin + 4;
  const int wrow = min(row0 + l31, cout - 1);  // (clamped rows multiply garbage that is never stored)
  // this wave's K range: whole 8-steps, dealt round-robin to the four waves
  const int nk8 = (cin + 7) / 8;
  for (int b0 = 0; b0 < B; b0 += bc) {
    const int nb = min(bc, B - b0);
    __syncthreads();
    for (int e = tid * 4; e < nb * cin; e += 1024) {
      const int bb = e / cin, c = e - bb * cin;
      *(f32x4 *)(lr_x + bb * xp + c) = *(const f32x4 *)(x + (size_t)(b0 + bb) * xs + c);
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const float *wp_ = w + (size_t)wrow * ws + 4 * h;
    const float *xq = lr_x + min(l31, nb - 1) * xp + 4 * h;
    // weights: EIGHT 16-byte loads per lane in flight (round 5: with four the 54 MB of style weights streamed at 0.7-0.9 TB/s --
    // 8 waves x 4 KB per CU in flight against an HBM latency of microseconds; the multiply is 16 us of matrix time at most)
    const int nit = (nk8 - wave + 3) / 4;
    for (int it0 = 0; it0 < nit; it0 += 8) {
      f32x4 wv[8], xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = (wave + 4 * (it0 + u)) * 8;
        wv[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (it0 + u < nit && k + 4 * h < cin) wv[u] = *(const f32x4 *)(wp_ + k);  // (cin % 4 == 0: a quad is inside or outside)
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = (wave + 4 * (it0 + u)) * 8;
        xv[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (it0 + u < nit && k + 4 * h < cin) xv[u] = *(const f32x4 *)(xq + k);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u][i], xv[u][i], acc, 0, 0, 0);
    }
    __syncthreads();  // everyone is done with x
    float *red = lr_x;  // [4 waves][32 rows][17]
    if (l31 < nb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 17 + l31] = acc[r];
    }
    __syncthreads();
    for (int e = tid; e < 32 * nb; e += 256) {
      const int r = e / nb, bb = e - r * nb;
      if (row0 + r < cout) {
        const float v = (red[(0 * 32 + r) * 17 + bb] + red[(1 * 32 + r) * 17 + bb]) + (red[(2 * 32 + r) * 17 + bb] + red[(3 * 32 + r) * 17 + bb]);
        out[(size_t)(b0 + bb) * os + row0 + r] = v + (bias ? bias[row0 + r] : 0.0f);
      }
    }
  }
}

// x f32[b, cin] (row pitch x_stride floats), w f32[cout, cin] (row pitch w_stride: a column slice of a wider matrix is fine),
// bias f32[cout] or NULL -> out f32[b, cout] (row pitch out_stride). cin % 4 == 0, 16-byte aligned rows.
extern "C" int p2pb_linear_rows(int b, int cin, int cout, const float *x, long x_stride, const float *w, long w_stride,
                                const float *bias, float *out, long out_stride, void *stream) {
  if (b <= 0 || cin <= 0 || cout <= 0 || !x || !w || !out || (cin & 3) || (x_stride & 3) || (w_stride & 3) ||
      (((uintptr_t)x | (uintptr_t)w) & 15) || x_stride < cin || w_stride < cin || out_stride < cout)
    return P2PB_EINVAL;
  int bc = (int)(65536 / ((long)(cin + 4) * 4));  // batch rows per LDS chunk (64 KB: two workgroups per CU)
  if (bc < 1) return P2PB_EINVAL;           // (cin > 16384)
  if (bc > LR_BC) bc = LR_BC;
  if (bc > b) bc = b;
  size_t lr_lds = (size_t)bc * (cin + 4) * 4;
  if (lr_lds < 4 * 32 * 17 * 4) lr_lds = 4 * 32 * 17 * 4;  // (the cross-wave sum's table)
  hipLaunchKernelGGL(linear_rows_kernel, dim3(cdiv(cout, 32)), dim3(256), lr_lds, (hipStream_t)stream, b, cin,
                     cout, x, x_stride, w, w_stride, bias, out, out_stride, bc);
  return p2pb_launch_status();
}

// ------------------------------------------------------------------------------------------------
// y = act(x*scale[b,c] + shift[b,c]) (+ residual)   over [b, c, P]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void affine_act_kernel(int c, int P, const float *__restrict__ x,
                                                         const float *__restrict__ scale,
                                                         const float *__restrict__ shift, int swish,
                                                         const float *__restrict__ residual, float *__restrict__ y) {
  const int bc = blockIdx.y;  // b*c + ch
  const float sc = scale[bc], sh = shift[bc];
  const float *xr = x + (size_t)bc * P;
  const float *rr = residual ? residual + (size_t)bc * P : nullptr;
  float *yr = y + (size_t)bc * P;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) {
    float v = xr[p] * sc + sh;
    if (swish) v = swishf(v);
    if (rr) v = rr[p] + v;
    yr[p] = v;
  }
}

// 16-byte form (rows of whole, aligned quads): a pure streaming pass, HBM-bound
__global__ __launch_bounds__(256) void affine_act4_kernel(int c, int P4, const f32x4 *__restrict__ x,
                                                          const float *__restrict__ scale,
                                                          const float *__restrict__ shift, int swish,
                                                          const f32x4 *__restrict__ residual, f32x4 *__restrict__ y) {
  const int bc = blockIdx.y;
  const float sc = scale[bc], sh = shift[bc];
  const f32x4 *xr = x + (size_t)bc * P4;
  const f32x4 *rr = residual ? residual + (size_t)bc * P4 : nullptr;
  f32x4 *yr = y + (size_t)bc * P4;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < P4; p += gridDim.x * 256) {
    f32x4 v = xr[p];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t = v[i] * sc + sh;
      if (swish) t = swishf(t);
      v[i] = t;
    }
    if (rr) v += rr[p];
    yr[p] = v;
  }
}

extern "C" int p2pb_affine_act(int b, int c, int npos, const float *x, const float *scale, const float *shift,
                               int swish, const float *residual, float *y, void *stream) {
  if (b <= 0 || c <= 0 || npos <= 0) return P2PB_EINVAL;
  if (npos % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15) == 0) {
    const int p4 = npos / 4;
    const unsigned gx = (unsigned)((p4 + 255) / 256 > 64 ? 64 : (p4 + 255) / 256);
    hipLaunchKernelGGL(affine_act4_kernel, dim3(gx, b * c), dim3(256), 0, (hipStream_t)stream, c, p4, (const f32x4 *)x,
                       scale, shift, swish, (const f32x4 *)residual, (f32x4 *)y);
    return p2pb_launch_status();
  }
  const unsigned gx = (unsigned)((npos + 255) / 256 > 64 ? 64 : (npos + 255) / 256);
  hipLaunchKernelGGL(affine_act_kernel, dim3(gx, b * c), dim3(256), 0, (hipStream_t)stream, c, npos, x, scale, shift,
                     swish, residual, y);
  return p2pb_launch_status();
}

// ------------------------------------------------------------------------------------------------
// y[b,c,m] = max_{u < U} act(x[b,c,m,u]*scale + shift), U a power of two <= 64 (32 in every config):
// lanes read the [m,u] plane contiguously, the max runs over aligned groups of U lanes.
// U == 0 selects "max over the whole row" (Pnet2Stage's global max-pool): y[b,c] = max_p act(...).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void affine_act_max_kernel(int M, int U, const float *__restrict__ x,
                                                             const float *__restrict__ scale,
                                                             const float *__restrict__ shift, int swish,
                                                             float *__restrict__ y) {
  const int bc = blockIdx.y;
  const float sc = scale[bc], sh = shift[bc];
  const float *xr = x + (size_t)bc * M * U;
  float *yr = y + (size_t)bc * M;
  const int total = M * U;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {  // total % 64 == 0 by construction
    float v = xr[e] * sc + sh;
    if (swish) v = swishf(v);
    for (int off = U >> 1; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    if ((e & (U - 1)) == 0) yr[e / U] = v;
  }
}

__global__ __launch_bounds__(256) void affine_act_rowmax_kernel(int P, const float *__restrict__ x,
                                                                const float *__restrict__ scale,
                                                                const float *__restrict__ shift, int swish,
                                                                float *__restrict__ y) {
  __shared__ float red[256];
  const int bc = blockIdx.x;
  const float sc = scale[bc], sh = shift[bc];
  const float *xr = x + (size_t)bc * P;
  float mx = -INFINITY;
  for (int p = threadIdx.x; p < P; p += 256) {
    float v = xr[p] * sc + sh;
    if (swish) v = swishf(v);
    mx = fmaxf(mx, v);
  }
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + w]);
    __syncthreads();
  }
  if (threadIdx.x == 0) y[bc] = red[0];
}

extern "C" int p2pb_affine_act_max(int b, int c, int m, int u, const float *x, const float *scale, const float *shift,
                                   int swish, float *y, void *stream) {
  if (b <= 0 || c <= 0 || m <= 0 || u < 0) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (u == 0) {
    hipLaunchKernelGGL(affine_act_rowmax_kernel, dim3(b * c), dim3(256), 0, s, m, x, scale, shift, swish, y);
    return p2pb_launch_status();
  }
  if ((u & (u - 1)) != 0 || u > 64 || ((long)m * u) % 64 != 0) return P2PB_EINVAL;
  const long total = (long)m * u;
  const unsigned gx = (unsigned)((total + 255) / 256 > 64 ? 64 : (total + 255) / 256);
  hipLaunchKernelGGL(affine_act_max_kernel, dim3(gx, b * c), dim3(256), 0, s, m, u, x, scale, shift, swish, y);
  return p2pb_launch_status();
}

