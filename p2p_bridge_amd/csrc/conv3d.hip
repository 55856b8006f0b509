// conv3d.hip -- 3x3x3 / stride 1 / pad 1 voxel convolution on the gfx950 matrix cores, exact fp32.
//
// This is the dominant kernel of the path: 72 % of the network's FLOPs (SURVEY.md 8a row a11,
// models/pvcnn.py:265-284) and MFMA-bound. The reference calls cuDNN (TF32 on NVIDIA); CDNA4 has no
// TF32 but has an exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, 157 TFLOP/s dense), which keeps the
// build's 1e-4 parity budget: every product is rounded once and accumulated in fp32, like an fmaf chain.
//
// Formulation: implicit GEMM   out[co, p] = sum_{tap, ci} W[tap][ci][co] * in[ci, p + off(tap)]
//   M = output channels (MFMA rows), N = voxels (MFMA columns), K = 27 * Cin.
// N is the voxel index on purpose: an accumulator register then holds consecutive w-voxels of one
// output channel across lanes 0..31, so the NCDHW store is lane-consecutive.
//
// Workgroup = 256 threads (4 waves) -> one brick of 256 voxels (8 N-tiles of 32) x NC output channels of
// one sample. Per chunk of CK input channels the workgroup stages the zero-padded halo brick
// [CK][TD+2][TH+2][TW+2] into LDS once (optional per-channel affine + Swish applied on the way in: that is
// how the preceding AdaGN + Swish is fused away), then every wave walks the 27 taps reading its B
// fragments from LDS at constant offsets; A fragments (packed weights) are 16-byte L1/L2 loads.
// Epilogue: + bias, store, and per-(sample, brick, wave, channel) {sum, sum of squares} partials for the
// GroupNorm that follows (reduced deterministically by gn_affine_kernel).
//
// Sparsity (exact, not approximate). A PU-Net patch is a 2-manifold: ~2.5 % of a 32^3 grid is occupied.
//   * first convolution of a PVConv: the input is zero away from the surface -> a (brick, chunk) whose
//     staged halo tile is all zero contributes exactly +0 and its 27x4 MFMA steps are skipped;
//   * second convolution: its input swish(affine(conv0)) equals a per-channel constant a[b,ci] wherever
//     conv0's input was zero (conv0 = bias there, exactly). By linearity
//         conv(x) = conv(x - a) + conv(a),
//     x - a is exactly zero in the far field (same skip applies) and conv(a) -- a constant field with
//     zero padding -- depends only on which of the 27 boundary classes (low/interior/high per axis) the
//     voxel is in: K[b, class, co] = bias + sum_{taps inside} sum_ci W*a, added in the epilogue.
// Every output voxel and every statistic is still produced by this kernel; only all-zero MFMA work is
// skipped. Compact 4x8x8 bricks (instead of full-row bricks) make the zero test fine-grained in 3-D.
#include "common.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Translation units: this file is compiled THREE times (p2p_bridge_amd/build.py), in parallel. The primary object holds
// everything and instantiates the split kernels in the default f16x3 arithmetic; -DCONV_TU=6 builds only the bf16x6
// instantiations of the split kernels behind two bridge functions (conv3d_tu6_split / conv3d_tu6_compact); -DCONV_TU=3
// only the bf16x3 instantiations the training data gradient launches (dense split kernel, channel-major, no operand
// transform: conv3d_tu3_split) -- every extern "C" entry point is compiled out there, every non-template kernel is static.
#ifndef CONV_TU
#define CONV_TU 0
#endif
#if CONV_TU == 6
#define CONV_TERMS SPLIT_BF16X6
#elif CONV_TU == 3
#define CONV_TERMS SPLIT_BF16X3
#else
#define CONV_TERMS SPLIT_F16X3
#endif

#define CONV_CK 8  // input channels per LDS stage

// brick = TD x TH x TW voxels = 8 N-tiles of 32 (2 for R = 4); an N-tile = ND x NH x TW voxels
template <int R, bool COMPACT>
struct ConvGeom;
template <>
struct ConvGeom<32, false> {
  static constexpr int TD = 2, TH = 4, TW = 32, ND = 1, NH = 1;
};
template <>
struct ConvGeom<16, false> {
  static constexpr int TD = 2, TH = 8, TW = 16, ND = 1, NH = 2;
};
template <>
struct ConvGeom<8, false> {
  static constexpr int TD = 4, TH = 8, TW = 8, ND = 1, NH = 4;
};
template <>
struct ConvGeom<4, false> {
  static constexpr int TD = 4, TH = 4, TW = 4, ND = 2, NH = 4;
};
template <>
struct ConvGeom<32, true> : ConvGeom<8, false> {};
template <>
struct ConvGeom<16, true> : ConvGeom<8, false> {};
template <>
struct ConvGeom<8, true> : ConvGeom<8, false> {};
template <>
struct ConvGeom<4, true> : ConvGeom<4, false> {};

// Swish with the hardware exp2 / reciprocal units: v * rcp(1 + exp2(-v*log2(e))). ~1e-6 relative error
// (both units are 1 ulp), an order of magnitude below the fp32 summation-order noise of the dense layers
// and two below the 1e-4 parity budget; 6 VALU ops instead of ~45 for expf + IEEE divide. It matters
// because the activation is recomputed on every operand stage (once per output-channel block).
__device__ __forceinline__ float fast_swish(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}

// the folded operand transform, ONE definition used by the staging code and by far_value_kernel so that
// "x - a" is bit-exactly zero wherever x is the far-field constant
__device__ __forceinline__ float xf_apply(float v, float sc, float sh, int swish) {
  v = v * sc + sh;
  return swish ? fast_swish(v) : v;
}

// MT = 32-row output-channel tiles per workgroup (NC = 32*MT), XF = apply affine(+swish)(-sub) to the input
template <int R, bool COMPACT, int MT, bool XF, bool CL>
__global__ __launch_bounds__(256) void conv3d_k3_kernel(int cin, int cout, int nchunk, int cout_pad,
                                                        const float *__restrict__ in, const float *__restrict__ wt,
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
  constexpr int NTILES = (G::TD * G::TH * G::TW) / 32;  // N-tiles in the brick (8, or 2 for R=4)
  constexpr int BH = R / G::TH, BW = R / G::TW;          // bricks per sample along h, w
  constexpr int R3 = R * R * R;
  __shared__ float tile[CONV_CK * PLANE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  // blockIdx.x -> brick. Workgroups are dealt to the 8 XCDs round-robin (id mod 8); with the zero-tile skip
  // the active bricks hug the surface, and a linear map would park a whole (h,w) column of bricks -- i.e.
  // all of the surface or none of it -- on one XCD. The compact geometry therefore uses a diagonal hash:
  // d-index = (x mod BD) - (3*bh + 5*bw), so consecutive ids walk diagonally through the grid.
  constexpr int BD = R / G::TD;
  constexpr int NBRICK = BD * BH * BW;
  int bd, bh, bw, b = blockIdx.z;
  if (brick_list) {  // compacted list of ACTIVE (sample, brick) pairs; the rest is written by conv3d_fill_kernel
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

  // this wave's two N-tiles: tile index t = 2*wave + s ; origin of an N-tile inside the brick
  int nbase[2];
  bool nact[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int t = 2 * wave + s;
    nact[s] = t < NTILES;
    constexpr int HB = G::TH / G::NH;
    const int td = (t / HB) * G::ND, th = (t % HB) * G::NH;
    const int jw = l31 % G::TW, jr = l31 / G::TW;  // lane's voxel inside the N-tile
    const int jh = jr % G::NH, jd = jr / G::NH;
    nbase[s] = ((td + jd) * HH + (th + jh)) * HW + jw;
  }

  // staging map, fixed for the whole kernel: this thread stages halo positions tid, tid+256, ... of EVERY
  // channel of a chunk (channel-outer order keeps the folded scale/shift wave-uniform, i.e. scalar loads,
  // and needs only NP offsets instead of one per staged element)
  constexpr int NP = (PLANE + 255) / 256;
  int soff[NP];  // offset inside a channel's r^3 grid, or -1 outside the grid / beyond the halo
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
  float stg[CONV_CK][NP];
  // unpredicated loads through scalar descriptors; halo positions outside the grid carry an out-of-range offset
  // and read the hardware's zero. Channel-major (reference) layout: one descriptor per channel row (rows past cin
  // are clamped and zeroed at staging time). Voxel-major layout (CL): a staged voxel's channels are contiguous,
  // 32 bytes per stage = 16-byte loads when cin % 4 == 0 (quads past cin are zeroed at staging time).
  unsigned voff[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j)
    voff[j] = soff[j] >= 0 ? (unsigned)soff[j] * (CL ? (unsigned)cin * 4u : 4u) : 0x80000000u;
  auto stage_load = [&](int ci0) {
    if (CL) {
      auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)inb, 0, R3 * cin * 4, 0x00020000);
      if ((cin & 3) == 0) {
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
          for (int q = 0; q < CONV_CK / 4; ++q) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[j] + (unsigned)(ci0 + 4 * q) * 4u, 0, 0));
#pragma unroll
            for (int i = 0; i < 4; ++i) stg[4 * q + i][j] = v[i];
          }
      } else {
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
          for (int c = 0; c < CONV_CK; ++c)
            stg[c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[j] + (unsigned)(ci0 + c) * 4u, 0, 0));
      }
    } else {
#pragma unroll
      for (int c = 0; c < CONV_CK; ++c) {
        auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)(inb + (size_t)min(ci0 + c, cin - 1) * R3), 0, R3 * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < NP; ++j) stg[c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[j], 0, 0));
      }
    }
  };
  stage_load(0);

  for (int ci0 = 0; ci0 < cin; ci0 += CONV_CK) {
    __syncthreads();  // everyone is done reading the previous chunk's tile
    int nonzero = 0;
#pragma unroll
    for (int c = 0; c < CONV_CK; ++c) {
      float sc = 1.0f, sh = 0.0f, sub = 0.0f;
      const bool cok = ci0 + c < cin;
      if (XF && cok) {
        sc = in_scale[b * cin + ci0 + c];
        sh = in_shift[b * cin + ci0 + c];
        if (in_sub) sub = in_sub[b * cin + ci0 + c];
      }
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        float v = cok ? stg[c][j] : 0.0f;
        if (XF && cok && soff[j] >= 0) v = xf_apply(v, sc, sh, in_swish) - sub;
        nonzero |= (v != 0.0f);
        if (tid + j * 256 < PLANE) tile[c * PLANE + tid + j * 256] = v;
      }
    }
    // barrier + "is any staged value non-zero" in one; an all-zero tile contributes exactly +0
    const int any = skip_zero ? __syncthreads_or(nonzero) : (__syncthreads(), 1);
    if (ci0 + CONV_CK < cin) {  // next chunk's loads fly during the MFMAs
      int nxt = ci0 + CONV_CK;
      asm volatile("" : "+s"(nxt));  // opaque: unpredicated loads would otherwise be hoisted above the staging phase
      stage_load(nxt);
    }
    if (!any) continue;

    // ---- 27 taps x CK/2 k-pairs of MFMAs; A fragments: one 16-byte load per (tap, M-tile), next tap
    //      prefetched while the current one is multiplied
    const float *wchunk = wt + ((((size_t)(ci0 / CONV_CK)) * 2 + khalf) * cout_pad + co0 + l31) * 4;
    const size_t wtap_stride = (size_t)nchunk * 2 * cout_pad * 4;
    f32x4 a_cur[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a_cur[m] = *(const f32x4 *)(wchunk + (size_t)m * 32 * 4);
    // B fragments are read one k-pair ahead, A fragments one tap ahead; the scheduling barriers pin both
    // prefetches (left alone, the scheduler sinks every load to just before its first use, so each group of
    // MFMAs would start with an exposed LDS / L2 round trip)
    float bf[2], bf_nxt[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) bf[s] = tile[khalf * PLANE + nbase[s]];
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      f32x4 a_nxt[MT];
      if (tap + 1 < 27) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
          a_nxt[m] = *(const f32x4 *)(wchunk + (size_t)(tap + 1) * wtap_stride + (size_t)m * 32 * 4);
      }
#pragma unroll
      for (int kk = 0; kk < CONV_CK / 2; ++kk) {
        const int step = tap * (CONV_CK / 2) + kk + 1;  // the (tap, k-pair) after this one
        if (step < 27 * (CONV_CK / 2)) {
          const int ntap = step / (CONV_CK / 2), nkk = step % (CONV_CK / 2);
          const int ntoff = ((ntap / 9) * HH + (ntap / 3) % 3) * HW + ntap % 3;
#pragma unroll
          for (int s = 0; s < 2; ++s) bf_nxt[s] = tile[(2 * nkk + khalf) * PLANE + nbase[s] + ntoff];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            acc[m][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[m][kk], bf[s], acc[m][s], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 2; ++s) bf[s] = bf_nxt[s];
      }
      if (tap + 1 < 27) {
#pragma unroll
        for (int m = 0; m < MT; ++m) a_cur[m] = a_nxt[m];
      }
    }
  }

  // ---- epilogue: bias (or the boundary-class constant), store, GroupNorm partial statistics
  float *outb = out + (size_t)b * cout * R3;
  // voxel coordinates / boundary class of this lane's column in each of the wave's two N-tiles
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
    for (int g = 0; g < 4; ++g) {
      float vv[2][4];  // voxel-major stores: the four consecutive channels of register group g, per N-tile
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * g + i;
        const int co = co0 + m * 32 + i + 8 * g + 4 * khalf;
        const bool cok = co < cout;
        const float bv = (cok && !out_class) ? bias[co] : 0.0f;
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (!nact[s]) continue;
          float v = acc[m][s][r] + bv;
          if (out_class && cok) v += out_class[((size_t)b * 27 + cls[s]) * cout + co];
          if (CL) vv[s][i] = v;
          else if (cok) outb[(size_t)co * R3 + vox[s]] = v;
          s1 += v;
          s2 += v * v;
        }
        if (stats_part) {
          // sum over the 32 lanes of this half-wave (a channel row lives in exactly one half of the wave),
          // one private slot per (sample, brick, wave, channel): plain stores, reduced later in fixed order
          s1 = halfwave_sum_to_last(s1);
          s2 = halfwave_sum_to_last(s2);
          if (l31 == 31 && cok) {
            float *p = stats_part + ((((size_t)b * NBRICK + brick) * 4 + wave) * cout + co) * 2;
            p[0] = s1;
            p[1] = s2;
          }
        }
      }
      if (CL) {
        const int cq = co0 + m * 32 + 8 * g + 4 * khalf;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (!nact[s]) continue;
          float *q = outb + (size_t)vox[s] * cout + cq;
          if (cq + 3 < cout && (cout & 3) == 0) *(f32x4 *)q = f32x4{vv[s][0], vv[s][1], vv[s][2], vv[s][3]};
          else
            for (int i = 0; i < 4; ++i)
              if (cq + i < cout) q[i] = vv[s][i];
        }
      }
    }
  }
}

// ================================================================================================
// Split-operand form (the default): the same implicit GEMM on the bf16 matrix pipe, fp32-faithful.
//
// gfx950 multiplies fp32 on the matrix cores at 1/16 of the bf16 rate (v_mfma_f32_32x32x2_f32: 2048 MACs
// per 64 cycles; v_mfma_f32_32x32x16_bf16: 16384 per 32), and has no TF32. Each fp32 operand is therefore
// split into three bf16 terms, x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)
// (round-to-nearest; the two residuals are exact in fp32), which carries 24+ significand bits, and a product
// is evaluated as the six terms
//        x*y ~= x2*y0 + x1*y1 + x0*y2 + x1*y0 + x0*y1 + x0*y0        (bf16 x bf16 is exact in fp32)
// accumulated in the fp32 MFMA accumulator, small terms first. The three dropped terms are below
// 2^-26 |x*y| (a quarter of an fp32 ulp), so the result differs from the exact-fp32 MFMA kernel above by
// less than a change of summation order: measured against fp64 on the network's layer shapes the rms error
// is 1.6e-7 for this kernel vs 1.9e-7 for the fp32 MFMA one (tests/test_fused_gpu.py pins this).
// Six bf16 MFMAs (192 cycles) replace eight fp32 ones (512 cycles) per 32x32x16 block: 2.67x fewer matrix
// cycles; measured 196 vs 120 TFLOP/s (fp32-equivalent) on the 128->128 r=16 layer.
//
// Layout: LDS tile[split][khalf][halo voxel] of 16-byte groups = 8 consecutive input channels as bf16, so a
// lane's B fragment of one MFMA is one ds_read_b128; weights pre-split and packed
// [tap][chunk16][split][khalf][cout_pad][8 bf16] so an A fragment is one 16-byte load. The operand
// transform (folded norm + Swish, far-field subtraction), the zero-tile skip, the brick lists and the epilogue
// are those of the fp32 kernel; the split happens once per staged element and is reused by 27 taps.
// ================================================================================================
#define CONV_SCK 16  // input channels per LDS stage of the split kernel = K of one bf16 MFMA
// byte offset of the trailer {max|w| bits, 1 / (S_x S_w)} behind a split pack (fp16 mode, common.h)
static __host__ __device__ size_t conv_split_trailer_bytes(int nchunk, int cout_pad) {
  return (size_t)27 * nchunk * 3 * 2 * cout_pad * 8 * sizeof(unsigned short);
}

// packed weights: wt[tap][chunk16][split 3][khalf 2][cout_pad][8 bf16]; element idx = channel chunk*16 + khalf*8 + idx
// mode SPLIT_F16X3 (common.h): planes 0, 1 = the fp16 pair of w * S_w, plane 2 unused; trailer = {max|w| bits, 1 / (S_x S_w)}
// w element (co, ci, tap) at w[co * s_co + ci * s_ci + (flip ? 26 - tap : tap)]: (cin * 27, 27, no flip) for a layer's own
// weight; (27, cout * 27, flip) packs the ADJOINT (data-gradient) operator straight from the forward weight [cin][cout][27]
// -- a correlation's adjoint is the correlation with the point-reflected kernel and the channel roles swapped
static __global__ void conv3d_pack_split_kernel(int cout, int cin, int nchunk, int cout_pad, const float *__restrict__ w,
                                        unsigned short *__restrict__ wt, int mode, float *__restrict__ trailer, long s_co,
                                        long s_ci, int flip, const unsigned *__restrict__ amax) {
  const size_t total = (size_t)27 * nchunk * 2 * cout_pad * 8;  // one thread per (tap, chunk, khalf, co, idx)
  // max |w| (bits): from the caller's slot (amax: the optimiser keeps it per tensor, csrc/optim.hip) or from the reduction
  // launched in front of this kernel (trailer[0])
  const bool x2w = (mode & SPLIT_X2W_FLAG) != 0;  // (pricing experiment, common.h: a zero low plane)
  mode &= 0xff;
  const float wmax = amax ? __builtin_bit_cast(float, *amax) : trailer[0];
  const float sw = mode == SPLIT_F16X3 ? f16_weight_scale(wmax) : 1.0f;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    trailer[1] = mode == SPLIT_F16X3 ? 1.0f / (SPLIT_F16_SX * sw) : 1.0f;
    if (amax) trailer[0] = wmax, trailer[2] = trailer[3] = 0.0f;  // (the whole trailer, as the zero fill of the other path)
  }
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int idx = (int)(e & 7);
    size_t q = e >> 3;
    const int co = (int)(q % cout_pad);
    q /= cout_pad;
    const int kh = (int)(q & 1);
    q >>= 1;
    const int chunk = (int)(q % nchunk), tap = (int)(q / nchunk);
    const int ci = chunk * CONV_SCK + kh * 8 + idx;
    const float x = (co < cout && ci < cin) ? w[(size_t)co * s_co + (size_t)ci * s_ci + (flip ? 26 - tap : tap)] : 0.0f;
    unsigned p0, p1, p2;
    if (mode == SPLIT_F16X3) {
      split2h(x * sw, 0.0f, p0, p1);
      if (x2w) p1 = 0u;
      p2 = 0u;
    } else {
      split3(x, 0.0f, p0, p1, p2);
    }
    const unsigned p[3] = {p0, p1, p2};
    for (int s = 0; s < 3; ++s)
      wt[((((size_t)(tap * nchunk + chunk) * 3 + s) * 2 + kh) * cout_pad + co) * 8 + idx] = (unsigned short)(p[s] & 0xffff);
  }
}

// Brick geometry of the split kernel: 4 x 8 x 8 bricks whose N-tiles are 4(d) x 1(h) x 8(w) columns. A B fragment is
// a ds_read_b128, which the LDS serves in four groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...),
// one cycle per group if the 16 lanes hit 16 different 16-byte bank groups. With the d-plane pitch of the halo
// tile (100 slots = 4 mod 16) the four d-rows of an N-tile start 4 bank groups apart, and swapping the two w-halves
// in rows 1 and 2 (lane_w below) gives every service group the residues {0..15} exactly once -- conflict-free for
// every tap offset (a tap only adds a constant). The h-row shape of the fp32 kernel is 3-way conflicted (pitch 10).
template <int R>
struct SplitGeom {
  static constexpr int TD = 4, TH = 8, TW = 8, ND = 4, NH = 1;
};
template <>
struct SplitGeom<4> : ConvGeom<4, false> {};
template <int TW>
__device__ __forceinline__ int lane_w(int l31) {
  const int jw = l31 % TW, jr = l31 / TW;
  return (TW == 8 && (jr == 1 || jr == 2)) ? jw ^ 4 : jw;
}

#ifndef CONV_NTAPS
#define CONV_NTAPS 27  // (timing experiments compile fewer)
#endif
// The 27-tap MFMA loop of one input stage (16 channels) for NT column tiles of one M-tile.
// Six products per (tap, tile), the small ones first: x0y2, x1y1, x2y0 | x1y0, x0y1 | x0y0. The B fragments roll
// through ONE register set: y2 of the next tap is read as soon as this tap's x0y2 products have issued, y1 after
// x0y1, y0 at the top of the tap (it is first needed by the third product) -- every LDS read has >= 2 NT MFMAs
// in front of its first use without a second fragment buffer; A fragments come straight from L2, one tap ahead.
// The scheduling barriers pin this order, else every load sinks to its first use.
// TERMS == SPLIT_F16X3 (the default, p2pb_set_split_terms): the fp16-pair split of common.h -- two operand planes, three
// products h1g0, h0g1, h0g0 (<= 3 * 2^-22 |x*y| inside fp16's range); the third plane of tile / pack is then unused.
// (The same three products of the bf16 split -- TERMS == SPLIT_BF16X3, <= 3 * 2^-18 -- were measured at the same speed
// and 6.4e-5 network error, at the 1e-4 parity bar instead of inside it: superseded, not instantiated.)
#ifndef SPLIT_TAPS_AD
#define SPLIT_TAPS_AD 3  // taps of weight prefetch in split_taps (experiment builds: -DSPLIT_TAPS_AD=1 is the round-4 schedule)
#endif
template <int NT, int HH, int HW, int PLANE, int TERMS>
__device__ __forceinline__ void split_taps(f32x16 (&acc)[NT], const u32x4 *__restrict__ tile, const u32x4 *wchunk,
                                           size_t wsplit_stride, size_t wtap_stride, const int (&nbase)[NT], int khalf) {
  constexpr int NP = split_planes(TERMS);  // operand planes in use
  // A fragments (weights) in a ring, requested AD taps ahead of their first product (round 5: one tap ahead -- 6 NT MFMAs, 192 NT
  // cycles -- does not cover an L2 round trip when a SIMD holds ONE wave with NT = 2: the 8^3 layers of a training batch of 8 took
  // 102 us for 36 us of matrix time)
  constexpr int AD = TERMS == SPLIT_BF16X6 ? 1 : SPLIT_TAPS_AD;  // (three operand planes: the deeper ring spills the 64-channel forms)
  u32x4 a_ring[AD + 1][3], bf[3][NT];
#pragma unroll
  for (int t = 0; t < AD; ++t)
#pragma unroll
    for (int s = 0; s < NP; ++s) a_ring[t][s] = wchunk[(size_t)t * wtap_stride + s * wsplit_stride];
  auto load_b = [&](int s, int toff) {
#pragma unroll
    for (int n = 0; n < NT; ++n) bf[s][n] = tile[(s * 2 + khalf) * PLANE + nbase[n] + toff];
  };
  if constexpr (TERMS == 6) load_b(2, 0);
  load_b(1, 0);
#pragma unroll
  for (int tap = 0; tap < CONV_NTAPS; ++tap) {
    const int toff = ((tap / 9) * HH + (tap / 3) % 3) * HW + tap % 3;
    const int toff_n = (((tap + 1) / 9) * HH + ((tap + 1) / 3) % 3) * HW + (tap + 1) % 3;
    const u32x4(&a_cur)[3] = a_ring[tap % (AD + 1)];
    auto mfma_term = [&](int pa, int pb) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
        acc[n] = split_mfma<TERMS>(a_cur[pa], bf[pb][n], acc[n]);
    };
    if (tap + AD < CONV_NTAPS) {
#pragma unroll
      for (int s = 0; s < NP; ++s) a_ring[(tap + AD) % (AD + 1)][s] = wchunk[(size_t)(tap + AD) * wtap_stride + s * wsplit_stride];
    }
    load_b(0, toff);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TERMS == 6) {
      mfma_term(0, 2);
      __builtin_amdgcn_sched_barrier(0);
      if (tap + 1 < CONV_NTAPS) load_b(2, toff_n);
      __builtin_amdgcn_sched_barrier(0);
      mfma_term(1, 1);
      mfma_term(2, 0);
      mfma_term(1, 0);
    }
    mfma_term(0, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (tap + 1 < CONV_NTAPS) load_b(1, toff_n);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TERMS != 6 && (X2W_KEEP_LOW_WEIGHT_PRODUCT || TERMS != SPLIT_F16X3)) mfma_term(1, 0);
    mfma_term(0, 0);
  }
}

// The tap loop of the PRE = true kernels (f16x3): split_taps' B schedule, with the A fragments (weights) in a ring of
// three tap slots loaded TWO taps ahead and carried across stages -- a[t % 3] is tap t's; on entry a[0], a[1] hold taps
// 0, 1 of this stage, on exit those of the next one (has_next). Memory returns are in order per wave, so the first A load
// issued behind the LDS-DMA burst of the next stage cannot return before that burst has landed: with two taps of weights
// already in registers the burst has two taps of MFMAs (>= 768 cycles) to do so. Weights come through a buffer descriptor:
// per-lane byte offset wv (one register) + a scalar offset per (stage, tap, plane) -- no 64-bit address per tap.
__device__ __forceinline__ u32x4 conv_wload(__amdgpu_buffer_rsrc_t rs, unsigned wv, unsigned so) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, wv, so, 0));
}
#ifndef CONV_PRE_AD
// taps of weight prefetch; the ring has CONV_PRE_AD + 1 slots, which must divide 27. Round 3 measured 8 against 2 at +-0 -- on one
// chain of 32 patches, two waves per SIMD. With the sampler's two chains of 16 the 8^3 layers launch 256 workgroups: ONE wave per
// SIMD with two tiles, 192 cycles of MFMAs per tap, and two taps of cover are less than an L2 round trip (what section 3.7 found in
// split_taps): 8 taps ahead, bench 223.1 -> 220.6 ms per sample call (profiles/r05b_conv_pre_ad8_ab.txt; 72 ring registers, none spilled)
#define CONV_PRE_AD 8
#endif
static_assert(27 % (CONV_PRE_AD + 1) == 0, "the ring position of tap 0 must be the same in every stage");
template <int NT, int HH, int HW, int PLANE>
__device__ __forceinline__ void split_taps_pre(f32x16 (&acc)[NT], const u32x4 *__restrict__ tile, __amdgpu_buffer_rsrc_t rsw,
                                               unsigned wv, unsigned sbase, unsigned stage_bytes, unsigned tap_bytes,
                                               unsigned plane_bytes, bool has_next, const int (&nbase)[NT], int khalf,
                                               u32x4 (&a)[CONV_PRE_AD + 1][2]) {
  // B fragments: plane 1 (h1) in one register set, plane 0 (h0) in TWO (tap parity): every ds_read_b128 is issued a full
  // eight MFMAs (>= 256 cycles) before its first use -- with a single h0 set its reads could only start once the
  // previous tap's last product had issued, four MFMAs (128 cycles, about one LDS latency under load) ahead of their use.
  //   tap t:  [A loads of tap t + AD]  G1: a0(t) x h1(t)  | read h1(t+1) |  G2: a1(t) x h0(t)  | read h0(t+1) |  G3: a0(t) x h0(t)
  // (same three products in the same order as split_taps: bit-identical accumulators)
  u32x4 b1[NT], b0[2][NT];
  auto load_b1 = [&](int toff) {
#pragma unroll
    for (int n = 0; n < NT; ++n) b1[n] = tile[(2 + khalf) * PLANE + nbase[n] + toff];
  };
  auto load_b0 = [&](int set, int toff) {
#pragma unroll
    for (int n = 0; n < NT; ++n) b0[set][n] = tile[khalf * PLANE + nbase[n] + toff];
  };
  load_b1(0);
  load_b0(0, 0);
#pragma unroll
  for (int tap = 0; tap < CONV_NTAPS; ++tap) {
    const int toff_n = (((tap + 1) / 9) * HH + ((tap + 1) / 3) % 3) * HW + (tap + 1) % 3;
    constexpr int AD = CONV_PRE_AD;
    const int cur = tap % (AD + 1), nx2 = (tap + AD) % (AD + 1), par = tap & 1;
    if (tap + AD < CONV_NTAPS) {
      const unsigned so = sbase + (unsigned)(tap + AD) * tap_bytes;
#pragma unroll
      for (int s = 0; s < 2; ++s) a[nx2][s] = conv_wload(rsw, wv, so + s * plane_bytes);
    } else if (has_next) {
      const unsigned so = sbase + stage_bytes + (unsigned)(tap + AD - CONV_NTAPS) * tap_bytes;
#pragma unroll
      for (int s = 0; s < 2; ++s) a[nx2][s] = conv_wload(rsw, wv, so + s * plane_bytes);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = split_mfma<SPLIT_F16X3>(a[cur][0], b1[n], acc[n]);
    __builtin_amdgcn_sched_barrier(0);
    if (tap + 1 < CONV_NTAPS) load_b1(toff_n);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NT; ++n)
      if (X2W_KEEP_LOW_WEIGHT_PRODUCT) acc[n] = split_mfma<SPLIT_F16X3>(a[cur][1], b0[par][n], acc[n]);
    __builtin_amdgcn_sched_barrier(0);
    if (tap + 1 < CONV_NTAPS) load_b0(par ^ 1, toff_n);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = split_mfma<SPLIT_F16X3>(a[cur][0], b0[par][n], acc[n]);
  }
}

// ---- pre-split operand grids ("S format", round 3) ---------------------------------------------------------------
// The staging phase of the split kernels -- load fp32, folded norm + Swish, fp16-pair split, LDS write, redone for every
// brick whose 6x10x10 halo holds the voxel (2.34 x) and for every output-channel block -- is half of their time on the
// f16x3 arithmetic. PRE = true kernels take the operand ALREADY transformed and split, in the exact byte layout of the LDS
// tile, and bring a stage into LDS with LDS-DMA (buffer_load_dwordx4 ... lds: no registers, no VALU, no ds_write):
//     S[b][voxel][chunk16][plane 2][khalf 2] of 16 bytes = 8 fp16  (h0 | h1 of 4 x value, channels chunk*16 + khalf*8 + i)
// i.e. 4 bytes per (voxel, channel) like the fp32 grid it replaces, channel count padded to a multiple of 16. (A PLANAR
// order [chunk][plane][khalf][voxel] -- 160-byte runs per DMA instruction instead of one cache line per lane -- was built
// and measured: the convolutions alone 6 % faster, the bench 1.4 % SLOWER, because both producers then write through an
// LDS transpose or read strided; profiles/README.md.) Producers:
// the voxeliser for a first convolution (voxelize.hip vox_gather_cl_split_kernel: no extra pass), conv3d_presplit_kernel for
// a second one (one elementwise pass over y1 once its GroupNorm statistics are folded). Same transform, same split, same
// products in the same order as the staging code below: outputs are bit-identical to the PRE = false kernels.
// The stage loop is double-buffered (2 x 37.5 KB, two workgroups per CU) with ONE barrier per stage: wait for my DMA of
// stage k, barrier, issue the DMA of stage k + 1 into the other buffer, 27 taps on buffer k. Halo slots outside the grid
// carry an out-of-range buffer offset: the hardware's zero lands in LDS.
#ifdef CONV_TIMELINE  // experiment builds only (tools/exp_conv_timeline.py): s_memtime stamps of one wave per workgroup
__device__ unsigned long long *conv_tl_buf;
extern "C" int p2pb_conv_timeline_set(void *p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(conv_tl_buf), &p, sizeof(p));
}
#define CONV_TL_INIT const unsigned tl_lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); int tl_n = 0;
#define CONV_TL(tid_) do { if ((tid_) == 0 && conv_tl_buf && tl_n < 15) conv_tl_buf[(size_t)tl_lin * 16 + 1 + tl_n++] = __builtin_readcyclecounter(); } while (0)
#define CONV_TL_AT(tid_, slot_) do { if ((tid_) == 0 && conv_tl_buf) conv_tl_buf[(size_t)tl_lin * 16 + 1 + (slot_)] = __builtin_readcyclecounter(); } while (0)
#define CONV_TL_ID(tid_) do { if ((tid_) == 0 && conv_tl_buf) conv_tl_buf[(size_t)tl_lin * 16] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); } while (0)
#else
#define CONV_TL_INIT
#define CONV_TL(tid_)
#define CONV_TL_AT(tid_, slot_)
#define CONV_TL_ID(tid_)
#endif
typedef int conv_i32x4 __attribute__((ext_vector_type(4)));
template <int R, int HD, int HH, int HW>
struct PreStage {
  static constexpr int PLANE = HD * HH * HW, NF = 4 * PLANE, NJ = (NF + 255) / 256;
  unsigned off[NJ];  // byte offset of (voxel, stage 0, quarter) from the sample's base; 0x80000000: outside the grid
  __device__ __forceinline__ void init(int tid, int d0, int h0, int w0, int nchunk) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int f = tid + j * 256;
      const int q = f / PLANE, e = f % PLANE;
      const int dz = e / (HH * HW), hy = (e / HW) % HH, wx = e % HW;
      const int d = d0 - 1 + dz, h = h0 - 1 + hy, w = w0 - 1 + wx;
      const bool ok = f < NF && (unsigned)d < (unsigned)R && (unsigned)h < (unsigned)R && (unsigned)w < (unsigned)R;
      off[j] = ok ? ((unsigned)((d * R + h) * R + w) * (unsigned)(nchunk * 4) + (unsigned)q) * 16u : 0x80000000u;
    }
  }
  // stage `chunk` of the sample behind rs -> buf[0 .. NF); every wave issues its 64-slot runs (lane l lands at run + l).
  // Issued as inline assembly ON PURPOSE: the compiler's wait-count pass assumes that any ds_read may alias the
  // destination of an LDS-DMA it knows of and puts `s_waitcnt vmcnt(0)` in front of the first fragment read after the
  // burst -- which serialises the DMA of stage k + 1 with the taps of stage k (separate __shared__ objects do not help
  // with this compiler). The hand-written form is invisible to that pass; the kernel orders it itself: the builtin
  // s_waitcnt vmcnt(0) + barrier at the top of the next stage. (The pass's own waits for the weight loads it DOES know of
  // can only be stricter than needed: per-wave memory returns are in order.)
  __device__ __forceinline__ void issue(conv_i32x4 rs, int chunk, u32x4 *buf, int tid) const {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void *)buf);
    const unsigned so = (unsigned)chunk * 64u;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int f0 = j * 256 + wave * 64;  // wave-uniform
      if (f0 < NF) {
        const unsigned m0v = base + (unsigned)f0 * 16u;
        if (f0 + lane < NF)
          asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                       :
                       : "v"(off[j]), "s"(rs), "s"(so), "s"(m0v)
                       : "memory");  // (m0 is a reserved register: the compiler does not track it as a clobber -- and uses it nowhere in this
                                     //  object, tools/disasm.sh conv3d: every m0 reference is one of these s_mov_b32)
      }
    }
  }
};
// buffer descriptor words for the inline-assembly DMA above (what __builtin_amdgcn_make_buffer_rsrc(p, 0, bytes, 0x00020000)
// builds), forced into scalar registers
__device__ __forceinline__ conv_i32x4 conv_make_rsrc(const void *p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  conv_i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}

#ifndef CONV_F16_WAVES
#define CONV_F16_WAVES 2  // (waves per SIMD the f16x3 forms are compiled for; their two-plane tile would fit three workgroups)
#endif
template <int R, bool COMPACT, int MT, bool XF, bool CL, int TERMS, bool PRE = false>
__global__ __launch_bounds__(256, TERMS == SPLIT_F16X3 ? CONV_F16_WAVES : 2) void conv3d_k3_split_kernel(int cin, int cout, int nchunk, int cout_pad,
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
  using G = SplitGeom<R>;
  constexpr int HD = G::TD + 2, HH = G::TH + 2, HW = G::TW + 2;
  constexpr int PLANE = HD * HH * HW;
  constexpr int NTILES = (G::TD * G::TH * G::TW) / 32;
  constexpr int BH = R / G::TH, BW = R / G::TW;
  constexpr int R3 = R * R * R;
  // tile[split][khalf][voxel] : 8 bf16 (16 bytes) = channels khalf*8 .. khalf*8+7 of the staged chunk (PRE: two buffers)
  static_assert(!PRE || (TERMS == SPLIT_F16X3 && !XF && CL), "pre-split operands: f16x3, voxel-major, transform applied");
  __shared__ u32x4 tile[split_planes(TERMS) * 2 * PLANE];
  // (PRE: the second stage buffer is its OWN object, and the stage loop is unrolled by two with the roles fixed, so that
  //  the compiler can tell the LDS-DMA into one buffer from the fragment reads of the other -- with one array it waits
  //  vmcnt(0) for the DMA burst of stage k + 1 in front of the first ds_read of stage k)
  __shared__ u32x4 tile2[PRE ? split_planes(TERMS) * 2 * PLANE : 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  constexpr int BD = R / G::TD;
  constexpr int NBRICK = BD * BH * BW;
  // Workgroup -> (sample, brick, channel block). The hardware deals workgroups to the 8 XCDs round-robin in launch
  // order (id mod 8), and every XCD has its own 4 MB L2. The launch id is therefore re-read as (xcd, j) and XCD x is
  // given the x-th CONTIGUOUS eighth of the work list, ordered (sample, brick, channel block): the bricks of a
  // sample -- whose 6x10x10 halos overlap 2.34x -- and both channel blocks of a brick then share one L2.
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const unsigned nblk = gridDim.x * gridDim.y * gridDim.z;
  const int ncoblk = (cout + 32 * MT - 1) / (32 * MT);
  int bd, bh, bw, b, coblk;
  if (brick_list) {  // compacted list of ACTIVE (sample, brick) pairs; the rest is written by conv3d_fill_kernel
    const unsigned total = (unsigned)(*brick_count) * ncoblk;
    unsigned v = lin;
    if (nblk % 8 == 0) {
      const unsigned per = (total + 7) / 8;
      if (lin / 8 >= per) return;
      v = (lin % 8) * per + lin / 8;
    }
    if (v >= total) return;
    const int entry = brick_list[v / ncoblk];
    coblk = v % ncoblk;
    b = entry / NBRICK;
    const int bk = entry % NBRICK;
    bd = bk / (BH * BW);
    bh = (bk / BW) % BH;
    bw = bk % BW;
  } else {
    const unsigned v = nblk % 8 == 0 ? (lin % 8) * (nblk / 8) + lin / 8 : lin;
    const unsigned per_sample = NBRICK * ncoblk;
    b = v / per_sample;
    const unsigned rem = v % per_sample;
    const int bk = rem / ncoblk;
    coblk = rem % ncoblk;
    bd = bk / (BH * BW);
    bh = (bk / BW) % BH;
    bw = bk % BW;
  }
  const int brick = (bd * BH + bh) * BW + bw;
  const int d0 = bd * G::TD, h0 = bh * G::TH, w0 = bw * G::TW;
  // waves as WM x WN: every wave owns ONE 32-channel M-tile and NT N-tiles of the brick. With 64 channels per
  // workgroup (MT = 2) that is 2 x 2 waves of 4 N-tiles each: an A fragment (weights, a 16-byte L1/L2 load
  // per lane) then feeds four N-tiles instead of two -- measured, the A stream through the L1 is what bounds
  // this kernel (removing it: 179 -> 230 TFLOP/s), while B fragments come from LDS, which has room.
  constexpr int WM = MT, WN = 4 / WM, NT = (NTILES / WN) > 0 ? NTILES / WN : 1;
  const int wm = wave / WN, wn = wave % WN;
  const int co0 = coblk * (32 * MT) + 32 * wm;

  int nbase[NT];
  bool nact[NT];
#pragma unroll
  for (int s = 0; s < NT; ++s) {
    const int t = NT * wn + s;
    nact[s] = t < NTILES;
    constexpr int HB = G::TH / G::NH;
    const int td = (t / HB) * G::ND, th = (t % HB) * G::NH;
    const int jw = lane_w<G::TW>(l31), jr = l31 / G::TW;
    const int jh = jr % G::NH, jd = jr / G::NH;
    nbase[s] = ((td + jd) * HH + (th + jh)) * HW + jw;
  }

  f32x16 acc[NT];
#pragma unroll
  for (int s = 0; s < NT; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.0f;
  // The epilogue's additive constants are fetched HERE, sixteen loads in one batch under the stage loop. Left inside the
  // epilogue's (branchy) row loop the compiler issued them one at a time, each followed by its own vmcnt(0): sixteen
  // serialised L2 round trips per wave, 64 more for the boundary-class constants of a second convolution -- a timeline of
  // the r = 32 C64 launch (tools/exp_conv_timeline.py) showed 43 k of a workgroup's 144 k cycles in the epilogue.
  float bvr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
    bvr[r] = (co < cout && !out_class) ? bias[co] : 0.0f;
  }

  if constexpr (PRE) {
    // `in` = the pre-split operand grid (S format): LDS-DMA stages, two buffers, one barrier per stage
    CONV_TL_INIT
    CONV_TL_ID(tid);
    CONV_TL(tid);  // 0: start (after the index arithmetic above)
    PreStage<R, HD, HH, HW> ps;
    ps.init(tid, d0, h0, w0, nchunk);
    const conv_i32x4 sg = conv_make_rsrc((const u32x4 *)in + (size_t)b * R3 * nchunk * 4, (unsigned)(R3 * nchunk * 64));
    ps.issue(sg, 0, tile, tid);
    CONV_TL(tid);  // 1: first DMA issued
    // weights: [tap][stage][plane 3][khalf 2][cout_pad] of 16 bytes
    const unsigned stage_bytes = 6u * cout_pad * 16u, tap_bytes = (unsigned)nchunk * stage_bytes, plane_bytes = 2u * cout_pad * 16u;
    auto rsw = __builtin_amdgcn_make_buffer_rsrc((void *)wt, 0, 27 * (int)tap_bytes, 0x00020000);
    const unsigned wv = (unsigned)(khalf * cout_pad + co0 + l31) * 16u;
    u32x4 aring[CONV_PRE_AD + 1][2];
#pragma unroll
    for (int t = 0; t < CONV_PRE_AD; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) aring[t][s] = conv_wload(rsw, wv, t * tap_bytes + s * plane_bytes);
    auto stage = [&](int k, const u32x4 *cur, u32x4 *nxt) {
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): my share of stage k has landed (and the weights of its first taps)
      __syncthreads();                     // everyone's has; the other buffer is no longer read
      CONV_TL(tid);  // 2 + 2k: stage k released
      if (k + 1 < nchunk) ps.issue(sg, k + 1, nxt, tid);
      split_taps_pre<NT, HH, HW, PLANE>(acc, cur, rsw, wv, (unsigned)k * stage_bytes, stage_bytes, tap_bytes, plane_bytes,
                                        k + 1 < nchunk, nbase, khalf, aring);
      CONV_TL(tid);  // 3 + 2k: taps of stage k issued
    };
    for (int k = 0; k < nchunk; k += 2) {
      stage(k, tile, tile2);
      if (k + 1 < nchunk) stage(k + 1, tile2, tile);
    }
#ifdef CONV_TIMELINE
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if (tid == 0 && conv_tl_buf) conv_tl_buf[(size_t)tl_lin * 16 + 14] = __builtin_readcyclecounter();  // 14: accumulators final
#endif
  } else {
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

  const float *inb = in + (size_t)b * cin * R3;
  float stg[CONV_SCK][NP];
  // unpredicated loads through scalar descriptors; halo positions outside the grid carry an out-of-range offset
  // and read the hardware's zero. Channel-major (reference) layout: one descriptor per channel row (rows past cin
  // are clamped and zeroed at staging time). Voxel-major layout (CL): a staged voxel's channels are contiguous,
  // 64 bytes per stage = 16-byte loads when cin % 4 == 0 (quads past cin are zeroed at staging time).
  unsigned voff[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j)
    voff[j] = soff[j] >= 0 ? (unsigned)soff[j] * (CL ? (unsigned)cin * 4u : 4u) : 0x80000000u;
  auto stage_load = [&](int ci0) {
    if (CL) {
      auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)inb, 0, R3 * cin * 4, 0x00020000);
      if ((cin & 3) == 0) {
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
          for (int q = 0; q < CONV_SCK / 4; ++q) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[j] + (unsigned)(ci0 + 4 * q) * 4u, 0, 0));
#pragma unroll
            for (int i = 0; i < 4; ++i) stg[4 * q + i][j] = v[i];
          }
      } else {
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
          for (int c = 0; c < CONV_SCK; ++c)
            stg[c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[j] + (unsigned)(ci0 + c) * 4u, 0, 0));
      }
    } else {
#pragma unroll
      for (int c = 0; c < CONV_SCK; ++c) {
        auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)(inb + (size_t)min(ci0 + c, cin - 1) * R3), 0, R3 * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < NP; ++j) stg[c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[j], 0, 0));
      }
    }
  };
  stage_load(0);

  for (int ci0 = 0; ci0 < cin; ci0 += CONV_SCK) {
    __syncthreads();
    int nonzero = 0;
#pragma unroll
    for (int c = 0; c < CONV_SCK; ++c) {
      float sc = 1.0f, sh = 0.0f, sub = 0.0f;
      const bool cok = ci0 + c < cin;
      if (XF && cok) {
        sc = in_scale[b * cin + ci0 + c];  // (wave-uniform: through the scalar cache)
        sh = in_shift[b * cin + ci0 + c];
        if (in_sub) sub = in_sub[b * cin + ci0 + c];
      }
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        float v = cok ? stg[c][j] : 0.0f;
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
            split_pair<TERMS>(stg[h * 8 + 2 * i][j], stg[h * 8 + 2 * i + 1][j], p0, p1, p2);
            q[0][i] = p0;
            q[1][i] = p1;
            q[2][i] = p2;
          }
#pragma unroll
          for (int s = 0; s < split_planes(TERMS); ++s) tile[(s * 2 + h) * PLANE + e] = q[s];
        }
      }
    }
    const int any = skip_zero ? __syncthreads_or(nonzero) : (__syncthreads(), 1);
    if (ci0 + CONV_SCK < cin) {  // next stage's loads fly during the MFMAs
      int nxt = ci0 + CONV_SCK;
      asm volatile("" : "+s"(nxt));  // opaque: unpredicated loads would otherwise be hoisted above the staging phase
      stage_load(nxt);
    }
    if (!any) continue;

    const u32x4 *wchunk = (const u32x4 *)wt + (((size_t)(ci0 / CONV_SCK) * 3) * 2 + khalf) * cout_pad + co0 + l31;
    const size_t wsplit_stride = (size_t)2 * cout_pad, wtap_stride = (size_t)nchunk * 3 * 2 * cout_pad;
    split_taps<NT, HH, HW, PLANE, TERMS>(acc, tile, wchunk, wsplit_stride, wtap_stride, nbase, khalf);
  }
  }  // !PRE
  if constexpr (TERMS == SPLIT_F16X3) {  // 1 / (S_x S_w): a power of two stored behind the pack
    const float oscale = ((const float *)((const char *)wt + conv_split_trailer_bytes(nchunk, cout_pad)))[1];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] *= oscale;
  }

  float *outb = out + (size_t)b * cout * R3;
  // boundary-class constants of a second convolution: the workgroup's [27][32 MT] slice of K[b] goes through LDS (the
  // operand tile is free now) -- one cooperative fetch instead of a dependent global load per (row, N-tile)
  constexpr int NCW = 32 * MT;
  float *kl = (float *)tile;
  if (out_class) {
    __syncthreads();  // every wave is done with the last stage's fragments
    const float *kb = out_class + (size_t)b * 27 * cout;
    const int cob = co0 - 32 * wm;
    for (int e = tid; e < 27 * NCW; e += 256) {
      const int c = e % NCW, co = cob + c;
      kl[e] = co < cout ? kb[(e / NCW) * cout + co] : 0.0f;
    }
    __syncthreads();
  }
  int vox[NT], cls[NT];
#pragma unroll
  for (int s = 0; s < NT; ++s) {
    const int t = NT * wn + s;
    constexpr int HB = G::TH / G::NH;
    const int td = (t / HB) * G::ND, th = (t % HB) * G::NH;
    const int jw = lane_w<G::TW>(l31), jr = l31 / G::TW;
    const int d = d0 + td + jr / G::NH, h = h0 + th + jr % G::NH, w = w0 + jw;
    vox[s] = (d * R + h) * R + w;
    const int cd = d == 0 ? 0 : (d == R - 1 ? 2 : 1), ch = h == 0 ? 0 : (h == R - 1 ? 2 : 1),
              cw = w == 0 ? 0 : (w == R - 1 ? 2 : 1);
    cls[s] = (cd * 3 + ch) * 3 + cw;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float vv[NT][4];  // voxel-major stores: the four consecutive channels of register group g, per N-tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * g + i;
      const int co = co0 + i + 8 * g + 4 * khalf;
      const bool cok = co < cout;
      const float bv = bvr[r];
      float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int s = 0; s < NT; ++s) {
        if (!nact[s]) continue;
        float v = acc[s][r] + bv;
        if (out_class && cok) v += kl[cls[s] * NCW + 32 * wm + i + 8 * g + 4 * khalf];
        if (CL) vv[s][i] = v;
        else if (cok) outb[(size_t)co * R3 + vox[s]] = v;
        s1 += v;
        s2 += v * v;
      }
      if (stats_part) {
        // the brick's four statistics slots: wave column wn fills slot wn for its channels; with two wave rows
        // only two columns exist and slots 2, 3 are zeroed
        s1 = halfwave_sum_to_last(s1);
        s2 = halfwave_sum_to_last(s2);
        if (l31 == 31 && cok) {
          float *p = stats_part + ((((size_t)b * NBRICK + brick) * 4 + wn) * cout + co) * 2;
          p[0] = s1;
          p[1] = s2;
          if (WN < 4) {
            float *z = stats_part + ((((size_t)b * NBRICK + brick) * 4 + WN + wn) * cout + co) * 2;
            z[0] = 0.0f;
            z[1] = 0.0f;
          }
        }
      }
    }
    if (CL) {
      const int cq = co0 + 8 * g + 4 * khalf;
#pragma unroll
      for (int s = 0; s < NT; ++s) {
        if (!nact[s]) continue;
        float *q = outb + (size_t)vox[s] * cout + cq;
        if (cq + 3 < cout && (cout & 3) == 0) *(f32x4 *)q = f32x4{vv[s][0], vv[s][1], vv[s][2], vv[s][3]};
        else
          for (int i = 0; i < 4; ++i)
            if (cq + i < cout) q[i] = vv[s][i];
      }
    }
  }
#ifdef CONV_TIMELINE
  if constexpr (PRE) {
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if (tid == 0 && conv_tl_buf)  // 15: epilogue stores issued and acknowledged
      conv_tl_buf[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 16 + 15] = __builtin_readcyclecounter();
  }
#endif
}

// weights [cout][cin][3][3][3] -> packed [27][cin_pad/8][2][cout_pad][4] (zero padded):
// element (tap, chunk, khalf, co, kk) = W[co][chunk*8 + 2*kk + khalf][tap], so that the four k-pair
// values one lane needs for a tap are one aligned 16-byte load and lanes 0..31 read 512 contiguous bytes
static __global__ void conv3d_pack_kernel(int cout, int cin, int nchunk, int cout_pad, const float *__restrict__ w,
                                   float *__restrict__ wt) {
  const size_t total = (size_t)27 * nchunk * 8 * cout_pad;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int kk = (int)(e & 3);
    const int co = (int)((e >> 2) % cout_pad);
    size_t q = (e >> 2) / cout_pad;
    const int kh = (int)(q & 1);
    q >>= 1;
    const int chunk = (int)(q % nchunk), tap = (int)(q / nchunk);
    const int ci = chunk * 8 + 2 * kk + kh;
    wt[e] = (co < cout && ci < cin) ? w[((size_t)co * cin + ci) * 27 + tap] : 0.0f;
  }
}

#if CONV_TU == 0
extern "C" int p2pb_conv3d_k3_pack_weights(int cout, int cin, const float *w, float *wt_packed, void *stream) {
  if (cout <= 0 || cin <= 0) return P2PB_EINVAL;
  const int nchunk = (cin + CONV_CK - 1) / CONV_CK, cout_pad = (cout + 63) / 64 * 64;
  const size_t total = (size_t)27 * nchunk * 8 * cout_pad;
  hipLaunchKernelGGL(conv3d_pack_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)),
                     dim3(256), 0, (hipStream_t)stream, cout, cin, nchunk, cout_pad, w, wt_packed);
  return p2pb_launch_status();
}

extern "C" size_t p2pb_conv3d_k3_packed_floats(int cout, int cin) {
  const int cin_pad = (cin + CONV_CK - 1) / CONV_CK * CONV_CK, cout_pad = (cout + 63) / 64 * 64;
  return (size_t)27 * cin_pad * cout_pad;
}

extern "C" size_t p2pb_conv3d_k3_split_packed_bytes(int cout, int cin) {
  const int nchunk = (cin + CONV_SCK - 1) / CONV_SCK, cout_pad = (cout + 63) / 64 * 64;
  return (size_t)27 * nchunk * 3 * 2 * cout_pad * 8 * sizeof(unsigned short) + 16;  // + trailer (fp16 mode's scales)
}

static int conv_pack_split(int cout, int cin, const float *w, void *wt_split, bool adjoint, void *stream,
                           const unsigned *amax = nullptr) {
  if (cout <= 0 || cin <= 0) return P2PB_EINVAL;
  const int nchunk = (cin + CONV_SCK - 1) / CONV_SCK, cout_pad = (cout + 63) / 64 * 64;
  const size_t total = (size_t)27 * nchunk * 2 * cout_pad * 8;
  // the pack is made for the arithmetic selected NOW (p2pb_set_split_terms); callers re-pack after a switch to / from 16
  float *trailer = (float *)((char *)wt_split + conv_split_trailer_bytes(nchunk, cout_pad));
  static const long x2w = p2pb_experiment_long("x2w", 0);
  const int mode = p2pb_g_split_terms;
  if (mode == SPLIT_F16X3 && !amax) {
    const int rc = p2pb_zero_async(trailer, 16, (hipStream_t)stream);
    if (rc) return rc;
    hipLaunchKernelGGL(absmax_bits_kernel, dim3(absmax_blocks((size_t)cout * cin * 27)), dim3(256), 0, (hipStream_t)stream, w, (size_t)cout * cin * 27,
                       (unsigned *)trailer);
  }
  hipLaunchKernelGGL(conv3d_pack_split_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)),
                     dim3(256), 0, (hipStream_t)stream, cout, cin, nchunk, cout_pad, w, (unsigned short *)wt_split,
                     mode | ((x2w && mode == SPLIT_F16X3 && !adjoint) ? SPLIT_X2W_FLAG : 0), trailer, adjoint ? 27L : (long)cin * 27, adjoint ? (long)cout * 27 : 27L, adjoint ? 1 : 0,
                     mode == SPLIT_F16X3 ? amax : nullptr);
  return p2pb_launch_status();
}
extern "C" int p2pb_conv3d_k3_pack_weights_split(int cout, int cin, const float *w, void *wt_split, void *stream) {
  return conv_pack_split(cout, cin, w, wt_split, false, stream);
}
extern "C" int p2pb_conv3d_k3_pack_weights_split_adjoint(int cout, int cin, const float *w_forward, void *wt_split, void *stream) {
  return conv_pack_split(cout, cin, w_forward, wt_split, true, stream);
}
extern "C" int p2pb_conv3d_k3_pack_weights_split_amax(int cout, int cin, const float *w, void *wt_split, const unsigned *amax_bits,
                                                      void *stream) {
  return amax_bits ? conv_pack_split(cout, cin, w, wt_split, false, stream, amax_bits) : P2PB_EINVAL;
}
#endif

static int conv_bricks(int r) { return r == 32 ? 128 : r == 16 ? 16 : r == 8 ? 2 : 1; }  // both geometries

#if CONV_TU == 0
extern "C" size_t p2pb_conv3d_k3_stats_floats(int b, int cout, int r) {
  return (size_t)b * conv_bricks(r) * 4 * cout * 2;
}
#endif

// far-field constants of a folded operand transform: a[b,c] = xf(base[c]) with the SAME device function the
// staging code uses (bit-identical), i.e. the value of swish(affine(conv0 output)) where conv0 saw only zeros
static __global__ void far_value_kernel(int c, const float *__restrict__ base, const float *__restrict__ scale,
                                 const float *__restrict__ shift, int swish, float *__restrict__ a) {
  const int b = blockIdx.y, ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= c) return;
  a[(size_t)b * c + ch] = xf_apply(base[ch], scale[(size_t)b * c + ch], shift[(size_t)b * c + ch], swish);
}

// T[b, tap, co] = sum_ci W[tap][ci][co] * a[b,ci]  (one thread per output channel, weights read coalesced)
static __global__ __launch_bounds__(256) void tap_sum_kernel(int cin, int cout, int nchunk, int cout_pad,
                                                      const float *__restrict__ wt, const float *__restrict__ a,
                                                      float *__restrict__ tsum) {
  const int tap = blockIdx.y, b = blockIdx.z;
  const int co = blockIdx.x * 256 + threadIdx.x;
  if (co >= cout) return;
  float acc = 0.0f;
  for (int ci = 0; ci < cin; ++ci) {
    const size_t idx = ((((size_t)tap * nchunk + (ci >> 3)) * 2 + (ci & 1)) * cout_pad + co) * 4 + ((ci & 7) >> 1);
    acc = __fmaf_rn(wt[idx], a[(size_t)b * cin + ci], acc);
  }
  tsum[((size_t)b * 27 + tap) * cout + co] = acc;
}

// K[b, class, co] = bias[co] + sum over the taps that stay inside the grid for that boundary class of T[b,tap,co]
// (the convolution of the constant field a with zero padding)
static __global__ __launch_bounds__(256) void class_bias_kernel(int cout, const float *__restrict__ tsum,
                                                         const float *__restrict__ bias, float *__restrict__ k_out) {
  const int cls = blockIdx.y, b = blockIdx.z;
  const int co = blockIdx.x * 256 + threadIdx.x;
  if (co >= cout) return;
  const int cd = cls / 9, ch = (cls / 3) % 3, cw = cls % 3;
  float acc = 0.0f;
  for (int tap = 0; tap < 27; ++tap) {
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    // class 0 = low face: the tap reading index -1 is outside; class 2 = high face: the tap reading R is outside
    if ((cd == 0 && kd == 0) || (cd == 2 && kd == 2) || (ch == 0 && kh == 0) || (ch == 2 && kh == 2) ||
        (cw == 0 && kw == 0) || (cw == 2 && kw == 2))
      continue;
    acc += tsum[((size_t)b * 27 + tap) * cout + co];
  }
  k_out[((size_t)b * 27 + cls) * cout + co] = acc + bias[co];
}

// The three steps above in ONE launch (round 4: 15 -> 5 launches per network evaluation): workgroup (FF_CO output channels, sample):
// every thread evaluates a[ci] into LDS (the first channel block also stores it), thread (c, tap) sums ITS tap for channel c over the
// input channels, then the 27 class sums from the LDS table. Same operations in the same order per output as far_value / tap_sum /
// class_bias: bit-identical.
// Round 5 (profiles/r05_overlap.txt: 52-79 us per launch on 32-64 workgroups, 290 us of every chain-evaluation): 32 channels x 27
// taps per workgroup instead of 64 channels x 16 tap groups of two (twice the workgroups, half the serial length per thread), and
// the weights as the 16-byte groups the pack holds -- [tap][ci / 8][ci & 1][co][(ci & 7) >> 1]: one float4 = input channels
// h, h + 2, h + 4, h + 6 of a chunk -- so a chunk of 8 input channels is two coalesced 16-byte loads (512 contiguous bytes per 32
// lanes) instead of eight 4-byte loads at a 16-byte stride; the fused multiply-adds stay in ascending input-channel order.
// part != NULL: the GroupNorm(+AdaGN) between the two convolutions is folded HERE as well -- `scale` / `shift` are then OUTPUTS
// (fin.scale / fin.shift, f32[b, cin], written by the first channel block for the kernels that stage the operand) and the
// gn_affine launch between the first convolution and this kernel is gone (round 5; gn_finish_sample: the same bits).
constexpr int FF_CO = 32;
static __global__ __launch_bounds__(1024) void far_field_kernel(int cin, int cout, int nchunk, int cout_pad,
                                                        const float *__restrict__ base, const float *__restrict__ scale,
                                                        const float *__restrict__ shift, int swish,
                                                        const float *__restrict__ wt, const float *__restrict__ bias,
                                                        float *__restrict__ a_out, float *__restrict__ k_out,
                                                        const float *__restrict__ part, int nslots, GnFinish fin) {
  extern __shared__ double ff_sm_d[];  // [scale[cin] | shift[cin]] (folded form) | union { 4 x 1024 doubles, a[cin] | T[27][FF_CO] }
  float *ff_sm = (float *)ff_sm_d;
  const int b = blockIdx.y, t = threadIdx.x, c = t & (FF_CO - 1), tap = t / FF_CO;
  const int co = blockIdx.x * FF_CO + c;
  float *a, *T;
  if (part != nullptr) {
    float *tsc = ff_sm, *tsh = ff_sm + cin;
    double *gl = ff_sm_d + (2 * cin + 1) / 2;
    GnFinish f = fin;
    if (blockIdx.x != 0) f.scale = f.shift = nullptr;  // (every channel block computes the values, one writes them)
    gn_finish_sample<4>(cin, nslots, part, f, b, gl, tsc, tsh);
    a = (float *)gl, T = a + nchunk * 8;
    for (int ch = t; ch < cin; ch += 1024) {
      const float v = xf_apply(base[ch], tsc[ch], tsh[ch], swish);
      a[ch] = v;
      if (blockIdx.x == 0) a_out[(size_t)b * cin + ch] = v;
    }
  } else {
    a = ff_sm, T = ff_sm + nchunk * 8;
    for (int ch = t; ch < cin; ch += 1024) {
      const float v = xf_apply(base[ch], scale[(size_t)b * cin + ch], shift[(size_t)b * cin + ch], swish);
      a[ch] = v;
      if (blockIdx.x == 0) a_out[(size_t)b * cin + ch] = v;
    }
  }
  __syncthreads();
  if (tap < 27) {
    float acc = 0.0f;
    if (co < cout) {
      const float4 *w4 = (const float4 *)wt + ((size_t)tap * nchunk * 2) * cout_pad + co;
#pragma unroll 4
      for (int k8 = 0; k8 < nchunk; ++k8) {
        const float4 h0 = w4[(size_t)(2 * k8) * cout_pad], h1 = w4[(size_t)(2 * k8 + 1) * cout_pad];
        const float *av = a + k8 * 8;
        const int left = cin - k8 * 8;  // (a ragged last chunk: the channels that exist, in order)
        if (left > 0) acc = __fmaf_rn(h0.x, av[0], acc);
        if (left > 1) acc = __fmaf_rn(h1.x, av[1], acc);
        if (left > 2) acc = __fmaf_rn(h0.y, av[2], acc);
        if (left > 3) acc = __fmaf_rn(h1.y, av[3], acc);
        if (left > 4) acc = __fmaf_rn(h0.z, av[4], acc);
        if (left > 5) acc = __fmaf_rn(h1.z, av[5], acc);
        if (left > 6) acc = __fmaf_rn(h0.w, av[6], acc);
        if (left > 7) acc = __fmaf_rn(h1.w, av[7], acc);
      }
    }
    T[tap * FF_CO + c] = acc;
  }
  __syncthreads();
  if (co >= cout || tap >= 27) return;
  {
    const int cls = tap;
    const int cd = cls / 9, ch = (cls / 3) % 3, cw = cls % 3;
    float acc = 0.0f;
    for (int tp = 0; tp < 27; ++tp) {
      const int kd = tp / 9, kh = (tp / 3) % 3, kw = tp % 3;
      if ((cd == 0 && kd == 0) || (cd == 2 && kd == 2) || (ch == 0 && kh == 0) || (ch == 2 && kh == 2) ||
          (cw == 0 && kw == 0) || (cw == 2 && kw == 2))
        continue;
      acc += T[tp * FF_CO + c];
    }
    k_out[((size_t)b * 27 + cls) * cout + co] = acc + bias[co];
  }
}

// a f32[b,cin] = far-field operand constants, k_out f32[b,27,cout] = per-boundary-class output constants,
// tap_ws f32[b,27,cout] scratch
#if CONV_TU == 0
static bool gn_shape_ok(int c, int groups, const float *style, int style_stride) {
  return groups > 0 && c % groups == 0 && c / groups <= 256 && !(style && style_stride < 2 * c);
}
static int far_field_launch(int b, int cin, int cout, const float *prev_bias, const float *in_scale, const float *in_shift,
                            int in_swish, const float *wt_packed, const float *bias, float *a, float *k_out, float *tap_ws,
                            const float *part, int nslots, const GnFinish &fin, hipStream_t s) {
  const int nchunk = (cin + CONV_CK - 1) / CONV_CK, cout_pad = (cout + 63) / 64 * 64;
  const size_t body = (size_t)(nchunk * 8 + 27 * FF_CO) * 4;
  const size_t lds = part ? (size_t)((2 * cin + 1) / 2) * 8 + (body > 32768 ? body : 32768) : body;
  if (lds <= 48 * 1024) {  // (tap_ws unused in this form)
    hipLaunchKernelGGL(far_field_kernel, dim3(cdiv(cout, FF_CO), b), dim3(1024), lds, s, cin, cout, nchunk, cout_pad, prev_bias,
                       in_scale, in_shift, in_swish, wt_packed, bias, a, k_out, part, nslots, fin);
    return p2pb_launch_status();
  }
  if (part) {  // (very wide layers: the norm as its own launch)
    const int e = p2pb_gn_affine_launch(b, cin, nslots, part, fin, s);
    if (e != 0) return e;
    in_scale = fin.scale, in_shift = fin.shift;
  }
  hipLaunchKernelGGL(far_value_kernel, dim3(cdiv(cin, 256), b), dim3(256), 0, s, cin, prev_bias, in_scale, in_shift,
                     in_swish, a);
  hipLaunchKernelGGL(tap_sum_kernel, dim3(cdiv(cout, 256), 27, b), dim3(256), 0, s, cin, cout, nchunk, cout_pad,
                     wt_packed, a, tap_ws);
  hipLaunchKernelGGL(class_bias_kernel, dim3(cdiv(cout, 256), 27, b), dim3(256), 0, s, cout, tap_ws, bias, k_out);
  return p2pb_launch_status();
}
extern "C" int p2pb_conv3d_k3_far_field(int b, int cin, int cout, const float *prev_bias, const float *in_scale,
                                        const float *in_shift, int in_swish, const float *wt_packed,
                                        const float *bias, float *a, float *k_out, float *tap_ws, void *stream) {
  if (b <= 0 || cin <= 0 || cout <= 0 || !in_scale || !in_shift) return P2PB_EINVAL;
  return far_field_launch(b, cin, cout, prev_bias, in_scale, in_shift, in_swish, wt_packed, bias, a, k_out, tap_ws, nullptr, 0,
                          GnFinish(), (hipStream_t)stream);
}
// the same with the GroupNorm(+AdaGN) of the operand folded in: part f32[b, nslots, cin, 2] = the first convolution's statistics
// partials; scale / shift f32[b, cin] are OUTPUTS (what p2pb_gn_affine_params would have written, same bits)
extern "C" int p2pb_conv3d_k3_far_field_gn(int b, int cin, int cout, const float *prev_bias, const float *part, int nslots,
                                           double count_per_channel, int groups, const float *gamma, const float *beta,
                                           const float *style, int style_stride, float eps, int in_swish,
                                           const float *wt_packed, const float *bias, float *scale, float *shift, float *a,
                                           float *k_out, float *tap_ws, void *stream) {
  if (b <= 0 || cin <= 0 || cout <= 0 || !part || nslots <= 0 || !scale || !shift || !gn_shape_ok(cin, groups, style, style_stride))
    return P2PB_EINVAL;
  GnFinish f = {};
  f.gamma = gamma, f.beta = beta, f.style = style, f.scale = scale, f.shift = shift, f.chmean = nullptr;
  f.count_per_channel = count_per_channel, f.style_stride = style_stride, f.groups = groups, f.eps = eps;
  return far_field_launch(b, cin, cout, prev_bias, nullptr, nullptr, in_swish, wt_packed, bias, a, k_out, tap_ws, part, nslots, f,
                          (hipStream_t)stream);
}
#endif

template <int R, bool COMPACT, int MT>
static int conv_launch(int b, int cin, int cout, const float *in, const float *wt, const float *bias,
                       const float *out_class, const float *in_scale, const float *in_shift, int in_swish,
                       const float *in_sub, int skip_zero, const int *brick_list, const int *brick_count, float *out,
                       float *stats_part, bool cl, hipStream_t s) {
  const int nchunk = (cin + CONV_CK - 1) / CONV_CK, cout_pad = (cout + 63) / 64 * 64;
  dim3 grid(conv_bricks(R), (cout + 32 * MT - 1) / (32 * MT), b);
  if (brick_list) grid = dim3(conv_bricks(R) * b, (cout + 32 * MT - 1) / (32 * MT), 1);
#define LAUNCH(XF, CL)                                                                                               \
  hipLaunchKernelGGL((conv3d_k3_kernel<R, COMPACT, MT, XF, CL>), grid, dim3(256), 0, s, cin, cout, nchunk, cout_pad, \
                     in, wt, bias, out_class, in_scale, in_shift, in_swish, in_sub, skip_zero, brick_list,            \
                     brick_count, out, stats_part)
  if (in_scale != nullptr) {
    if (cl) LAUNCH(true, true);
    else LAUNCH(true, false);
  } else {
    if (cl) LAUNCH(false, true);
    else LAUNCH(false, false);
  }
#undef LAUNCH
  return p2pb_launch_status();
}

// the bf16x6 / bf16x3 instantiations live in the -DCONV_TU=6 / -DCONV_TU=3 objects
int conv3d_tu6_split(int r, int mt, int b, int cin, int cout, const float *in, const void *wt, const float *bias,
                     const float *out_class, const float *in_scale, const float *in_shift, int in_swish,
                     const float *in_sub, int skip_zero, const int *brick_list, const int *brick_count, float *out,
                     float *stats_part, bool cl, hipStream_t s);
int conv3d_tu3_split(int r, int mt, int b, int cin, int cout, const float *in, const void *wt, const float *bias,
                     const float *out_class, const float *in_scale, const float *in_shift, int in_swish,
                     const float *in_sub, int skip_zero, const int *brick_list, const int *brick_count, float *out,
                     float *stats_part, bool cl, hipStream_t s);
template <int R, int MT>
static int conv_launch_split(int b, int cin, int cout, const float *in, const void *wt, const float *bias,
                             const float *out_class, const float *in_scale, const float *in_shift, int in_swish,
                             const float *in_sub, int skip_zero, const int *brick_list, const int *brick_count,
                             float *out, float *stats_part, bool cl, hipStream_t s, bool pre = false) {
  const int nchunk = (cin + CONV_SCK - 1) / CONV_SCK, cout_pad = (cout + 63) / 64 * 64;
  dim3 grid(conv_bricks(R), (cout + 32 * MT - 1) / (32 * MT), b);
  if (brick_list) grid = dim3(conv_bricks(R) * b, (cout + 32 * MT - 1) / (32 * MT), 1);
  const unsigned short *w = (const unsigned short *)wt;
#if CONV_TU == 0
  if (p2pb_g_split_terms == SPLIT_BF16X6) {
    if (pre) return P2PB_EINVAL;  // (the S format is the f16x3 arithmetic's)
    return conv3d_tu6_split(R, MT, b, cin, cout, in, wt, bias, out_class, in_scale, in_shift, in_swish, in_sub, skip_zero,
                            brick_list, brick_count, out, stats_part, cl, s);
  }
  if (p2pb_g_split_terms == SPLIT_BF16X3) {  // the training data gradient's form only: plain operand, channel-major
    if (pre || in_scale || in_sub || cl || brick_list) return P2PB_EINVAL;
    return conv3d_tu3_split(R, MT, b, cin, cout, in, wt, bias, out_class, in_scale, in_shift, in_swish, in_sub, skip_zero,
                            brick_list, brick_count, out, stats_part, cl, s);
  }
#endif
#define LAUNCHT(XF, CL, TM)                                                                                           \
  hipLaunchKernelGGL((conv3d_k3_split_kernel<R, true, MT, XF, CL, TM>), grid, dim3(256), 0, s, cin, cout, nchunk,        \
                     cout_pad, in, w, bias, out_class, in_scale, in_shift, in_swish, in_sub, skip_zero, brick_list,     \
                     brick_count, out, stats_part)
#define LAUNCH(XF, CL) LAUNCHT(XF, CL, CONV_TERMS)
  if (pre) {  // `in` is the pre-split operand grid (S format): f16x3, voxel-major, transform already applied
#if CONV_TU == 0
    if constexpr (R >= 8) {
      if (!cl || in_scale || in_sub) return P2PB_EINVAL;
      hipLaunchKernelGGL((conv3d_k3_split_kernel<R, true, MT, false, true, SPLIT_F16X3, true>), grid, dim3(256), 0, s,
                         cin, cout, nchunk, cout_pad, in, w, bias, out_class, in_scale, in_shift, in_swish, in_sub, 0,
                         brick_list, brick_count, out, stats_part);
      return p2pb_launch_status();
    }
#endif
    return P2PB_EINVAL;
  }
#if CONV_TU == 3
  if (in_scale != nullptr || cl) return P2PB_EINVAL;
  LAUNCH(false, false);
#else
  if (in_scale != nullptr) {
    if (cl) LAUNCH(true, true);
    else LAUNCH(true, false);
  } else {
    if (cl) LAUNCH(false, true);
    else LAUNCH(false, false);
  }
#endif
#undef LAUNCH
#undef LAUNCHT
  return p2pb_launch_status();
}
#if CONV_TU != 0
#if CONV_TU == 6
int conv3d_tu6_split(int r, int mt, int b, int cin, int cout, const float *in, const void *wt, const float *bias,
#else
int conv3d_tu3_split(int r, int mt, int b, int cin, int cout, const float *in, const void *wt, const float *bias,
#endif
                     const float *out_class, const float *in_scale, const float *in_shift, int in_swish,
                     const float *in_sub, int skip_zero, const int *brick_list, const int *brick_count, float *out,
                     float *stats_part, bool cl, hipStream_t s) {
#define GOB(RR)                                                                                                        \
  return mt == 2 ? conv_launch_split<RR, 2>(b, cin, cout, in, wt, bias, out_class, in_scale, in_shift, in_swish, in_sub, \
                                            skip_zero, brick_list, brick_count, out, stats_part, cl, s)      \
                 : conv_launch_split<RR, 1>(b, cin, cout, in, wt, bias, out_class, in_scale, in_shift, in_swish, in_sub, \
                                            skip_zero, brick_list, brick_count, out, stats_part, cl, s)
  switch (r) {
    case 32: GOB(32);
    case 16: GOB(16);
    case 8: GOB(8);
    case 4: GOB(4);
    default: return P2PB_EINVAL;
  }
#undef GOB
}
#endif

// ------------------------------------------------------------------------------------------------
// Brick activity from the voxel occupancy (cnt of avg_voxelize), compact 4x8x8 bricks:
//   first conv : a brick has non-zero input in its halo  <=> an occupied voxel within brick +- 1
//   second conv: its operand differs from the far-field constant only inside dil(occupied, 1), so a brick
//                has work <=> an occupied voxel within brick +- 2
// Output: four compacted lists of (sample*NBRICK + brick): active/inactive for each conv, and their counts.
// ------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void brick_flags_kernel(const int *__restrict__ cnt, unsigned char *__restrict__ flags) {
  constexpr int TD = 4, TH = 8, TW = 8, BH = R / TH, BW = R / TW, NBRICK = (R / TD) * BH * BW;
  __shared__ int f1, f2;
  const int b = blockIdx.y, bk = blockIdx.x;
  const int d0 = (bk / (BH * BW)) * TD, h0 = ((bk / BW) % BH) * TH, w0 = (bk % BW) * TW;
  if (threadIdx.x == 0) f1 = f2 = 0;
  __syncthreads();
  constexpr int ED = TD + 4, EH = TH + 4, EW = TW + 4;
  int a1 = 0, a2 = 0;
  for (int e = threadIdx.x; e < ED * EH * EW; e += 256) {
    const int dz = e / (EH * EW), hy = (e / EW) % EH, wx = e % EW;
    const int d = d0 - 2 + dz, h = h0 - 2 + hy, w = w0 - 2 + wx;
    if ((unsigned)d < (unsigned)R && (unsigned)h < (unsigned)R && (unsigned)w < (unsigned)R) {
      if (cnt[(size_t)b * R * R * R + (d * R + h) * R + w] > 0) {
        a2 = 1;
        if (dz >= 1 && dz <= TD + 2 && hy >= 1 && hy <= TH + 2 && wx >= 1 && wx <= TW + 2) a1 = 1;
      }
    }
  }
  if (a1) f1 = 1;  // benign race: every writer stores 1
  if (a2) f2 = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    flags[((size_t)b * NBRICK + bk) * 2 + 0] = (unsigned char)f1;
    flags[((size_t)b * NBRICK + bk) * 2 + 1] = (unsigned char)f2;
  }
}

// single workgroup: compaction of up to 1024*PER entries into active/inactive lists for both convolutions
static __global__ __launch_bounds__(1024) void brick_compact_kernel(int total, const unsigned char *__restrict__ flags,
                                                            int *__restrict__ lists, int *__restrict__ counts) {
  __shared__ int wsum[16];
  const int t = threadIdx.x;
  const int per = (total + 1023) / 1024;
  const int beg = t * per, end = min(beg + per, total);
  for (int which = 0; which < 2; ++which) {
    int k = 0;
    for (int e = beg; e < end; ++e) k += flags[(size_t)e * 2 + which];
    // inclusive wave scan + cross-wave offsets
    int inc = k;
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(inc, d);
      if ((t & 63) >= d) inc += y;
    }
    __syncthreads();
    if ((t & 63) == 63) wsum[t >> 6] = inc;
    __syncthreads();
    int base = 0, all = 0;
    for (int w = 0; w < 16; ++w) {
      if (w < (t >> 6)) base += wsum[w];
      all += wsum[w];
    }
    int apos = base + inc - k;      // active entries before this thread's range
    int ipos = beg - apos;          // inactive entries before it
    int *act = lists + (size_t)(2 * which) * total, *ina = lists + (size_t)(2 * which + 1) * total;
    for (int e = beg; e < end; ++e) {
      if (flags[(size_t)e * 2 + which]) act[apos++] = e;
      else ina[ipos++] = e;
    }
    if (t == 0) {
      counts[2 * which] = all;
      counts[2 * which + 1] = total - all;
    }
  }
}

// lists i32[4][b*NBRICK] = {active conv0, inactive conv0, active conv1, inactive conv1}, counts i32[4];
// flags_ws: b*NBRICK*2 bytes of scratch. r in {16, 32}.
#if CONV_TU == 0
extern "C" int p2pb_conv3d_brick_lists(int b, int r, const int *cnt, unsigned char *flags_ws, int *lists, int *counts,
                                       void *stream) {
  if (b <= 0 || (r != 16 && r != 32)) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nb = conv_bricks(r);
  if (r == 32) hipLaunchKernelGGL(brick_flags_kernel<32>, dim3(nb, b), dim3(256), 0, s, cnt, flags_ws);
  else hipLaunchKernelGGL(brick_flags_kernel<16>, dim3(nb, b), dim3(256), 0, s, cnt, flags_ws);
  hipLaunchKernelGGL(brick_compact_kernel, dim3(1), dim3(1024), 0, s, nb * b, flags_ws, lists, counts);
  return p2pb_launch_status();
}
#endif

// inactive bricks: the convolution's output there is a known constant per channel (bias, or the
// boundary-class constant K): write it and the brick's exact {sum, sum of squares} partials
template <int R, bool CL>
__global__ __launch_bounds__(256) void conv3d_fill_kernel(int cout, const float *__restrict__ bias,
                                                          const float *__restrict__ out_class,
                                                          const int *__restrict__ brick_list,
                                                          const int *__restrict__ brick_count, float *__restrict__ out,
                                                          float *__restrict__ stats_part) {
  constexpr int TD = 4, TH = 8, TW = 8, BH = R / TH, BW = R / TW, NBRICK = (R / TD) * BH * BW, R3 = R * R * R;
  __shared__ int ncls[27];
  if ((int)blockIdx.x >= *brick_count) return;
  const int entry = brick_list[blockIdx.x];
  const int b = entry / NBRICK, bk = entry % NBRICK;
  const int d0 = (bk / (BH * BW)) * TD, h0 = ((bk / BW) % BH) * TH, w0 = (bk % BW) * TW;
  const int t = threadIdx.x;
  const int d = d0 + t / (TH * TW), h = h0 + (t / TW) % TH, w = w0 + t % TW;
  const int cd = d == 0 ? 0 : (d == R - 1 ? 2 : 1), ch = h == 0 ? 0 : (h == R - 1 ? 2 : 1),
            cw = w == 0 ? 0 : (w == R - 1 ? 2 : 1);
  const int cls = (cd * 3 + ch) * 3 + cw;
  if (t < 27) ncls[t] = 0;
  __syncthreads();
  atomicAdd(&ncls[cls], 1);
  __syncthreads();
  const float *kb = out_class ? out_class + (size_t)b * 27 * cout : nullptr;
  if (!out) {  // statistics only (p2pb_conv3d_k3_forward_sparse flags bit 5: nobody reads the inactive bricks' outputs)
  } else if (CL) {  // voxel-major: the brick's voxels x channels, channels fastest (coalesced)
    __shared__ unsigned char vcls[256];
    vcls[t] = (unsigned char)cls;
    __syncthreads();
    float *ob = out + (size_t)b * cout * R3;
    if ((cout & 3) == 0) {  // 16 bytes per thread, (voxel, channel quad) advanced without a division per element
      const int c4n = cout >> 2, dq = 256 / c4n, dr = 256 % c4n;
      int vl = t / c4n, c4 = t % c4n;
      for (; vl < 256; vl += dq) {
        const int dd = d0 + vl / (TH * TW), hh = h0 + (vl / TW) % TH, ww = w0 + vl % TW;
        const float *src = kb ? kb + vcls[vl] * cout : bias;
        *(f32x4 *)(ob + (size_t)((dd * R + hh) * R + ww) * cout + 4 * c4) = *(const f32x4 *)(src + 4 * c4);
        if (dr) {
          c4 += dr;
          if (c4 >= c4n) {
            c4 -= c4n;
            ++vl;
          }
        }
      }
    } else {
      for (int e = t; e < 256 * cout; e += 256) {
        const int vl = e / cout, co = e - vl * cout;
        const int dd = d0 + vl / (TH * TW), hh = h0 + (vl / TW) % TH, ww = w0 + vl % TW;
        ob[(size_t)((dd * R + hh) * R + ww) * cout + co] = kb ? kb[vcls[vl] * cout + co] : bias[co];
      }
    }
  } else {
    float *ob = out + (size_t)b * cout * R3 + (d * R + h) * R + w;
    for (int co = 0; co < cout; ++co) ob[(size_t)co * R3] = kb ? kb[cls * cout + co] : bias[co];
  }
  if (stats_part) {
    for (int co = t; co < cout; co += 256) {
      float s1 = 0.0f, s2 = 0.0f;
      if (kb) {
        for (int c = 0; c < 27; ++c) {
          const float v = kb[c * cout + co], n = (float)ncls[c];
          s1 += n * v;
          s2 += n * v * v;
        }
      } else {
        const float v = bias[co];
        s1 = 256.0f * v;
        s2 = 256.0f * v * v;
      }
      float *p = stats_part + (((size_t)b * NBRICK + bk) * 4) * cout * 2;
      p[(size_t)co * 2] = s1;
      p[(size_t)co * 2 + 1] = s2;
#pragma unroll
      for (int wv = 1; wv < 4; ++wv) {
        p[((size_t)wv * cout + co) * 2] = 0.0f;
        p[((size_t)wv * cout + co) * 2 + 1] = 0.0f;
      }
    }
  }
}

// out[b,cout,r,r,r] = conv3d(xf(in[b,cin,r,r,r]), W) + bias, where xf(x) = x (in_scale == NULL) or
// swish?(x*in_scale[b,ci] + in_shift[b,ci]) - in_sub[b,ci]; out_class (optional, f32[b,27,cout]) replaces
// bias per boundary class; stats_part (optional) receives per-(b, slot, cout) {sum, sum of squares} of the
// output. flags: bit 0 = skip all-zero operand tiles (exact), bit 1 = compact 4x8x8 bricks, bit 2 = wt_packed is
// the split pack (p2pb_conv3d_k3_pack_weights_split) -> bf16x6 kernel. r in {4,8,16,32}.
#if CONV_TU == 0
extern "C" int p2pb_conv3d_k3_forward(int b, int cin, int cout, int r, const float *in, const float *wt_packed,
                                      const float *bias, const float *in_scale, const float *in_shift, int in_swish,
                                      float *out, float *stats_part, void *stream) {
  return p2pb_conv3d_k3_forward_ex(b, cin, cout, r, in, wt_packed, bias, nullptr, in_scale, in_shift, in_swish, nullptr,
                                   0, out, stats_part, stream);
}

extern "C" int p2pb_conv3d_k3_forward_ex(int b, int cin, int cout, int r, const float *in, const void *wt_packed,
                                         const float *bias, const float *out_class, const float *in_scale,
                                         const float *in_shift, int in_swish, const float *in_sub, int flags,
                                         float *out, float *stats_part, void *stream) {
  if (b <= 0 || cin <= 0 || cout <= 0) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int skip = flags & 1;
  const bool compact = (flags & 2) != 0;
  const bool cl = (flags & 8) != 0;  // voxel-major tensors in[b,r,r,r,cin], out[b,r,r,r,cout]
  const bool pre = (flags & 16) != 0;  // `in` is the pre-split operand grid (p2pb_conv3d_presplit / ..._cl_gather_split)
  if (pre && (flags & 12) != 12) return P2PB_EINVAL;
  // 64 output channels per workgroup unless that leaves fewer than 2 workgroups per CU (small grids)
  const bool wide = cout > 32 && (long)conv_bricks(r) * ((cout + 63) / 64) * b >= 512;
  if (flags & 4) {  // wt_packed is the split (3 x bf16) pack; always the compact tiling (same results, same slots)
    // 64 output channels per workgroup, unless that leaves under one workgroup per CU (small grids x small batches:
    // the 8^3 grids of a training batch of 8): 32 channels then double the workgroup count
    static const long wide_min = p2pb_experiment_long("conv_wide_min", 256);  // (A/B switch)
    const bool wide = cout > 32 && (long)conv_bricks(r) * ((cout + 63) / 64) * b >= wide_min;
#define GOS(RR)                                                                                                       \
  return wide ? conv_launch_split<RR, 2>(b, cin, cout, in, wt_packed, bias, out_class, in_scale, in_shift, in_swish,  \
                                         in_sub, skip, nullptr, nullptr, out, stats_part, cl, s, pre)         \
              : conv_launch_split<RR, 1>(b, cin, cout, in, wt_packed, bias, out_class, in_scale, in_shift, in_swish,  \
                                         in_sub, skip, nullptr, nullptr, out, stats_part, cl, s, pre)
    switch (r) {
      case 32: GOS(32);
      case 16: GOS(16);
      case 8: GOS(8);
      case 4: GOS(4);
      default: return P2PB_EINVAL;
    }
#undef GOS
  }
  const float *wt32 = (const float *)wt_packed;
#define GO(RR, CP)                                                                                                    \
  return wide ? conv_launch<RR, CP, 2>(b, cin, cout, in, wt32, bias, out_class, in_scale, in_shift, in_swish,         \
                                       in_sub, skip, nullptr, nullptr, out, stats_part, cl, s)                            \
              : conv_launch<RR, CP, 1>(b, cin, cout, in, wt32, bias, out_class, in_scale, in_shift, in_swish,         \
                                       in_sub, skip, nullptr, nullptr, out, stats_part, cl, s)
  switch (r) {
    case 32:
      if (compact) { GO(32, true); } else { GO(32, false); }
    case 16:
      if (compact) { GO(16, true); } else { GO(16, false); }
    case 8: GO(8, false);
    case 4: GO(4, false);
    default: return P2PB_EINVAL;
  }
#undef GO
}
#endif

// list-driven sparse form: MFMA workgroups only for the `active` (sample, brick) pairs, constants for the
// `inactive` ones (lists from p2pb_conv3d_brick_lists). Compact geometry; r in {16, 32}.
#if CONV_TU == 0
extern "C" int p2pb_conv3d_k3_forward_sparse(int b, int cin, int cout, int r, const float *in, const void *wt_packed,
                                             const float *bias, const float *out_class, const float *in_scale,
                                             const float *in_shift, int in_swish, const float *in_sub, int flags,
                                             const int *active_list, const int *active_count,
                                             const int *inactive_list, const int *inactive_count, float *out,
                                             float *stats_part, void *stream) {
  if (b <= 0 || cin <= 0 || cout <= 0 || (r != 16 && r != 32) || !active_list || !inactive_list) return P2PB_EINVAL;
  if ((flags & 32) && !stats_part) return P2PB_EINVAL;  // (bit 5: the inactive bricks' statistics only -- there must be statistics)
  hipStream_t s = (hipStream_t)stream;
  const int total = conv_bricks(r) * b;
  const bool cl = (flags & 8) != 0;
  const bool pre = (flags & 16) != 0;  // `in` is the pre-split operand grid
  if (pre && (flags & 12) != 12) return P2PB_EINVAL;
#define FILL(RR, CL)                                                                                          \
  hipLaunchKernelGGL((conv3d_fill_kernel<RR, CL>), dim3(total), dim3(256), 0, s, cout, bias, out_class, inactive_list, \
                     inactive_count, (flags & 32) ? (float *)nullptr : out, stats_part)
  if (r == 32) {
    if (cl) FILL(32, true);
    else FILL(32, false);
  } else {
    if (cl) FILL(16, true);
    else FILL(16, false);
  }
#undef FILL
  const bool wide = cout > 32;
  if (flags & 4) {
#define GOS(RR)                                                                                                       \
  return wide ? conv_launch_split<RR, 2>(b, cin, cout, in, wt_packed, bias, out_class, in_scale, in_shift, in_swish,  \
                                         in_sub, 1, active_list, active_count, out, stats_part, cl, s, pre)   \
              : conv_launch_split<RR, 1>(b, cin, cout, in, wt_packed, bias, out_class, in_scale, in_shift, in_swish,  \
                                         in_sub, 1, active_list, active_count, out, stats_part, cl, s, pre)
    if (r == 32) { GOS(32); }
    GOS(16);
#undef GOS
  }
  const float *wt32 = (const float *)wt_packed;
#define GO(RR)                                                                                                        \
  return wide ? conv_launch<RR, true, 2>(b, cin, cout, in, wt32, bias, out_class, in_scale, in_shift, in_swish,       \
                                         in_sub, 1, active_list, active_count, out, stats_part, cl, s)                    \
              : conv_launch<RR, true, 1>(b, cin, cout, in, wt32, bias, out_class, in_scale, in_shift, in_swish,       \
                                         in_sub, 1, active_list, active_count, out, stats_part, cl, s)
  if (r == 32) { GO(32); }
  GO(16);
#undef GO
}
#endif

// ================================================================================================
// Compact form of the split kernel: voxel-level sparsity inside the bricks.
//
// The first convolution of a PVConv is non-constant only on D1 = dilate(occupied, 1) (elsewhere every input in
// the 3x3x3 window is zero and the output is the bias), the second -- in its far-field form, operand x - a --
// only on D2 = dilate(D1, 1) (elsewhere the output is the boundary-class constant K). D1 / D2 are 15 % / 26 % of a
// 32^3 grid, 30 % / 50 % at 16^3, 57 % / 87 % at 8^3, while a brick (4x8x8) is "active" as soon as it holds one
// such voxel. So a workgroup computes only the ACTIVE outputs of its brick: their local ids come from a per-brick
// list (coordinate-only: built once per (level, resolution) on the geometry stream), they are packed 32 to an
// MFMA column tile, and the B fragment of (tile, tap) is still "halo slot of my voxel + constant tap offset" --
// the main loop is the split kernel's, with 1..8 gathered tiles instead of 8 fixed ones. The remaining voxels of the
// brick get their constant (and its exact contribution to the GroupNorm statistics) from the same workgroup.
// Waves: WM = 2 -> 64 channels per workgroup, the tiles are dealt to two wave columns; WM = 1 (layers of 32
// channels) -> four wave columns. Values are bit-identical to the dense split kernel on the computed outputs.
// ================================================================================================

// per (sample, brick): local ids (ld*8 + lh)*8 + lw of the voxels in D1 (which = 0) / D2 (which = 1), in an
// LDS-conflict-avoiding order (below), followed by the ids NOT in the set; counts[which][b][brick] = size of the set.
template <int R>
__global__ __launch_bounds__(256) void active_lists_kernel(const int *__restrict__ cnt, unsigned char *__restrict__ lists,
                                                           int *__restrict__ counts, int nb) {
  constexpr int TD = 4, TH = 8, TW = 8, BH = R / TH, BW = R / TW, NBRICK = (R / TD) * BH * BW;
  constexpr int ED = TD + 4, EH = TH + 4, EW = TW + 4;  // occupancy, brick +- 2
  constexpr int FD = TD + 2, FH = TH + 2, FW = TW + 2;  // D1, brick +- 1
  __shared__ unsigned char occ[ED * EH * EW], d1[FD * FH * FW];
  __shared__ int wcount[2][4], wbc[2][4][16];
  const int b = blockIdx.y, bk = blockIdx.x, t = threadIdx.x;
  const int d0 = (bk / (BH * BW)) * TD, h0 = ((bk / BW) % BH) * TH, w0 = (bk % BW) * TW;
  for (int e = t; e < ED * EH * EW; e += 256) {
    const int d = d0 - 2 + e / (EH * EW), h = h0 - 2 + (e / EW) % EH, w = w0 - 2 + e % EW;
    const bool in = (unsigned)d < (unsigned)R && (unsigned)h < (unsigned)R && (unsigned)w < (unsigned)R;
    occ[e] = in && cnt[(size_t)b * R * R * R + (d * R + h) * R + w] > 0;
  }
  __syncthreads();
  for (int e = t; e < FD * FH * FW; e += 256) {
    const int z = e / (FH * FW), y = (e / FW) % FH, x = e % FW;  // voxel (d0-1+z, ...): occ index offset by +1
    int any = 0;
    for (int k = 0; k < 27; ++k) any |= occ[((z + k / 9) * EH + (y + (k / 3) % 3)) * EW + x + k % 3];
    // a voxel outside the grid is never an input: its D1 flag must not leak into D2 of its neighbours
    const int d = d0 - 1 + z, h = h0 - 1 + y, w = w0 - 1 + x;
    const bool in = (unsigned)d < (unsigned)R && (unsigned)h < (unsigned)R && (unsigned)w < (unsigned)R;
    d1[e] = in ? any : 0;
  }
  __syncthreads();
  const int ld = t / 64, lh = (t / 8) % 8, lw = t % 8;
  int f[2];
  f[0] = d1[((ld + 1) * FH + lh + 1) * FW + lw + 1];
  f[1] = 0;
  for (int k = 0; k < 27; ++k) f[1] |= d1[((ld + k / 9) * FH + lh + (k / 3) % 3) * FW + lw + k % 3];
  const int lane = t & 63, wave = t >> 6;
  // Order of the active ids: the convolution reads the B fragment of a column tile with one ds_read_b128 per lane at
  // "halo slot of my voxel + tap offset", served in groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31} of a
  // half-wave), one cycle per group when the 16 slots differ mod 16. Sorting the ids by (rank inside their residue
  // class, residue) makes any 16 consecutive ones (nearly) distinct mod 16; full tiles then deal the first / second
  // 16 of their 32 ids to the lanes of the first / second service group. Ascending ids would be 2-3-way conflicted.
  const int rho = ((ld * (TH + 2) + lh) * (TW + 2) + lw) & 15;
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    int rk = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const unsigned long long m = __ballot(f[w] && rho == q);
      if (rho == q) rk = mbcnt(m);
      if (lane == 0) wbc[w][wave][q] = __popcll(m);
    }
    f[w] |= rk << 1;  // bit 0: active, the rest: rank among the wave's active ids of the same residue
  }
  __syncthreads();
  constexpr int POS[32] = {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27,
                           4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31};
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    const int act = f[w] & 1;
    int rk = f[w] >> 1, total = 0, below = 0, inact_before = 0;
    for (int q = 0; q < wave; ++q) rk += wbc[w][q][rho];
    for (int q = 0; q < 16; ++q) {
      int c = 0;
      for (int v = 0; v < 4; ++v) c += wbc[w][v][q];
      total += c;
      below += min(c, rk) + (q < rho && c > rk);  // ids sorted before (rk, rho)
    }
    // inactive ids keep their ascending order behind the active ones
    const unsigned long long ia = __ballot(!act);
    if (lane == 0) wcount[w][wave] = __popcll(ia);
    __syncthreads();
    for (int q = 0; q < wave; ++q) inact_before += wcount[w][q];
    int slot;
    if (act) slot = below < (total & ~31) ? (below & ~31) + POS[below & 31] : below;
    else slot = total + inact_before + mbcnt(ia);
    unsigned char *dst = lists + (((size_t)w * nb + b) * NBRICK + bk) * 256;
    dst[slot] = (unsigned char)t;
    if (t == 0) counts[((size_t)w * nb + b) * NBRICK + bk] = total;
  }
}

// lists u8[2][b][NBRICK][256], counts i32[2][b][NBRICK]; r in {8, 16, 32}
#if CONV_TU == 0
extern "C" int p2pb_conv3d_active_lists(int b, int r, const int *cnt, unsigned char *lists, int *counts, void *stream) {
  if (b <= 0 || (r != 8 && r != 16 && r != 32)) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nbrick = conv_bricks(r);
  if (r == 32) hipLaunchKernelGGL(active_lists_kernel<32>, dim3(nbrick, b), dim3(256), 0, s, cnt, lists, counts, b);
  else if (r == 16) hipLaunchKernelGGL(active_lists_kernel<16>, dim3(nbrick, b), dim3(256), 0, s, cnt, lists, counts, b);
  else hipLaunchKernelGGL(active_lists_kernel<8>, dim3(nbrick, b), dim3(256), 0, s, cnt, lists, counts, b);
  return p2pb_launch_status();
}
#endif

template <int R, int WM, bool XF, int TERMS, bool PRE = false>  // TERMS, PRE: see conv3d_k3_split_kernel
__global__ __launch_bounds__(256, TERMS == SPLIT_F16X3 ? CONV_F16_WAVES : 2) void conv3d_k3_compact_kernel(int cin, int cout, int nchunk, int cout_pad,
                                                                const float *__restrict__ in,
                                                                const unsigned short *__restrict__ wt,
                                                                const float *__restrict__ bias,
                                                                const float *__restrict__ out_class,
                                                                const float *__restrict__ in_scale,
                                                                const float *__restrict__ in_shift, int in_swish,
                                                                const float *__restrict__ in_sub, int skip_zero,
                                                                const unsigned char *__restrict__ alist,
                                                                const int *__restrict__ acount,
                                                                float *__restrict__ out, float *__restrict__ stats_part) {
  using G = SplitGeom<R>;
  constexpr int HD = G::TD + 2, HH = G::TH + 2, HW = G::TW + 2;
  constexpr int PLANE = HD * HH * HW;
  constexpr int BH = R / G::TH, BW = R / G::TW, BD = R / G::TD, NBRICK = BD * BH * BW;
  constexpr int R3 = R * R * R;
  constexpr int WN = 4 / WM;
  static_assert(!PRE || (TERMS == SPLIT_F16X3 && !XF), "pre-split operands: f16x3, transform applied");
  __shared__ u32x4 tile[split_planes(TERMS) * 2 * PLANE];
  __shared__ u32x4 tile2[PRE ? split_planes(TERMS) * 2 * PLANE : 1];  // (its own object: see the split kernel)
  __shared__ unsigned char lst[256];
  __shared__ int ncls[27];
  __shared__ float wstat[4][2][16][2];  // per wave, half-wave, accumulator row: {sum, sumsq} over the active outputs

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  // XCD-aware order (see the split kernel): XCD x gets the x-th contiguous eighth of (sample, brick, channel block)
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const unsigned nblk = gridDim.x * gridDim.y * gridDim.z;
  const int ncoblk = (cout + 32 * WM - 1) / (32 * WM);
  const unsigned v = nblk % 8 == 0 ? (lin % 8) * (nblk / 8) + lin / 8 : lin;
  const unsigned per_sample = NBRICK * ncoblk;
  const int b = v / per_sample;
  const int brick = (v % per_sample) / ncoblk, coblk = (v % per_sample) % ncoblk;
  const int bd = brick / (BH * BW), bh = (brick / BW) % BH, bw = brick % BW;
  const int d0 = bd * G::TD, h0 = bh * G::TH, w0 = bw * G::TW;
  const int wm = wave / WN, wn = wave % WN;
  const int cob = coblk * (32 * WM);  // first channel of the workgroup
  const int co0 = cob + 32 * wm;      // first channel of this wave's M-tile

  CONV_TL_INIT
  CONV_TL_ID(tid);
  CONV_TL(tid);  // 0: start
  const int count = acount[(size_t)b * NBRICK + brick];
  lst[tid] = alist[((size_t)b * NBRICK + brick) * 256 + tid];
  if (tid < 27) ncls[tid] = 0;
  __syncthreads();
  CONV_TL(tid);  // 1: brick list in LDS
  const int ntiles = (count + 31) >> 5;
  auto vox_of = [&](int l, int &cls) {
    const int d = d0 + (l >> 6), h = h0 + ((l >> 3) & 7), w = w0 + (l & 7);
    const int cd = d == 0 ? 0 : (d == R - 1 ? 2 : 1), ch = h == 0 ? 0 : (h == R - 1 ? 2 : 1),
              cw = w == 0 ? 0 : (w == R - 1 ? 2 : 1);
    cls = (cd * 3 + ch) * 3 + cw;
    return (d * R + h) * R + w;
  };

  constexpr int NP = (PLANE + 255) / 256;
  int soff[NP];
  unsigned voff[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int e = tid + j * 256;
    const int dz = e / (HH * HW), hy = (e / HW) % HH, wx = e % HW;
    const int d = d0 - 1 + dz, h = h0 - 1 + hy, w = w0 - 1 + wx;
    const bool ok = e < PLANE && (unsigned)d < (unsigned)R && (unsigned)h < (unsigned)R && (unsigned)w < (unsigned)R;
    soff[j] = ok ? (d * R + h) * R + w : -1;
    voff[j] = ok ? (unsigned)soff[j] * (unsigned)cin * 4u : 0x80000000u;
  }
  const float *inb = in + (size_t)b * cin * R3;
  float *outb = out + (size_t)b * cout * R3;
  float stg[CONV_SCK][NP];
  auto stage_load = [&](int ci0) {
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)inb, 0, R3 * cin * 4, 0x00020000);
    if ((cin & 3) == 0) {
#pragma unroll
      for (int j = 0; j < NP; ++j)
#pragma unroll
        for (int q = 0; q < CONV_SCK / 4; ++q) {
          const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[j] + (unsigned)(ci0 + 4 * q) * 4u, 0, 0));
#pragma unroll
          for (int i = 0; i < 4; ++i) stg[4 * q + i][j] = x[i];
        }
    } else {
#pragma unroll
      for (int j = 0; j < NP; ++j)
#pragma unroll
        for (int c = 0; c < CONV_SCK; ++c)
          stg[c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[j] + (unsigned)(ci0 + c) * 4u, 0, 0));
    }
  };

  if (l31 == 31) {  // this wave's statistics accumulate in LDS across the passes (touched by lanes 31 / 63 only)
#pragma unroll
    for (int r = 0; r < 16; ++r) wstat[wave][khalf][r][0] = wstat[wave][khalf][r][1] = 0.0f;
  }

  // wave column wn takes tiles wn, wn + WN, ...: nt of them (<= 8 / WN <= 4), wave-uniform. The whole stage loop is
  // specialised on nt (1..4): each count keeps split_taps' rolling schedule and only its own accumulators
  int nt = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i * WN + wn < ntiles) nt = i + 1;

  auto run = [&](auto ntc_tag) {
    constexpr int NTC = decltype(ntc_tag)::value;  // 0: this wave has no tile, it only helps staging
    constexpr int NA = NTC > 0 ? NTC : 1;
    int nbase[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int o = (i * WN + wn) * 32 + l31;
      const int l = lst[o < count ? o : 0];
      nbase[i] = ((l >> 6) * HH + ((l >> 3) & 7)) * HW + (l & 7);
    }
    f32x16 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    float bvr[16];  // the epilogue's bias values, fetched in one batch under the stage loop (see the split kernel)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      bvr[r] = (NTC > 0 && co < cout && !out_class) ? bias[co] : 0.0f;
    }

    if constexpr (PRE) {  // `in` = the pre-split operand grid: LDS-DMA stages, two buffers, one barrier per stage
      PreStage<R, HD, HH, HW> ps;
      ps.init(tid, d0, h0, w0, nchunk);
      const conv_i32x4 sg = conv_make_rsrc((const u32x4 *)in + (size_t)b * R3 * nchunk * 4, (unsigned)(R3 * nchunk * 64));
      ps.issue(sg, 0, tile, tid);
      CONV_TL(tid);  // 2: first DMA issued
      const unsigned stage_bytes = 6u * cout_pad * 16u, tap_bytes = (unsigned)nchunk * stage_bytes, plane_bytes = 2u * cout_pad * 16u;
      auto rsw = __builtin_amdgcn_make_buffer_rsrc((void *)wt, 0, 27 * (int)tap_bytes, 0x00020000);
      const unsigned wv = (unsigned)(khalf * cout_pad + co0 + l31) * 16u;
      u32x4 aring[CONV_PRE_AD + 1][2];
      if (NTC > 0) {
#pragma unroll
        for (int t = 0; t < CONV_PRE_AD; ++t)
#pragma unroll
          for (int s = 0; s < 2; ++s) aring[t][s] = conv_wload(rsw, wv, t * tap_bytes + s * plane_bytes);
      }
      auto stage = [&](int k, const u32x4 *cur, u32x4 *nxt) {
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
        __syncthreads();
        if (k < 4) CONV_TL_AT(tid, 3 + 2 * k);  // stage k released (k < 4)
        if (k + 1 < nchunk) ps.issue(sg, k + 1, nxt, tid);
        if (NTC > 0)
          split_taps_pre<NA, HH, HW, PLANE>(acc, cur, rsw, wv, (unsigned)k * stage_bytes, stage_bytes, tap_bytes, plane_bytes,
                                            k + 1 < nchunk, nbase, khalf, aring);
        if (k < 4) CONV_TL_AT(tid, 4 + 2 * k);  // its taps issued
      };
      for (int k = 0; k < nchunk; k += 2) {
        stage(k, tile, tile2);
        if (k + 1 < nchunk) stage(k + 1, tile2, tile);
      }
    } else {
    stage_load(0);
    for (int ci0 = 0; ci0 < cin; ci0 += CONV_SCK) {
      __syncthreads();
      int nonzero = 0;
#pragma unroll
      for (int c = 0; c < CONV_SCK; ++c) {
        float sc = 1.0f, sh = 0.0f, sub = 0.0f;
        const bool cok = ci0 + c < cin;
        if (XF && cok) {
          sc = in_scale[b * cin + ci0 + c];
          sh = in_shift[b * cin + ci0 + c];
          if (in_sub) sub = in_sub[b * cin + ci0 + c];
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          float x = cok ? stg[c][j] : 0.0f;
          if (XF && cok && soff[j] >= 0) x = xf_apply(x, sc, sh, in_swish) - sub;
          nonzero |= (x != 0.0f);
          stg[c][j] = x;
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
              split_pair<TERMS>(stg[h * 8 + 2 * i][j], stg[h * 8 + 2 * i + 1][j], p0, p1, p2);
              q[0][i] = p0;
              q[1][i] = p1;
              q[2][i] = p2;
            }
#pragma unroll
            for (int s = 0; s < split_planes(TERMS); ++s) tile[(s * 2 + h) * PLANE + e] = q[s];
          }
        }
      }
      const int any = (skip_zero & 1) ? __syncthreads_or(nonzero) : (__syncthreads(), 1);
      if (ci0 + CONV_SCK < cin) {
        int nxt = ci0 + CONV_SCK;
        asm volatile("" : "+s"(nxt));
        stage_load(nxt);
      }
      if (!any || NTC == 0) continue;
      const u32x4 *wchunk = (const u32x4 *)wt + (((size_t)(ci0 / CONV_SCK) * 3) * 2 + khalf) * cout_pad + co0 + l31;
      const size_t wsplit_stride = (size_t)2 * cout_pad, wtap_stride = (size_t)nchunk * 3 * 2 * cout_pad;
      split_taps<NA, HH, HW, PLANE, TERMS>(acc, tile, wchunk, wsplit_stride, wtap_stride, nbase, khalf);
    }
    }  // !PRE
    CONV_TL_AT(tid, 11);  // stage loop done
    // boundary-class constants of a second convolution: the workgroup's [27][32 WM] slice of K[b] through LDS (the operand
    // tile is free now; every wave takes part, also those without a tile) instead of a dependent global load per
    // (row, tile) in the epilogue
    constexpr int NCW = 32 * WM;
    float *kl = (float *)tile;
    if (out_class) {
      __syncthreads();
      const float *kbs = out_class + (size_t)b * 27 * cout;
      for (int e = tid; e < 27 * NCW; e += 256) {
        const int c = e % NCW, co = cob + c;
        kl[e] = co < cout ? kbs[(e / NCW) * cout + co] : 0.0f;
      }
      __syncthreads();
    }
    CONV_TL_AT(tid, 12);  // class constants staged
    if (NTC == 0) return;
    if constexpr (TERMS == SPLIT_F16X3) {
      const float oscale = ((const float *)((const char *)wt + conv_split_trailer_bytes(nchunk, cout_pad)))[1];
#pragma unroll
      for (int n = 0; n < NA; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] *= oscale;
    }

    // ---- the active outputs: bias / class constant, 16-byte voxel-major stores, statistics
    int ovox[NA], ocls[NA];
    bool oact[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int o = (i * WN + wn) * 32 + l31;
      oact[i] = o < count;
      ovox[i] = vox_of(lst[oact[i] ? o : 0], ocls[i]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float vv[NA][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * g + i;
        const int co = co0 + i + 8 * g + 4 * khalf;
        const bool cok = co < cout;
        const float bv = bvr[r];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int n = 0; n < NA; ++n) {
          float x = acc[n][r] + bv;
          if (out_class && cok) x += kl[ocls[n] * NCW + 32 * wm + i + 8 * g + 4 * khalf];
          vv[n][i] = x;
          if (oact[n]) {
            s1 += x;
            s2 += x * x;
          }
        }
        s1 = halfwave_sum_to_last(s1);
        s2 = halfwave_sum_to_last(s2);
        if (l31 == 31) {
          wstat[wave][khalf][r][0] = s1;
          wstat[wave][khalf][r][1] = s2;
        }
      }
      const int cq = co0 + 8 * g + 4 * khalf;
#pragma unroll
      for (int n = 0; n < NA; ++n) {
        if (!oact[n]) continue;
        float *q = outb + (size_t)ovox[n] * cout + cq;
        if (cq + 3 < cout && (cout & 3) == 0) *(f32x4 *)q = f32x4{vv[n][0], vv[n][1], vv[n][2], vv[n][3]};
        else
          for (int i = 0; i < 4; ++i)
            if (cq + i < cout) q[i] = vv[n][i];
      }
    }
  };
  if (ntiles > 0) {  // (workgroup-uniform: every wave runs the stage loop, with its own tile count)
    if (nt == 0) run(std::integral_constant<int, 0>{});
    else if (nt == 1) run(std::integral_constant<int, 1>{});
    else if (nt == 2) run(std::integral_constant<int, 2>{});
    else if (nt == 3) run(std::integral_constant<int, 3>{});
    else run(std::integral_constant<int, 4>{});
  }
  __syncthreads();
  CONV_TL_AT(tid, 13);  // active outputs stored

  // ---- the brick's other voxels: their constant, and its exact share of the statistics
  const int ninact = 256 - count;
  for (int e = tid; e < ninact; e += 256) {
    int cls;
    (void)vox_of(lst[count + e], cls);
    atomicAdd(&ncls[cls], 1);
  }
  __syncthreads();
  const int cw = min(32 * WM, cout - cob);  // channels of this workgroup
  const float *kb = out_class ? out_class + (size_t)b * 27 * cout : nullptr;
  // (skip_zero bit 1 = "listed outputs only": the caller reads `out` at listed voxels alone -- a PVConv's second convolution, whose
  //  only reader is the devoxelisation: its corners lie within one voxel of an occupied voxel, inside D1 -- so the constants are
  //  not stored; their statistics below stay exact)
  if (skip_zero & 2) {
  } else if ((cout & 3) == 0) {  // 16 bytes per thread; (voxel, channel quad) advance incrementally, no division in the loop
    const int cw4 = cw >> 2, dq = 256 / cw4, dr = 256 % cw4;
    int vi = tid / cw4, c4 = tid % cw4;
    f32x4 bq = {0.0f, 0.0f, 0.0f, 0.0f};
    if (!kb) bq = *(const f32x4 *)(bias + cob + 4 * c4);
#pragma unroll 4
    for (; vi < ninact; vi += dq) {
      int cls;
      const int vx = vox_of(lst[count + vi], cls);
      *(f32x4 *)(outb + (size_t)vx * cout + cob + 4 * c4) = kb ? *(const f32x4 *)(kb + cls * cout + cob + 4 * c4) : bq;
      if (dr) {
        c4 += dr;
        if (c4 >= cw4) {
          c4 -= cw4;
          ++vi;
        }
        if (!kb) bq = *(const f32x4 *)(bias + cob + 4 * c4);
      }
    }
  } else {
    for (int e = tid; e < ninact * cw; e += 256) {
      const int vi = e / cw, c = e - vi * cw;
      int cls;
      const int vx = vox_of(lst[count + vi], cls);
      outb[(size_t)vx * cout + cob + c] = kb ? kb[cls * cout + cob + c] : bias[cob + c];
    }
  }
  if (stats_part) {
    // slots of the brick: [0, WN) = the wave columns' active sums, WN = the constants' sums, the rest zero
    float *sp = stats_part + (((size_t)b * NBRICK + brick) * 4) * cout * 2;
    if (l31 == 31) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (co < cout) {
          sp[((size_t)wn * cout + co) * 2] = wstat[wave][khalf][r][0];
          sp[((size_t)wn * cout + co) * 2 + 1] = wstat[wave][khalf][r][1];
        }
      }
    }
    if (WN == 4) __syncthreads();
    if (tid < cw) {
      const int co = cob + tid;
      float s1 = 0.0f, s2 = 0.0f;
      if (kb) {
        for (int c = 0; c < 27; ++c) {
          const float x = kb[c * cout + co], n = (float)ncls[c];
          s1 += n * x;
          s2 += n * x * x;
        }
      } else {
        const float x = bias[co];
        s1 = (float)ninact * x;
        s2 = (float)ninact * x * x;
      }
      if (WN == 4) {  // no free slot: on top of wave column 0's sums (written before the barrier above)
        sp[(size_t)co * 2] += s1;
        sp[(size_t)co * 2 + 1] += s2;
      } else {
        sp[((size_t)WN * cout + co) * 2] = s1;
        sp[((size_t)WN * cout + co) * 2 + 1] = s2;
      }
      for (int sl = WN + 1; sl < 4; ++sl) {
        sp[((size_t)sl * cout + co) * 2] = 0.0f;
        sp[((size_t)sl * cout + co) * 2 + 1] = 0.0f;
      }
    }
  }
#ifdef CONV_TIMELINE
  __builtin_amdgcn_s_waitcnt(0x0f70);
  CONV_TL_AT(tid, 14);  // constants + statistics written
#endif
}

// in f32[b,r,r,r,cin] -> out f32[b,r,r,r,cout] (voxel-major), wt = split pack; alist/acount = ONE set of
// p2pb_conv3d_active_lists (D1 for a first convolution, D2 for a second one in far-field form). r in {8,16,32}.
// the bf16x6 instantiations live in the -DCONV_TU=6 object (the -DCONV_TU=3 object has no compact form)
#if CONV_TU != 3
int conv3d_tu6_compact(int b, int cin, int cout, int r, const float *in, const void *wt_split, const float *bias,
                       const float *out_class, const float *in_scale, const float *in_shift, int in_swish,
                       const float *in_sub, const unsigned char *alist, const int *acount, float *out, float *stats_part,
                       hipStream_t s, int listed_only);
static int conv_launch_compact(int b, int cin, int cout, int r, const float *in, const void *wt_split, const float *bias,
                               const float *out_class, const float *in_scale, const float *in_shift, int in_swish,
                               const float *in_sub, const unsigned char *alist, const int *acount, float *out,
                               float *stats_part, hipStream_t s, bool pre = false, int listed_only = 0) {
  const int lo = listed_only ? 2 : 0;  // bit 1 of the kernels' skip_zero argument
  const bool xf = in_scale != nullptr;
#if CONV_TU == 0
  if (p2pb_g_split_terms == SPLIT_BF16X3) return P2PB_EINVAL;  // (the data gradient's arithmetic: dense form only)
  if (pre && (p2pb_g_split_terms == SPLIT_BF16X6 || xf || in_sub)) return P2PB_EINVAL;
  if (p2pb_g_split_terms == SPLIT_BF16X6)
    return conv3d_tu6_compact(b, cin, cout, r, in, wt_split, bias, out_class, in_scale, in_shift, in_swish, in_sub, alist,
                              acount, out, stats_part, s, listed_only);
#endif
  const int nchunk = (cin + CONV_SCK - 1) / CONV_SCK, cout_pad = (cout + 63) / 64 * 64;
  const unsigned short *w = (const unsigned short *)wt_split;
  const bool wm1 = cout <= 32;  // one M-tile per workgroup, tiles dealt to four wave columns
  dim3 grid(conv_bricks(r), (cout + (wm1 ? 31 : 63)) / (wm1 ? 32 : 64), b);
#define LAUNCH(RR, WMV, XF)                                                                                           \
  hipLaunchKernelGGL((conv3d_k3_compact_kernel<RR, WMV, XF, CONV_TERMS>), grid, dim3(256), 0, s, cin, cout, nchunk, cout_pad, \
                     in, w, bias, out_class, in_scale, in_shift, in_swish, in_sub, 1 | lo, alist, acount, out, stats_part)
#define GO(RR)                                                   \
  if (wm1) {                                                     \
    if (xf) LAUNCH(RR, 1, true);                                 \
    else LAUNCH(RR, 1, false);                                   \
  } else {                                                       \
    if (xf) LAUNCH(RR, 2, true);                                 \
    else LAUNCH(RR, 2, false);                                   \
  }
#if CONV_TU == 0
#define GOPRE(RR)                                                                                                          \
  do {                                                                                                                     \
    if (wm1)                                                                                                               \
      hipLaunchKernelGGL((conv3d_k3_compact_kernel<RR, 1, false, SPLIT_F16X3, true>), grid, dim3(256), 0, s, cin, cout, \
                         nchunk, cout_pad, in, w, bias, out_class, in_scale, in_shift, in_swish, in_sub, lo, alist, acount, out, \
                         stats_part);                                                                                      \
    else                                                                                                                   \
      hipLaunchKernelGGL((conv3d_k3_compact_kernel<RR, 2, false, SPLIT_F16X3, true>), grid, dim3(256), 0, s, cin, cout, \
                         nchunk, cout_pad, in, w, bias, out_class, in_scale, in_shift, in_swish, in_sub, lo, alist, acount, out, \
                         stats_part);                                                                                      \
  } while (0)
  if (pre) {
    if (r == 32) GOPRE(32); else if (r == 16) GOPRE(16); else GOPRE(8);
    return p2pb_launch_status();
  }
#undef GOPRE
#endif
  if (r == 32) { GO(32) } else if (r == 16) { GO(16) } else { GO(8) }
#undef GO
#undef LAUNCH
  return p2pb_launch_status();
}
#if CONV_TU == 6
int conv3d_tu6_compact(int b, int cin, int cout, int r, const float *in, const void *wt_split, const float *bias,
                       const float *out_class, const float *in_scale, const float *in_shift, int in_swish,
                       const float *in_sub, const unsigned char *alist, const int *acount, float *out, float *stats_part,
                       hipStream_t s, int listed_only) {
  return conv_launch_compact(b, cin, cout, r, in, wt_split, bias, out_class, in_scale, in_shift, in_swish, in_sub, alist, acount,
                             out, stats_part, s, false, listed_only);
}
#endif
#endif  // CONV_TU != 3

#if CONV_TU == 0
extern "C" int p2pb_conv3d_k3_forward_compact(int b, int cin, int cout, int r, const float *in, const void *wt_split,
                                              const float *bias, const float *out_class, const float *in_scale,
                                              const float *in_shift, int in_swish, const float *in_sub,
                                              const unsigned char *alist, const int *acount, float *out,
                                              float *stats_part, int flags, void *stream) {
  if (b <= 0 || cin <= 0 || cout <= 0 || !alist || !acount || (r != 8 && r != 16 && r != 32) || (flags & ~32)) return P2PB_EINVAL;
  return conv_launch_compact(b, cin, cout, r, in, wt_split, bias, out_class, in_scale, in_shift, in_swish, in_sub, alist,
                             acount, out, stats_part, (hipStream_t)stream, false, flags & 32);
}
#endif

#if CONV_TU == 0
// the compact form on a pre-split operand grid (S format, see PreStage): in_split u32x4[b][r^3][ceil(cin/16)][4]
extern "C" int p2pb_conv3d_k3_forward_compact_pre(int b, int cin, int cout, int r, const void *in_split,
                                                  const void *wt_split, const float *bias, const float *out_class,
                                                  const unsigned char *alist, const int *acount, float *out,
                                                  float *stats_part, int flags, void *stream) {
  if (b <= 0 || cin <= 0 || cout <= 0 || !in_split || !alist || !acount || (r != 8 && r != 16 && r != 32) || (flags & ~32))
    return P2PB_EINVAL;
  return conv_launch_compact(b, cin, cout, r, (const float *)in_split, wt_split, bias, out_class, nullptr, nullptr, 0,
                             nullptr, alist, acount, out, stats_part, (hipStream_t)stream, true, flags & 32);
}

// y f32[b][nvox][c] (voxel-major) -> S format u32x4[b][nvox][ceil(c/16)][2 planes][2 khalf]: the operand transform of
// the split kernels' staging phase (folded norm + Swish - far-field value, then the fp16-pair split of 4 x value), once
// per element.
#define PRESPLIT_VT 128  // voxels per workgroup
static __global__ __launch_bounds__(256) void conv3d_presplit_kernel(int c, int nchunk, int nvox, const float *__restrict__ y,
                                                                     const float *__restrict__ in_scale,
                                                                     const float *__restrict__ in_shift, int in_swish,
                                                                     const float *__restrict__ in_sub,
                                                                     u32x4 *__restrict__ out) {
  // a thread keeps ONE group of 8 channels (its scale / shift / far-field value in registers) and walks the voxels of the
  // workgroup's tile: 256 / ng voxels per pass, a voxel's row read and written by ng neighbouring lanes (coalesced)
  const int b = blockIdx.y, ng = nchunk * 2, vpp = 256 / ng;
  const int g = threadIdx.x % ng, vl = threadIdx.x / ng, c0 = g * 8;
  if (vl >= vpp) return;
  float sc[8], sh[8], sub[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ch = c0 + i;
    const bool ok = ch < c && in_scale != nullptr;
    sc[i] = ok ? in_scale[b * c + ch] : 1.0f;
    sh[i] = ok ? in_shift[b * c + ch] : 0.0f;
    sub[i] = (ok && in_sub) ? in_sub[b * c + ch] : 0.0f;
  }
  const int v0 = blockIdx.x * PRESPLIT_VT, v1 = min(v0 + PRESPLIT_VT, nvox);
  const bool quad = (c & 3) == 0;
  for (int v = v0 + vl; v < v1; v += vpp) {
    const float *src = y + ((size_t)b * nvox + v) * c + c0;
    float x[8];
    if (quad) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4 t = {0.0f, 0.0f, 0.0f, 0.0f};
        if (c0 + 4 * q < c) t = *(const f32x4 *)(src + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[4 * q + i] = t[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = c0 + i < c ? src[i] : 0.0f;
    }
    if (in_scale) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (c0 + i < c) x[i] = xf_apply(x[i], sc[i], sh[i], in_swish) - sub[i];
    }
    u32x4 p0, p1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned a0, a1, a2;
      split_pair<SPLIT_F16X3>(x[2 * i], x[2 * i + 1], a0, a1, a2);
      p0[i] = a0;
      p1[i] = a1;
    }
    u32x4 *dst = out + (((size_t)b * nvox + v) * nchunk + (g >> 1)) * 4 + (g & 1);
    dst[0] = p0;
    dst[2] = p1;
  }
}

// y f32[b, nvox, c] -> out_split (S format, b * nvox * ceil(c/16) * 64 bytes); in_scale / in_shift / in_sub f32[b,c] or NULL
extern "C" int p2pb_conv3d_presplit(int b, int c, long nvox, const float *y, const float *in_scale, const float *in_shift,
                                    int in_swish, const float *in_sub, void *out_split, void *stream) {
  if (b <= 0 || c <= 0 || nvox <= 0 || nvox > 0x7fffffffL / 64 || !y || !out_split || (in_scale && !in_shift)) return P2PB_EINVAL;
  const int nchunk = (c + CONV_SCK - 1) / CONV_SCK;
  if (nchunk * 2 > 256) return P2PB_EINVAL;  // (<= 2048 channels)
  const dim3 grid(cdiv(nvox, PRESPLIT_VT), b);
  hipLaunchKernelGGL(conv3d_presplit_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, nchunk, (int)nvox, y, in_scale,
                     in_shift, in_swish, in_sub, (u32x4 *)out_split);
  return p2pb_launch_status();
}
#endif

// ------------------------------------------------------------------------------------------------
// GroupNorm(+AdaGN) folded to a per-(sample, channel) affine:  AdaGN(GN(x)) == x*scale + shift.
//   GroupNorm (biased variance, eps) : y = (x - mean_g) * rstd_g * gamma_c + beta_c
//   AdaGN (models/modules.py:341-358): z = y * factor_bc + bias_bc , (factor, bias) = chunk(style, 2)
// Statistics come from the producing kernel's per-slot partial {sum, sum of squares}; they are
// combined in double in a fixed order (deterministic). One thread per (sample, channel); the group
// moments are recomputed by each of the group's channels (C/G <= 64 channels x nslots partials, tiny).
// Also returns chmean[b,c] = mean over positions of the transformed output (SE3d's squeeze input).
// ------------------------------------------------------------------------------------------------
// one workgroup per (sample, group). Thread t accumulates channel (t mod cg) over the slots
// t/cg, t/cg + 256/cg, ... (adjacent threads read adjacent channels: coalesced), the 256/cg partial
// accumulators per channel are then summed in ascending order -- a fixed order, so deterministic.
static __global__ __launch_bounds__(256) void gn_affine_kernel(int c, int groups, int nslots, double count_per_channel,
                                                        const float *__restrict__ part, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, const float *__restrict__ style,
                                                        int style_stride, float eps, float *__restrict__ scale,
                                                        float *__restrict__ shift, float *__restrict__ chmean,
                                                        float *__restrict__ mean_rstd) {
  __shared__ double lds[4 * 256];
  GnFinish f;  // (the arithmetic lives in common.h: producing kernels handed a GnFinish run it themselves)
  f.gamma = gamma, f.beta = beta, f.style = style, f.scale = scale, f.shift = shift, f.chmean = chmean;
  f.count_per_channel = count_per_channel, f.style_stride = style_stride, f.groups = groups, f.eps = eps;
  gn_finish_group(c, nslots, part, f, blockIdx.y, blockIdx.x, lds, mean_rstd);
}

// part: f32[b, nslots, c, 2]; gamma/beta f32[c] or NULL; style = rows of (factor[c] | bias[c]) with a row pitch of
// style_stride floats (a column slice of the one style GEMM of the evaluation), or NULL -> scale/shift/chmean f32[b,c]
#if CONV_TU == 0
// (for the producers of other translation units that were handed a finisher they cannot run themselves)
int p2pb_gn_affine_launch(int b, int c, int nslots, const float *part, const GnFinish &f, hipStream_t s) {
  hipLaunchKernelGGL(gn_affine_kernel, dim3(f.groups, b), dim3(256), 0, s, c, f.groups, nslots, f.count_per_channel, part, f.gamma,
                     f.beta, f.style, f.style_stride, f.eps, f.scale, f.shift, f.chmean, (float *)nullptr);
  return p2pb_launch_status();
}
extern "C" int p2pb_gn_affine_params_ex(int b, int c, int groups, int nslots, double count_per_channel,
                                        const float *part, const float *gamma, const float *beta, const float *style,
                                        int style_stride, float eps, float *scale, float *shift, float *chmean,
                                        float *mean_rstd, void *stream) {
  if (b <= 0 || c <= 0 || groups <= 0 || c % groups != 0 || nslots <= 0 || c / groups > 256) return P2PB_EINVAL;
  if (style && style_stride < 2 * c) return P2PB_EINVAL;
  hipLaunchKernelGGL(gn_affine_kernel, dim3(groups, b), dim3(256), 0, (hipStream_t)stream, c, groups, nslots,
                     count_per_channel, part, gamma, beta, style, style_stride, eps, scale, shift, chmean, mean_rstd);
  return p2pb_launch_status();
}

extern "C" int p2pb_gn_affine_params(int b, int c, int groups, int nslots, double count_per_channel,
                                     const float *part, const float *gamma, const float *beta, const float *style,
                                     int style_stride, float eps, float *scale, float *shift, float *chmean,
                                     void *stream) {
  return p2pb_gn_affine_params_ex(b, c, groups, nslots, count_per_channel, part, gamma, beta, style, style_stride, eps,
                                  scale, shift, chmean, nullptr, stream);
}
#endif

// hid[h] = relu(sum_i w1[h][i] * mean[i]) for the SE3d bottleneck (models/pvcnn.py SE3d.fc[0..1]): ONE WAVE PER ROW -- lane l adds
// the terms i = l, l + 64, ... in that order, then the 64 lane sums through a fixed xor tree. (Round 5: the first form gave a row
// to ONE THREAD, c dependent loads from global memory in a row: 16-32 busy threads and ~12 of the tail kernel's 16 us.)
// Both kernels below use it: the same bits whichever runs.
__device__ __forceinline__ void se_hidden(int c, int hidden, const float *__restrict__ w1, const float *mean, float *hid) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int h = wave; h < hidden; h += nw) {
    float acc = 0.0f;
    for (int i = lane; i < c; i += 64) acc = __fmaf_rn(w1[(size_t)h * c + i], mean[i], acc);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) hid[h] = fmaxf(acc, 0.0f);
  }
}


// ------------------------------------------------------------------------------------------------
// Squeeze-excite gate (models/modules.py:362-378: Linear(c, c/8, no bias) -> ReLU -> Linear(c/8, c, no bias) ->
// Sigmoid on the per-channel mean of the normalised grid) folded into the devoxelisation affine:
//   gate = sigmoid(W2 relu(W1 chmean)),  aff_a = scale * gate,  aff_b = shift * gate.   One workgroup per sample.
// ------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void se_gate_affine_kernel(int c, int hidden, const float *__restrict__ chmean,
                                                             const float *__restrict__ w1, const float *__restrict__ w2,
                                                             const float *__restrict__ scale,
                                                             const float *__restrict__ shift, float *__restrict__ aff_a,
                                                             float *__restrict__ aff_b) {
  extern __shared__ float se_sm[];  // c means + hidden activations
  float *mean = se_sm, *hid = se_sm + c;
  const int b = blockIdx.x, t = threadIdx.x;
  for (int i = t; i < c; i += 256) mean[i] = chmean[(size_t)b * c + i];
  __syncthreads();
  se_hidden(c, hidden, w1, mean, hid);
  __syncthreads();
  for (int i = t; i < c; i += 256) {
    float acc = 0.0f;
    for (int h = 0; h < hidden; ++h) acc = __fmaf_rn(w2[(size_t)i * hidden + h], hid[h], acc);
    const float gate = 1.0f / (1.0f + expf(-acc));
    aff_a[(size_t)b * c + i] = scale[(size_t)b * c + i] * gate;
    aff_b[(size_t)b * c + i] = shift[(size_t)b * c + i] * gate;
  }
}

// ------------------------------------------------------------------------------------------------
// The tail of a PVConv's voxel branch in ONE launch (round 5: three GroupNorm-folding launches + the gate -> one):
//   workgroup (0, b): GroupNorm(+AdaGN) of the SECOND convolution's output from its statistics partials (gn_finish_sample: the
//     arithmetic and bits of gn_affine_kernel) -> scale, shift, channel mean in LDS -> SE3d gate (se_gate_affine_kernel's
//     arithmetic) -> aff_a = scale * gate, aff_b = shift * gate (hidden == 0: no SE3d, aff = scale, shift);
//   workgroup (1, b): the GroupNorm(+AdaGN) of the POINT branch's 1x1 convolution (its partials have been waiting since
//     before the voxel branch started) -> scale_p, shift_p for the devoxelisation pass that adds swish(h * scale_p + shift_p).
// 1024 threads: four groups at a time. Replaces models/pvcnn.py:283-286 (AdaGN, SE3d) + models/pvcnn.py:162-205's norm.
// ------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(1024) void pvconv_tail_kernel(int c, int hidden, const float *__restrict__ part2, int nslots2,
                                                           GnFinish f2, const float *__restrict__ w1, const float *__restrict__ w2,
                                                           float *__restrict__ aff_a, float *__restrict__ aff_b,
                                                           int cp, const float *__restrict__ partp, int nslotsp, GnFinish fp) {
  extern __shared__ double pt_sm[];  // 4 x 1024 doubles | scale[c] | shift[c] | mean[c] | hid[hidden]
  const int b = blockIdx.y, t = threadIdx.x;
  if (blockIdx.x == 1) {
    gn_finish_sample<4>(cp, nslotsp, partp, fp, b, pt_sm);
    return;
  }
  float *sc = (float *)(pt_sm + 4096), *sh = sc + c, *mean = sh + c, *hid = mean + c;
  gn_finish_sample<4>(c, nslots2, part2, f2, b, pt_sm, sc, sh, mean);
  if (hidden <= 0) {
    for (int i = t; i < c; i += 1024) {
      aff_a[(size_t)b * c + i] = sc[i];
      aff_b[(size_t)b * c + i] = sh[i];
    }
    return;
  }
  se_hidden(c, hidden, w1, mean, hid);
  __syncthreads();
  for (int i = t; i < c; i += 1024) {
    float acc = 0.0f;
    for (int h = 0; h < hidden; ++h) acc = __fmaf_rn(w2[(size_t)i * hidden + h], hid[h], acc);
    const float gate = 1.0f / (1.0f + expf(-acc));
    aff_a[(size_t)b * c + i] = sc[i] * gate;
    aff_b[(size_t)b * c + i] = sh[i] * gate;
  }
}

#if CONV_TU == 0
// part2 f32[b, nslots2, c, 2] + its norm (count2 = positions per channel, groups2, gamma2, beta2, style2 rows of 2c floats or NULL)
// -> aff_a, aff_b f32[b, c] (SE3d gate from w1 f32[hidden, c], w2 f32[c, hidden]; hidden == 0: none); partp (may be NULL)
// f32[b, nslotsp, cp, 2] + its norm -> scale_p, shift_p f32[b, cp]
extern "C" int p2pb_pvconv_tail(int b, int c, int hidden, const float *part2, int nslots2, double count2, int groups2,
                                const float *gamma2, const float *beta2, const float *style2, int style_stride2, float eps2,
                                const float *w1, const float *w2, float *aff_a, float *aff_b, int cp, const float *partp,
                                int nslotsp, double countp, int groupsp, const float *gammap, const float *betap,
                                const float *stylep, int style_stridep, float epsp, float *scale_p, float *shift_p, void *stream) {
  if (b <= 0 || c <= 0 || hidden < 0 || !part2 || nslots2 <= 0 || !aff_a || !aff_b || (hidden > 0 && (!w1 || !w2)) ||
      !gn_shape_ok(c, groups2, style2, style_stride2))
    return P2PB_EINVAL;
  if (partp && (cp <= 0 || nslotsp <= 0 || !scale_p || !shift_p || !gn_shape_ok(cp, groupsp, stylep, style_stridep))) return P2PB_EINVAL;
  GnFinish f2 = {}, fp = {};
  f2.gamma = gamma2, f2.beta = beta2, f2.style = style2, f2.style_stride = style_stride2, f2.groups = groups2, f2.eps = eps2;
  f2.count_per_channel = count2;  // (scale / shift / chmean stay in the kernel's LDS tables)
  fp.gamma = gammap, fp.beta = betap, fp.style = stylep, fp.style_stride = style_stridep, fp.groups = groupsp, fp.eps = epsp;
  fp.count_per_channel = countp, fp.scale = scale_p, fp.shift = shift_p;
  const size_t lds = 4096 * 8 + (size_t)(3 * c + hidden) * 4;
  if (lds > 64 * 1024) return P2PB_EINVAL;
  hipLaunchKernelGGL(pvconv_tail_kernel, dim3(partp ? 2 : 1, b), dim3(1024), lds, (hipStream_t)stream, c, hidden, part2, nslots2, f2,
                     w1, w2, aff_a, aff_b, cp, partp, nslotsp, fp);
  return p2pb_launch_status();
}
#endif

#if CONV_TU == 0
extern "C" int p2pb_se_gate_affine(int b, int c, int hidden, const float *chmean, const float *w1, const float *w2,
                                   const float *scale, const float *shift, float *aff_a, float *aff_b, void *stream) {
  if (b <= 0 || c <= 0 || hidden <= 0) return P2PB_EINVAL;
  hipLaunchKernelGGL(se_gate_affine_kernel, dim3(b), dim3(256), (size_t)(c + hidden) * sizeof(float),
                     (hipStream_t)stream, c, hidden, chmean, w1, w2, scale, shift, aff_a, aff_b);
  return p2pb_launch_status();
}
#endif

