// wgrad.hip -- weight gradients of the dense layers for training (BASELINE config 3), exact fp32 on the gfx950
// matrix cores (v_mfma_f32_32x32x2_f32). Replaces what cuDNN / cuBLAS compute for the reference's
// nn.Conv3d (models/pvcnn.py:265-282) and k=1 Conv1d / Conv2d (models/pvcnn.py:162-205, 803-823) in backward:
//
//   3x3x3 convolution:   dW[co][ci][tap] = sum_{b, v} dY[b, co, v] * X[b, ci, v + off(tap)]     (zero padding)
//   1x1 convolution:     dW[co][ci]      = sum_{b, p} dY[b, co, p] * X[b, ci, p]
//   both:                db[co]          = sum_{b, p} dY[b, co, p]
//
// (The data gradients need no kernel of their own: dX of the 3x3x3 convolution is the forward kernel of conv3d.hip
// run on dY with the taps flipped and the channel roles swapped, dX of a 1x1 layer is the forward GEMM of
// pointwise.hip with the transposed weight -- p2p_bridge_amd/dense.py.)
//
// Formulation: a GEMM whose reduction dimension is the VOXEL / POSITION index: M = output channels (A operand = dY
// rows), N = input channels (B operand = X rows, shifted by the tap offset), K = B * r^3. The output is tiny and K is
// huge, so the launch is split over K: every workgroup owns one (co tile, ci tile) and walks a strided subset of
// the K units (4x8x8 voxel bricks / 256-position chunks), keeping the accumulators in registers all the way, and
// writes ONE partial [co tile][ci tile][taps]; wgrad_reduce_kernel adds the partials in a fixed order
// (deterministic: no float atomics). Per unit the workgroup stages the dY brick [64][256] and the zero-padded halo
// brick of X [32][6*10*10] in LDS (odd row pitches: the 32 lanes of a fragment read hit 32 distinct banks), then every
// k-pair (two voxels) is one ds_read_b32 per fragment. The 27 taps are dealt to the four waves (7/7/7/6): a wave keeps
// 2 x 7 accumulator tiles (224 VGPRs) and spends 9 LDS reads per 14 MFMAs (896 matrix cycles) -- the kernel is
// matrix-bound by construction, staging is ~2 % of a unit.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WG_COT 64  // output channels per workgroup (2 M-tiles)
#define WG_CIT 32  // input channels per workgroup (1 N-tile), conv
#define WG_TPW 7   // taps per wave

template <int TD, int TH, int TW>
struct WBrick {
  static constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
  static constexpr int NV = TD * TH * TW, PLANE = HD * HH * HW;
  static constexpr int PA = NV + 1, PB = PLANE | 1;  // odd LDS row pitches
};

template <int R, int TD, int TH, int TW>
__global__ __launch_bounds__(256) void conv3d_k3_wgrad_kernel(int nb, int cin, int cout, int nsplit,
                                                              const float *__restrict__ x,
                                                              const float *__restrict__ dy,
                                                              float *__restrict__ part, float *__restrict__ bpart) {
  using G = WBrick<TD, TH, TW>;
  constexpr int R3 = R * R * R;
  constexpr int BD = R / TD, BH = R / TH, BW = R / TW, NBRICK = BD * BH * BW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *dys = smem;                    // [WG_COT][PA]
  float *xs = smem + WG_COT * G::PA;    // [WG_CIT][PB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, khalf = lane >> 5;
  const int split = blockIdx.x, co0 = blockIdx.y * WG_COT, ci0 = blockIdx.z * WG_CIT;
  const int tap0 = wave * WG_TPW;
  const int ntap = min(WG_TPW, 27 - tap0);

  f32x16 acc[2][WG_TPW];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < WG_TPW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.0f;
  float bsum = 0.0f;  // thread tid < WG_COT: running sum of dY row tid (only the ci-tile-0 workgroups write it)

  const int units = nb * NBRICK;
  for (int u = split; u < units; u += nsplit) {
    const int b = u / NBRICK, bk = u % NBRICK;
    const int d0 = (bk / (BH * BW)) * TD, h0 = ((bk / BW) % BH) * TH, w0 = (bk % BW) * TW;
    __syncthreads();  // everyone is done with the previous unit's tiles
    // ---- stage dY brick: thread -> voxel(s) j, all rows
    for (int j = tid; j < G::NV; j += 256) {
      const int jd = j / (TH * TW), jh = (j / TW) % TH, jw = j % TW;
      const size_t gv = ((size_t)(d0 + jd) * R + (h0 + jh)) * R + (w0 + jw);
      const float *src = dy + ((size_t)b * cout + co0) * R3 + gv;
#pragma unroll 8
      for (int c = 0; c < WG_COT; ++c) dys[c * G::PA + j] = (co0 + c < cout) ? src[(size_t)c * R3] : 0.0f;
    }
    // ---- stage the zero-padded halo brick of X
    for (int e = tid; e < G::PLANE; e += 256) {
      const int dz = e / (G::HH * G::HW), hy = (e / G::HW) % G::HH, wx = e % G::HW;
      const int d = d0 - 1 + dz, h = h0 - 1 + hy, w = w0 - 1 + wx;
      const bool ok = (unsigned)d < (unsigned)R && (unsigned)h < (unsigned)R && (unsigned)w < (unsigned)R;
      const float *src = x + ((size_t)b * cin + ci0) * R3 + ((size_t)d * R + h) * R + w;
#pragma unroll 8
      for (int c = 0; c < WG_CIT; ++c) xs[c * G::PB + e] = (ok && ci0 + c < cin) ? src[(size_t)c * R3] : 0.0f;
    }
    __syncthreads();
    if (bpart && blockIdx.z == 0 && tid < WG_COT) {
      float s = 0.0f;
      for (int j = 0; j < G::NV; ++j) s += dys[tid * G::PA + j];
      bsum += s;
    }
    // ---- K loop over voxel pairs: voxel v = 2*kk + khalf
    const float *arow = dys + l31 * G::PA;
    const float *brow = xs + l31 * G::PB;
#pragma unroll 2
    for (int kk = 0; kk < G::NV / 2; ++kk) {
      const int v = 2 * kk + khalf;
      const int jd = v / (TH * TW), jh = (v / TW) % TH, jw = v % TW;
      const int hb = (jd * G::HH + jh) * G::HW + jw;
      const float a0 = arow[v], a1 = arow[32 * G::PA + v];
#pragma unroll
      for (int t = 0; t < WG_TPW; ++t) {
        if (t < ntap) {
          const int tap = tap0 + t;
          const int toff = ((tap / 9) * G::HH + (tap / 3) % 3) * G::HW + tap % 3;
          const float bv = brow[hb + toff];
          acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1][t], 0, 0, 0);
        }
      }
    }
  }
  // ---- partial out: part[split][tap][co][ci] (lanes = consecutive ci: coalesced rows), bias sums behind it
  float *po = part + (size_t)split * ((size_t)cout * cin * 27 + cout);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < WG_TPW; ++t) {
      if (t >= ntap) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf, ci = ci0 + l31;
        if (co < cout && ci < cin) po[((size_t)(tap0 + t) * cout + co) * cin + ci] = acc[m][t][r];
      }
    }
  if (bpart && blockIdx.z == 0 && tid < WG_COT && co0 + tid < cout) po[(size_t)cout * cin * 27 + co0 + tid] = bsum;
}

// 1x1 layers: units = 256-position chunks of one sample; workgroup tile 64 x 64, wave (w & 1, w >> 1) owns one
// 32 x 32 tile
#define PW_CH 256
__global__ __launch_bounds__(256) void pointwise_wgrad_kernel(int nb, int cin, int cout, int npos, int nsplit,
                                                              const float *__restrict__ x,
                                                              const float *__restrict__ dy,
                                                              float *__restrict__ part, float *__restrict__ bpart) {
  constexpr int PA = PW_CH + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *dys = smem;            // [64][PA]
  float *xs = smem + 64 * PA;   // [64][PA]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, khalf = lane >> 5;
  const int split = blockIdx.x, co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float bsum = 0.0f;
  const int nchunk = (npos + PW_CH - 1) / PW_CH;
  const int units = nb * nchunk;
  for (int u = split; u < units; u += nsplit) {
    const int b = u / nchunk, p0 = (u % nchunk) * PW_CH;
    const int np = min(PW_CH, npos - p0);
    __syncthreads();
    {  // thread -> position tid (coalesced rows)
      const bool ok = tid < np;
      const float *sd = dy + ((size_t)b * cout + co0) * npos + p0 + tid;
      const float *sx = x + ((size_t)b * cin + ci0) * npos + p0 + tid;
#pragma unroll 8
      for (int c = 0; c < 64; ++c) {
        dys[c * PA + tid] = (ok && co0 + c < cout) ? sd[(size_t)c * npos] : 0.0f;
        xs[c * PA + tid] = (ok && ci0 + c < cin) ? sx[(size_t)c * npos] : 0.0f;
      }
    }
    __syncthreads();
    if (bpart && blockIdx.z == 0 && tid < 64) {
      float s = 0.0f;
      for (int j = 0; j < np; ++j) s += dys[tid * PA + j];
      bsum += s;
    }
    const float *arow = dys + ((wave & 1) * 32 + l31) * PA;
    const float *brow = xs + ((wave >> 1) * 32 + l31) * PA;
#pragma unroll 4
    for (int kk = 0; kk < PW_CH / 2; ++kk) {
      const int v = 2 * kk + khalf;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[v], brow[v], acc, 0, 0, 0);
    }
  }
  float *po = part + (size_t)split * ((size_t)cout * cin + cout);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + (wave & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf, ci = ci0 + (wave >> 1) * 32 + l31;
    if (co < cout && ci < cin) po[(size_t)co * cin + ci] = acc[r];
  }
  if (bpart && blockIdx.z == 0 && tid < 64 && co0 + tid < cout) po[(size_t)cout * cin + co0 + tid] = bsum;
}

// ================================================================================================
// Split-operand (bf16) forms of the same two GEMMs -- the default. torch.set_float32_matmul_precision("high")
// (the reference's choice, train.py:221) means exactly this arithmetic: every fp32 operand is the sum of bf16
// terms and the product is accumulated in fp32 on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16: 16 K-values per
// 32-cycle instruction against 2 per 64 cycles for the exact-fp32 MFMA above). NTERM = 2 ("bf16x3": x0y0 + x0y1 +
// x1y0, 16 significand bits, the precision class of the reference's TF32 cuDNN / cuBLAS kernels, 3 MFMAs per 16
// K-values) or NTERM = 3 ("bf16x6", fp32-faithful like the forward kernels, 6 MFMAs).
//
// K = voxels, and an MFMA lane carries 8 CONSECUTIVE K-values of its row: 8 voxels along w. Both operands are
// activations that live in HBM as fp32 rows [channel][voxel], so a fragment is one 32-byte run of a row -- loaded
// straight from L1 / L2 (no LDS: every element is used by one lane only; the reuse is across waves and taps, which
// the caches serve) and split into its bf16 terms in registers. The tap shift is along the SAME axis as the
// fragment for kw != 1; per (kd, kh) row the lane loads the aligned run f[0..7] plus its two neighbours and forms
// the three kw fragments (l,f0..f6) / (f0..f7) / (f1..f7,r) from them: one load + one split per 3 taps.
// Workgroup = 64 co x 64 ci x ONE kd plane (9 taps): wave (m, n) owns the 32 x 32 tile (m, n) for those 9 taps =
// 144 accumulator registers -> two waves per SIMD hide the load latency. Split over K as above.
// ================================================================================================
typedef float f32x4w __attribute__((ext_vector_type(4)));
template <int NTERM>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&t)[NTERM]) {
  unsigned p0, p1, p2;
  split3(a, b, p0, p1, p2);
  t[0] = p0;
  if (NTERM > 1) t[1] = p1;
  if (NTERM > 2) t[2] = p2;
}

template <int NTERM>
__device__ __forceinline__ void mfma_products(f32x16 &acc, const u32x4 (&a)[NTERM], const u32x4 (&b)[NTERM]) {
  // small terms first; (i, j) with i + j < NTERM
#pragma unroll
  for (int s = NTERM - 1; s >= 0; --s)
#pragma unroll
    for (int i = 0; i <= s; ++i)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                    __builtin_bit_cast(bf16x8, b[s - i]), acc, 0, 0, 0);
}

__device__ __forceinline__ void load8(const float *p, bool vec, float (&f)[8]) {
  if (vec) {
    const float4 v0 = *(const float4 *)p, v1 = *(const float4 *)(p + 4);
    f[0] = v0.x, f[1] = v0.y, f[2] = v0.z, f[3] = v0.w, f[4] = v1.x, f[5] = v1.y, f[6] = v1.z, f[7] = v1.w;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = p[i];
  }
}

template <int R, int NTERM>
__global__ __launch_bounds__(256, 2) void conv3d_k3_wgrad_bf16_kernel(int nb, int cin, int cout, int nsplit,
                                                                      const float *__restrict__ x,
                                                                      const float *__restrict__ dy,
                                                                      float *__restrict__ part,
                                                                      float *__restrict__ bpart) {
  constexpr int R3 = R * R * R, KG = R3 / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, khalf = lane >> 5;
  const int split = blockIdx.x, kd = blockIdx.z;
  const int ncit = (cin + 63) / 64;
  const int co_t = (blockIdx.y / ncit) * 64 + (wave & 1) * 32, ci_t = (blockIdx.y % ncit) * 64 + (wave >> 1) * 32;
  const int co = co_t + l31, ci = ci_t + l31;
  const bool cok = co < cout, cik = ci < cin;
  const bool want_bias = bpart && kd == 0 && (blockIdx.y % ncit) == 0 && (wave >> 1) == 0;
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  float bsum = 0.0f;
  const int total = nb * KG;
  if constexpr (R >= 8) {
    // Round 5: the loop used to wait for every load right behind its issue -- dY, then each of the three kh rows: four
    // dependent memory round trips per K unit (~3 us) in front of 27 MFMAs (0.4 us); the r = 32 launches ran at 17 % of their
    // own MFMA time. Now every load of unit g + nsplit is issued while unit g is multiplied: the raw registers of an operand
    // are re-requested as soon as its fragments are split (one register set, >= 27 MFMAs of cover per load), through buffer
    // descriptors whose out-of-range offset (0x80000000) returns the zero padding -- no exec-masked branch around a load,
    // no select behind it. Same values, same order of products and additions: bit-identical partials.
    const auto rsa = __builtin_amdgcn_make_buffer_rsrc((void *)dy, 0, (int)((size_t)nb * cout * R3 * 4), 0x00020000);
    const auto rsx = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)((size_t)nb * cin * R3 * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    float fa[8], f[3][8], lf[3], rt[3];
    auto ld8 = [&](__amdgpu_buffer_rsrc_t rs, unsigned off, float (&o)[8]) {
      const f32x4w v0 = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
      const f32x4w v1 = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 16, 0));
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = v0[i], o[4 + i] = v1[i];
    };
    auto issue_a = [&](int g) {
      const int b = g / KG, q = (g % KG) * 16 + 8 * khalf;
      const unsigned off = (cok && g < total) ? (unsigned)(((size_t)b * cout + co) * R3 + q) * 4u : OOB;
      ld8(rsa, off, fa);
    };
    auto issue_row = [&](int g, int kh) {
      const int b = g / KG, q = (g % KG) * 16 + 8 * khalf;
      const int d = q / (R * R), h = (q / R) % R, w = q % R;  // w is a multiple of 8
      const int nd = d + kd - 1, nh = h + kh - 1;
      const bool rok = cik && g < total && (unsigned)nd < (unsigned)R && (unsigned)nh < (unsigned)R;
      const unsigned off = rok ? (unsigned)(((size_t)b * cin + ci) * R3 + ((size_t)nd * R + nh) * R + w) * 4u : OOB;
      ld8(rsx, off, f[kh]);
      lf[kh] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsx, (rok && w > 0) ? off - 4u : OOB, 0, 0));
      rt[kh] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsx, (rok && w + 8 < R) ? off + 32u : OOB, 0, 0));
    };
    issue_a(split);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) issue_row(split, kh);
    for (int g = split; g < total; g += nsplit) {
      const int gn = g + nsplit;
      u32x4 a[NTERM];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned t[NTERM];
        split_pair<NTERM>(fa[2 * i], fa[2 * i + 1], t);
#pragma unroll
        for (int s = 0; s < NTERM; ++s) a[s][i] = t[s];
      }
      if (want_bias) bsum += ((fa[0] + fa[1]) + (fa[2] + fa[3])) + ((fa[4] + fa[5]) + (fa[6] + fa[7]));
      __builtin_amdgcn_sched_barrier(0);
      issue_a(gn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        u32x4 fm[NTERM], fz[NTERM], fp[NTERM];  // kw = 0 (dw = -1), 1, 2
        unsigned t[NTERM];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          split_pair<NTERM>(f[kh][2 * i], f[kh][2 * i + 1], t);
#pragma unroll
          for (int s = 0; s < NTERM; ++s) fz[s][i] = t[s];
        }
        split_pair<NTERM>(lf[kh], f[kh][0], t);
#pragma unroll
        for (int s = 0; s < NTERM; ++s) fm[s][0] = t[s];
#pragma unroll
        for (int i = 1; i < 4; ++i) {
          split_pair<NTERM>(f[kh][2 * i - 1], f[kh][2 * i], t);
#pragma unroll
          for (int s = 0; s < NTERM; ++s) fm[s][i] = t[s], fp[s][i - 1] = t[s];
        }
        split_pair<NTERM>(f[kh][7], rt[kh], t);
#pragma unroll
        for (int s = 0; s < NTERM; ++s) fp[s][3] = t[s];
        __builtin_amdgcn_sched_barrier(0);
        issue_row(gn, kh);
        __builtin_amdgcn_sched_barrier(0);
        mfma_products<NTERM>(acc[kh * 3 + 0], a, fm);
        mfma_products<NTERM>(acc[kh * 3 + 1], a, fz);
        mfma_products<NTERM>(acc[kh * 3 + 2], a, fp);
      }
    }
  } else {  // R == 4: a fragment spans two h-rows of four voxels; gathered element-wise (64-voxel grids only)
    for (int g = split; g < total; g += nsplit) {
      const int b = g / KG, q = (g % KG) * 16 + 8 * khalf;
      float fa[8];
      if (cok) load8(dy + ((size_t)b * cout + co) * R3 + q, true, fa);
      else {
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = 0.0f;
      }
      u32x4 a[NTERM];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned t[NTERM];
        split_pair<NTERM>(fa[2 * i], fa[2 * i + 1], t);
#pragma unroll
        for (int s = 0; s < NTERM; ++s) a[s][i] = t[s];
      }
      if (want_bias) bsum += ((fa[0] + fa[1]) + (fa[2] + fa[3])) + ((fa[4] + fa[5]) + (fa[6] + fa[7]));
      const float *xrow = x + ((size_t)b * cin + (cik ? ci : 0)) * R3;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          float f[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int v = q + i;
            const int nd = v / (R * R) + kd - 1, nh = (v / R) % R + kh - 1, nw = v % R + kw - 1;
            const bool ok = cik && (unsigned)nd < (unsigned)R && (unsigned)nh < (unsigned)R && (unsigned)nw < (unsigned)R;
            f[i] = ok ? xrow[((size_t)nd * R + nh) * R + nw] : 0.0f;
          }
          u32x4 fb[NTERM];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            unsigned t[NTERM];
            split_pair<NTERM>(f[2 * i], f[2 * i + 1], t);
#pragma unroll
            for (int s = 0; s < NTERM; ++s) fb[s][i] = t[s];
          }
          mfma_products<NTERM>(acc[kh * 3 + kw], a, fb);
        }
    }
  }
  float *po = part + (size_t)split * ((size_t)cout * cin * 27 + cout);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int oc = co_t + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      if (oc < cout && cik) po[((size_t)(kd * 9 + t) * cout + oc) * cin + ci] = acc[t][r];
    }
  if (want_bias) {
    bsum += __shfl_xor(bsum, 32);
    if (khalf == 0 && cok) po[(size_t)cout * cin * 27 + co] = bsum;
  }
}

// ---- round 5: the same GEMM with its operands staged through LDS ----------------------------------------------------------------
// The register form above gives every lane its own 32-byte run of its own channel row: 64 lanes of a load touch 64 different cache
// lines, and the CU's address path processes ~one line per clock -- 14 such loads per K unit cost ~900 clocks of that path per wave
// against 864 matrix clocks, for four to eight waves per CU: the r = 32 launches ran at 17-21 % of their own MFMA time whatever the
// loads' latency. Here a K unit is 32 voxels (whole grid rows: one at r = 32, two at r = 16, four at r = 8), the workgroup brings
// the unit's dY tile [64 co][32] and the three kh-shifted X tiles [64 ci][32] (zero rows outside the grid: out-of-range buffer
// offsets) with COALESCED 16-byte loads (8 lanes per 128-byte row segment: 8 lines per load instead of 64), splits every element
// into its bf16 terms ONCE (the register form split each one in two waves and three kw alignments) and writes the term planes to
// LDS; a wave's fragments are one ds_read_b128 per plane, the kw = 0 / 2 fragments are the kw = 1 one shifted by a bf16 with
// v_alignbit and one neighbour word (zero at a row end). The next unit's loads are in flight while a unit is multiplied.
// Row pitch 80 bytes: the 16 lanes of a service group of a ds_read_b128 fall on all eight 16-byte bank groups twice.
// Same bf16 terms, products and per-accumulator order of K units within a split as the register form (the split of K over the
// workgroups differs: partial sums round differently, deterministically).
template <int R, int NTERM>
__global__ __launch_bounds__(256, 2) void conv3d_k3_wgrad_lds_kernel(int nb, int cin, int cout, int nsplit,
                                                                     const float *__restrict__ x,
                                                                     const float *__restrict__ dy,
                                                                     float *__restrict__ part,
                                                                     float *__restrict__ bpart) {
  static_assert((R >= 8 && 32 % R == 0) || R == 32, "a K unit is whole grid rows");
  constexpr int R3 = R * R * R, UPS = R3 / 32;  // units per sample
  constexpr int P = 80, ROWS = 256, PB = ROWS * P + 32;  // row pitch (bytes), rows per plane (64 dY + 3 x 64 X), plane bytes
  __shared__ __attribute__((aligned(16))) unsigned char lds[NTERM * PB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, khalf = lane >> 5;
  const int split = blockIdx.x, kd = blockIdx.z;
  const int ncit = (cin + 63) / 64;
  const int co_blk = (blockIdx.y / ncit) * 64, ci_blk = (blockIdx.y % ncit) * 64;
  const int co_t = co_blk + (wave & 1) * 32, ci_t = ci_blk + (wave >> 1) * 32;
  const int ci = ci_t + l31;
  const bool cik = ci < cin;
  const bool do_bias = bpart && kd == 0 && (blockIdx.y % ncit) == 0;
  const int srow = tid >> 3, sp = tid & 7;  // staging role: rows srow, srow + 32 of every tile; piece sp (4 voxels)
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  float bs[2] = {0.0f, 0.0f};
  const int total = nb * UPS;
  const auto rsa = __builtin_amdgcn_make_buffer_rsrc((void *)dy, 0, (int)((size_t)nb * cout * R3 * 4), 0x00020000);
  const auto rsx = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)((size_t)nb * cin * R3 * 4), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  f32x4w ra[2], rb[3][2];
  auto issue = [&](int u) {
    const int b = u / UPS, v = (u % UPS) * 32 + 4 * sp;
    const int d = v / (R * R), h = (v / R) % R, w = v % R;
    const bool in = u < total;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int co = co_blk + srow + 32 * j;
      ra[j] = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(
                                             rsa, (in && co < cout) ? (unsigned)(((size_t)b * cout + co) * R3 + v) * 4u : OOB, 0, 0));
    }
    const int nd = d + kd - 1;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int nh = h + kh - 1;
      const bool rok = in && (unsigned)nd < (unsigned)R && (unsigned)nh < (unsigned)R;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = ci_blk + srow + 32 * j;
        rb[kh][j] = __builtin_bit_cast(
            f32x4w, __builtin_amdgcn_raw_buffer_load_b128(
                        rsx, (rok && c < cin) ? (unsigned)(((size_t)b * cin + c) * R3 + ((size_t)nd * R + nh) * R + w) * 4u : OOB, 0, 0));
      }
    }
  };
  auto put = [&](const f32x4w &v, int row) {  // four voxels of one row -> NTERM planes of 8 bytes at piece sp
    unsigned t0[NTERM], t1[NTERM];
    split_pair<NTERM>(v[0], v[1], t0);
    split_pair<NTERM>(v[2], v[3], t1);
#pragma unroll
    for (int s = 0; s < NTERM; ++s) {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      *(u32x2 *)(lds + s * PB + 16 + row * P + sp * 8) = u32x2{t0[s], t1[s]};
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      put(ra[j], srow + 32 * j);
      if (do_bias) bs[j] += (ra[j][0] + ra[j][1]) + (ra[j][2] + ra[j][3]);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) put(rb[kh][j], 64 + kh * 64 + srow + 32 * j);
    }
  };
  issue(split);
  for (int u = split; u < total; u += nsplit) {
    __syncthreads();  // the previous unit's fragments are read
    stash();
    __syncthreads();
    issue(u + nsplit);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int col = 16 * (2 * ks + khalf);  // byte offset of the lane's 8 bf16 inside a row
      const int wv = (16 * ks + 8 * khalf) % R;  // its first voxel's w
      const bool wl = wv > 0, wr = wv + 8 < R;
      u32x4 a[NTERM];
#pragma unroll
      for (int s = 0; s < NTERM; ++s) a[s] = *(const u32x4 *)(lds + s * PB + 16 + ((wave & 1) * 32 + l31) * P + col);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        u32x4 fm[NTERM], fz[NTERM], fp[NTERM];  // kw = 0 (dw = -1), 1, 2
#pragma unroll
        for (int s = 0; s < NTERM; ++s) {
          const unsigned char *base = lds + s * PB + 16 + (64 + kh * 64 + (wave >> 1) * 32 + l31) * P + col;
          fz[s] = *(const u32x4 *)base;
          const unsigned lw = *(const unsigned *)(base - 4), rw = *(const unsigned *)(base + 16);
          const unsigned left = wl ? lw >> 16 : 0u, right = wr ? rw << 16 : 0u;
          fm[s][0] = (fz[s][0] << 16) | left;
#pragma unroll
          for (int i = 1; i < 4; ++i) fm[s][i] = __builtin_amdgcn_alignbit(fz[s][i], fz[s][i - 1], 16);
#pragma unroll
          for (int i = 0; i < 3; ++i) fp[s][i] = __builtin_amdgcn_alignbit(fz[s][i + 1], fz[s][i], 16);
          fp[s][3] = (fz[s][3] >> 16) | right;
        }
        mfma_products<NTERM>(acc[kh * 3 + 0], a, fm);
        mfma_products<NTERM>(acc[kh * 3 + 1], a, fz);
        mfma_products<NTERM>(acc[kh * 3 + 2], a, fp);
      }
    }
  }
  float *po = part + (size_t)split * ((size_t)cout * cin * 27 + cout);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int oc = co_t + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      if (oc < cout && cik) po[((size_t)(kd * 9 + t) * cout + oc) * cin + ci] = acc[t][r];
    }
  if (do_bias) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float v = bs[j];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      const int co = co_blk + srow + 32 * j;
      if (sp == 0 && co < cout) po[(size_t)cout * cin * 27 + co] = v;
    }
  }
}

template <int NTERM>
__global__ __launch_bounds__(256, 2) void pointwise_wgrad_bf16_kernel(int nb, int cin, int cout, int npos, int nsplit,
                                                                      const float *__restrict__ x,
                                                                      const float *__restrict__ dy,
                                                                      float *__restrict__ part,
                                                                      float *__restrict__ bpart) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, khalf = lane >> 5;
  const int split = blockIdx.x;
  const int ncit = (cin + 63) / 64;
  const int co_t = (blockIdx.y / ncit) * 64 + (wave & 1) * 32, ci_t = (blockIdx.y % ncit) * 64 + (wave >> 1) * 32;
  const int co = co_t + l31, ci = ci_t + l31;
  const bool cok = co < cout, cik = ci < cin;
  const bool want_bias = bpart && (blockIdx.y % ncit) == 0 && (wave >> 1) == 0;
  const bool vec = (npos & 3) == 0;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float bsum = 0.0f;
  const int KG = (npos + 15) / 16;
  const int total = nb * KG;
  auto consume = [&](const float (&fa)[8], const float (&fb)[8]) {
    if (want_bias) bsum += ((fa[0] + fa[1]) + (fa[2] + fa[3])) + ((fa[4] + fa[5]) + (fa[6] + fa[7]));
    u32x4 a[NTERM], bq[NTERM];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned t[NTERM];
      split_pair<NTERM>(fa[2 * i], fa[2 * i + 1], t);
#pragma unroll
      for (int s = 0; s < NTERM; ++s) a[s][i] = t[s];
      split_pair<NTERM>(fb[2 * i], fb[2 * i + 1], t);
#pragma unroll
      for (int s = 0; s < NTERM; ++s) bq[s][i] = t[s];
    }
    mfma_products<NTERM>(acc, a, bq);
  };
  // rows of 16-byte pieces inside 2 GB tensors (every layer of the networks): the loads of the next three K units are in
  // flight while one is multiplied (round 5: the loop used to issue two loads, wait a full memory round trip, run three
  // MFMAs). Buffer descriptors: a piece past the row end, a channel past the tile, a unit past the end has the out-of-range
  // offset and reads zeros -- adding nothing. Same order of products and additions as the plain loop below.
  const bool piped = (npos & 3) == 0 && (size_t)nb * (cout > cin ? cout : cin) * npos * 4 < (1ull << 31) &&
                     (((size_t)x | (size_t)dy) & 15) == 0;
  if (piped) {
    const auto rsa = __builtin_amdgcn_make_buffer_rsrc((void *)dy, 0, (int)((size_t)nb * cout * npos * 4), 0x00020000);
    const auto rsb = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)((size_t)nb * cin * npos * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    float ra[4][8], rb[4][8];
    auto issue = [&](int g, float (&fa)[8], float (&fb)[8]) {
      const int b = g / KG, p = (g % KG) * 16 + 8 * khalf;
      const bool in = g < total;
      const unsigned oa = (unsigned)((b * cout + co) * npos + p) * 4u, ob = (unsigned)((b * cin + ci) * npos + p) * 4u;
      const bool h0 = in && p + 4 <= npos, h1 = in && p + 8 <= npos;
      const f32x4w a0 = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rsa, (h0 && cok) ? oa : OOB, 0, 0));
      const f32x4w a1 = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rsa, (h1 && cok) ? oa + 16u : OOB, 0, 0));
      const f32x4w b0 = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rsb, (h0 && cik) ? ob : OOB, 0, 0));
      const f32x4w b1 = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rsb, (h1 && cik) ? ob + 16u : OOB, 0, 0));
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = a0[i], fa[4 + i] = a1[i], fb[i] = b0[i], fb[4 + i] = b1[i];
    };
    int g = split;
#pragma unroll
    for (int j = 0; j < 3; ++j) issue(g + j * nsplit, ra[j], rb[j]);
    while (g < total) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        issue(g + 3 * nsplit, ra[(j + 3) & 3], rb[(j + 3) & 3]);
        __builtin_amdgcn_sched_barrier(0);
        consume(ra[j], rb[j]);
        __builtin_amdgcn_sched_barrier(0);
        g += nsplit;
        if (g >= total) break;
      }
    }
  } else {
    for (int g = split; g < total; g += nsplit) {
      const int b = g / KG, p = (g % KG) * 16 + 8 * khalf;
      float fa[8], fb[8];
      const float *sa = dy + ((size_t)b * cout + (cok ? co : 0)) * npos + p;
      const float *sb = x + ((size_t)b * cin + (cik ? ci : 0)) * npos + p;
      if (p + 8 <= npos) {
        load8(sa, vec, fa);
        load8(sb, vec, fb);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          fa[i] = p + i < npos ? sa[i] : 0.0f;
          fb[i] = p + i < npos ? sb[i] : 0.0f;
        }
      }
      if (!cok) {
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = 0.0f;
      }
      if (!cik) {
#pragma unroll
        for (int i = 0; i < 8; ++i) fb[i] = 0.0f;
      }
      consume(fa, fb);
    }
  }
  float *po = part + (size_t)split * ((size_t)cout * cin + cout);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int oc = co_t + (r & 3) + 8 * (r >> 2) + 4 * khalf;
    if (oc < cout && cik) po[(size_t)oc * cin + ci] = acc[r];
  }
  if (want_bias) {
    bsum += __shfl_xor(bsum, 32);
    if (khalf == 0 && cok) po[(size_t)cout * cin + co] = bsum;
  }
}

// The 1x1 weight gradient with its operands through LDS (round 5; see conv3d_k3_wgrad_lds_kernel): a K unit is 64 positions of
// one sample, dY [64 co][64] and X [64 ci][64] arrive by coalesced 16-byte loads (16 lanes per 256-byte row segment), are split into
// their bf16 terms once and written as term planes (row pitch 144 bytes); four k-steps x three products per unit and wave. Rows of
// 16-byte pieces only (npos % 4 == 0: every layer of the networks); pieces past the row end read zeros.
template <int NTERM>
__global__ __launch_bounds__(256, 2) void pointwise_wgrad_lds_kernel(int nb, int cin, int cout, int npos, int nsplit,
                                                                     const float *__restrict__ x,
                                                                     const float *__restrict__ dy,
                                                                     float *__restrict__ part,
                                                                     float *__restrict__ bpart) {
  constexpr int P = 144, PB = 128 * P;  // row pitch (bytes); plane = 64 dY rows + 64 X rows
  __shared__ __attribute__((aligned(16))) unsigned char lds[NTERM * PB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, khalf = lane >> 5;
  const int split = blockIdx.x;
  const int ncit = (cin + 63) / 64;
  const int co_blk = (blockIdx.y / ncit) * 64, ci_blk = (blockIdx.y % ncit) * 64;
  const int co_t = co_blk + (wave & 1) * 32, ci_t = ci_blk + (wave >> 1) * 32;
  const int ci = ci_t + l31;
  const bool cik = ci < cin;
  const bool do_bias = bpart && (blockIdx.y % ncit) == 0;
  const int srow = tid >> 4, sp = tid & 15;  // staging role: rows srow + 16 j of both tiles, piece sp (4 positions)
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float bs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const int UPS = (npos + 63) / 64, total = nb * UPS;
  const auto rsa = __builtin_amdgcn_make_buffer_rsrc((void *)dy, 0, (int)((size_t)nb * cout * npos * 4), 0x00020000);
  const auto rsx = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)((size_t)nb * cin * npos * 4), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  f32x4w ra[4], rb[4];
  auto issue = [&](int u) {
    const int b = u / UPS, pos = (u % UPS) * 64 + 4 * sp;
    const bool in = u < total && pos + 4 <= npos;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = co_blk + srow + 16 * j, c = ci_blk + srow + 16 * j;
      ra[j] = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(
                                             rsa, (in && co < cout) ? (unsigned)((b * cout + co) * npos + pos) * 4u : OOB, 0, 0));
      rb[j] = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(
                                             rsx, (in && c < cin) ? (unsigned)((b * cin + c) * npos + pos) * 4u : OOB, 0, 0));
    }
  };
  auto put = [&](const f32x4w &v, int row) {
    unsigned t0[NTERM], t1[NTERM];
    split_pair<NTERM>(v[0], v[1], t0);
    split_pair<NTERM>(v[2], v[3], t1);
#pragma unroll
    for (int s = 0; s < NTERM; ++s) {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      *(u32x2 *)(lds + s * PB + row * P + sp * 8) = u32x2{t0[s], t1[s]};
    }
  };
  issue(split);
  for (int u = split; u < total; u += nsplit) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      put(ra[j], srow + 16 * j);
      if (do_bias) bs[j] += (ra[j][0] + ra[j][1]) + (ra[j][2] + ra[j][3]);
      put(rb[j], 64 + srow + 16 * j);
    }
    __syncthreads();
    issue(u + nsplit);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int col = 32 * ks + 16 * khalf;
      u32x4 a[NTERM], bq[NTERM];
#pragma unroll
      for (int s = 0; s < NTERM; ++s) {
        a[s] = *(const u32x4 *)(lds + s * PB + ((wave & 1) * 32 + l31) * P + col);
        bq[s] = *(const u32x4 *)(lds + s * PB + (64 + (wave >> 1) * 32 + l31) * P + col);
      }
      mfma_products<NTERM>(acc, a, bq);
    }
  }
  float *po = part + (size_t)split * ((size_t)cout * cin + cout);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int oc = co_t + (r & 3) + 8 * (r >> 2) + 4 * khalf;
    if (oc < cout && cik) po[(size_t)oc * cin + ci] = acc[r];
  }
  if (do_bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = bs[j];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8);
      const int co = co_blk + srow + 16 * j;
      if (sp == 0 && co < cout) po[(size_t)cout * cin + co] = v;
    }
  }
}

// out = sum_s part[s], s ascending (deterministic). A partial row is [ntap][cout*cin] weights | cout bias sums; the
// weights leave in the parameter's layout dw[co][ci][ntap] (coalesced reads of the partials, one scattered write).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(int nsplit, int ntap, size_t cc, size_t nbias,
                                                           const float *__restrict__ part, float *__restrict__ dw,
                                                           float *__restrict__ db) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t n = cc * ntap, row = n + nbias;
  if (i >= (db ? row : n)) return;
  // same ascending order (same bits); eight partials in flight per thread -- the plain loop was one L2 round trip per
  // addend (23 us per launch, 52 launches per training step)
  float s = 0.0f;
  int k = 0;
  for (; k + 8 <= nsplit; k += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(k + j) * row + i];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  for (; k < nsplit; ++k) s += part[(size_t)k * row + i];
  if (i < n) dw[(i % cc) * ntap + i / cc] = s;
  else db[i - n] = s;
}

static void wg_reduce(int nsplit, int ntap, size_t cc, size_t nbias, const float *part, float *dw, float *db, hipStream_t s) {
  // (round 5: noting these ~50 reductions per step and performing them in batched launches behind backward measured SLOWER --
  //  12.9 -> 13.1 ms per config-3 step, profiles/r05b_defer_reduce_ab.txt: the partials are read back from HBM instead of L2)
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv((long)(cc * ntap + nbias), 256)), dim3(256), 0, s, nsplit, ntap, cc, nbias, part,
                     dw, db);
}

// math: 0 = bf16x3 (default; "high" matmul precision, train.py:221), 1 = bf16x6 (fp32-faithful), 2 = exact fp32 MFMA
static int conv_wgrad_units(int r, int math) {  // K units of the form that runs: bricks (fp32), 32 voxels (LDS form), 16 (r = 4)
  return math == 2 ? (r >= 8 ? (r / 4) * (r / 8) * (r / 8) : 1) : r >= 8 ? r * r * r / 32 : r * r * r / 16;
}
static int conv_wgrad_wgs_per_split(int cin, int cout, int math) {
  return math == 2 ? ((cout + WG_COT - 1) / WG_COT) * ((cin + WG_CIT - 1) / WG_CIT)
                   : ((cout + 63) / 64) * ((cin + 63) / 64) * 3;
}
// K-splits of a launch: enough workgroups to fill the chip (`target`), but the partials (written once, read once by
// the reduction) stay below ~48 MB -- wide layers on small grids have few K units and large outputs
static int wgrad_nsplit(long units, int wgs_per_split, int target, size_t out_floats) {
  // `target` = the workgroup slots of the chip for this kernel (256 CUs x 2 for the bf16 forms): the largest split count whose
  // grid still fits them at once -- one more workgroup than slots is a second round for one CU and the launch waits for it
  // (round 5: 384 workgroups of the r = 32 layers on 256 CUs left half the CUs with two and half with one)
  long s = target >= wgs_per_split ? target / wgs_per_split : 1;
  const long cap = (long)((48u << 20) / (out_floats * sizeof(float)));
  if (s > cap) s = cap;
  if (s > units) s = units;
  if (s > 192) s = 192;
  return s < 1 ? 1 : (int)s;
}

// The bf16 forms address both tensors through 32-bit buffer offsets (0x80000000 = "outside": the zero padding). A batch whose
// larger operand reaches 2 GiB therefore runs the exact-fp32 form (64-bit addressing; slower, never less accurate) instead of
// being refused mid-backward -- decided here, by the launcher AND its workspace helper alike.
static int conv_wgrad_math_that_runs(int b, int cin, int cout, int r, int math) {
  return (math != 2 && (size_t)b * (cin > cout ? cin : cout) * r * r * r * 4 >= (1ull << 31)) ? 2 : math;
}

extern "C" size_t p2pb_conv3d_k3_wgrad_ws_floats(int b, int cin, int cout, int r, int math) {
  math = conv_wgrad_math_that_runs(b, cin, cout, r, math);
  const int ns = wgrad_nsplit((long)b * conv_wgrad_units(r, math), conv_wgrad_wgs_per_split(cin, cout, math),
                              512, (size_t)cout * cin * 27);
  return (size_t)ns * ((size_t)cout * cin * 27 + cout);
}

template <int R, int TD, int TH, int TW>
static void conv_wgrad_launch_fp32(int b, int cin, int cout, int ns, const float *x, const float *dy, float *ws,
                                   bool bias, hipStream_t s) {
  using G = WBrick<TD, TH, TW>;
  const int cot = (cout + WG_COT - 1) / WG_COT, cit = (cin + WG_CIT - 1) / WG_CIT;
  const size_t lds = (size_t)(WG_COT * G::PA + WG_CIT * G::PB) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)conv3d_k3_wgrad_kernel<R, TD, TH, TW>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<R, TD, TH, TW>), dim3(ns, cot, cit), dim3(256), lds, s, b, cin, cout, ns,
                     x, dy, ws, bias ? ws : nullptr);
}

template <int R>
static void conv_wgrad_launch_bf16(int b, int cin, int cout, int ns, int math, const float *x, const float *dy,
                                   float *ws, bool bias, hipStream_t s) {
  const dim3 grid(ns, ((cout + 63) / 64) * ((cin + 63) / 64), 3);
  if constexpr (R >= 8) {  // operands through LDS (tensors below 2 GB: the launcher checked)
    if (math == 1)
      hipLaunchKernelGGL((conv3d_k3_wgrad_lds_kernel<R, 3>), grid, dim3(256), 0, s, b, cin, cout, ns, x, dy, ws,
                         bias ? ws : nullptr);
    else
      hipLaunchKernelGGL((conv3d_k3_wgrad_lds_kernel<R, 2>), grid, dim3(256), 0, s, b, cin, cout, ns, x, dy, ws,
                         bias ? ws : nullptr);
    return;
  }
  if (math == 1)
    hipLaunchKernelGGL((conv3d_k3_wgrad_bf16_kernel<R, 3>), grid, dim3(256), 0, s, b, cin, cout, ns, x, dy, ws,
                       bias ? ws : nullptr);
  else
    hipLaunchKernelGGL((conv3d_k3_wgrad_bf16_kernel<R, 2>), grid, dim3(256), 0, s, b, cin, cout, ns, x, dy, ws,
                       bias ? ws : nullptr);
}

extern "C" int p2pb_conv3d_k3_wgrad(int b, int cin, int cout, int r, const float *x, const float *dy, float *dw,
                                    float *db, float *ws, int math, void *stream) {
  if (b <= 0 || cin <= 0 || cout <= 0 || !x || !dy || !dw || !ws || math < 0 || math > 2) return P2PB_EINVAL;
  if (r != 4 && r != 8 && r != 16 && r != 32) return P2PB_EINVAL;
  math = conv_wgrad_math_that_runs(b, cin, cout, r, math);
  hipStream_t s = (hipStream_t)stream;
  const int ns = wgrad_nsplit((long)b * conv_wgrad_units(r, math), conv_wgrad_wgs_per_split(cin, cout, math),
                              512, (size_t)cout * cin * 27);
  const bool bias = db != nullptr;
  if (math == 2) {
    switch (r) {
      case 32: conv_wgrad_launch_fp32<32, 4, 8, 8>(b, cin, cout, ns, x, dy, ws, bias, s); break;
      case 16: conv_wgrad_launch_fp32<16, 4, 8, 8>(b, cin, cout, ns, x, dy, ws, bias, s); break;
      case 8: conv_wgrad_launch_fp32<8, 4, 8, 8>(b, cin, cout, ns, x, dy, ws, bias, s); break;
      default: conv_wgrad_launch_fp32<4, 4, 4, 4>(b, cin, cout, ns, x, dy, ws, bias, s); break;
    }
  } else {
    switch (r) {
      case 32: conv_wgrad_launch_bf16<32>(b, cin, cout, ns, math, x, dy, ws, bias, s); break;
      case 16: conv_wgrad_launch_bf16<16>(b, cin, cout, ns, math, x, dy, ws, bias, s); break;
      case 8: conv_wgrad_launch_bf16<8>(b, cin, cout, ns, math, x, dy, ws, bias, s); break;
      default: conv_wgrad_launch_bf16<4>(b, cin, cout, ns, math, x, dy, ws, bias, s); break;
    }
  }
  wg_reduce(ns, 27, (size_t)cout * cin, (size_t)cout, ws, dw, db, s);
  return p2pb_launch_status();
}

// does the bf16 weight gradient of a 1x1 layer take the LDS form? (rows of 16-byte pieces, 32-bit buffer offsets)
static bool pw_wgrad_lds(int b, int cin, int cout, int npos) {
  return (npos & 3) == 0 && (size_t)b * (cin > cout ? cin : cout) * npos * 4 < (1ull << 31);
}
static int pw_wgrad_units(int b, int cin, int cout, int npos, int math) {  // K units per sample of the form that runs
  return math == 2 ? (npos + PW_CH - 1) / PW_CH : pw_wgrad_lds(b, cin, cout, npos) ? (npos + 63) / 64 : (npos + 15) / 16;
}

extern "C" size_t p2pb_pointwise_wgrad_ws_floats(int b, int cin, int cout, int npos, int math) {
  const int tiles = ((cout + 63) / 64) * ((cin + 63) / 64);
  const int ns = wgrad_nsplit((long)b * pw_wgrad_units(b, cin, cout, npos, math), tiles, 512, (size_t)cout * cin);
  return (size_t)ns * ((size_t)cout * cin + cout);
}

extern "C" int p2pb_pointwise_wgrad(int b, int cin, int cout, int npos, const float *x, const float *dy, float *dw,
                                    float *db, float *ws, int math, void *stream) {
  if (b <= 0 || cin <= 0 || cout <= 0 || npos <= 0 || !x || !dy || !dw || !ws || math < 0 || math > 2)
    return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int cot = (cout + 63) / 64, cit = (cin + 63) / 64;
  const int ns = wgrad_nsplit((long)b * pw_wgrad_units(b, cin, cout, npos, math), cot * cit, 512, (size_t)cout * cin);
  const bool bias = db != nullptr;
  if (math == 2) {
    const size_t lds = (size_t)(2 * 64 * (PW_CH + 1)) * sizeof(float);
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void *)pointwise_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
      attr = true;
    }
    hipLaunchKernelGGL(pointwise_wgrad_kernel, dim3(ns, cot, cit), dim3(256), lds, s, b, cin, cout, npos, ns, x, dy,
                       ws, bias ? ws : nullptr);
  } else if (pw_wgrad_lds(b, cin, cout, npos)) {
    if (math == 1)
      hipLaunchKernelGGL(pointwise_wgrad_lds_kernel<3>, dim3(ns, cot * cit), dim3(256), 0, s, b, cin, cout, npos, ns, x, dy, ws,
                         bias ? ws : nullptr);
    else
      hipLaunchKernelGGL(pointwise_wgrad_lds_kernel<2>, dim3(ns, cot * cit), dim3(256), 0, s, b, cin, cout, npos, ns, x, dy, ws,
                         bias ? ws : nullptr);
  } else if (math == 1) {
    hipLaunchKernelGGL(pointwise_wgrad_bf16_kernel<3>, dim3(ns, cot * cit), dim3(256), 0, s, b, cin, cout, npos, ns, x,
                       dy, ws, bias ? ws : nullptr);
  } else {
    hipLaunchKernelGGL(pointwise_wgrad_bf16_kernel<2>, dim3(ns, cot * cit), dim3(256), 0, s, b, cin, cout, npos, ns, x,
                       dy, ws, bias ? ws : nullptr);
  }
  wg_reduce(ns, 1, (size_t)cout * cin, (size_t)cout, ws, dw, db, s);
  return p2pb_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of a PVConv's FIRST convolution, sparse in K (round 5; review r4 item 4, first step).
// Its operand X is the voxelised point features: zero outside the occupied voxels -- 2048 points in a 32^3 grid occupy <= 6 % of
// it -- so   dW[co][ci][tap] = sum_b sum_{v' occupied} dY[b, co, v' - off(tap)] * X[b, ci, v']   has K = (occupied voxels), not r^3.
// The dense kernel above spends 399 us per launch on the r = 32 layers of the config-3 step (1.6 ms of its 14.8 ms) multiplying
// zeros. Here: (1) the occupied voxels of every sample are listed (ascending) from avg_voxelize's counts; (2) X is gathered at
// them into Xocc[b][k][ci], dY is transposed once to voxel-major dYt[b][v][co] (a voxel's channels contiguous: the 27 shifted
// reads of an occupied voxel are 27 contiguous rows) and its bias sums are taken on the way; (3) workgroup (sample x K-split, tap,
// 64 x 64 channel tile) walks its occupied voxels 32 at a time -- rows of dYt (zero outside the grid) and of Xocc through LDS -- and
// accumulates a 4 x 4 block per thread in exact fp32 FMAs, in voxel order; (4) wgrad_reduce_kernel adds the partials in split order.
// Deterministic, exact fp32 products (the dense bf16x3 form keeps 16 + 8 bits per operand).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void wg_occ_list_kernel(int r3, int n, const int *__restrict__ cnt, int *__restrict__ occ,
                                                           int *__restrict__ nocc) {
  __shared__ int sc[1024];
  const int b = blockIdx.x, t = threadIdx.x;
  const int *c = cnt + (size_t)b * r3;
  const int per = (r3 + 1023) / 1024, beg = t * per, end = min(beg + per, r3);
  int k = 0;
  for (int v = beg; v < end; ++v) k += c[v] > 0;
  sc[t] = k;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // inclusive scan (Hillis-Steele)
    const int y = t >= d ? sc[t - d] : 0;
    __syncthreads();
    sc[t] += y;
    __syncthreads();
  }
  int pos = sc[t] - k;
  if (t == 1023) nocc[b] = min(sc[t], n);
  for (int v = beg; v < end; ++v)
    if (c[v] > 0) {
      if (pos < n) occ[(size_t)b * n + pos] = v;
      ++pos;
    }
}

__global__ __launch_bounds__(256) void wg_xocc_kernel(int cin, int r3, int n, const float *__restrict__ x, const int *__restrict__ occ,
                                                      const int *__restrict__ nocc, float *__restrict__ xocc) {
  const int b = blockIdx.y;
  const size_t total = (size_t)nocc[b] * cin;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int k = (int)(e / cin), ci = (int)(e % cin);
    xocc[((size_t)b * n + k) * cin + ci] = x[((size_t)b * cin + ci) * r3 + occ[(size_t)b * n + k]];
  }
}

// dy f32[b, co, r3] -> dyt f32[b, r3, co] (32 x 32 LDS tiles)
__global__ __launch_bounds__(256) void wg_dyt_kernel(int co, int r3, const float *__restrict__ dy, float *__restrict__ dyt) {
  __shared__ float t[32][33];
  const int b = blockIdx.z, v0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float *src = dy + (size_t)b * co * r3;
  float *dst = dyt + (size_t)b * co * r3;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cc = c0 + ty + 8 * k, vv = v0 + tx;
    t[ty + 8 * k][tx] = (cc < co && vv < r3) ? src[(size_t)cc * r3 + vv] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int vv = v0 + ty + 8 * k, cc = c0 + tx;
    if (cc < co && vv < r3) dst[(size_t)vv * co + cc] = t[tx][ty + 8 * k];
  }
}

// bias sums of sample b into the partial row of its first K-split, zeros into its other splits (fixed order: 256 strided
// partials + tree)
__global__ __launch_bounds__(256) void wg_bias_rows_kernel(int co, int r3, int S, size_t row, size_t nw, const float *__restrict__ dy,
                                                           float *__restrict__ part) {
  __shared__ float red[256];
  const int c = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const float *src = dy + ((size_t)b * co + c) * r3;
  float s = 0.0f;
  for (int v = t; v < r3; v += 256) s += src[v];
  red[t] = s;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (t < d) red[t] += red[t + d];
    __syncthreads();
  }
  if (t < S) part[((size_t)b * S + t) * row + nw + c] = t == 0 ? red[0] : 0.0f;
}

__global__ __launch_bounds__(256) void conv3d_k3_wgrad_occ_kernel(int r, int n, int cin, int cout, int S, size_t row,
                                                                  const float *__restrict__ dyt, const float *__restrict__ xocc,
                                                                  const int *__restrict__ occ, const int *__restrict__ nocc,
                                                                  float *__restrict__ part) {
  __shared__ __attribute__((aligned(16))) float As[32][64];
  __shared__ __attribute__((aligned(16))) float Bs[32][64];
  __shared__ int su[32];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const int b = blockIdx.x / S, s = blockIdx.x % S, tap = blockIdx.y;
  const int cit = (cin + 63) / 64;
  const int co0 = (blockIdx.z / cit) * 64, ci0 = (blockIdx.z % cit) * 64;
  const int r3 = r * r * r;
  const int od = tap / 9 - 1, oh = (tap / 3) % 3 - 1, ow = tap % 3 - 1;
  const int no = nocc[b];
  const int L = (no + S - 1) / S, k0 = s * L, k1 = min(no, k0 + L);
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
  for (int kk = k0; kk < k1; kk += 32) {
    __syncthreads();
    if (t < 32) {  // the dY voxel each of the chunk's occupied voxels meets under this tap, or -1 outside the grid
      int u = -1;
      if (kk + t < k1) {
        const int v = occ[(size_t)b * n + kk + t];
        const int d = v / (r * r) - od, h = (v / r) % r - oh, w = v % r - ow;
        if ((unsigned)d < (unsigned)r && (unsigned)h < (unsigned)r && (unsigned)w < (unsigned)r) u = (d * r + h) * r + w;
      }
      su[t] = u;
    }
    __syncthreads();
#pragma unroll
    for (int e = t; e < 32 * 64; e += 256) {
      const int k = e >> 6, c = e & 63;
      const int u = su[k];
      As[k][c] = (u >= 0 && co0 + c < cout) ? dyt[((size_t)b * r3 + u) * cout + co0 + c] : 0.0f;
      Bs[k][c] = (kk + k < k1 && ci0 + c < cin) ? xocc[((size_t)b * n + kk + k) * cin + ci0 + c] : 0.0f;
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const float4 a = *(const float4 *)&As[k][4 * ty], bb = *(const float4 *)&Bs[k][4 * tx];
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __fmaf_rn(av[i], bv[j], acc[i][j]);
    }
  }
  float *po = part + (size_t)blockIdx.x * row + (size_t)tap * cout * cin;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = co0 + 4 * ty + i, ci = ci0 + 4 * tx + j;
      if (co < cout && ci < cin) po[(size_t)co * cin + ci] = acc[i][j];
    }
}

static int wg_occ_splits(int b, int cin, int cout) {
  const long wgs = (long)b * 27 * ((cout + 63) / 64) * ((cin + 63) / 64);
  long S = (400 + wgs - 1) / wgs;
  return (int)(S < 1 ? 1 : S > 8 ? 8 : S);
}
// floats of workspace: partial rows | dYt | Xocc | occ (ints) | nocc (ints)
extern "C" size_t p2pb_conv3d_k3_wgrad_occ_ws_floats(int b, int cin, int cout, int r, int n) {
  const size_t r3 = (size_t)r * r * r, row = (size_t)cout * cin * 27 + cout;
  return (size_t)b * wg_occ_splits(b, cin, cout) * row + (size_t)b * r3 * cout + (size_t)b * n * cin + (size_t)b * n + b + 16;
}
// x f32[b,cin,r,r,r] (zero outside the voxels with cnt > 0), dy f32[b,cout,r,r,r], cnt i32[b,r^3] (avg_voxelize's counts), n = points
// per cloud (an upper bound of the occupied voxels) -> dw f32[cout,cin,3,3,3], db f32[cout] | NULL
extern "C" int p2pb_conv3d_k3_wgrad_occ(int b, int cin, int cout, int r, int n, const float *x, const float *dy, const int *cnt,
                                        float *dw, float *db, float *ws, void *stream) {
  if (b <= 0 || cin <= 0 || cout <= 0 || r <= 0 || n <= 0 || !x || !dy || !cnt || !dw || !ws) return P2PB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int r3 = r * r * r, S = wg_occ_splits(b, cin, cout), ns = b * S;
  const size_t nw = (size_t)cout * cin * 27, row = nw + cout;
  float *part = ws, *dyt = part + (size_t)ns * row, *xocc = dyt + (size_t)b * r3 * cout;
  int *occ = (int *)(xocc + (size_t)b * n * cin), *nocc = occ + (size_t)b * n;
  hipLaunchKernelGGL(wg_occ_list_kernel, dim3(b), dim3(1024), 0, s, r3, n, cnt, occ, nocc);
  hipLaunchKernelGGL(wg_xocc_kernel, dim3((unsigned)cdiv((long)n * cin, 256), b), dim3(256), 0, s, cin, r3, n, x, occ, nocc, xocc);
  hipLaunchKernelGGL(wg_dyt_kernel, dim3(cdiv(r3, 32), cdiv(cout, 32), b), dim3(256), 0, s, cout, r3, dy, dyt);
  hipLaunchKernelGGL(wg_bias_rows_kernel, dim3(cout, b), dim3(256), 0, s, cout, r3, S, row, nw, dy, part);
  hipLaunchKernelGGL(conv3d_k3_wgrad_occ_kernel, dim3(ns, 27, ((cout + 63) / 64) * ((cin + 63) / 64)), dim3(256), 0, s, r, n, cin, cout,
                     S, row, dyt, xocc, occ, nocc, part);
  wg_reduce(ns, 27, (size_t)cout * cin, (size_t)cout, part, dw, db, s);
  return p2pb_launch_status();
}
