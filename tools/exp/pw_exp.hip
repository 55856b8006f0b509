// Experiment harness (not part of the product): variants of the pointwise GEMM main loop, to find
// what keeps the MFMA pipe at 55 %. Built and run by tools/exp/pw_exp.py.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK 16
__device__ __forceinline__ float hw_sum(float v) {
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// V bit0: unpredicated loads; bit1: no stats; bit2: no stores; bit3: no B loads in loop; bit4: no A loads in loop
template <int V, int OCC>
__global__ __launch_bounds__(256, OCC) void k(int cin, int cout, int cout_pad, int P, const float *__restrict__ in,
                                              const float *__restrict__ wp, const float *__restrict__ bias,
                                              float *__restrict__ out, float *__restrict__ stats_part) {
  constexpr int MT = 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int p0 = blockIdx.x * 256, co0 = blockIdx.y * (32 * MT), b = blockIdx.z;
  const int pl[2] = {p0 + wave * 64 + l31, p0 + wave * 64 + 32 + l31};
  const bool pok[2] = {pl[0] < P, pl[1] < P};
  const float *inb = in + (size_t)b * cin * P;
  const int nchunk8 = (cin + 7) >> 3;
  f32x16 acc[MT][2];
  for (int m = 0; m < MT; ++m)
    for (int s = 0; s < 2; ++s)
      for (int r = 0; r < 16; ++r) acc[m][s][r] = 0.0f;
  float bcur[CK / 2][2], bnxt[CK / 2][2];
  auto load_b = [&](int ci0, float(&dst)[CK / 2][2]) {
#pragma unroll
    for (int kk = 0; kk < CK / 2; ++kk) {
      const int ci = ci0 + 2 * kk + khalf;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (V & 1) dst[kk][s] = inb[(size_t)ci * P + pl[s]];
        else dst[kk][s] = (ci < cin && pok[s]) ? inb[(size_t)ci * P + pl[s]] : 0.0f;
      }
    }
  };
  load_b(0, bcur);
  const float *wbase = wp + ((size_t)khalf * cout_pad + co0 + l31) * 4;
  const size_t wchunk_stride = (size_t)2 * cout_pad * 4;
  f32x4 a_cur[CK / 8][MT], a_nxt[CK / 8][MT];
  auto load_a = [&](int chunk0, f32x4(&dst)[CK / 8][MT]) {
#pragma unroll
    for (int sub = 0; sub < CK / 8; ++sub) {
      const int ch = (V & 1) ? chunk0 + sub : (chunk0 + sub < nchunk8 ? chunk0 + sub : nchunk8 - 1);
#pragma unroll
      for (int m = 0; m < MT; ++m) dst[sub][m] = *(const f32x4 *)(wbase + (size_t)ch * wchunk_stride + (size_t)m * 32 * 4);
    }
  };
  load_a(0, a_cur);
  for (int ci0 = 0; ci0 < cin; ci0 += CK) {
    const int chunk0 = ci0 >> 3;
    const bool more = ci0 + CK < cin;
    if (more) {
      if (!(V & 16)) load_a(chunk0 + CK / 8, a_nxt);
      if (!(V & 8)) load_b(ci0 + CK, bnxt);
    }
#pragma unroll
    for (int sub = 0; sub < CK / 8; ++sub) {
      if (!(V & 1) && chunk0 + sub >= nchunk8) break;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            acc[m][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[sub][m][kk], bcur[sub * 4 + kk][s], acc[m][s], 0, 0, 0);
    }
    if (more) {
      if (!(V & 8))
#pragma unroll
        for (int kk = 0; kk < CK / 2; ++kk)
#pragma unroll
          for (int s = 0; s < 2; ++s) bcur[kk][s] = bnxt[kk][s];
      if (!(V & 16))
#pragma unroll
        for (int sub = 0; sub < CK / 8; ++sub)
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
      float bv = co < cout ? bias[co] : 0.0f;
      float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int p = pl[s];
        const float v = acc[m][s][r] + bv;
        if (co < cout && pok[s]) {
          if (!(V & 4) || v == 123.456f) outb[(size_t)co * P + p] = v;
          if (!(V & 2)) {
            s1 += v;
            s2 += v * v;
          }
        }
      }
      if (!(V & 2)) {
        s1 = hw_sum(s1);
        s2 = hw_sum(s2);
        if (l31 == 31 && co < cout) {
          float *q = stats_part + ((((size_t)b * gridDim.x + blockIdx.x) * 4 + wave) * cout + co) * 2;
          q[0] = s1;
          q[1] = s2;
        }
      }
    }
  }
}


// ---- v4 candidate: wave tile = 32*MT couts x 128 positions; lane j holds positions 4j..4j+3 (one 16-byte load per
// channel), n-tile s = positions {4j+s}; stores are 16-byte too. Rows addressed through the buffer soffset (SGPR).
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int CKK, int OCC, bool STATS>
__global__ __launch_bounds__(256, OCC) void k2(int cin, int cout, int cout_pad, int P, const float *__restrict__ in,
                                               const float *__restrict__ wp, const float *__restrict__ bias,
                                               float *__restrict__ out, float *__restrict__ stats_part) {
  constexpr int MT = 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int co0 = blockIdx.y * (32 * MT), b = blockIdx.z;
  const int p = blockIdx.x * 512 + wave * 128 + l31 * 4;
  const bool pok = p < P;
  const int pc = pok ? p : P - 4;
  const float *inb = in + (size_t)b * cin * P;
  const int nchunk8 = cin >> 3;
  f32x16 acc[MT][4];
  for (int m = 0; m < MT; ++m)
    for (int s = 0; s < 4; ++s)
      for (int r = 0; r < 16; ++r) acc[m][s][r] = 0.0f;
  const int rowb = P * 4;
  const unsigned voff = (unsigned)(khalf * P + pc) * 4u;
  f32x4 bcur[CKK / 2], bnxt[CKK / 2];
  auto load_b = [&](int ci0, f32x4(&dst)[CKK / 2]) {
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)(inb + (size_t)ci0 * P), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int kk = 0; kk < CKK / 2; ++kk) {
      i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 2 * kk * rowb, 0);
      dst[kk] = __builtin_bit_cast(f32x4, v);
    }
  };
  load_b(0, bnxt);
  const float *wbase = wp + ((size_t)khalf * cout_pad + co0 + l31) * 4;
  const size_t wchunk_stride = (size_t)2 * cout_pad * 4;
  f32x4 a_cur[CKK / 8][MT], a_nxt[CKK / 8][MT];
  auto load_a = [&](int chunk0, f32x4(&dst)[CKK / 8][MT]) {
#pragma unroll
    for (int sub = 0; sub < CKK / 8; ++sub)
#pragma unroll
      for (int m = 0; m < MT; ++m)
        dst[sub][m] = *(const f32x4 *)(wbase + (size_t)(chunk0 + sub) * wchunk_stride + (size_t)m * 32 * 4);
  };
  load_a(0, a_nxt);
  for (int ci0 = 0; ci0 < cin; ci0 += CKK) {
    const int chunk0 = ci0 >> 3;
    const bool more = ci0 + CKK < cin;
    // rotate first (this is where the vmcnt wait lands), then request the next chunk, then multiply
#pragma unroll
    for (int kk = 0; kk < CKK / 2; ++kk) bcur[kk] = bnxt[kk];
#pragma unroll
    for (int sub = 0; sub < CKK / 8; ++sub)
#pragma unroll
      for (int m = 0; m < MT; ++m) a_cur[sub][m] = a_nxt[sub][m];
    if (more) {
      load_a(chunk0 + CKK / 8, a_nxt);
      load_b(ci0 + CKK, bnxt);
    }
#pragma unroll
    for (int sub = 0; sub < CKK / 8; ++sub)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            acc[m][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[sub][m][kk], bcur[sub * 4 + kk][s], acc[m][s], 0, 0, 0);
  }
  float *outb = out + (size_t)b * cout * P;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      const float bv = co < cout ? bias[co] : 0.0f;
      f32x4 v = {acc[m][0][r] + bv, acc[m][1][r] + bv, acc[m][2][r] + bv, acc[m][3][r] + bv};
      float s1 = 0.0f, s2 = 0.0f;
      if (co < cout && pok) {
        *(f32x4 *)(outb + (size_t)co * P + p) = v;
        if (STATS) {
          s1 = (v[0] + v[1]) + (v[2] + v[3]);
          s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
      }
      if (STATS) {
        s1 = hw_sum(s1);
        s2 = hw_sum(s2);
        if (l31 == 31 && co < cout) {
          float *q = stats_part + ((((size_t)b * gridDim.x + blockIdx.x) * 4 + wave) * cout + co) * 2;
          q[0] = s1;
          q[1] = s2;
        }
      }
    }
  }
}
extern "C" int pw_exp(int variant, int occ, int b, int cin, int cout, int P, const float *in, const float *wp,
                      const float *bias, float *out, float *stats, void *stream) {
  dim3 grid((P + 255) / 256, (cout + 63) / 64, b);
  const int cout_pad = (cout + 127) / 128 * 128;
#define L(V, O) hipLaunchKernelGGL((k<V, O>), grid, dim3(256), 0, (hipStream_t)stream, cin, cout, cout_pad, P, in, wp, bias, out, stats)
#define C(V) case V: if (occ == 2) L(V, 2); else L(V, 3); break;
  switch (variant) {
    C(0) C(1) C(3) C(5) C(7) C(9) C(17) C(25) C(31)
    default: return -1;
  }
  return (int)hipGetLastError();
}

extern "C" int pw_exp2(int ck, int occ, int stats_on, int b, int cin, int cout, int P, const float *in, const float *wp,
                       const float *bias, float *out, float *stats, void *stream) {
  dim3 grid((P + 511) / 512, (cout + 63) / 64, b);
  const int cout_pad = (cout + 127) / 128 * 128;
#define L2(CKK, O, S) hipLaunchKernelGGL((k2<CKK, O, S>), grid, dim3(256), 0, (hipStream_t)stream, cin, cout, cout_pad, P, in, wp, bias, out, stats)
  if (ck == 16 && occ == 2 && stats_on) L2(16, 2, true);
  else if (ck == 16 && occ == 2) L2(16, 2, false);
  else if (ck == 8 && occ == 2 && stats_on) L2(8, 2, true);
  else if (ck == 8 && occ == 2) L2(8, 2, false);
  else if (ck == 16 && occ == 1 && stats_on) L2(16, 1, true);
  else if (ck == 32 && occ == 1 && stats_on) L2(32, 1, true);
  else return -1;
  return (int)hipGetLastError();
}
