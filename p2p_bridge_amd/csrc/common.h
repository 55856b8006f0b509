// common.h -- shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p2pb_hip.h"

#define P2PB_WAVE 64

static inline int p2pb_launch_status() { return (int)hipGetLastError(); }

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
struct RowMin {
  __device__ static float f(float a, float b) { return fminf(a, b); }
};
struct RowMax {
  __device__ static float f(float a, float b) { return fmaxf(a, b); }
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

// zero-fill `nbytes` (multiple of 4) on stream `s` with a kernel node (see abi.hip)
int p2pb_zero_async(void *p, size_t nbytes, hipStream_t s);
