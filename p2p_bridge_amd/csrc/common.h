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

// zero-fill `nbytes` (multiple of 4) on stream `s` with a kernel node (see abi.hip)
int p2pb_zero_async(void *p, size_t nbytes, hipStream_t s);
