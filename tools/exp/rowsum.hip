// experiment: 32 values per lane -> per-lane total of row (lane & 31) over the 32 lanes of its half-wave
#include <hip/hip_runtime.h>
template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ void swap16(float &a, float &b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
struct OpAdd { __device__ static float f(float a, float b) { return a + b; } };
struct OpMin { __device__ static float f(float a, float b) { return fminf(a, b); } };
struct OpMax { __device__ static float f(float a, float b) { return fmaxf(a, b); } };
template <class Op>
__device__ __forceinline__ float rowreduce32(float (&v)[32]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 16; ++i) {  // lanes L, L^16: odd 16-lane rows keep v[i+16]
    float a = v[i], b = v[i + 16];
    swap16(a, b);
    v[i] = Op::f(a, b);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {  // L, L^8 (row_ror:8)
    const float x = Op::f(v[i], dppf<0x128>(v[i])), y = Op::f(v[i + 8], dppf<0x128>(v[i + 8]));
    v[i] = (lane & 8) ? y : x;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // L, 7-L inside each 8 (row_half_mirror)
    const float x = Op::f(v[i], dppf<0x141>(v[i])), y = Op::f(v[i + 4], dppf<0x141>(v[i + 4]));
    v[i] = (lane & 4) ? y : x;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {  // L, L^2 (quad_perm [2,3,0,1])
    const float x = Op::f(v[i], dppf<0x4E>(v[i])), y = Op::f(v[i + 2], dppf<0x4E>(v[i + 2]));
    v[i] = (lane & 2) ? y : x;
  }
  const float x = Op::f(v[0], dppf<0xB1>(v[0])), y = Op::f(v[1], dppf<0xB1>(v[1]));  // L, L^1
  return (lane & 1) ? y : x;
}
__global__ void k(const float *in, float *out) {  // in[lane][32]
  float v[32], w[32], u[32];
  for (int i = 0; i < 32; ++i) v[i] = w[i] = u[i] = in[threadIdx.x * 32 + i];
  out[threadIdx.x] = rowreduce32<OpAdd>(v);
  out[64 + threadIdx.x] = rowreduce32<OpMin>(w);
  out[128 + threadIdx.x] = rowreduce32<OpMax>(u);
}
extern "C" int run(const float *in, float *out) {
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, out);
  return (int)hipDeviceSynchronize();
}
