// Sustained rate of the 16-bit matrix pipe of this part under its power envelope (DESIGN.md 3.1 quotes it as the practical
// ceiling of the split-operand kernels): every wave runs a register-only loop of v_mfma_f32_32x32x16_f16 on four independent
// accumulators; operands are random fp16 (zero operands clock higher: MI355X_MICROARCH.md "DVFS give-back"), 1 / 2 / 4 waves
// per SIMD, 0.5 s each. Prints TFLOP/s and the effective clock (MFMAs x 32 cycles / time).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak tools/exp/mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(const f16x8 *__restrict__ ab, float *__restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  f16x8 a = ab[2 * (t & 4095)], b = ab[2 * (t & 4095) + 1];
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, c3, 0, 0, 0);
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  if (s == 12345.678f) out[t] = s;  // (keeps the loop alive)
}
int main() {
  const int n = 4096 * 2;
  std::vector<_Float16> h(n * 8);
  srand(1);
  for (auto &v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f);
  f16x8 *d; float *o;
  hipMalloc(&d, n * 16); hipMalloc(&o, 1 << 24);
  hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  for (int wps : {1, 2, 4}) {
    const int blocks = cus * wps;  // 256 threads = 4 waves = one per SIMD
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<<<blocks, 256>>>(d, o, 200);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      mfma_loop<<<blocks, 256>>>(d, o, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double mf = (double)blocks * 4 * iters * 32;  // MFMAs
      const double tf = mf * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
      const double clk = (double)iters * 32 * wps * 32 / (ms * 1e-3) / 1e9;  // MFMAs per SIMD x 32 cycles
      printf("%d wave(s)/SIMD, %d CUs: %.1f ms, %.1f TFLOP/s fp16 dense (%.3f of 2516.6), effective matrix clock %.2f GHz\n", wps, cus, ms, tf, tf / 2516.6, clk);
      if (ms < 400) iters = (int)(iters * 500.0 / ms);
    }
  }
  return 0;
}
