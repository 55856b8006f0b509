// stamp.hip -- experiment helper (NOT part of the product library): device-side wall-clock stamps as kernel nodes, and an empty
// kernel, for measuring the real (un-profiled) timeline of the captured sampler step. rocprofv3's kernel trace cannot be used
// for that: under it the two sampler chains never overlap (profiles/r05_overlap.txt).
// build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/exp/libstamp.so tools/exp/stamp.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void stamp_kernel(unsigned long long *slot) { *slot = wall_clock64(); }  // 100 MHz constant clock
__global__ void noop_kernel() {}
extern "C" int exp_stamp(unsigned long long *slot, hipStream_t s) {
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, s, slot);
  return (int)hipGetLastError();
}
extern "C" int exp_noop(int n, hipStream_t s) {
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, s);
  return (int)hipGetLastError();
}
