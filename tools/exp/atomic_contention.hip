// How fast are 64-bit device-scope atomic adds without return when many waves hit the same place? One lane per wave
// adds to acc[(wave_global % spread) * stride_bytes / 8]. Prints ns per atomic for spread = 1 (one address) and for
// several strides (same line / neighbouring sectors / separate lines).   hipcc --offload-arch=gfx950 -O3 tools/exp/atomic_contention.hip -o tools/exp/atomic_contention (built here, run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long *acc, int spread, int stride8, int reps) {
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) == 0)
    for (int r = 0; r < reps; ++r) atomicAdd(acc + (size_t)((w + r) % spread) * stride8, 1ull);
}
int main() {
  unsigned long long *acc;
  hipMalloc(&acc, 64 << 20);
  hipMemset(acc, 0, 64 << 20);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 8192, reps = 4;
  const double total = (double)blocks * 4 * reps;
  const int spreads[] = {1, 2, 4, 8, 16, 64, 256, 1024, 8192};
  const int strides[] = {8, 32, 64, 128, 256, 4096};
  for (int st : strides)
    for (int sp : spreads) {
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, acc, sp, st / 8, reps);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, acc, sp, st / 8, reps);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("stride %5d B  spread %5d addresses: %8.1f us total, %7.2f ns per atomic, %7.1f ns per atomic per address\n", st, sp,
             ms * 1e3, ms * 1e6 / total, ms * 1e6 / (total / sp));
    }
  return 0;
}
