// FETCH_SIZE calibration for 8-byte-per-lane streaming reads (MI355X_MICROARCH.md: only 16 B/lane is calibrated):
// read a 1 GiB buffer once with 8 B / lane (a wave covers 512 contiguous bytes, like pw_split_kernel's B rows)
// and once with 16 B / lane; compare FETCH_SIZE with the known byte count.
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void read8(const f2 *p, size_t n, float *out) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    f2 v = p[i];
    s += v.x + v.y;
  }
  if (s == 123.456f) out[0] = s;
}
__global__ void read4(const float *p, size_t n, float *out) {  // 4 B / lane: the 256-channel GEMM's activation loads
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s == 123.456f) out[0] = s;
}
__global__ void read16(const f4 *p, size_t n, float *out) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    f4 v = p[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) out[0] = s;
}
extern "C" void calib(void *buf, size_t bytes, float *out) {
  for (int r = 0; r < 3; ++r) {
    hipLaunchKernelGGL(read4, dim3(4096), dim3(256), 0, 0, (const float *)buf, bytes / 4, out);
    hipLaunchKernelGGL(read8, dim3(4096), dim3(256), 0, 0, (const f2 *)buf, bytes / 8, out);
    hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const f4 *)buf, bytes / 16, out);
  }
  hipDeviceSynchronize();
}
