// abi.hip -- library identification for the C ABI (include/p2pb_hip.h) + the zero-fill helper.
#include "common.h"

extern "C" int p2pb_version(void) { return 1; }
extern "C" const char *p2pb_target_arch(void) { return "gfx950"; }

// Zero-fill as an ordinary kernel node. hipMemsetAsync is avoided on purpose: under hipGraph stream
// capture its memset node did not re-execute reliably on replay here (stale voxel counts -> OOB list
// writes -> GPU memory fault after a few replays), a kernel node always does.
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned *__restrict__ p, size_t nwords) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += stride) p[i] = 0u;
}

int p2pb_zero_async(void *p, size_t nbytes, hipStream_t s) {
  if (nbytes == 0) return 0;
  const size_t nwords = (nbytes + 3) / 4;  // every buffer zeroed here is a whole number of 32-bit words
  const unsigned grid = (unsigned)((nwords + 255) / 256 > 2048 ? 2048 : (nwords + 255) / 256);
  hipLaunchKernelGGL(zero_words_kernel, dim3(grid), dim3(256), 0, s, (unsigned *)p, nwords);
  return (int)hipGetLastError();
}
