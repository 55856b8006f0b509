// abi.hip -- library identification for the C ABI (include/p2pb_hip.h) + the zero-fill helper.
#include "common.h"

extern "C" int p2pb_version(void) { return 1; }
extern "C" const char *p2pb_target_arch(void) { return "gfx950"; }

// products per split operand pair in the bf16 matrix kernels (common.h)
int p2pb_g_split_terms = SPLIT_F16X3;
extern "C" int p2pb_set_split_terms(int terms) {
  if (terms != SPLIT_BF16X6 && terms != SPLIT_F16X3) return P2PB_EINVAL;
  p2pb_g_split_terms = terms;
  return 0;
}
extern "C" int p2pb_get_split_terms(void) { return p2pb_g_split_terms; }

// Zero-fill as an ordinary kernel node. hipMemsetAsync is avoided on purpose: under hipGraph stream
// capture its memset node did not re-execute reliably on replay here (stale voxel counts -> OOB list
// writes -> GPU memory fault after a few replays), a kernel node always does.
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned *__restrict__ p, size_t nwords) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += stride) p[i] = 0u;
}

__global__ __launch_bounds__(256) void zero_quads_kernel(uint4 *__restrict__ p, size_t nquads) {
  const size_t stride = (size_t)gridDim.x * 256;
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nquads; i += stride) p[i] = z;
}

int p2pb_zero_async(void *p, size_t nbytes, hipStream_t s) {
  if (nbytes == 0) return 0;
  if ((((size_t)p) & 15) == 0 && (nbytes & 15) == 0) {  // 16 B per lane: the streaming-store sweet spot
    const size_t nq = nbytes / 16;
    const unsigned grid = (unsigned)((nq + 255) / 256 > 4096 ? 4096 : (nq + 255) / 256);
    hipLaunchKernelGGL(zero_quads_kernel, dim3(grid), dim3(256), 0, s, (uint4 *)p, nq);
    return (int)hipGetLastError();
  }
  const size_t nwords = (nbytes + 3) / 4;  // every buffer zeroed here is a whole number of 32-bit words
  const unsigned grid = (unsigned)((nwords + 255) / 256 > 2048 ? 2048 : (nwords + 255) / 256);
  hipLaunchKernelGGL(zero_words_kernel, dim3(grid), dim3(256), 0, s, (unsigned *)p, nwords);
  return (int)hipGetLastError();
}
