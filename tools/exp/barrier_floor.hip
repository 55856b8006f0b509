// What does one round of "every wave publishes a value, barrier, every wave reads the winner" cost on one CU?
// hipcc --offload-arch=gfx950 -O3 tools/exp/barrier_floor.hip -o /tmp/barrier_floor && /tmp/barrier_floor
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
#define LDSP __attribute__((address_space(3)))
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_umax_step(unsigned v) {
  const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
  return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = dpp_umax_step<0x111, 0xf>(v);
  v = dpp_umax_step<0x112, 0xf>(v);
  v = dpp_umax_step<0x114, 0xf>(v);
  v = dpp_umax_step<0x118, 0xf>(v);
  v = dpp_umax_step<0x142, 0xa>(v);
  v = dpp_umax_step<0x143, 0xc>(v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ u64 wave_max_u64(u64 v) {
  const unsigned hi = (unsigned)(v >> 32);
  const unsigned H = wave_max_u32(hi);
  const unsigned L = wave_max_u32(hi == H ? (unsigned)v : 0u);
  return ((u64)H << 32) | L;
}
__device__ __forceinline__ unsigned row_max_u32(unsigned v) {
  v = dpp_umax_step<0x111, 0xf>(v);
  v = dpp_umax_step<0x112, 0xf>(v);
  v = dpp_umax_step<0x114, 0xf>(v);
  v = dpp_umax_step<0x118, 0xf>(v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 15);
}
__device__ __forceinline__ u64 row_max_u64(u64 v) {
  const unsigned hi = (unsigned)(v >> 32);
  const unsigned H = row_max_u32(hi);
  const unsigned L = row_max_u32(hi == H ? (unsigned)v : 0u);
  return ((u64)H << 32) | L;
}
template <int MODE>
__global__ void k(int rounds, u64 *out, int nthreads, const int *chase) {
  __shared__ u64 slot[2][16];
  __shared__ u64 gmax[3];
  __shared__ float xyz[2][16][4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < 32) slot[t >> 4][t & 15] = 0;
  if (t < 3) gmax[t] = 0;
  __syncthreads();
  u64 acc = t;
  float sx = t;
  int jm3 = 1;
  const u64 t0 = __builtin_amdgcn_s_memtime();
  for (int j = 1; j < rounds; ++j) {
    if (MODE == 0) {  // barrier only
      __builtin_amdgcn_s_barrier();
    } else if (MODE == 1) {  // LDS write + barrier + LDS read
      if (lane == 0) slot[j & 1][wave] = acc + j;
      __syncthreads();
      acc += slot[j & 1][t & 15];
    } else if (MODE == 2) {  // + atomic max + three float reads + readlanes (the grid kernel's floor)
      if (lane == 0) {
        xyz[j & 1][wave][0] = sx;
        atomicMax(&gmax[jm3], (acc & ~15ull) | wave);
      }
      __syncthreads();
      const u64 fin = *(volatile u64 *)&gmax[jm3];
      const float q = xyz[j & 1][t & 15][0];
      jm3 = jm3 == 2 ? 0 : jm3 + 1;
      if (t == 0) gmax[jm3 == 2 ? 0 : jm3 + 1] = 0;
      const int ws = __builtin_amdgcn_readfirstlane((int)(fin & 15));
      sx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q), ws)) + 1.0f;
      acc += fin;
    } else if (MODE == 3) {  // a wave reduction (two dependent 6-step DPP chains as wave_max_u64), no barrier
      unsigned v = (unsigned)acc;
      for (int r = 0; r < 2; ++r) {
        for (int o = 1; o < 64; o <<= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
        v += r;
      }
      acc += v;
    } else if (MODE == 4) {  // one wave_max_u64 (DPP), no barrier; the next round's input depends on it
      acc = wave_max_u64(acc ^ lane) + j;
    } else if (MODE == 5) {  // MODE 2 with LDS-address-space accesses (ds_ instructions instead of flat_)
      if (lane == 0) {
        xyz[j & 1][wave][0] = sx;
        atomicMax(&gmax[jm3], (acc & ~15ull) | wave);
      }
      __syncthreads();
      const u64 fin = *(volatile LDSP u64 *)&gmax[jm3];
      const float q = *(volatile LDSP float *)&xyz[j & 1][t & 15][0];
      jm3 = jm3 == 2 ? 0 : jm3 + 1;
      if (t == 0) gmax[jm3 == 2 ? 0 : jm3 + 1] = 0;
      const int ws = __builtin_amdgcn_readfirstlane((int)(fin & 15));
      sx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q), ws)) + 1.0f;
      acc += fin;
    } else if (MODE == 6) {  // round 3's scheme: slots, barrier, 16-lane DPP maximum, ballot, dependent LDS read
      if (lane == 0) {
        slot[j & 1][wave] = acc + wave;
        xyz[j & 1][wave][0] = sx;
      }
      __syncthreads();
      const u64 mine = slot[j & 1][t & 15];
      const u64 v = row_max_u64(mine);
      const int ws = __builtin_ctzll(__ballot(lane < 16 && mine == v));
      sx = xyz[j & 1][ws][0] + 1.0f;
      acc += v;
    } else if (MODE == 7) {  // one dependent global load per round (L2-resident 1 MB ring), lane-uniform
      acc = chase[(acc & 0x3ffff)];
    } else if (MODE == 8) {  // readlane with a scalar index that depends on a VALU result, 4 in a row
      int v = (int)acc;
      for (int r = 0; r < 4; ++r) {
        const int src = __builtin_amdgcn_readfirstlane(v) & 63;
        v = __builtin_amdgcn_readlane(v + lane, src) + r;
      }
      acc = v;
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memtime();
  if (t == 0) out[0] = t1 - t0;
  out[1 + t] = acc + (u64)sx;
}
int main() {
  u64 *d;
  hipMalloc(&d, 8 * 2048);
  int *chase, *hc = new int[1 << 18];
  for (int i = 0; i < (1 << 18); ++i) hc[i] = (int)(((long long)i * 40503 + 12345) & 0x3ffff);
  hipMalloc(&chase, 4 << 18);
  hipMemcpy(chase, hc, 4 << 18, hipMemcpyHostToDevice);
  const int rounds = 20000;
  for (int nt : {1024, 512, 256, 64}) {
    u64 h[9];
#define RUN(M) hipLaunchKernelGGL(k<M>, dim3(1), dim3(nt), 0, 0, rounds, d, nt, chase); hipMemcpy(&h[M], d, 8, hipMemcpyDeviceToHost);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
    printf("%4d threads, cycles per round: barrier %.0f | write+barrier+read %.0f | atomic scheme (flat reads) %.0f, (ds reads) %.0f | round-3 scheme %.0f | "
           "wave_max_u64: shfl x2 %.0f, DPP %.0f | dependent L2 load %.0f | 4 dependent readlanes %.0f\n", nt,
           (double)h[0] / rounds, (double)h[1] / rounds, (double)h[2] / rounds, (double)h[5] / rounds, (double)h[6] / rounds, (double)h[3] / rounds,
           (double)h[4] / rounds, (double)h[7] / rounds, (double)h[8] / rounds);
  }
  return 0;
}
