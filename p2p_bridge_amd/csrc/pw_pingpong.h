// pw_pingpong.h -- the wide 1x1-convolution GEMM (>= 256 output channels, positions in whole 256-blocks) of the f16x3
// arithmetic as a double-buffered "ping-pong" kernel (round 3). Included by pointwise.hip.
//
// Why: pw_split_kernel<WM = 4> (256 channels x 128 positions, 72 KB, two workgroups per CU) ran the 512 -> 1024 launch
// with the matrix pipe 34 % busy: every wave does "stage, barrier, multiply, barrier", the weight tile's LDS-DMA is
// issued and awaited inside one stage, and only the accident of the co-resident workgroup's phase fills the matrix pipe
// while a workgroup stages. It also moves a 32 KB weight tile per 128 positions (4.3 GB through L2 per launch).
//
// This form: ONE workgroup of 8 waves per CU on a 256-channel x 256-position tile (a wave: 64 channels x 128 positions,
// 128 accumulator registers), 32 input channels per stage, BOTH operand tiles double-buffered in LDS (2 x 64 KB):
//   * the two waves of a SIMD belong to different halves of the workgroup and run a stage in OPPOSITE order --
//     half 0: multiply stage s, then stage its share of s + 1; half 1: stage its share of s + 1, then multiply s --
//     so each SIMD's matrix pipe always has one wave in its MFMA block while the partner does the VALU / LDS-write /
//     load work (separate issue ports), by construction instead of by luck; ONE barrier per stage;
//   * the weight tile of stage s + 1 is DMA'd (global_load_lds) at the top of stage s and awaited at its end: a whole
//     stage of latency cover; the raw activations of stage s + 2 are in flight in registers during stage s + 1;
//   * a weight tile feeds 256 positions: half the weight bytes per MFMA (2.1 GB per launch);
//   * the activation tile is transformed (folded norm + Swish) and split once per 256 channels, as before.
// Per stage and SIMD: 2 waves x 48 MFMAs x 32 cycles = 3072 matrix cycles; 2 x (16 elements x ~12 VALU + 4 b128 LDS
// writes + 4 DMA issues + 8 loads) of staging beside them.
// Layouts: A tile = the split pack's planes 0 / 1 of two 128-channel blocks, [blk][kstep][plane][khalf][128] x 16 B
// (32 KB, THREE buffers: DMA'd two stages ahead); B tile = [kstep][plane][khalf][256 positions] x 16 B (32 KB, two buffers), position p at slot p (lane l of N-tile n reads
// slot 32 n + l: 16 consecutive 16-byte slots per LDS service group, conflict-free; staging lane l writes slots 2l and
// 2l + 1 with two b128 stores, 32-byte stride: conflict-free).
// Outputs in pw_split_kernel's layout: per-(sample, 64-position slot, channel) {sum, sum of squares} partials (fixed
// order: deterministic; a wave's 128 positions land in its even slot, zero in the odd one), {min, max} per slot for the
// global pooling (pool_u == 0 only; the wave's extrema in both of its slots), optional channel-major stores.
#pragma once

#ifdef PP_TIMELINE  // experiment builds (tools/exp_pp_timeline.py): s_memtime at four points of wave 0, kept in scalar
__device__ unsigned long long *pp_tl_buf;  // registers and stored at the very end (a store in between would shift the
extern "C" int p2pb_pp_timeline_set(void *p) {  // hand-counted vmcnt waits of the stage loop)
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(pp_tl_buf), &p, sizeof(p));
}
#define PP_TL(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define PP_TL(var)
#endif
#define PP_CK 32
#define PP_TILE 2048  // 16-byte groups per operand tile per stage (32 KB)
#define PP_LDS_BYTES (5 * PP_TILE * 16)  // A x 3, B x 2: all 160 KB of the CU

template <bool XF, bool POOL, bool PRE = false>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(224))) void pw_pingpong_kernel(int cin, int cout, int P, int nslots,
                                                             const float *__restrict__ in, const u32x4 *__restrict__ wp,
                                                             const float *__restrict__ bias,
                                                             const float *__restrict__ bias_b,
                                                             const float *__restrict__ in_scale,
                                                             const float *__restrict__ in_shift, int in_swish,
                                                             float *__restrict__ out, float *__restrict__ stats_part,
                                                             float *__restrict__ mm_out, int pool_u) {
  extern __shared__ u32x4 pp_lds[];  // [A0 | A1 | A2 | B0 | B1], 32 KB each
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;
  const int wm = wave & 3, wn = wave >> 2;  // MFMA tile: 64 channels x 128 positions; waves w, w + 4 share a SIMD
  const int grp = wave >> 2;                // 0: multiply first, 1: stage first
  const int cg = wave & 3, half = wave >> 2;  // staging share: channels 8 cg .. 8 cg + 7 of the stage, positions 128 half ..
  // XCD-aware order (pw_split_kernel): the channel blocks of one activation tile run side by side on one XCD
  const int ncoblk = gridDim.y;
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const unsigned nblk = gridDim.x * gridDim.y * gridDim.z;
  const unsigned vid = nblk % 8 == 0 ? (lin % 8) * (nblk / 8) + lin / 8 : lin;
  const int bx = (vid / ncoblk) % gridDim.x, by = vid % ncoblk;
  const int b = vid / (ncoblk * gridDim.x);
  const int pblk = bx * 256, co0 = by * 256;
  const int nstage = cin / PP_CK;
  const int nblk128 = cout / 128;
  float bpre = 0.0f;  // bias (+ per-sample bias) of channel co0 + tid for the epilogue's table
  if (tid < 256) {
    bpre = bias ? bias[co0 + tid] : 0.0f;
    if (bias_b) bpre += bias_b[(size_t)b * cout + co0 + tid];
  }

  f32x16 acc[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

  // one descriptor for the sample's [cin, P] operand; rows through the scalar offset. The loads are INLINE ASM and their
  // waits are counted by hand (PP_VMCNT): beside global_load_lds hipcc waits vmcnt(0) for every ordinary load, which
  // drains the weight DMA and the two-stage-deep activation prefetch at every barrier.
  const unsigned long long inb = (unsigned long long)(in + (size_t)b * cin * P);
  const u32x4 rs = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)inb),
                    (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(inb >> 32)),
                    (unsigned)__builtin_amdgcn_readfirstlane(cin * P * 4), 0x00020000u};
  const unsigned voff = (unsigned)(pblk + 128 * half + 2 * lane) * 4u;
  // Raw activations, two stages in flight: even stages in the PRIVATE registers v[224:239], odd stages in v[240:255].
  // The kernel is compiled for 224 VGPRs (amdgpu_num_vgpr) and these 32 are touched by inline asm only -- loads write
  // them, hand-counted s_waitcnt retire them, v_mov copies hand the values to the compiler. Two things hipcc does
  // otherwise: (1) beside global_load_lds it waits vmcnt(0) for every ordinary load (and wraps the loads in waterfall
  // loops), draining the weight DMA and the prefetch at every barrier; (2) with the loads as inline asm on compiler-
  // allocated registers it copies registers whose load is still in flight (loop back edge) -- garbage.
#define PP_RP(set, i) "v[" PP_STR(PP_R##set##i##L) ":" PP_STR(PP_R##set##i##H) "]"
#define PP_STR2(x) #x
#define PP_STR(x) PP_STR2(x)
#define PP_R00L 224
#define PP_R00H 225
#define PP_R01L 226
#define PP_R01H 227
#define PP_R02L 228
#define PP_R02H 229
#define PP_R03L 230
#define PP_R03H 231
#define PP_R04L 232
#define PP_R04H 233
#define PP_R05L 234
#define PP_R05H 235
#define PP_R06L 236
#define PP_R06H 237
#define PP_R07L 238
#define PP_R07H 239
#define PP_R10L 240
#define PP_R10H 241
#define PP_R11L 242
#define PP_R11H 243
#define PP_R12L 244
#define PP_R12H 245
#define PP_R13L 246
#define PP_R13H 247
#define PP_R14L 248
#define PP_R14H 249
#define PP_R15L 250
#define PP_R15H 251
#define PP_R16L 252
#define PP_R16H 253
#define PP_R17L 254
#define PP_R17H 255
  // (s_nop 4: an SGPR written by the SALU needs 5 wait states before a VMEM instruction reads it; the compiler's hazard
  //  recogniser does not look inside inline asm and the row offset is computed right in front of it)
#define PP_LOAD1(set, i)                                                                                            \
  asm volatile("s_nop 4\n\tbuffer_load_dwordx2 " PP_RP(set, i) ", %0, %1, %2 offen" ::"v"(voff), "s"(rs),             \
               "s"(__builtin_amdgcn_readfirstlane((st * PP_CK + 8 * cg + i) * P * 4))                                  \
               : "memory", "v" PP_STR(PP_R##set##i##L), "v" PP_STR(PP_R##set##i##H))
#define PP_LOAD8(set) PP_LOAD1(set, 0); PP_LOAD1(set, 1); PP_LOAD1(set, 2); PP_LOAD1(set, 3); PP_LOAD1(set, 4); PP_LOAD1(set, 5); PP_LOAD1(set, 6); PP_LOAD1(set, 7)
#define PP_TAKE1(set, i, r)                                                                                         \
  asm volatile("v_mov_b32 %0, v" PP_STR(PP_R##set##i##L) "\n\tv_mov_b32 %1, v" PP_STR(PP_R##set##i##H) : "=v"(r[i][0]), "=v"(r[i][1]))
#define PP_TAKE8(set, r) PP_TAKE1(set, 0, r); PP_TAKE1(set, 1, r); PP_TAKE1(set, 2, r); PP_TAKE1(set, 3, r); PP_TAKE1(set, 4, r); PP_TAKE1(set, 5, r); PP_TAKE1(set, 6, r); PP_TAKE1(set, 7, r)
  auto load_b = [&](int s, int set) {
    const int st = s < nstage ? s : nstage - 1;  // past the end: a valid row, never used
    if (set == 0) { PP_LOAD8(0); } else { PP_LOAD8(1); }
  };
  auto take_b = [&](int set, float (&r)[8][2]) {
    if (set == 0) { PP_TAKE8(0, r); } else { PP_TAKE8(1, r); }
  };
#define PP_N(n) #n
#define PP_VMCNT(n) asm volatile("s_waitcnt vmcnt(" PP_N(n) ")" ::: "memory")
// the stage barrier: the DMA of the next stage has landed (vmcnt), this wave's LDS writes and reads are done (lgkmcnt),
// then a RAW s_barrier -- __syncthreads() carries a vmcnt(0) while an LDS-DMA is in flight, which would drain the
// two-stage-deep prefetch at every barrier
#define PP_BARRIER(n) asm volatile("s_waitcnt vmcnt(" PP_N(n) ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define PP_BARRIER(n) asm volatile("s_waitcnt vmcnt(" PP_N(n) ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
  auto dma_a = [&](int s, int buf) {
    const int st = s < nstage ? s : nstage - 1;
    const u32x4 *src = wp + ((size_t)st * nblk128 + by * 2) * PWS_TILE;
    u32x4 *dst = pp_lds + buf * PP_TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = i * 512 + wave * 64;  // wave-uniform; lane l lands at e + l
      const int blk = e >> 10, rem = e & 1023;
      const int srow = blk * PWS_TILE + (((rem >> 9) * 3 + ((rem >> 8) & 1)) * 2 + ((rem >> 7) & 1)) * 128 + (rem & 127);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + srow + lane),
                                       (__attribute__((address_space(3))) void *)(dst + e), 16, 0, 0);
    }
  };
  auto stage_b = [&](int s, int buf, int set) {  // braw (raw activations of stage s) -> transformed, split, into buffer buf
    // Written element-parallel (every step over all 16 values before the next step) so that the sixteen dependent
    // chains fma -> exp -> rcp -> mul -> cvt -> sub -> cvt overlap: issued two at a time (the compiler's choice for the
    // nested form) the phase was latency-bound at ~11 cycles per instruction.
    // y4 = 4 swish(v) = v * (4 / (1 + 2^(-v log2 e))) = v * rcp(0.25 + 0.25 e): the activation scale of the fp16 split
    // (SPLIT_F16_SX = 4, exact: a power of two) rides in the reciprocal's argument
    float braw[8][2], y[8][2];
    take_b(set, braw);
    if (XF) {
      float sc[8], sh[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = (s < nstage ? s : nstage - 1) * PP_CK + 8 * cg + i;
        sc[i] = in_scale[b * cin + c];
        sh[i] = in_shift[b * cin + c];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e) y[i][e] = __fmaf_rn(braw[i][e], sc[i], sh[i]);
      if (in_swish) {
        float t[8][2];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int e = 0; e < 2; ++e) t[i][e] = __builtin_amdgcn_exp2f(y[i][e] * -1.44269504088896340736f);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int e = 0; e < 2; ++e) t[i][e] = __builtin_amdgcn_rcpf(__fmaf_rn(t[i][e], 0.25f, 0.25f));
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int e = 0; e < 2; ++e) y[i][e] *= t[i][e];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int e = 0; e < 2; ++e) y[i][e] *= SPLIT_F16_SX;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e) y[i][e] = braw[i][e] * SPLIT_F16_SX;
    }
    u32x4 *lb = pp_lds + (3 + buf) * PP_TILE;
    const int kstep = cg >> 1, kh = cg & 1;
    u32x4 q0[2], q1[2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned p0, p1;
        split2h(y[2 * i][e], y[2 * i + 1][e], p0, p1);
        q0[e][i] = p0;
        q1[e][i] = p1;
      }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int slot = 128 * half + 2 * lane + e;
      lb[((kstep * 2 + 0) * 2 + kh) * 256 + slot] = q0[e];
      lb[((kstep * 2 + 1) * 2 + kh) * 256 + slot] = q1[e];
    }
  };
  auto multiply = [&](int abuf, int buf) {
    const u32x4 *la = pp_lds + abuf * PP_TILE + (wm >> 1) * 1024 + (wm & 1) * 64 + l31;
    const u32x4 *lb = pp_lds + (3 + buf) * PP_TILE + wn * 128 + l31;
#pragma unroll
    for (int kstep = 0; kstep < 2; ++kstep) {
      u32x4 af[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int m = 0; m < 2; ++m) af[s][m] = la[((kstep * 2 + s) * 2 + khalf) * 128 + m * 32];
      // all four column tiles' operands first, then the products term by term (small terms first: a1 b0, a0 b1, a0 b0):
      // the three MFMAs that update one accumulator are 8 instructions apart instead of 2 -- same order per accumulator,
      // same bits; launch 1.003 -> 0.987 ms (profiles/r03c_pingpong_order_ab.txt)
      u32x4 bf[4][2];
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int s = 0; s < 2; ++s) bf[n][s] = lb[((kstep * 2 + s) * 2 + khalf) * 256 + n * 32];
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m][n] = split_mfma<SPLIT_F16X3>(af[1][m], bf[n][0], acc[m][n]);
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m][n] = split_mfma<SPLIT_F16X3>(af[0][m], bf[n][1], acc[m][n]);
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m][n] = split_mfma<SPLIT_F16X3>(af[0][m], bf[n][0], acc[m][n]);
    }
  };

  PP_TL(tl0);
  if constexpr (PRE) {
    // ---- pre-split operand (p2pb_pointwise_presplit: folded norm + Swish + fp16-pair split applied ONCE per element, in the
    // byte layout of the B tile): both operand tiles arrive by LDS-DMA, nothing is staged through registers, every wave
    // runs  [DMA B(s+1), DMA A(s+2); multiply(s); barrier]  -- the timeline of the staged form says why: a wave's
    // staging (2.0-2.9 k cycles: sixteen exp / rcp chains, splits, LDS writes, and once per 256-channel block) takes as long
    // as its multiply (2.0-2.2 k), so a stage lasts 5.2 k cycles where the matrix pipe needs 3.1 k.
    // `in` = S[b][P / 256][cin / 32][2048] x 16 B. VMEM queue at a barrier, oldest first: A(s+1) B(s+1) A(s+2) -> vmcnt(4).
    const u32x4 *bsrc = (const u32x4 *)in + ((size_t)b * gridDim.x + bx) * (size_t)nstage * PP_TILE;
    auto dma_b = [&](int s, int buf) {
      const int st = s < nstage ? s : nstage - 1;
      const u32x4 *src = bsrc + (size_t)st * PP_TILE;
      u32x4 *dst = pp_lds + (3 + buf) * PP_TILE;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = i * 512 + wave * 64;  // wave-uniform; lane l lands at e + l
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + e + lane),
                                         (__attribute__((address_space(3))) void *)(dst + e), 16, 0, 0);
      }
    };
    dma_b(0, 0);
    dma_a(0, 0);
    dma_a(1, 1);
    PP_BARRIER(4);
    for (int s = 0; s < nstage; ++s) {
      dma_b(s + 1, (s + 1) & 1);
      dma_a(s + 2, (s + 2) % 3);
      multiply(s % 3, s & 1);
      PP_BARRIER(4);
    }
  }
  if constexpr (!PRE) {
  if (grp == 1) __builtin_amdgcn_s_setprio(1);  // the later-dispatched half loses every arbitration otherwise (+4..8 %)
  // ---- prologue. VMEM queue, oldest first: loads(0) loads(1) DMA(0) DMA(1) | loads(2)
  load_b(0, 0);
  load_b(1, 1);
  dma_a(0, 0);
  dma_a(1, 1);
  PP_VMCNT(16);
  stage_b(0, 0, 0);
  load_b(2, 0);
  PP_BARRIER(12);  // DMA(0) has landed
  }
  PP_TL(tl1);
  // Interval s (between barriers s and s + 1): half 0 runs [DMA A(s+2); multiply(s); stage B(s+1); loads(s+3)], half 1
  // [DMA A(s+2); stage B(s+1); loads(s+3); multiply(s)] -- written as ONE loop body with the multiply in common code
  // (half 1 is the same stream rotated by half an interval: its staging sits behind the barrier), because a two-sided
  // `if (half) {stage; multiply} else {multiply; stage}` made the register allocator spill 167 registers.
  // In both halves the queue at the end of interval s reads  DMA(s+1) loads(s+2) DMA(s+2) loads(s+3):
  //   stage B(s+1) needs loads(s+1), older than all 16 of the first three groups  -> vmcnt(16)
  //   the barrier needs DMA(s+1)                                                   -> vmcnt(20)
  // Past the last stage the indices clamp (valid addresses, buffers nobody reads again), so the counts never change.
  auto head = [&](int s, int set) {  // the staging half of an interval: DMA A(s+1), stage B(s), loads(s+2)
    dma_a(s + 1, (s + 1) % 3);
    PP_VMCNT(16);
    stage_b(s, s & 1, set);
    load_b(s + 2, set);
  };
#ifdef PP_TIMELINE  // inside one steady-state interval (the even stage 6), both halves: wave 0 and wave 4
  unsigned long long ts[5] = {0, 0, 0, 0, 0};
#define PP_TLS(k) do { if (s == 6) ts[k] = __builtin_readcyclecounter(); } while (0)
#else
#define PP_TLS(k)
#endif
  if constexpr (!PRE) {
  if (grp == 1) head(1, 1);
  for (int s = 0; s < nstage; s += 2) {
    // even stage s: its successor s + 1 lives in register set 1
    PP_TLS(0);
    if (grp == 0) dma_a(s + 2, (s + 2) % 3);
    multiply(s % 3, 0);
    PP_TLS(1);
    if (grp == 0) {
      PP_VMCNT(16);
      PP_TLS(2);
      stage_b(s + 1, 1, 1);
      load_b(s + 3, 1);
      PP_TLS(3);
      PP_BARRIER(20);
      PP_TLS(4);
    } else {
      PP_BARRIER(20);
      PP_TLS(2);
      head(s + 2, 0);
      PP_TLS(3);
    }
    // odd stage s + 1: its successor s + 2 lives in register set 0
    if (grp == 0) dma_a(s + 3, (s + 3) % 3);
    multiply((s + 1) % 3, 1);
    if (grp == 0) {
      PP_VMCNT(16);
      stage_b(s + 2, 0, 0);
      load_b(s + 4, 0);
      PP_BARRIER(20);
    } else {
      PP_BARRIER(20);
      if (s + 2 < nstage) head(s + 3, 1);
    }
  }
  }  // !PRE
  PP_VMCNT(0);  // nothing of this workgroup may still be on its way into LDS when the waves retire
  PP_TL(tl2);

  // ---- epilogue (the arithmetic and the outputs of pws_epilogue; a slot = 64 consecutive positions)
  {
    // bias (+ per-sample bias) of the tile's 256 channels through an LDS table: fetched by 256 lanes at once. Read per row
    // straight from memory, the 32 rows of a lane were 32 (64 with bias_b) dependent load -> wait -> use round trips in a
    // chain of branches -- most of the epilogue's 14.8 k cycles (profiles/r03b_pingpong_timeline.txt). The operand buffers
    // are free: every wave's DMA has landed (vmcnt(0) above) once all of them are past this barrier.
    float *btab = (float *)pp_lds;
    __syncthreads();
    if (tid < 256) btab[tid] = bpre;  // (fetched at the top of the kernel: its latency is under the stage loop)
    const float oscale = ((const float *)(wp + (size_t)nstage * nblk128 * PWS_TILE))[1];  // 1 / (S_x S_w)
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float bv = btab[wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf];
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n][r] = acc[m][n][r] * oscale + bv;
      }
  }
  PP_TL(tl3);
#ifdef PP_TIMELINE
  unsigned long long tl4 = 0;
#endif
  const int pw0 = pblk + wn * 128;  // the wave's first position
  if (out) {
    float *ob = out + (size_t)b * cout * P;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
#pragma unroll
        for (int n = 0; n < 4; ++n) ob[(size_t)co * P + pw0 + n * 32 + l31] = acc[m][n][r];
      }
  }
  const int rm = l31 >> 4, rr = l31 & 15;
  const int rco = co0 + wm * 64 + rm * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * khalf;  // this lane's row after a rowreduce32
  // ONE reduction set per wave (its 128 positions = two 64-position slots of the partials' layout): the four tiles of a row
  // are combined per lane first, then one reduce-scatter per statistic -- 4 instead of 8 (the epilogue was 21 k of a
  // workgroup's 112 k cycles, profiles/r03b_pingpong_timeline.txt, nearly all of it these reductions). The sums go to the
  // wave's even slot and zero to the odd one; the extrema (whose consumer takes a min / max over slots) go to both.
  {
    const int slot = pw0 >> 6;  // even
    float tv[32];
    if (stats_part) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) tv[m * 16 + r] = (acc[m][0][r] + acc[m][1][r]) + (acc[m][2][r] + acc[m][3][r]);
      const float s1 = rowreduce32<RowAdd>(tv);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tv[m * 16 + r] = (acc[m][0][r] * acc[m][0][r] + acc[m][1][r] * acc[m][1][r]) +
                           (acc[m][2][r] * acc[m][2][r] + acc[m][3][r] * acc[m][3][r]);
      const float s2 = rowreduce32<RowAdd>(tv);
      float *q = stats_part + (((size_t)b * nslots + slot) * cout + rco) * 2;
      q[0] = s1;
      q[1] = s2;
      float *z = q + (size_t)cout * 2;  // slot + 1
      z[0] = 0.0f;
      z[1] = 0.0f;
    }
#ifdef PP_TIMELINE
    tl4 = __builtin_readcyclecounter();
#endif
    if (POOL && pool_u == 0) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tv[m * 16 + r] = fminf(fminf(acc[m][0][r], acc[m][1][r]), fminf(acc[m][2][r], acc[m][3][r]));
      const float mn = rowreduce32<RowMin>(tv);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tv[m * 16 + r] = fmaxf(fmaxf(acc[m][0][r], acc[m][1][r]), fmaxf(acc[m][2][r], acc[m][3][r]));
      const float mx = rowreduce32<RowMax>(tv);
      float *q = mm_out + (((size_t)b * (P / 64) + slot) * cout + rco) * 2;
      q[0] = mn;
      q[1] = mx;
      q[(size_t)cout * 2] = mn;  // slot + 1
      q[(size_t)cout * 2 + 1] = mx;
    }
  }
#ifdef PP_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0 && pp_tl_buf) {
    unsigned long long *q = pp_tl_buf + (size_t)lin * 16;
    q[0] = tl0, q[1] = tl1, q[2] = tl2, q[3] = __builtin_readcyclecounter(), q[4] = tl3, q[5] = tl4;
    for (int k = 0; k < 5; ++k) q[6 + k] = ts[k];
  }
  if (tid == 256 && pp_tl_buf) {
    unsigned long long *q = pp_tl_buf + (size_t)lin * 16;
    for (int k = 0; k < 5; ++k) q[11 + k] = ts[k];
  }
#endif
}

