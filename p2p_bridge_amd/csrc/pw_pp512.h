// pw_pp512.h -- the wide 1x1-convolution GEMM of the f16x3 arithmetic (>= 512 output channels in whole 512-blocks, an even
// number of 32-channel stages), round 4: the ping-pong kernel of round 3 (256 x 256 tile: git history, csrc/pw_pingpong.h) re-tiled to 512 channels x 128 positions.
// Included by pointwise.hip.
//
// Why (round-3 evidence, profiles/r03f_pmc_pw_pingpong_*, r03b_pingpong_timeline.txt, r03d_pw_presplit_ab.txt): on the
// 256-channel x 256-position tile a wave's STAGING -- folded norm + Swish (exp, rcp), the fp16-pair split and the LDS write of
// its share of the activation tile -- took as long as its 48 MFMAs (2.0-2.9 k vs 2.0-2.2 k cycles), the same activation
// tile was transformed by each of the layer's 256-channel blocks (4 x for 512 -> 1024), and the tile's LDS writes were 2-way
// bank-conflicted (ds_write_b128 banks are (addr / 4) mod 32 over 8-lane groups; lanes 32 bytes apart collide pairwise:
// 17.8 M conflict cycles of 87 M). The matrix pipe was 46 % busy.
//
// This form keeps everything that worked -- one 8-wave workgroup per CU, the two waves of a SIMD in opposite phase (one in
// its MFMA block while the other stages), weight tiles by LDS-DMA, raw activations prefetched two stages ahead into
// pinned VGPRs by inline-asm buffer loads with hand-counted s_waitcnt, ONE barrier per stage, all 160 KB of LDS -- and
// changes the tile: 512 channels x 128 positions. A wave still owns 64 channels x 128 positions (128 accumulator
// registers, the same fragment reads and MFMA order), but
//   * the activation tile of a stage is 128 positions x 32 channels: HALF the staging work per MFMA (8 elements per lane
//     per stage instead of 16), and a 512 -> 1024 layer transforms every activation twice instead of four times
//     (256 -> 512: once);
//   * a staging lane owns ONE position: its two 16-byte LDS stores (term planes h0 / h1) go to consecutive slots of
//     consecutive lanes -- conflict-free under the stores' 32-bank rule;
//   * the weight tile is 64 KB per stage (512 channels): LDS = 2 x 64 KB (A, double-buffered, DMA'd one stage ahead at the
//     top of an interval and awaited at its end) + 2 x 16 KB (B);
//   * positions need not come in whole tiles (P % 4 == 0 for the row alignment the split path requires anyway): loads past
//     the row end read the next row or the descriptor's zero, and the epilogue masks those columns -- PVDL's 50000- and
//     12500-point layers take this kernel.
// LDS map (16-byte slots): A[buf][blk 4][kstep 2][plane 2][khalf 2][128 channels], B[buf][kstep 2][plane 2][khalf 2][128
// positions]; fragment reads are one ds_read_b128 per (kstep, plane, 32-row tile), lane l31 at slot base + l31 (16
// consecutive slots per service group: conflict-free), exactly the round-3 layout with a 128-slot row pitch.
// Outputs in pw_split_kernel's layout: {sum, sum of squares} partials per (sample, 64-position slot, channel) (the wave's 128
// positions in its even slot, zero in the odd one; slots past the last tile zeroed by the last tile), {min, max} per slot
// for the global pooling (pool_u == 0), optional channel-major stores.
#pragma once

#ifdef PP_TIMELINE  // experiment builds (tools/exp_p5_timeline.py): s_memtime stamps of waves 0 and 4, stored at the very end
__device__ unsigned long long *pp_tl_buf;
extern "C" int p2pb_pp_timeline_set(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(pp_tl_buf), &p, sizeof(p)); }
#endif
#define P5_CK 32
#define P5_A_SLOTS 4096  // 16-byte slots of one weight stage tile (64 KB)
#define P5_B_SLOTS 1024  // ... of one activation stage tile (16 KB)
#define P5_LDS_BYTES ((2 * P5_A_SLOTS + 2 * P5_B_SLOTS) * 16)  // all 160 KB of the CU

#define P5_STR2(x) #x
#define P5_STR(x) P5_STR2(x)
// the pinned registers: the raw activations of the next stage in v[248:255] (the kernel is compiled for 248 VGPRs; these 8 are
// touched by inline asm only -- loads write them, a hand-placed s_waitcnt retires them, v_mov copies hand them to the compiler)
#define P5_R00 248
#define P5_R01 249
#define P5_R02 250
#define P5_R03 251
#define P5_R04 252
#define P5_R05 253
#define P5_R06 254
#define P5_R07 255
// (s_nop 4: an SGPR written by the SALU needs wait states before a VMEM instruction reads it as its scalar offset; the
//  hazard recogniser does not look inside inline asm)
#define P5_LOAD1(set, i)                                                                                     \
  asm volatile("s_nop 4\n\tbuffer_load_dword v" P5_STR(P5_R##set##i) ", %0, %1, %2 offen" ::"v"(voff), "s"(rs), \
               "s"(__builtin_amdgcn_readfirstlane((st * P5_CK + 8 * cg + i) * P * 4))                         \
               : "memory", "v" P5_STR(P5_R##set##i))
#define P5_LOAD8(set) P5_LOAD1(set, 0); P5_LOAD1(set, 1); P5_LOAD1(set, 2); P5_LOAD1(set, 3); P5_LOAD1(set, 4); P5_LOAD1(set, 5); P5_LOAD1(set, 6); P5_LOAD1(set, 7)
#define P5_TAKE1(set, i, r) asm volatile("v_mov_b32 %0, v" P5_STR(P5_R##set##i) : "=v"(r[i]))
#define P5_TAKE8(set, r) P5_TAKE1(set, 0, r); P5_TAKE1(set, 1, r); P5_TAKE1(set, 2, r); P5_TAKE1(set, 3, r); P5_TAKE1(set, 4, r); P5_TAKE1(set, 5, r); P5_TAKE1(set, 6, r); P5_TAKE1(set, 7, r)
// y[i0], y[i0 + 1] = raw[i0 .. i0 + 1] * sc + sh with the raw pair read from the pinned registers v[248 + i0 : 249 + i0].
// Packed (v_pk_fma_f32, two channels per instruction). Review r4 item 5 asked for two plain v_fma_f32 instead (the guide prices
// the packed instruction at +22 cycles per MFMA gap beside a running matrix pipe): built (-DP5_PK_FMA=0) and measured on one box,
// three alternations -- packed 0.883 / 0.901 / 0.878 ms per 512 -> 1024 launch, plain 0.909 / 0.898 / 0.905; the sampler equal
// (profiles/r05_pp512_fma_ab.txt). Here the window runs beside the OTHER wave's MFMAs and four instructions fewer win.
#ifndef P5_PK_FMA
#define P5_PK_FMA 1
#endif
#if P5_PK_FMA
#define P5_TAKE_FMA(j, i0)                                                                                          \
  do {                                                                                                              \
    const f32x2 scp = {sc[i0], sc[i0 + 1]}, shp = {sh[i0], sh[i0 + 1]};                                             \
    f32x2 yp;                                                                                                       \
    asm volatile("v_pk_fma_f32 %0, v[" P5_STR(P5_R0##i0) ":" P5_STR(P5_R0##i0##H) "], %1, %2" : "=v"(yp) : "s"(scp), "v"(shp)); \
    y[i0] = yp[0];                                                                                                  \
    y[i0 + 1] = yp[1];                                                                                              \
  } while (0)
#else
#define P5_TAKE_FMA(j, i0)                                                                                          \
  do {                                                                                                              \
    asm volatile("v_fma_f32 %0, v" P5_STR(P5_R0##i0) ", %1, %2" : "=v"(y[i0]) : "s"(sc[i0]), "v"(sh[i0]));          \
    asm volatile("v_fma_f32 %0, v" P5_STR(P5_R0##i0##H) ", %1, %2" : "=v"(y[i0 + 1]) : "s"(sc[i0 + 1]), "v"(sh[i0 + 1])); \
  } while (0)
#endif
#define P5_R00H 249
#define P5_R02H 251
#define P5_R04H 253
#define P5_R06H 255
#define P5_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// the stage barrier: the weight DMA of the next stage has landed (vmcnt), this wave's LDS traffic is done (lgkmcnt), then a
// RAW s_barrier (__syncthreads() would carry a vmcnt(0) and drain the prefetch)
#define P5_BARRIER(n) asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <bool XF, bool POOL>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(248))) void pw_pp512_kernel(
    int cin, int cout, int P, int nslots, const float *__restrict__ in, const u32x4 *__restrict__ wp,
    const float *__restrict__ bias, const float *__restrict__ bias_b, const float *__restrict__ in_scale,
    const float *__restrict__ in_shift, int in_swish, float *__restrict__ out, float *__restrict__ stats_part,
    float *__restrict__ mm_out, int pool_u) {
  extern __shared__ u32x4 p5_lds[];  // [A0 | A1 | B0 | B1]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;
  const int wm = wave;        // MFMA tile: channels 64 wm .. + 63 of the block, all 128 positions; waves w, w + 4 share a SIMD
  const int grp = wave >> 2;  // 0: multiply first, 1: stage first
  const int cg = wave & 3, half = wave >> 2;  // staging share: channels 8 cg .. + 7 of the stage, position 64 half + lane
  // XCD-aware order (pw_split_kernel): the channel blocks of one activation tile run side by side on one XCD
  const int ncoblk = gridDim.y;
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const unsigned nblk = gridDim.x * gridDim.y * gridDim.z;
  const unsigned vid = nblk % 8 == 0 ? (lin % 8) * (nblk / 8) + lin / 8 : lin;
  const int bx = (vid / ncoblk) % gridDim.x, by = vid % ncoblk;
  const int b = vid / (ncoblk * gridDim.x);
  const int pblk = bx * 128, co0 = by * 512;
  const int nstage = cin / P5_CK;
  const int nblk128 = cout / 128;
  // bias (+ per-sample bias) of channel co0 + tid for the epilogue's table (its latency is under the stage loop)
  float bpre = bias ? bias[co0 + tid] : 0.0f;
  if (bias_b) bpre += bias_b[(size_t)b * cout + co0 + tid];

#ifdef PP_TIMELINE
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool tl_on = false;
  const unsigned long long tl0 = __builtin_readcyclecounter();
#define P5_TLS(k) do { if (tl_on) ts[k] = __builtin_readcyclecounter(); } while (0)
#else
#define P5_TLS(k)
#endif
  f32x16 acc[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

  // one descriptor for the sample's [cin, P] operand; rows through the scalar offset
  const unsigned long long inb = (unsigned long long)(in + (size_t)b * cin * P);
  const u32x4 rs = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)inb),
                    (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(inb >> 32)),
                    (unsigned)__builtin_amdgcn_readfirstlane(cin * P * 4), 0x00020000u};
  // (positions past the row end of a ragged last tile are masked below; their loads are clamped INTO the row: the scalar row
  //  offset is outside the descriptor's bounds check, so an unclamped one would read past the last row of the last sample --
  //  ADVICE r4)
  const int pcl = pblk + 64 * half + lane;
  const unsigned voff = (unsigned)(pcl < P ? pcl : P - 1) * 4u;
  auto load_b = [&](int s) {
    const int st = s < nstage ? s : nstage - 1;  // past the end: a valid row, never used
    P5_LOAD8(0);
  };
  auto take_b = [&](float (&r)[8]) { P5_TAKE8(0, r); };
  // the weight tile of a stage: 8 LDS-DMA instructions per wave. A wave BLOCKS at issue while the texture path accepts them
  // (60-170 cycles per 1 KB instruction, tools/exp_p5_timeline.py), so each half issues its share in its own non-matrix
  // window: half 1 at the top of the interval (beside half 0's MFMA block), half 0 right after its MFMA block
  auto dma_a = [&](int s, int buf) {
    const int st = s < nstage ? s : nstage - 1;
    const u32x4 *src = wp + ((size_t)st * nblk128 + by * 4) * PWS_TILE;
    u32x4 *dst = p5_lds + buf * P5_A_SLOTS;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = i * 512 + wave * 64;  // wave-uniform; lane l lands at e + l
      const int blk = e >> 10, rem = e & 1023;
      const int srow = blk * PWS_TILE + (((rem >> 9) * 3 + ((rem >> 8) & 1)) * 2 + ((rem >> 7) & 1)) * 128 + (rem & 127);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + srow + lane),
                                       (__attribute__((address_space(3))) void *)(dst + e), 16, 0, 0);
    }
  };
  // The non-matrix window of a wave, stage s -> B[buf]: fetch the stage's folded-norm parameters (scalar loads: their latency
  // runs under what follows), take the raw activations out of the pinned registers, REQUEST what the next windows need -- the
  // raw activations of stage s + 1 into the registers just freed, then (dma >= 0) this wave's share of weight stage `dma` --
  // and only then do the VALU work (transform, split) and the two LDS stores: the texture path accepts a stage's bytes at 64
  // B/clk and a wave blocks at issue meanwhile, so everything is issued FIRST and the arithmetic runs in its shadow
  // (issued last, the eight loads alone waited ~1 k cycles behind the DMA: tools/exp_p5_timeline.py).
  // element-parallel arithmetic (every step over all 8 values before the next): the dependent chains fma -> exp -> rcp -> mul
  // -> cvt -> sub -> cvt overlap. 4 swish(v) = v * rcp(0.25 + 0.25 e^-v): the activation scale of the fp16 split
  // (SPLIT_F16_SX = 4, a power of two: exact) rides in the reciprocal's argument
  auto stage = [&](int s, int buf, int dma, int dbuf) {
    float braw[8], y[8], sc[8], sh[8];
    if (XF) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = (s < nstage ? s : nstage - 1) * P5_CK + 8 * cg + i;
        sc[i] = in_scale[b * cin + c];
        sh[i] = in_shift[b * cin + c];
      }
    }
    P5_VMCNT(0);  // the raw activations of stage s (requested a whole interval ago)
    P5_TLS(2);
    if (dma >= 0) dma_a(dma, dbuf);  // (first: the scalar loads above land while the texture path accepts the DMA)
    // the folded norm straight out of the pinned registers (packed fma, two channels per instruction; no copies), THEN the
    // request for the next stage's activations into the same registers
    if (XF) {
      P5_TAKE_FMA(0, 0); P5_TAKE_FMA(1, 2); P5_TAKE_FMA(2, 4); P5_TAKE_FMA(3, 6);
    } else {
      take_b(braw);
    }
    load_b(s + 1);
    P5_TLS(1);
    if (XF) {
      if (in_swish) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = __builtin_amdgcn_exp2f(y[i] * -1.44269504088896340736f);
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = __builtin_amdgcn_rcpf(__fmaf_rn(t[i], 0.25f, 0.25f));
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] *= t[i];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] *= SPLIT_F16_SX;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = braw[i] * SPLIT_F16_SX;
    }
    u32x4 q0, q1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned p0, p1;
      split2h(y[2 * i], y[2 * i + 1], p0, p1);
      q0[i] = p0;
      q1[i] = p1;
    }
    u32x4 *lb = p5_lds + 2 * P5_A_SLOTS + buf * P5_B_SLOTS;
    const int kstep = cg >> 1, kh = cg & 1, slot = 64 * half + lane;
    lb[((kstep * 2 + 0) * 2 + kh) * 128 + slot] = q0;
    lb[((kstep * 2 + 1) * 2 + kh) * 128 + slot] = q1;
    P5_TLS(3);
  };
  auto multiply = [&](int buf) {  // stage tiles A[buf], B[buf]
    const u32x4 *la = p5_lds + buf * P5_A_SLOTS + (wm >> 1) * 1024 + (wm & 1) * 64 + l31;
    const u32x4 *lb = p5_lds + 2 * P5_A_SLOTS + buf * P5_B_SLOTS + l31;
#pragma unroll
    for (int kstep = 0; kstep < 2; ++kstep) {
      u32x4 af[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int m = 0; m < 2; ++m) af[s][m] = la[((kstep * 2 + s) * 2 + khalf) * 128 + m * 32];
      // all four column tiles' operands first, then the products term by term (small terms first: a1 b0, a0 b1, a0 b0): the
      // three MFMAs that update one accumulator are 8 instructions apart (profiles/r03c_pingpong_order_ab.txt)
      u32x4 bf[4][2];
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int s = 0; s < 2; ++s) bf[n][s] = lb[((kstep * 2 + s) * 2 + khalf) * 128 + n * 32];
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 2; ++m)
          if (X2W_KEEP_LOW_WEIGHT_PRODUCT) acc[m][n] = split_mfma<SPLIT_F16X3>(af[1][m], bf[n][0], acc[m][n]);
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

  if (grp == 1) __builtin_amdgcn_s_setprio(1);  // the later-dispatched half loses every arbitration otherwise
  // ---- prologue: L(0), D(0) requested; stage 0 staged (which requests L(1)); barrier: D(0) has landed.
  // (L(s) = the 8 activation loads of stage s into the one pinned register set, D(s) = this wave's 8 weight DMAs)
  load_b(0);
  dma_a(0, 0);
  stage(0, 0, -1, 0);
  P5_BARRIER(8);  // D(0) (older than L(1)) has landed
#ifdef PP_TIMELINE
  const unsigned long long tl1 = __builtin_readcyclecounter();
#endif
  // Interval s (between barriers s and s + 1), W = the window above for stage s + 1 (requests L(s+2), D(s+1)):
  //   half 0: multiply(s) | W     -- its matrix block starts at the barrier
  //   half 1: W | multiply(s)     -- issue + VALU beside half 0's MFMAs, then its own block beside half 0's window
  // The barrier waits for everything this wave requested in the interval (vmcnt(0)): D(s+1) must have landed for all waves'
  // reads of A[(s+1) & 1], and L(s+2), older than it, has had the whole interval. Past the last stage the indices clamp
  // (valid addresses, buffers nobody reads again).
  for (int s = 0; s < nstage; ++s) {
#ifdef PP_TIMELINE
    tl_on = s == 6;
#endif
    const int cur = s & 1, nxt = cur ^ 1;
    P5_TLS(0);
    if (grp == 1) stage(s + 1, nxt, s + 1, nxt);
    multiply(cur);
    P5_TLS(4);
    if (grp == 0) stage(s + 1, nxt, s + 1, nxt);
    P5_TLS(5);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    P5_TLS(6);
    P5_BARRIER(0);
    P5_TLS(7);
  }
  P5_VMCNT(0);  // nothing of this workgroup may still be on its way into LDS when the waves retire
#ifdef PP_TIMELINE
  const unsigned long long tl2 = __builtin_readcyclecounter();
#endif

  // ---- epilogue (the arithmetic and the outputs of pws_epilogue; a slot = 64 consecutive positions)
  {
    // bias (+ per-sample bias) of the tile's 512 channels through an LDS table (the operand buffers are free: every wave's
    // DMA has landed once all of them are past this barrier)
    float *btab = (float *)p5_lds;
    __syncthreads();
    btab[tid] = bpre;
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
  const bool full = pblk + 128 <= P;  // (wave-uniform; a ragged last tile masks its columns >= P in place below)
  if (out) {
    float *ob = out + (size_t)b * cout * P;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
#pragma unroll
        for (int n = 0; n < 4; ++n)
          if (pblk + n * 32 + l31 < P) ob[(size_t)co * P + pblk + n * 32 + l31] = acc[m][n][r];
      }
  }
  const int rm = l31 >> 4, rr = l31 & 15;
  const int rco = co0 + wm * 64 + rm * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * khalf;  // this lane's row after a rowreduce32
  // ONE reduction set per wave (its 128 positions = two 64-position slots of the partials' layout): the four tiles of a row
  // are combined per lane first, then one reduce-scatter per statistic. The sums go to the wave's even slot and zero to the
  // odd one; the extrema (whose consumer takes a min / max over slots) go to both.
  // Ragged tile (once per sample at most): columns >= P are set to ZERO in the accumulators before the sums, and to a copy of
  // the row's first column (always valid) before the extrema -- the statistics code itself has one form and no masks.
  {
    const int slot = pblk >> 6;  // even
    const int nmine = 2 * (int)gridDim.x;  // slots the tiles of this launch own; [nmine, nslots) are zeroed by the last tile
    float tv[32];
    if (!full) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const bool ok = pblk + n * 32 + l31 < P;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] = ok ? acc[m][n][r] : 0.0f;
      }
    }
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
      if (bx == (int)gridDim.x - 1)
        for (int sl = nmine; sl < nslots; ++sl) {
          float *zz = stats_part + (((size_t)b * nslots + sl) * cout + rco) * 2;
          zz[0] = 0.0f;
          zz[1] = 0.0f;
        }
    }
    if (POOL && pool_u == 0) {
      if (!full) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float first = __shfl(acc[m][0][r], khalf * 32);  // column pblk of this row (pblk < P)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n][r] = pblk + n * 32 + l31 < P ? acc[m][n][r] : first;
          }
      }
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
      // minmax f32[b, 2 * ceil(P / 128), cout, 2] (p2pb_pointwise_minmax_floats, split tiling): both slots of the tile
      float *q = mm_out + (((size_t)b * nmine + slot) * cout + rco) * 2;
      q[0] = mn;
      q[1] = mx;
      q[(size_t)cout * 2] = mn;  // slot + 1
      q[(size_t)cout * 2 + 1] = mx;
    }
  }
#ifdef PP_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((tid == 0 || tid == 256) && pp_tl_buf) {
    unsigned long long *q = pp_tl_buf + (size_t)lin * 32 + (tid ? 16 : 0);
    q[0] = tl0, q[1] = tl1, q[2] = tl2, q[3] = __builtin_readcyclecounter();
    for (int k = 0; k < 8; ++k) q[4 + k] = ts[k];
  }
#endif
}
