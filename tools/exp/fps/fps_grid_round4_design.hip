// NOT SHIPPED (round 4 experiment, kept for the record): fps_grid_kernel re-designed around a timeline of the round -- one LDS
// atomic for the waves' maxima (wave id in the key), unchanged waves keep their maximum, the updating wave finds its new maximum
// in the cell's own reduction and settles the cell after the barrier, record ranges in LDS. Bit-identical indices, and SLOWER than
// the round-3 kernel it was meant to replace: 22.0 vs 18.6 ms at 50000 -> 12500 on box-surface clouds, 25.8 vs 21.5 on surface
// patches (both with the per-axis grid), 47.2 vs 43.4 ms per PVDL evaluation at B = 8. The work started from a baseline that the
// first experiment of the series had already degraded (26.2 ms), so every later "gain" was measured against the wrong number;
// the three-way A/B on one box is profiles/r04g_fps_grid.txt. This is the kernel part of csrc/sampling.hip at that point.
typedef float fg_f32x4 __attribute__((ext_vector_type(4)));
#ifdef FG_TIMELINE  // experiment builds (round 4; the driver script left the tree): the round of a wave that updates exactly one cell, by phase.
// A stamp waits for the scalar it is given (s_memtime issues in order, but does not wait for the vector pipe by itself)
__device__ unsigned long long *fg_tl_buf;
extern "C" int p2pb_fg_timeline_set(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(fg_tl_buf), &p, sizeof(p)); }
#define FG_STAMP(i, dep)                                                                                              \
  do {                                                                                                                \
    int fg_dummy;                                                                                                     \
    asm volatile("s_mov_b32 %1, %2\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts[i]), "=s"(fg_dummy) : "s"((int)(dep)) : "memory"); \
  } while (0)
#else
#define FG_STAMP(i, dep)
#endif
#define FG_LDS __attribute__((address_space(3)))  // (a volatile access through a GENERIC pointer becomes a flat_ instruction)
// Keys inside this kernel: (distance bits + 1) << 32 | 28-bit tie order (lower (k mod 512, k) is better, stored inverted: the
// order of fps_key) << 4 | four zero bits -- the wave id travels in them in the round's LDS atomic. k < 2^28.
__device__ __forceinline__ u64 fg_key(float best, int k) {
  const unsigned hi = best >= 0.0f ? (__float_as_uint(best) + 1u) : 0u;
  const unsigned sec = ((unsigned)(k & 511) << 19) | (unsigned)(k >> 9);
  return ((u64)hi << 32) | (u64)((~sec & 0x0FFFFFFFu) << 4);
}
__device__ __forceinline__ int fg_key_index(u64 key) {
  const unsigned sec = ~((unsigned)key >> 4) & 0x0FFFFFFFu;
  return (int)(((sec & 0x7FFFFu) << 9) | (sec >> 19));
}
// What bounds a round (round 4, tools/dbg/fg_abl.py on timing-only builds -- tools/exp/patches/sampling_fg_ablations.patch --,
// 50000 -> 12500, cycles at 2.4 GHz): 1944 with
// neither box tests nor updates (slot write, barrier, reading the winner -- sixteen waves, four to a SIMD, each issuing the
// same instructions), 2461 with the box tests, 4908 in all: the wave that updates a cell is the critical path (one trip to
// L2, then reductions), everything else is instruction count x 16 waves. Hence: the waves' maxima meet in ONE LDS atomic
// (the wave id in the key's low bits names the winner: no slot array, no 16-lane reduction after the barrier); a wave none
// of whose cells changed keeps its maximum; the wave that updates a cell finds ITS new maximum in the same reduction as the
// cell's (the lanes' other cells join the candidates) and brings the cell's own record up to date after the barrier.
#ifndef FG_NW
#define FG_NW 16  // waves per cloud, 64 / FG_NW cells per lane. 8 waves x 8 cells (-DFG_NW=8) measured 32.3 ms against 23.5 at
#endif            // 50000 -> 12500: the cell-select chains double and two updates per wave and round become common

constexpr int FG_NS = 64 / FG_NW;  // cells per lane
#ifndef FG_NQV
#define FG_NQV 2
#endif
constexpr int FG_NQ = FG_NQV;     // records per lane in flight when a cell is updated (3: two registers spill at 128)
__global__ __launch_bounds__(64 * FG_NW) void fps_grid_kernel(int n, int m, const float *__restrict__ coords,
                                                        const int *__restrict__ cell_start,
                                                        const float4 *__restrict__ rec, const float *__restrict__ cbox,
                                                        float *__restrict__ mind, int *__restrict__ indices) {
  __shared__ __attribute__((aligned(16))) float sxyz[2][16][4];  // the coordinates of the waves' farthest points (by round parity)
  __shared__ u64 gmax[3];  // the rounds' maxima (three buffers: one in use, one being read, one being zeroed)
  __shared__ int2 crange[FG_CELLS];  // every cell's record range {first, count}: read (one uniform ds_read_b64) by the wave that updates it
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x;
  const float *c = coords + (size_t)b * 3 * n;
  const int *cs = cell_start + (size_t)b * (FG_CELLS + 1);
  const float4 *rc = rec + (size_t)b * n;
  const float *bxp = cbox + (size_t)b * FG_CELLS * 6;
  float *md = mind + (size_t)b * n;
  int *out = indices + (size_t)b * m;

  // this lane's four cells: tight box, key of the farthest point (distance bits | tie-key: the reference's total order;
  // the bound of the box test is its distance word - 1, a NaN before round 1: every cell with a point is visited then) and
  // that point's coordinates -- all in registers; the record ranges in LDS
  float blo[FG_NS][3], bhi[FG_NS][3], cx[FG_NS], cy[FG_NS], cz[FG_NS];
  u64 ckey[FG_NS];
  unsigned full = 0;  // bit i: cell i holds a point
#pragma unroll
  for (int i = 0; i < FG_NS; ++i) {
    const int cell = fg_cell(wave + FG_NW * (i >> 2), lane + 64 * (i & 3));
    const int first = cs[cell], count = cs[cell + 1] - first;
    crange[cell] = make_int2(first, count);
    full |= (count > 0 ? 1u : 0u) << i;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      blo[i][a] = bxp[(size_t)cell * 6 + a];
      bhi[i][a] = bxp[(size_t)cell * 6 + 3 + a];
    }
    ckey[i] = 0;
    cx[i] = cy[i] = cz[i] = 0.0f;
  }
  for (int k = t; k < n; k += 64 * FG_NW) md[k] = 1e38f;
  if (t < 3) gmax[t] = 0;
  if (t < 128) ((float *)sxyz)[t] = 0.0f;
  int jm3 = 1;  // j mod 3
  if (t == 0) out[0] = 0;
  float sx = c[0], sy = c[n], sz = c[(size_t)2 * n];  // sample 0 = point 0
  __syncthreads();

  u64 wkey = 0;  // the wave's farthest point: key and coordinates (wave-uniform)
  float wkx = 0.0f, wky = 0.0f, wkz = 0.0f;
  bool pend = false;  // (wave-uniform) the cell whose record is brought up to date after the barrier
  int pslot = 0, plane = 0;
  u64 pbest = 0;
  float pbx = 0.0f, pby = 0.0f, pbz = 0.0f;
  auto settle = [&](int ii, int src, u64 best, float bxv, float byv, float bzv) {  // the cell's new farthest point, to its owner
    const u64 wbest = wave_max_u64(best);
    const int from = __builtin_ctzll(__ballot(best == wbest));  // (keys are unique: the point index is part of them)
    const float wx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bxv), from));
    const float wy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, byv), from));
    const float wz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bzv), from));
#pragma unroll
    for (int i = 0; i < FG_NS; ++i)
      if (lane == src && ii == i) {
        ckey[i] = wbest;
        cx[i] = wx, cy[i] = wy, cz[i] = wz;
      }
  };
#ifdef FG_TIMELINE
  unsigned long long ts[8], tl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int one = 0;
#endif
  for (int j = 1; j < m; ++j) {
    FG_STAMP(0, __float_as_int(sx));
    // ---- the wave's cells this sample can change: the squared distance to the cell's box bounds sqdist3 of every
    // point inside from below (also in fp32: the operations are monotone under rounding), so bound >= current
    // maximum means nothing in the cell changes. Round 1 visits every cell: that is what initialises the keys.
    // (Handing several cells of a wave to 16-lane rows through LDS so that their load latencies overlap measured
    //  20 % SLOWER than taking them one after the other with all 64 lanes; taking them two at a time with both cells'
    //  loads in flight changed nothing, round 4: 0.2 cells per wave and round, two or more in 1 % of them.)
    auto hits = [&](int i) {
      const float dx = fmaxf(fmaxf(blo[i][0] - sx, sx - bhi[i][0]), 0.0f);
      const float dy = fmaxf(fmaxf(blo[i][1] - sy, sy - bhi[i][1]), 0.0f);
      const float dz = fmaxf(fmaxf(blo[i][2] - sz, sz - bhi[i][2]), 0.0f);
      const float cmax = __uint_as_float((unsigned)(ckey[i] >> 32) - 1u);  // (key 0, before round 1: NaN -- a hit)
      return __ballot(((full >> i) & 1u) != 0 && !(sqdist3(dx, dy, dz) >= cmax));
    };
    unsigned long long todo0 = hits(0), todo1 = hits(1), todo2 = hits(2), todo3 = hits(3);  // (scalars: an array went to scratch)
    unsigned long long todo4 = 0, todo5 = 0, todo6 = 0, todo7 = 0;
    if constexpr (FG_NS == 8) todo4 = hits(4), todo5 = hits(5), todo6 = hits(6), todo7 = hits(7);
#define FG_ANY (todo0 | todo1 | todo2 | todo3 | todo4 | todo5 | todo6 | todo7)
    auto pick = [&](int &ii, int &src) {  // (wave-uniform) the next cell to update: (slot, owning lane); false when none is left
#define FG_PICK(I, T)          \
  if (T) {                     \
    ii = I;                    \
    src = __builtin_ctzll(T);  \
    T &= T - 1;                \
    return true;               \
  }
      FG_PICK(0, todo0)
      FG_PICK(1, todo1)
      FG_PICK(2, todo2)
      FG_PICK(3, todo3)
      FG_PICK(4, todo4)
      FG_PICK(5, todo5)
      FG_PICK(6, todo6)
      FG_PICK(7, todo7)
#undef FG_PICK
      return false;
    };
    int ia, la;
#ifdef FG_TIMELINE
    one = __builtin_popcountll(todo0) + __builtin_popcountll(todo1) + __builtin_popcountll(todo2) + __builtin_popcountll(todo3) +
          __builtin_popcountll(todo4) + __builtin_popcountll(todo5) + __builtin_popcountll(todo6) + __builtin_popcountll(todo7);
    FG_STAMP(1, one);
#endif
    if (FG_ANY != 0) {  // (round 1: every cell with a point)
      __builtin_amdgcn_s_setprio(3);             // the round waits for these waves
      while (pick(ia, la)) {  // all 64 lanes recompute the cell with exactly fps_kernel's arithmetic
        const bool last = FG_ANY == 0;
        const int2 rg = crange[fg_cell(wave + FG_NW * (ia >> 2), la + 64 * (ia & 3))];  // (wave-uniform address)
        const int p0 = __builtin_amdgcn_readfirstlane(rg.x), pn = __builtin_amdgcn_readfirstlane(rg.y);
        // FG_NQ points per lane in flight (a surface scan fills few cells of the 16^3 grid: 30 - 100 points per cell, and one
        // 64-point pass at a time paid an L2 trip per pass: 24.6 -> 28.3 ms inside the PVDL sampler, profiles/r04g_fps_grid.txt)
        float4 r[FG_NQ];
        float dold[FG_NQ];
        FG_STAMP(2, p0 + pn);
#pragma unroll
        for (int q = 0; q < FG_NQ; ++q)
          if (64 * q + lane < pn) {
            r[q] = rc[p0 + 64 * q + lane];
            dold[q] = md[p0 + 64 * q + lane];
          }
        // while the records travel: the farthest point among this lane's OTHER cells
        u64 okey = 0;
        float ox = 0.0f, oy = 0.0f, oz = 0.0f;
        if (last) {
#pragma unroll
          for (int i = 0; i < FG_NS; ++i)
            if (ckey[i] > okey && !(lane == la && ia == i)) {
              okey = ckey[i];
              ox = cx[i], oy = cy[i], oz = cz[i];
            }
        }
#ifdef FG_TIMELINE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FG_STAMP(3, p0);
#endif
        u64 best = 0;
        float bx = 0.0f, by = 0.0f, bz = 0.0f;
        for (int i0 = 0;;) {
#pragma unroll
          for (int q = 0; q < FG_NQ; ++q) {
            const int k = i0 + 64 * q + lane;
            if (k < pn) {
              const float d = sqdist3(r[q].x - sx, r[q].y - sy, r[q].z - sz);
              float d2;  // (bare v_min_f32: see fps_kernel)
              asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d), "v"(dold[q]));
              if (d2 != dold[q]) md[p0 + k] = d2;
              const u64 key = fg_key(d2, __float_as_int(r[q].w));
              if (key > best) {
                best = key;
                bx = r[q].x, by = r[q].y, bz = r[q].z;
              }
            }
          }
          i0 += 64 * FG_NQ;
          if (i0 >= pn) break;
#pragma unroll
          for (int q = 0; q < FG_NQ; ++q)
            if (i0 + 64 * q + lane < pn) {
              r[q] = rc[p0 + i0 + 64 * q + lane];
              dold[q] = md[p0 + i0 + 64 * q + lane];
            }
        }
        if (!last) {
          settle(ia, la, best, bx, by, bz);
        } else {
          // the wave's new maximum in ONE reduction: the cell's points and the lanes' other cells are the candidates; the
          // cell's own record (its maximum alone: a second reduction) is brought up to date after the barrier
#ifdef FG_NODEFER
          settle(ia, la, best, bx, by, bz);
#else
          pend = true, pslot = ia, plane = la, pbest = best, pbx = bx, pby = by, pbz = bz;
#endif
          if (okey > best) {
            best = okey;
            bx = ox, by = oy, bz = oz;
          }
          wkey = wave_max_u64(best);
          const int from = __builtin_ctzll(__ballot(best == wkey));  // (an all-empty wave: the zero key, lane 0)
          wkx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bx), from));
          wky = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, by), from));
          wkz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bz), from));
          FG_STAMP(4, __float_as_int(wkz));
        }
      }
      __builtin_amdgcn_s_setprio(0);
    }
    if (lane == 0) {
      // (one 16-byte store: three adjacent floats become a ds_write_b96, which misbehaved beside another stream's matrix
      //  kernels -- voxelize.hip devox_cl_kernel, round 4)
      *(volatile FG_LDS fg_f32x4 *)&sxyz[j & 1][wave][0] = fg_f32x4{wkx, wky, wkz, 0.0f};
      atomicMax((unsigned long long *)&gmax[jm3], (unsigned long long)(wkey | (u64)wave));
    }
    FG_STAMP(5, jm3);
    __syncthreads();
    FG_STAMP(6, jm3);
    const u64 fin = *(volatile FG_LDS u64 *)&gmax[jm3];
    // (three 4-byte reads: with one volatile 16-byte read the compiler took all three coordinates from element 0)
    volatile FG_LDS float *sqp = (volatile FG_LDS float *)&sxyz[j & 1][t & 15][0];
    const float sq[3] = {sqp[0], sqp[1], sqp[2]};
    jm3 = jm3 == 2 ? 0 : jm3 + 1;
    // (the buffer of round j + 2: whoever adds to it has passed the barrier of round j + 1, which thread 0 reaches after this)
    if (t == 0) gmax[jm3 == 2 ? 0 : jm3 + 1] = 0;
    const int ws = __builtin_amdgcn_readfirstlane((int)((unsigned)fin & 15u));  // the winning wave; lane ws holds its point
    sx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sq[0]), ws));
    sy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sq[1]), ws));
    sz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sq[2]), ws));
    if (t == 0) out[j] = fg_key_index(fin);
#undef FG_ANY
#ifdef FG_TIMELINE
    FG_STAMP(7, __float_as_int(sz));
    if (j > 1 && one == 1) {
      for (int i = 0; i < 7; ++i) tl[i] += ts[i + 1] - ts[i];
      tl[7] += 1;
    }
#endif
    if (pend) {  // (needed by this wave's next box tests and its next maximum; off the round's critical path)
      settle(pslot, plane, pbest, pbx, pby, pbz);
      pend = false;
    }
  }
#ifdef FG_TIMELINE
  if (lane == 0 && fg_tl_buf)
    for (int i = 0; i < 8; ++i) fg_tl_buf[((size_t)b * 16 + wave) * 8 + i] = tl[i];
#endif
}

