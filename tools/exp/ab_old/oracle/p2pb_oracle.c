/*
 * p2pb_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's CUDA kernels for the P2P-Bridge hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The shipped path (the .hip sources under p2p_bridge_amd/csrc, behind include/p2pb_hip.h) never calls it.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference root;
 * PN2 = third_party/openpoints/cpp/pointnet2_batch/src).
 *
 * Numerics contract (DESIGN.md "Arithmetic contract"):
 *   - the reference is built by nvcc with default flags (-fmad=true, IEEE div/sqrt), so
 *     `a*a + b*b + c*c` is   fmaf(c,c, fmaf(b,b, a*a))   on the device. This file is compiled with
 *     -ffp-contract=off and spells every contraction explicitly with fmaf(), left to right,
 *     exactly as nvcc contracts the reference's expression trees. The HIP kernels use the same
 *     spelled-out sequence, so integer decisions (indices) are bit-identical by construction.
 *   - where the reference's result depends on atomic ordering (float atomicAdd), this oracle sums
 *     in ascending point index; HIP kernels that are deterministic use the same order, the ones
 *     that use atomics are compared with a tolerance.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

#ifdef _OPENMP
#include <omp.h>
#endif
ORC_API void orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#endif
}

static inline float sqdist3(float dx, float dy, float dz) {
  /* nvcc contraction of dx*dx + dy*dy + dz*dz */
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

/* ------------------------------------------------------------------------------------------
 * Voxelization.forward  (models/pvcnn.py:215-231): centre, scale by 2*max-norm, clamp, round.
 * The reference does this with torch reductions whose summation order is backend dependent.
 * The build defines ONE order (SURVEY.md section 7 "hard parts" (ii)): per-axis mean =
 * (256-lane strided double partial sums, then a fixed binary tree) / N; everything after the
 * mean is order independent (max of squared norms, then one sqrt). Identical in the HIP kernel.
 *   coords f32[b,3,n] -> norm f32[b,3,n] (voxel units, clamped to [0,r-1]), vox i32[b,3,n]
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_voxel_coords(int b, int n, int r, int normalize, float eps, const float *coords,
                              float *norm, int *vox) {
#pragma omp parallel for schedule(static)
  for (int bi = 0; bi < b; ++bi) {
    const float *c = coords + (size_t)bi * 3 * n;
    float *o = norm + (size_t)bi * 3 * n;
    int *v = vox + (size_t)bi * 3 * n;
    float mean[3];
    for (int a = 0; a < 3; ++a) {
      double part[256];
      for (int t = 0; t < 256; ++t) {
        double s = 0.0;
        for (int k = t; k < n; k += 256) s += (double)c[a * n + k];
        part[t] = s;
      }
      for (int s = 128; s > 0; s >>= 1)
        for (int t = 0; t < s; ++t) part[t] += part[t + s];
      mean[a] = (float)(part[0] / (double)n);
    }
    float maxsq = 0.0f;
    for (int k = 0; k < n; ++k) {
      float x = c[k] - mean[0], y = c[n + k] - mean[1], z = c[2 * n + k] - mean[2];
      float s = sqdist3(x, y, z);
      if (s > maxsq) maxsq = s;
    }
    float denom = sqrtf(maxsq) * 2.0f + eps;
    for (int a = 0; a < 3; ++a)
      for (int k = 0; k < n; ++k) {
        float t = c[a * n + k] - mean[a];
        if (normalize)
          t = t / denom + 0.5f;
        else
          t = (t + 1.0f) / 2.0f;
        t = t * (float)r;
        t = fminf(fmaxf(t, 0.0f), (float)(r - 1));
        o[a * n + k] = t;
        v[a * n + k] = (int)rintf(t); /* torch.round = half-to-even */
      }
  }
}

/* ------------------------------------------------------------------------------------------
 * avg_voxelize  (PN2/vox_gpu.cu:18-45 grid_stats_kernel, :50-78 avg_voxelize_kernel,
 *                PN2/vox.cpp:17-44 zero-initialised ind/out/cnt)
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_avg_voxelize_fwd(int b, int c, int n, int r, const int *coords, const float *feat,
                                  int *ind, int *cnt, float *out) {
  const int r2 = r * r, r3 = r2 * r;
#pragma omp parallel for schedule(static)
  for (int bi = 0; bi < b; ++bi) {
    const int *co = coords + (size_t)bi * 3 * n;
    const float *f = feat + (size_t)bi * c * n;
    int *id = ind + (size_t)bi * n;
    int *cn = cnt + (size_t)bi * r3;
    float *o = out + (size_t)bi * c * r3;
    memset(cn, 0, sizeof(int) * r3);
    memset(o, 0, sizeof(float) * (size_t)c * r3);
    for (int i = 0; i < n; ++i) {
      id[i] = co[i] * r2 + co[i + n] * r + co[i + n + n]; /* vox_gpu.cu:33 */
      cn[id[i]] += 1;
    }
    for (int i = 0; i < n; ++i) {
      int pos = id[i];
      int cur = cn[pos];
      if (cur > 0) {
        float div = (float)(1.0 / (double)(float)cur); /* vox_gpu.cu:70: double literal 1.0 */
        for (int j = 0; j < c; ++j) o[(size_t)j * r3 + pos] += f[(size_t)j * n + i] * div;
      }
    }
  }
}

/* PN2/vox_gpu.cu:92-120 avg_voxelize_grad_kernel, vox.cpp:55-79 */
ORC_API void orc_avg_voxelize_bwd(int b, int c, int n, int r3, const int *ind, const int *cnt,
                                  const float *gy, float *gx) {
#pragma omp parallel for schedule(static)
  for (int bi = 0; bi < b; ++bi) {
    const int *id = ind + (size_t)bi * n;
    const int *cn = cnt + (size_t)bi * r3;
    const float *g = gy + (size_t)bi * c * r3;
    float *o = gx + (size_t)bi * c * n;
    for (int i = 0; i < n; ++i) {
      int pos = id[i];
      int cur = cn[pos];
      float div = cur > 0 ? (float)(1.0 / (double)(float)cur) : 0.0f;
      for (int j = 0; j < c; ++j)
        o[(size_t)j * n + i] = cur > 0 ? g[(size_t)j * r3 + pos] * div : 0.0f;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * trilinear_devoxelize (PN2/trilinear_devox_gpu.cu:21-108 forward, :123-166 backward,
 *                       PN2/trilinear_devox.cpp:18-60)
 * inds/wgts are written only when training != 0 ([b,8,n]); otherwise may be NULL.
 * ------------------------------------------------------------------------------------------ */
static inline void devox_corner(float x, float y, float z, int r, int idx[8], float w[8]) {
  const int r2 = r * r;
  float xl = floorf(x), yl = floorf(y), zl = floorf(z);
  float xd1 = x - xl, yd1 = y - yl, zd1 = z - zl;
  float xd0 = 1.0f - xd1, yd0 = 1.0f - yd1, zd0 = 1.0f - zd1;
  w[0] = xd0 * yd0 * zd0;
  w[1] = xd0 * yd0 * zd1;
  w[2] = xd0 * yd1 * zd0;
  w[3] = xd0 * yd1 * zd1;
  w[4] = xd1 * yd0 * zd0;
  w[5] = xd1 * yd0 * zd1;
  w[6] = xd1 * yd1 * zd0;
  w[7] = xd1 * yd1 * zd1;
  int xlo = (int)xl, ylo = (int)yl, zlo = (int)zl;
  int xhi = (xd1 > 0) ? -1 : 0, yhi = (yd1 > 0) ? -1 : 0, zhi = (zd1 > 0) ? 1 : 0;
  idx[0] = xlo * r2 + ylo * r + zlo;
  idx[1] = idx[0] + zhi;
  idx[2] = idx[0] + (yhi & r);
  idx[3] = idx[2] + zhi;
  idx[4] = idx[0] + (xhi & r2);
  idx[5] = idx[4] + zhi;
  idx[6] = idx[4] + (yhi & r);
  idx[7] = idx[6] + zhi;
}

ORC_API void orc_trilinear_devox_fwd(int b, int c, int n, int r, int training, const float *coords,
                                     const float *feat, int *inds, float *wgts, float *outs) {
  const int r3 = r * r * r;
#pragma omp parallel for schedule(static)
  for (int bi = 0; bi < b; ++bi) {
    const float *co = coords + (size_t)bi * 3 * n;
    const float *f = feat + (size_t)bi * c * r3;
    float *o = outs + (size_t)bi * c * n;
    for (int i = 0; i < n; ++i) {
      int idx[8];
      float w[8];
      devox_corner(co[i], co[i + n], co[i + 2 * n], r, idx, w);
      if (training) {
        for (int k = 0; k < 8; ++k) {
          wgts[(size_t)bi * 8 * n + (size_t)k * n + i] = w[k];
          inds[(size_t)bi * 8 * n + (size_t)k * n + i] = idx[k];
        }
      }
      for (int j = 0; j < c; ++j) {
        const float *fj = f + (size_t)j * r3;
        float acc = w[0] * fj[idx[0]]; /* nvcc: mul then 7 chained fma, left to right */
        for (int k = 1; k < 8; ++k) acc = fmaf(w[k], fj[idx[k]], acc);
        o[(size_t)j * n + i] = acc;
      }
    }
  }
}

ORC_API void orc_trilinear_devox_bwd(int b, int c, int n, int r3, const int *inds,
                                     const float *wgts, const float *gy, float *gx) {
#pragma omp parallel for schedule(static)
  for (int bi = 0; bi < b; ++bi) {
    const int *id = inds + (size_t)bi * 8 * n;
    const float *w = wgts + (size_t)bi * 8 * n;
    const float *g = gy + (size_t)bi * c * n;
    float *o = gx + (size_t)bi * c * r3;
    memset(o, 0, sizeof(float) * (size_t)c * r3);
    for (int j = 0; j < c; ++j)
      for (int i = 0; i < n; ++i) {
        float gv = g[(size_t)j * n + i];
        for (int k = 0; k < 8; ++k) o[(size_t)j * r3 + id[(size_t)k * n + i]] += w[(size_t)k * n + i] * gv;
      }
  }
}

/* ------------------------------------------------------------------------------------------
 * ball_query (PN2/pvcnn_ball_query_gpu.cu:19-56; zero-init PN2/pvcnn_ball_query.cpp:21-23;
 *             r2 = radius*radius in float, pvcnn_ball_query.cpp:25)
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_ball_query(int b, int n, int m, float r2, int u, const float *centers,
                            const float *points, int *idx) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < m; ++j) {
      const float *p = points + (size_t)bi * 3 * n;
      const float *ce = centers + (size_t)bi * 3 * m;
      int *o = idx + ((size_t)bi * m + j) * u;
      for (int v = 0; v < u; ++v) o[v] = 0;
      float cx = ce[j], cy = ce[j + m], cz = ce[j + 2 * m];
      int cnt = 0;
      for (int k = 0; k < n && cnt < u; ++k) {
        float d2 = sqdist3(cx - p[k], cy - p[k + n], cz - p[k + 2 * n]);
        if (d2 < r2) {
          if (cnt == 0)
            for (int v = 0; v < u; ++v) o[v] = k;
          o[cnt] = k;
          ++cnt;
        }
      }
    }
}

/* grouping (PN2/pvcnn_grouping_gpu.cu:18-38 fwd, :62-83 bwd) */
ORC_API void orc_grouping_fwd(int b, int c, int n, int m, int u, const float *feat, const int *idx,
                              float *out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *f = feat + ((size_t)bi * c + l) * n;
      const int *id = idx + (size_t)bi * m * u;
      float *o = out + ((size_t)bi * c + l) * m * u;
      for (int q = 0; q < m * u; ++q) o[q] = f[id[q]];
    }
}

ORC_API void orc_grouping_bwd(int b, int c, int n, int m, int u, const float *gy, const int *idx,
                              float *gx) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *g = gy + ((size_t)bi * c + l) * m * u;
      const int *id = idx + (size_t)bi * m * u;
      float *o = gx + ((size_t)bi * c + l) * n;
      memset(o, 0, sizeof(float) * n);
      for (int q = 0; q < m * u; ++q) o[id[q]] += g[q];
    }
}

/* gather (PN2/pvcnn_sampling_gpu.cu:17-33 fwd, :55-71 bwd) */
ORC_API void orc_gather_fwd(int b, int c, int n, int m, const float *feat, const int *idx,
                            float *out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *f = feat + ((size_t)bi * c + l) * n;
      const int *id = idx + (size_t)bi * m;
      float *o = out + ((size_t)bi * c + l) * m;
      for (int j = 0; j < m; ++j) o[j] = f[id[j]];
    }
}

ORC_API void orc_gather_bwd(int b, int c, int n, int m, const float *gy, const int *idx, float *gx) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *g = gy + ((size_t)bi * c + l) * m;
      const int *id = idx + (size_t)bi * m;
      float *o = gx + ((size_t)bi * c + l) * n;
      memset(o, 0, sizeof(float) * n);
      for (int j = 0; j < m; ++j) o[id[j]] += g[j];
    }
}

/* ------------------------------------------------------------------------------------------
 * furthest_point_sampling (PN2/pvcnn_sampling_gpu.cu:92-184; distances init 1e38
 * PN2/pvcnn_sampling.cpp:56). Literal emulation of the 512-thread block: thread t scans
 * k = t, t+512, ... keeping the first strictly greater d2, then the shared-memory tree where the
 * left operand survives ties (:170). Net effect: argmax by (d2 desc, k mod 512 asc, k asc).
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_fps(int b, int n, int m, const float *coords, float *dist_ws, int *indices) {
  if (m <= 0) return;
#pragma omp parallel for schedule(static)
  for (int bi = 0; bi < b; ++bi) {
    const float *c = coords + (size_t)bi * 3 * n;
    float *dist = dist_ws + (size_t)bi * n;
    int *out = indices + (size_t)bi * m;
    enum { BS = 512 };
    float dists[BS];
    int dists_i[BS];
    for (int k = 0; k < n; ++k) dist[k] = 1e38f;
    int old = 0;
    out[0] = 0;
    for (int j = 1; j < m; ++j) {
      float x1 = c[old], y1 = c[old + n], z1 = c[old + 2 * n];
      for (int t = 0; t < BS; ++t) {
        int besti = 0;
        float best = -1.0f;
        for (int k = t; k < n; k += BS) {
          float td = dist[k];
          float d = sqdist3(c[k] - x1, c[k + n] - y1, c[k + 2 * n] - z1);
          float d2 = fminf(d, td);
          if (d2 != td) dist[k] = d2;
          if (d2 > best) {
            best = d2;
            besti = k;
          }
        }
        dists[t] = best;
        dists_i[t] = besti;
      }
      for (int u = 0; (1 << u) < BS; ++u)
        for (int t = 0; t < (BS >> (u + 1)); ++t) {
          int i1 = (t * 2) << u, i2 = (t * 2 + 1) << u;
          if (dists[i1] < dists[i2]) {
            dists[i1] = dists[i2];
            dists_i[i1] = dists_i[i2];
          }
        }
      old = dists_i[0];
      out[j] = old;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * three_nearest_neighbors + interpolate
 * (PN2/pvcnn_neighbor_interpolate_gpu.cu:20-80 search+weights, :96-124 interpolate, :154-180 bwd)
 * bests are doubles initialised to 1e40; clamp to [1e-10f, 1e10f]; inverse-squared-distance.
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_three_nn(int b, int n, int m, const float *points, const float *centers,
                          float *weights, int *indices) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < n; ++j) {
      const float *p = points + (size_t)bi * 3 * n;
      const float *ce = centers + (size_t)bi * 3 * m;
      float *w = weights + (size_t)bi * 3 * n;
      int *id = indices + (size_t)bi * 3 * n;
      float ux = p[j], uy = p[j + n], uz = p[j + 2 * n];
      double best0 = 1e40, best1 = 1e40, best2 = 1e40;
      int bi0 = 0, bi1 = 0, bi2 = 0;
      for (int k = 0; k < m; ++k) {
        float d = sqdist3(ux - ce[k], uy - ce[k + m], uz - ce[k + 2 * m]);
        if (d < best2) {
          best2 = d;
          bi2 = k;
          if (d < best1) {
            best2 = best1;
            bi2 = bi1;
            best1 = d;
            bi1 = k;
            if (d < best0) {
              best1 = best0;
              bi1 = bi0;
              best0 = d;
              bi0 = k;
            }
          }
        }
      }
      best0 = fmax(fmin((double)1e10f, best0), (double)1e-10f);
      best1 = fmax(fmin((double)1e10f, best1), (double)1e-10f);
      best2 = fmax(fmin((double)1e10f, best2), (double)1e-10f);
      float d0d1 = (float)(best0 * best1);
      float d0d2 = (float)(best0 * best2);
      float d1d2 = (float)(best1 * best2);
      float inv = 1.0f / (d0d1 + d0d2 + d1d2);
      w[j] = d1d2 * inv;
      id[j] = bi0;
      w[j + n] = d0d2 * inv;
      id[j + n] = bi1;
      w[j + 2 * n] = d0d1 * inv;
      id[j + 2 * n] = bi2;
    }
}

ORC_API void orc_three_interp_fwd(int b, int c, int m, int n, const float *cfeat, const int *indices,
                                  const float *weights, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *f = cfeat + ((size_t)bi * c + l) * m;
      const int *id = indices + (size_t)bi * 3 * n;
      const float *w = weights + (size_t)bi * 3 * n;
      float *o = out + ((size_t)bi * c + l) * n;
      for (int j = 0; j < n; ++j)
        o[j] = fmaf(f[id[j + 2 * n]], w[j + 2 * n], fmaf(f[id[j + n]], w[j + n], f[id[j]] * w[j]));
    }
}

ORC_API void orc_three_interp_bwd(int b, int c, int n, int m, const float *gy, const int *indices,
                                  const float *weights, float *gx) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *g = gy + ((size_t)bi * c + l) * n;
      const int *id = indices + (size_t)bi * 3 * n;
      const float *w = weights + (size_t)bi * 3 * n;
      float *o = gx + ((size_t)bi * c + l) * m;
      memset(o, 0, sizeof(float) * m);
      for (int j = 0; j < n; ++j) {
        o[id[j]] += g[j] * w[j];
        o[id[j + n]] += g[j] * w[j + n];
        o[id[j + 2 * n]] += g[j] * w[j + 2 * n];
      }
    }
}

/* ------------------------------------------------------------------------------------------
 * chamfer_3D (metrics/chamfer3D/chamfer3D.cu:12-133 NmDistanceKernel: strict '<' inside a tile,
 * strict '>' across tiles => lowest index wins ties; :155-175 NmDistanceGradKernel)
 * xyz are POINT-major [b,n,3].
 * ------------------------------------------------------------------------------------------ */
static void nm_distance(int b, int n, const float *xyz, int m, const float *xyz2, float *result,
                        int *result_i) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < n; ++j) {
      const float *q = xyz2 + (size_t)i * m * 3;
      float x1 = xyz[((size_t)i * n + j) * 3 + 0];
      float y1 = xyz[((size_t)i * n + j) * 3 + 1];
      float z1 = xyz[((size_t)i * n + j) * 3 + 2];
      float best = 0;
      int best_i = 0;
      for (int k = 0; k < m; ++k) {
        float d = sqdist3(q[k * 3 + 0] - x1, q[k * 3 + 1] - y1, q[k * 3 + 2] - z1);
        if (k == 0 || d < best) {
          best = d;
          best_i = k;
        }
      }
      result[(size_t)i * n + j] = best;
      result_i[(size_t)i * n + j] = best_i;
    }
}

ORC_API void orc_chamfer_fwd(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1,
                             float *dist2, int *idx1, int *idx2) {
  nm_distance(b, n, xyz1, m, xyz2, dist1, idx1);
  nm_distance(b, m, xyz2, n, xyz1, dist2, idx2);
}

static void nm_distance_grad(int b, int n, const float *xyz1, int m, const float *xyz2,
                             const float *grad_dist1, const int *idx1, float *g1, float *g2) {
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < n; ++j) {
      float x1 = xyz1[((size_t)i * n + j) * 3 + 0];
      float y1 = xyz1[((size_t)i * n + j) * 3 + 1];
      float z1 = xyz1[((size_t)i * n + j) * 3 + 2];
      int j2 = idx1[(size_t)i * n + j];
      float x2 = xyz2[((size_t)i * m + j2) * 3 + 0];
      float y2 = xyz2[((size_t)i * m + j2) * 3 + 1];
      float z2 = xyz2[((size_t)i * m + j2) * 3 + 2];
      float g = grad_dist1[(size_t)i * n + j] * 2;
      g1[((size_t)i * n + j) * 3 + 0] += g * (x1 - x2);
      g1[((size_t)i * n + j) * 3 + 1] += g * (y1 - y2);
      g1[((size_t)i * n + j) * 3 + 2] += g * (z1 - z2);
      g2[((size_t)i * m + j2) * 3 + 0] += -(g * (x1 - x2));
      g2[((size_t)i * m + j2) * 3 + 1] += -(g * (y1 - y2));
      g2[((size_t)i * m + j2) * 3 + 2] += -(g * (z1 - z2));
    }
}

/* gradxyz1/gradxyz2 are accumulated into (caller zero-fills, dist_chamfer_3D.py:77-83) */
ORC_API void orc_chamfer_bwd(int b, int n, int m, const float *xyz1, const float *xyz2,
                             float *gradxyz1, float *gradxyz2, const float *graddist1,
                             const float *graddist2, const int *idx1, const int *idx2) {
  nm_distance_grad(b, n, xyz1, m, xyz2, graddist1, idx1, gradxyz1, gradxyz2);
  nm_distance_grad(b, m, xyz2, n, xyz1, graddist2, idx2, gradxyz2, gradxyz1);
}

/* ------------------------------------------------------------------------------------------
 * PyTorchEMD approxmatch / matchcost / grads (metrics/PyTorchEMD/cuda/emd_kernel.cu:33-175,
 * :211-262, :300-345, :347-375). The reference uses __expf (fast, approximate); this oracle uses
 * expf, so parity on this op is a tolerance (documented in tests/test_emd_parity.py).
 * match is [b,m,n] (match[i*n*m + l*n + k]).
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_approxmatch(int b, int n, int m, const float *xyz1, const float *xyz2, float *match,
                             float *temp /* [b,(n+m)*2] */) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < b; ++i) {
    float *remainL = temp + (size_t)i * (n + m) * 2, *remainR = remainL + n, *ratioL = remainR + m,
          *ratioR = ratioL + n;
    const float *p1 = xyz1 + (size_t)i * n * 3, *p2 = xyz2 + (size_t)i * m * 3;
    float *mt = match + (size_t)i * n * m;
    float multiL, multiR;
    if (n >= m) {
      multiL = 1;
      multiR = (float)(n / m);
    } else {
      multiL = (float)(m / n);
      multiR = 1;
    }
    for (size_t j = 0; j < (size_t)n * m; ++j) mt[j] = 0;
    for (int j = 0; j < n; ++j) remainL[j] = multiL;
    for (int j = 0; j < m; ++j) remainR[j] = multiR;
    for (int j = 7; j >= -2; --j) {
      float level = -powf(4.0f, (float)j);
      if (j == -2) level = 0;
      for (int k = 0; k < n; ++k) {
        float x1 = p1[k * 3], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2];
        float suml = 1e-9f;
        for (int l = 0; l < m; ++l) {
          float d = level * sqdist3(p2[l * 3] - x1, p2[l * 3 + 1] - y1, p2[l * 3 + 2] - z1);
          suml += expf(d) * remainR[l];
        }
        ratioL[k] = remainL[k] / suml;
      }
      for (int l = 0; l < m; ++l) {
        float x2 = p2[l * 3], y2 = p2[l * 3 + 1], z2 = p2[l * 3 + 2];
        float sumr = 0;
        for (int k = 0; k < n; ++k) {
          float w = expf(level * sqdist3(x2 - p1[k * 3], y2 - p1[k * 3 + 1], z2 - p1[k * 3 + 2])) *
                    ratioL[k];
          sumr += w;
        }
        sumr *= remainR[l];
        float consumption = fminf(remainR[l] / (sumr + 1e-9f), 1.0f);
        ratioR[l] = consumption * remainR[l];
        remainR[l] = fmaxf(0.0f, remainR[l] - sumr);
      }
      for (int k = 0; k < n; ++k) {
        float x1 = p1[k * 3], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2];
        float suml = 0;
        float rl = ratioL[k];
        for (int l = 0; l < m; ++l) {
          float w = expf(level * sqdist3(p2[l * 3] - x1, p2[l * 3 + 1] - y1, p2[l * 3 + 2] - z1)) *
                    rl * ratioR[l];
          mt[(size_t)l * n + k] += w;
          suml += w;
        }
        remainL[k] = fmaxf(0.0f, remainL[k] - suml);
      }
    }
  }
}

ORC_API void orc_matchcost(int b, int n, int m, const float *xyz1, const float *xyz2,
                           const float *match, float *out) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < b; ++i) {
    const float *p1 = xyz1 + (size_t)i * n * 3, *p2 = xyz2 + (size_t)i * m * 3;
    const float *mt = match + (size_t)i * n * m;
    double s = 0.0; /* the reference sums in a 512-thread tree; double is the neutral choice */
    for (int k = 0; k < n; ++k)
      for (int l = 0; l < m; ++l) {
        float d = sqdist3(p2[l * 3] - p1[k * 3], p2[l * 3 + 1] - p1[k * 3 + 1],
                          p2[l * 3 + 2] - p1[k * 3 + 2]);
        s += (double)(d * mt[(size_t)l * n + k]);
      }
    out[i] = (float)s;
  }
}

ORC_API void orc_matchcost_bwd(int b, int n, int m, const float *grad_cost, const float *xyz1,
                               const float *xyz2, const float *match, float *grad1, float *grad2) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < b; ++i) {
    const float *p1 = xyz1 + (size_t)i * n * 3, *p2 = xyz2 + (size_t)i * m * 3;
    const float *mt = match + (size_t)i * n * m;
    for (int l = 0; l < n; ++l) { /* matchcostgrad1 :347-375 */
      float x1 = p1[l * 3], y1 = p1[l * 3 + 1], z1 = p1[l * 3 + 2];
      double dx = 0, dy = 0, dz = 0;
      for (int k = 0; k < m; ++k) {
        float d = mt[(size_t)k * n + l] * 2;
        dx += (double)((x1 - p2[k * 3]) * d);
        dy += (double)((y1 - p2[k * 3 + 1]) * d);
        dz += (double)((z1 - p2[k * 3 + 2]) * d);
      }
      grad1[((size_t)i * n + l) * 3 + 0] = (float)dx * grad_cost[i];
      grad1[((size_t)i * n + l) * 3 + 1] = (float)dy * grad_cost[i];
      grad1[((size_t)i * n + l) * 3 + 2] = (float)dz * grad_cost[i];
    }
    for (int k = 0; k < m; ++k) { /* matchcostgrad2 :300-345 */
      float x2 = p2[k * 3], y2 = p2[k * 3 + 1], z2 = p2[k * 3 + 2];
      double sx = 0, sy = 0, sz = 0;
      for (int j = 0; j < n; ++j) {
        float d = mt[(size_t)k * n + j] * 2;
        sx += (double)((x2 - p1[j * 3]) * d);
        sy += (double)((y2 - p1[j * 3 + 1]) * d);
        sz += (double)((z2 - p1[j * 3 + 2]) * d);
      }
      grad2[((size_t)i * m + k) * 3 + 0] = (float)sx * grad_cost[i];
      grad2[((size_t)i * m + k) * 3 + 1] = (float)sy * grad_cost[i];
      grad2[((size_t)i * m + k) * 3 + 2] = (float)sz * grad_cost[i];
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * emd_assignment: auction algorithm (metrics/emd_assignment/emd_assignment/emd_cuda.cu:95-226
 * Bid/GetMax/Assign/CalcDist, host loop :256-268). The reference is data-race dependent in
 * GetMax/Assign (last writer wins); this oracle resolves every race by
 * ASCENDING bidder index j (the highest j with a maximal increment wins GetMax; in the final
 * "last" round later bidders overwrite earlier ones), a valid serialisation of the CUDA program.
 * The per-bidder argmax walks xyz2 in ascending k exactly like a 1-thread-per-bidder Bid launch
 * (the reference splits that walk over thread_per_unass threads and merges; ties may pick a
 * different but equally good k there).
 * State buffers follow emd_module.py:43-54; returns 1 on success, -1 on shape violations (:236-249).
 * ------------------------------------------------------------------------------------------ */
ORC_API int orc_auction_fwd(int b, int n, const float *xyz1, const float *xyz2, float *dist,
                            int *assignment, float *price, int *assignment_inv, int *bid,
                            float *bid_increments, float *max_increments, int *max_idx, float eps,
                            int iters) {
  if (b > 512 || n % 128 != 0) return -1;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < b; ++i) {
    const float *p1 = xyz1 + (size_t)i * n * 3, *p2 = xyz2 + (size_t)i * n * 3;
    int *as = assignment + (size_t)i * n, *inv = assignment_inv + (size_t)i * n;
    int *bd = bid + (size_t)i * n, *mi = max_idx + (size_t)i * n;
    float *pr = price + (size_t)i * n, *binc = bid_increments + (size_t)i * n,
          *minc = max_increments + (size_t)i * n;
    unsigned char *un = (unsigned char *)malloc((size_t)n);
    for (int it = 0; it < iters; ++it) {
      int last = (it == iters - 1);
      /* Bid */
      for (int j = 0; j < n; ++j) {
        if (as[j] != -1) continue;
        float x1 = p1[j * 3], y1 = p1[j * 3 + 1], z1 = p1[j * 3 + 2];
        float best = -1e9f, better = -1e9f;
        int best_i = -1;
        for (int k = 0; k < n; ++k) {
          float x2 = p2[k * 3] - x1, y2 = p2[k * 3 + 1] - y1, z2 = p2[k * 3 + 2] - z1;
          /* emd_cuda.cu:146: 3.0 is a double literal: (3.0 - sqrtf(..) - price) in double */
          float d = (float)((3.0 - (double)sqrtf(sqdist3(x2, y2, z2))) - (double)pr[k]);
          if (d > best) {
            better = best;
            best = d;
            best_i = k;
          } else if (d > better) {
            better = d;
          }
        }
        bd[j] = best_i;
        binc[j] = best - better + eps;
        if (binc[j] > minc[best_i]) minc[best_i] = binc[j];
      }
      /* GetMax */
      for (int j = 0; j < n; ++j) {
        if (as[j] != -1) continue;
        int bid_id = bd[j];
        float bi_ = binc[j], mx = minc[bid_id];
        if ((double)bi_ - 1e-6 <= (double)mx && (double)mx <= (double)bi_ + 1e-6) mi[bid_id] = j;
      }
      /* Assign: snapshot the "unassigned" predicate first (all threads test it before writing in
       * the common schedule), then apply in ascending j */
      for (int j = 0; j < n; ++j) un[j] = (as[j] == -1);
      for (int j = 0; j < n; ++j) {
        if (!un[j]) continue;
        int bid_id = bd[j];
        if (last || mi[bid_id] == j) {
          float bi_ = binc[j];
          int ai = inv[bid_id];
          if (!last && ai != -1) as[ai] = -1;
          inv[bid_id] = j;
          as[j] = bid_id;
          pr[bid_id] += bi_;
          minc[bid_id] = -1e9f;
        }
      }
    }
    free(un);
    for (int j = 0; j < n; ++j) { /* CalcDist :217-226 */
      int k = as[j];
      float dx = p1[j * 3] - p2[k * 3], dy = p1[j * 3 + 1] - p2[k * 3 + 1],
            dz = p1[j * 3 + 2] - p2[k * 3 + 2];
      dist[(size_t)i * n + j] = sqdist3(dx, dy, dz);
    }
  }
  return 1;
}

/* emd_cuda.cu:284-303: gradient wrt xyz1 only; accumulates into gradxyz (caller zero-fills) */
ORC_API void orc_auction_bwd(int b, int n, const float *xyz1, const float *xyz2, float *gradxyz,
                             const float *graddist, const int *idx) {
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < n; ++j) {
      int j2 = idx[(size_t)i * n + j];
      float g = graddist[(size_t)i * n + j] * 2;
      for (int a = 0; a < 3; ++a)
        gradxyz[((size_t)i * n + j) * 3 + a] +=
            g * (xyz1[((size_t)i * n + j) * 3 + a] - xyz2[((size_t)i * n + j2) * 3 + a]);
    }
}

/* ------------------------------------------------------------------------------------------
 * K nearest neighbours, patch-sized K (object patch extraction, denoise_object.py:91).
 * pytorch3d.ops.knn_points is a pip dependency of the reference (not under /root/reference): its contract is
 * "the K smallest squared distances per query point, ascending, and their indices"; the CUDA build accumulates
 * diff*diff per dimension (contracted to fma, like every distance above). Restated as a full sort by
 * (distance, index) -- ties by ascending index. Parity for this function is pinned to that published contract
 * and to tests/test_denoise_* (brute force), not to a golden vector of the reference ("parity unpinned").
 * ------------------------------------------------------------------------------------------ */
typedef struct { float d; int j; } orc_knn_pair;
static int orc_knn_cmp(const void *a, const void *b) {
  const orc_knn_pair *x = (const orc_knn_pair *)a, *y = (const orc_knn_pair *)b;
  if (x->d < y->d) return -1;
  if (x->d > y->d) return 1;
  return (x->j > y->j) - (x->j < y->j);
}
/* query f32[b,s,3], points f32[b,n,3] -> dist2 f32[b,s,k], idx i32[b,s,k] */
ORC_API void orc_knn_points(int b, int s, int n, int k, const float *query, const float *points, float *dist2,
                            int *idx) {
#pragma omp parallel for schedule(dynamic) collapse(2)
  for (int bi = 0; bi < b; ++bi)
    for (int q = 0; q < s; ++q) {
      const float *p = points + (size_t)bi * n * 3;
      const float *qp = query + ((size_t)bi * s + q) * 3;
      orc_knn_pair *pr = (orc_knn_pair *)malloc(sizeof(orc_knn_pair) * (size_t)n);
      for (int j = 0; j < n; ++j) {
        pr[j].d = sqdist3(p[3 * j] - qp[0], p[3 * j + 1] - qp[1], p[3 * j + 2] - qp[2]);
        pr[j].j = j;
      }
      qsort(pr, (size_t)n, sizeof(orc_knn_pair), orc_knn_cmp);
      for (int i = 0; i < k; ++i) {
        dist2[((size_t)bi * s + q) * k + i] = pr[i].d;
        idx[((size_t)bi * s + q) * k + i] = pr[i].j;
      }
      free(pr);
    }
}

/* ------------------------------------------------------------------------------------------
 * Point <-> triangle squared distances (P2M metric: metrics/p2m.py:66,131 -> pytorch3d._C.point_face_dist_forward /
 * face_point_dist_forward). pytorch3d is a pip dependency (absent from /root/reference): restated from its published
 * geometry_utils.cuh (PointTriangle3DistanceForward, IsInsideTriangle, BarycentricCoords3Forward,
 * PointLine3DistanceForward, kEpsilon 1e-8). "Parity unpinned" against pytorch3d itself; pinned to closed-form
 * cases in tests/test_metrics_unit_sphere_*.py. Plain fp32, no contraction (file is built with -ffp-contract=off).
 * ------------------------------------------------------------------------------------------ */
typedef struct { float x, y, z; } orc_v3;
static inline orc_v3 orc_sub(orc_v3 a, orc_v3 b) { orc_v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline float orc_dot(orc_v3 a, orc_v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline orc_v3 orc_cross(orc_v3 a, orc_v3 b) {
  orc_v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
  return r;
}
static float orc_point_segment(orc_v3 p, orc_v3 v0, orc_v3 v1) {
  orc_v3 d = orc_sub(v1, v0);
  float l2 = orc_dot(d, d);
  if (l2 <= 1e-8f) { orc_v3 q = orc_sub(p, v1); return orc_dot(q, q); }
  float t = orc_dot(d, orc_sub(p, v0)) / l2;
  t = fminf(fmaxf(t, 0.0f), 1.0f);
  orc_v3 q = {p.x - (v0.x + t * d.x), p.y - (v0.y + t * d.y), p.z - (v0.z + t * d.z)};
  return orc_dot(q, q);
}
static float orc_point_triangle(orc_v3 p, orc_v3 v0, orc_v3 v1, orc_v3 v2, float min_area) {
  orc_v3 n = orc_cross(orc_sub(v2, v0), orc_sub(v1, v0));
  float nn = sqrtf(orc_dot(n, n));
  float inv = 1.0f / (nn + 1e-8f);
  n.x *= inv; n.y *= inv; n.z *= inv;
  float t = orc_dot(orc_sub(v0, p), n);
  orc_v3 p0 = {p.x + t * n.x, p.y + t * n.y, p.z + t * n.z};
  int inside = 0;
  if (0.5f * nn >= min_area) {
    orc_v3 e0 = orc_sub(v1, v0), e1 = orc_sub(v2, v0), e2 = orc_sub(p0, v0);
    float d00 = orc_dot(e0, e0), d01 = orc_dot(e0, e1), d11 = orc_dot(e1, e1), d20 = orc_dot(e2, e0),
          d21 = orc_dot(e2, e1);
    float denom = d00 * d11 - d01 * d01 + 1e-8f;
    float w1 = (d11 * d20 - d01 * d21) / denom, w2 = (d00 * d21 - d01 * d20) / denom;
    float w0 = 1.0f - w1 - w2;
    inside = (0.0f <= w0 && w0 <= 1.0f) && (0.0f <= w1 && w1 <= 1.0f) && (0.0f <= w2 && w2 <= 1.0f);
  }
  if (inside) return t * t;
  return fminf(fminf(orc_point_segment(p, v0, v1), orc_point_segment(p, v0, v2)), orc_point_segment(p, v1, v2));
}
/* which = 0: per point the closest triangle; which = 1: per triangle the closest point */
ORC_API void orc_point_face(int which, int np, int nt, const float *pts, const float *tris, float min_area,
                            float *dist, int *idx) {
  const int no = which == 0 ? np : nt, ni = which == 0 ? nt : np;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < no; ++i) {
    float best = 3.4e38f;
    int bi = 0;
    for (int k = 0; k < ni; ++k) {
      const float *pp = pts + 3 * (which == 0 ? i : k), *q = tris + 9 * (size_t)(which == 0 ? k : i);
      orc_v3 p = {pp[0], pp[1], pp[2]}, v0 = {q[0], q[1], q[2]}, v1 = {q[3], q[4], q[5]}, v2 = {q[6], q[7], q[8]};
      float d = orc_point_triangle(p, v0, v1, v2, min_area);
      if (d < best) { best = d; bi = k; }
    }
    dist[i] = best;
    idx[i] = bi;
  }
}

/* ------------------------------------------------------------------------------------------
 * Room pipeline: exact radius query standing for sklearn.neighbors.KDTree.query_radius
 * (denoise_room.py:459-464). The KD-tree returns the SET { i : |p_i - c| <= r } in tree order; the contract
 * restated here fixes the order (ascending index) and the arithmetic (fp32 squared distance in the build's
 * fma sequence, compared with r*r) -- third-party code absent from /root/reference: parity unpinned.
 * Two calls: out == NULL counts, else fills out[offsets[s] ...].
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_radius_query(int s, int n, const float *centers, const float *points, float radius, int *counts,
                              const long long *offsets, int *out) {
  const float r2 = radius * radius;
#pragma omp parallel for schedule(dynamic, 4)
  for (int c = 0; c < s; ++c) {
    const float cx = centers[c * 3], cy = centers[c * 3 + 1], cz = centers[c * 3 + 2];
    long long pos = out ? offsets[c] : 0;
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
      const float d = sqdist3(points[(size_t)i * 3] - cx, points[(size_t)i * 3 + 1] - cy, points[(size_t)i * 3 + 2] - cz);
      if (d <= r2) {
        if (out) out[pos + cnt] = i;
        ++cnt;
      }
    }
    if (counts) counts[c] = cnt;
  }
}
