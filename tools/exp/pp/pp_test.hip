// standalone harness for pw_pingpong_kernel: the 512 -> 1024 x 8192 x 32 launch against pw_split_kernel (through the
// library's C ABI), outputs compared, both timed with HIP events. Build (container): tools/exp/pp/build.sh; run on the GPU box.
#include "../../../p2p_bridge_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
__device__ __forceinline__ float swishf(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}
#define PWS_TILE (2 * 3 * 2 * 128)
#include "../../../p2p_bridge_amd/csrc/pw_pingpong.h"
int p2pb_g_split_terms = 16;  // (the harness links the library for the ABI calls; this TU's own copy is unused)

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, ci = 512, co = 1024, P = argc > 2 ? atoi(argv[2]) : 8192;
  const int reps = 20;
  std::vector<float> hx((size_t)B * ci * P), hw((size_t)co * ci), hb(co), hsc((size_t)B * ci), hsh((size_t)B * ci);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
  for (auto &v : hx) v = rnd() * 1.7f;
  for (auto &v : hw) v = rnd() * 0.05f;
  for (auto &v : hb) v = rnd();
  for (auto &v : hsc) v = 0.5f + 0.5f * (rnd() + 1.0f);
  for (auto &v : hsh) v = rnd();
  float *x, *w, *bias, *sc, *sh, *st0, *st1, *mm0, *mm1;
  void *wp;
  const int nslots = P / 64;
  CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&bias, co * 4));
  CK(hipMalloc(&sc, hsc.size() * 4)); CK(hipMalloc(&sh, hsh.size() * 4));
  const size_t nst = (size_t)B * nslots * co * 2;
  CK(hipMalloc(&st0, nst * 4)); CK(hipMalloc(&st1, nst * 4)); CK(hipMalloc(&mm0, nst * 4)); CK(hipMalloc(&mm1, nst * 4));
  CK(hipMalloc(&wp, p2pb_pointwise_split_packed_bytes(co, ci)));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hb.data(), co * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sc, hsc.data(), hsc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sh, hsh.data(), hsh.size() * 4, hipMemcpyHostToDevice));
  if (p2pb_pointwise_pack_weights_split(co, ci, w, wp, nullptr)) { printf("pack failed\n"); return 1; }
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto old_k = [&] {
    int rc = p2pb_pointwise_conv_pool_forward(B, ci, co, P, x, wp, bias, nullptr, sc, sh, 1, 4, nullptr, st0, 0, mm0, nullptr);
    if (rc) { printf("old kernel rc %d\n", rc); exit(1); }
  };
  CK(hipFuncSetAttribute((const void *)pw_pingpong_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES));
#define TRACE_ARG
  auto new_k = [&] {
    hipLaunchKernelGGL((pw_pingpong_kernel<true, true>), dim3(P / 256, co / 256, B), dim3(512), PP_LDS_BYTES, 0, ci, co, P, nslots, x,
                       (const u32x4 *)wp, bias, nullptr, sc, sh, 1, nullptr, st1, mm1, 0 TRACE_ARG);
  };
  float ms_old = 0, ms_new = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < 3; ++i) old_k();
    CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) old_k(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_old, e0, e1));
    for (int i = 0; i < 3; ++i) new_k();
    CK(hipGetLastError());
    CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) new_k(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_new, e0, e1));
    const double fl = 2.0 * B * P * (double)ci * co;
    printf("pass %d: pw_split %.4f ms (%.1f TF/s)   pingpong %.4f ms (%.1f TF/s, frac %.3f of 838.9)\n", pass, ms_old / reps,
           fl / (ms_old / reps) / 1e9, ms_new / reps, fl / (ms_new / reps) / 1e9, fl / (ms_new / reps) / 1e9 / 838.9);
  }
  if (getenv("PP_CPUREF")) {  // small shapes: fp64 reference of the new kernel's per-slot sums, error by (slot, 64-channel block)
    std::vector<float> bq2(nst);
    CK(hipMemcpy(bq2.data(), st1, nst * 4, hipMemcpyDeviceToHost));
    std::vector<double> act((size_t)ci * P);
    for (int b = 0; b < 1; ++b) {
      for (int c = 0; c < ci; ++c)
        for (int p = 0; p < P; ++p) {
          const double v = (double)hx[((size_t)b * ci + c) * P + p] * hsc[b * ci + c] + hsh[b * ci + c];
          act[(size_t)c * P + p] = v / (1.0 + exp(-v));
        }
      for (int sl = 0; sl < nslots; ++sl)
        for (int cb = 0; cb < co / 64; ++cb) {
          double worst = 0;
          for (int c = cb * 64; c < cb * 64 + 64; ++c) {
            double s = 0;
            for (int p = sl * 64; p < sl * 64 + 64; ++p) {
              double a = hb[c];
              for (int k = 0; k < ci; ++k) a += (double)hw[(size_t)c * ci + k] * act[(size_t)k * P + p];
              s += a;
            }
            worst = fmax(worst, fabs(s - bq2[(((size_t)b * nslots + sl) * co + c) * 2]));
          }
          printf("%s%8.2e", cb == 0 ? "slot: " : " ", worst);
          if (cb == co / 64 - 1) printf("\n");
        }
    }
  }
  // compare: per-(b, channel) totals of the statistics, and the global {min, max}
  std::vector<float> a(nst), bq(nst), ma(nst), mb(nst);
  CK(hipMemcpy(a.data(), st0, nst * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(bq.data(), st1, nst * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(ma.data(), mm0, nst * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(mb.data(), mm1, nst * 4, hipMemcpyDeviceToHost));
  double worst_s = 0, worst_q = 0; int bad_mm = 0;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < co; ++c) {
      double s0 = 0, s1 = 0, q0 = 0, q1 = 0; float mn0 = INFINITY, mn1 = INFINITY, mx0 = -INFINITY, mx1 = -INFINITY;
      for (int sl = 0; sl < nslots; ++sl) {
        const size_t i = (((size_t)b * nslots + sl) * co + c) * 2;
        s0 += a[i]; s1 += bq[i]; q0 += a[i + 1]; q1 += bq[i + 1];
        mn0 = fminf(mn0, ma[i]); mn1 = fminf(mn1, mb[i]); mx0 = fmaxf(mx0, ma[i + 1]); mx1 = fmaxf(mx1, mb[i + 1]);
      }
      worst_s = fmax(worst_s, fabs(s0 - s1) / (fabs(s0) + 1e-3 * sqrt(q0 * P))); worst_q = fmax(worst_q, fabs(q0 - q1) / q0);
      if (mn0 != mn1 || mx0 != mx1) { if (bad_mm < 5) printf("minmax differs b %d c %d: %g %g vs %g %g\n", b, c, mn0, mx0, mn1, mx1); ++bad_mm; }
    }
  printf("stats totals: worst rel diff sum %.3e  sumsq %.3e ; {min,max} mismatches %d of %d\n", worst_s, worst_q, bad_mm, B * co);
  return bad_mm != 0 || worst_q > 1e-5;
}
