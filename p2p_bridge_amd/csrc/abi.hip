// abi.hip -- library identification for the C ABI (include/p2pb_hip.h) + the zero-fill helper.
#include "common.h"

extern "C" int p2pb_version(void) { return P2PB_ABI_VERSION; }  // include/p2pb_hip.h; _lib.lib() refuses any other
extern "C" const char *p2pb_target_arch(void) { return "gfx950"; }

// Arithmetic of the split-operand kernels (common.h SPLIT_*): a process-wide default (p2pb_set_split_terms: set_conv_math)
// and a per-THREAD override (p2pb_set_split_terms_thread: `with fused.split_math(...)`, e.g. the bf16x6 data-gradient pass
// that train() runs on the autograd thread) -- so a temporary switch on one thread cannot pair another thread's f16 weight
// pack with a bf16x6 kernel (round-2 advice). Every launcher and pack reads p2pb_split_terms_now().
#include <atomic>
static std::atomic<int> g_split_terms{SPLIT_F16X3};
static thread_local int tl_split_terms = 0;  // 0: no override on this thread
int p2pb_split_terms_now() { return tl_split_terms ? tl_split_terms : g_split_terms.load(std::memory_order_relaxed); }
extern "C" int p2pb_set_split_terms(int terms) {
  if (terms != SPLIT_BF16X6 && terms != SPLIT_F16X3) return P2PB_EINVAL;
  g_split_terms.store(terms, std::memory_order_relaxed);
  return 0;
}
extern "C" int p2pb_set_split_terms_thread(int terms) {  // 0 clears the calling thread's override
  // SPLIT_BF16X3 (two bf16 terms, three products, <= 3 * 2^-18 per product -- above the TF32 the reference trains in): only as
  // a thread override, only for the plain dense forms the training data gradient launches (anything else refuses it)
  if (terms != 0 && terms != SPLIT_BF16X6 && terms != SPLIT_F16X3 && terms != SPLIT_BF16X3) return P2PB_EINVAL;
  tl_split_terms = terms;
  return 0;
}
extern "C" int p2pb_get_split_terms(void) { return p2pb_split_terms_now(); }  // what a launch from THIS thread would use

// Deterministic mode (process-wide): the scatter-add backward passes (devoxelise, grouping, three-NN interpolation) accumulate
// their LDS rows with ONE wave per workgroup, so every destination receives its contributions in program order (ascending source
// index; the lanes of one ds_add_f32 are served in lane order) instead of in the arrival order of 8 waves. Slower (those passes
// 0.6 -> ~4 ms per config-3 step), bit-reproducible from run to run; rows that do not fit the LDS are refused (P2PB_EINVAL)
// rather than sent to the global-atomic kernels.
static std::atomic<int> g_deterministic{0};
bool p2pb_deterministic() { return g_deterministic.load(std::memory_order_relaxed) != 0; }
extern "C" int p2pb_set_deterministic(int on) {
  g_deterministic.store(on ? 1 : 0, std::memory_order_relaxed);
  return 0;
}
extern "C" int p2pb_get_deterministic(void) { return p2pb_deterministic() ? 1 : 0; }

// which form a pointwise launch took, per (cin, cout, positions): a small host-side table behind p2pb_debug_pointwise_form (tests assert
// that the layers of the bench's configuration run the kernels the roofline is quoted on)
#include <mutex>
namespace {
struct FormRow {
  int cin, cout, npos, form;
  unsigned long long n;
};
std::mutex g_form_mu;
FormRow g_forms[128];
int g_nforms = 0;
}  // namespace
void p2pb_note_pointwise_form(int cin, int cout, int npos, int form) {
  std::lock_guard<std::mutex> lk(g_form_mu);
  for (int i = 0; i < g_nforms; ++i)
    if (g_forms[i].cin == cin && g_forms[i].cout == cout && g_forms[i].npos == npos) {
      g_forms[i].form = form;
      ++g_forms[i].n;
      return;
    }
  if (g_nforms < 128) g_forms[g_nforms++] = {cin, cout, npos, form, 1ull};
}
extern "C" int p2pb_debug_pointwise_form(int cin, int cout, int npos, unsigned long long *launches) {
  std::lock_guard<std::mutex> lk(g_form_mu);
  if (launches) *launches = 0;
  if (cin < 0) {
    g_nforms = 0;
    return -1;
  }
  for (int i = 0; i < g_nforms; ++i)
    if (g_forms[i].cin == cin && g_forms[i].cout == cout && g_forms[i].npos == npos) {
      if (launches) *launches = g_forms[i].n;
      return g_forms[i].form;
    }
  return -1;
}

// P2PB_EXPERIMENT="key=value;key=value": the A/B switches of the library (conv_wide_min, am_chunks, pw_wm, pw_pp, fps_mid,
// fps_coop_test_fallback, vox_onepass); callers cache the answer per site.
#include <cstdlib>
#include <cstring>
#include <cctype>
// (same reading as the Python parser, p2p_bridge_amd/_experiment.py: items split at ';', key and value trimmed, keys compared
//  case-insensitively -- "Pw_Pp = 0" and "pw_pp=0" are the same switch on both sides)
long p2pb_experiment_long(const char *key, long dflt) {
  const char *e = getenv("P2PB_EXPERIMENT");
  const size_t kl = strlen(key);
  while (e && *e) {
    while (*e == ';' || isspace((unsigned char)*e)) ++e;
    size_t i = 0;
    while (i < kl && e[i] && tolower((unsigned char)e[i]) == tolower((unsigned char)key[i])) ++i;
    if (i == kl) {
      const char *q = e + kl;
      while (*q == ' ' || *q == '\t') ++q;
      if (*q == '=') return atol(q + 1);  // (atol skips leading blanks itself)
    }
    e = strchr(e, ';');
  }
  return dflt;
}

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

// ---- the GroupNorm finisher handed to the NEXT statistics-producing launch of this thread (common.h GnFinish) ----
// p2pb_gn_finisher_arm stores the descriptor; the producer's launcher takes it (p2pb_gn_finisher_take) and launches
// gn_affine_kernel right behind the producer, in stream order (pointwise.hip pw_finish_behind).
namespace {
thread_local bool tl_fin_armed = false;
thread_local GnFinish tl_fin;
}  // namespace
extern "C" int p2pb_gn_finisher_arm(int groups, double count_per_channel, const float *gamma, const float *beta, const float *style,
                                    int style_stride, float eps, float *scale, float *shift, float *chmean) {
  if (groups <= 0 || !scale || !shift || !(count_per_channel > 0.0) || tl_fin_armed) return P2PB_EINVAL;
  tl_fin.gamma = gamma, tl_fin.beta = beta, tl_fin.style = style, tl_fin.scale = scale, tl_fin.shift = shift, tl_fin.chmean = chmean;
  tl_fin.count_per_channel = count_per_channel, tl_fin.style_stride = style_stride;
  tl_fin.groups = groups, tl_fin.eps = eps;
  tl_fin_armed = true;
  return 0;
}
extern "C" int p2pb_gn_finisher_armed(void) { return tl_fin_armed ? 1 : 0; }
extern "C" void p2pb_gn_finisher_disarm(void) { tl_fin_armed = false; }  // (a caller whose producing launch never happened)
bool p2pb_gn_finisher_take(GnFinish *out) {
  if (!tl_fin_armed) return false;
  *out = tl_fin;
  tl_fin_armed = false;
  return true;
}
