// Library runtime: error state, launch checks, hipEvent kernel-class profiling, C-ABI GEMM entry.
#include <mutex>
#include <string>
#include <vector>

#include "gemm_core.h"
#include "rtv_internal.h"

namespace rtv {

static thread_local std::string g_err;

int set_error(int code, const char* msg) {
  g_err = msg ? msg : "unknown error";
  if (code > 0) {
    g_err += ": ";
    g_err += hipGetErrorString((hipError_t)code);
  }
  return code == 0 ? -1 : code;
}

int ensure_dynamic_lds(const void* kernel, int bytes, LdsAttr* state, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return set_error(-1, "cannot query the current device");
  const uint64_t bit = 1ull << dev;
  if (state->done.load(std::memory_order_acquire) & bit) return 0;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return set_error((int)e, what);
  state->done.fetch_or(bit, std::memory_order_release);
  return 0;
}

int device_num_cus() {
  static std::atomic<int> cus[64];   // zero-initialised; the CU count of a device never changes
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    n = prop.multiProcessorCount;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

static std::atomic<int64_t> g_dispatch[DK_COUNT];
static const char* const g_dispatch_names[DK_COUNT] = {
    "gemm_kernel<128x128>", "gemm_kernel<other tile config>", "gemm8_kernel<256x256 ping-pong>", "gemm8m_kernel<128x256 ping-pong>",
    "gemm5_kernel<160x256 one wave per SIMD>", "gemm_fp8_kernel<256x256>",
    "attn_fwd_w4_kernel<one wave per SIMD>", "attn_fwd_pp_kernel<four-phase>", "attn_fwd_kernel<lockstep, 256 rows>",
    "attn_fwd_kernel<lockstep, 128 rows>", "attn_combine_kernel<kv split>",
    "conv_halo4p_kernel<persistent>", "conv_halo4_kernel", "conv_halo_kernel", "conv_igemm_kernel"};
void note_kernel(int id) {
  if (id >= 0 && id < DK_COUNT) g_dispatch[id].fetch_add(1, std::memory_order_relaxed);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error((int)e, what);
  return 0;
}

// ---------------------------------------------------------------- profiling
struct ProfRec {
  int cls;
  hipEvent_t start, stop;
  double work;
};
static unsigned g_prof_mask = 0;  // bit c = time launches of class c
// Sampling: an event pair costs the stream ~2.5 us per record (r04: 938 bracketed LayerNorm / RoPE launches cost a block 9 ms, the
// 2.4 k GEMM brackets ~12 ms), so a class may be bracketed every `stride`-th launch only; ALL launches of an enabled class are
// counted (work, launches) so that a caller can scale the sampled time to the whole class.
static int g_prof_stride[PROF_NCLASS] = {1, 1, 1, 1, 1, 1};
static int64_t g_prof_seen[PROF_NCLASS] = {0};
static double g_prof_seen_work[PROF_NCLASS] = {0};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_event_pool;

static hipEvent_t get_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

ProfScope::ProfScope(int cls, hipStream_t s, double work) : slot(-1), stream(s) {
  if (!((g_prof_mask >> cls) & 1u)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_seen_work[cls] += work;
  if (g_prof_seen[cls]++ % g_prof_stride[cls]) return;   // not a sampled launch
  ProfRec r;
  r.cls = cls;
  r.work = work;
  r.start = get_event();
  r.stop = get_event();
  hipEventRecord(r.start, s);
  g_prof.push_back(r);
  slot = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  hipEventRecord(g_prof[slot].stop, stream);
}

}  // namespace rtv

using namespace rtv;

extern "C" {

int rtv_version(void) { return RTV_ABI_VERSION; }

const char* rtv_last_error(void) { return g_err.c_str(); }

int rtv_lab_build(void) {
#ifdef RTV_LAB
  return 1;
#else
  return 0;
#endif
}

int rtv_dispatch_counts(int64_t* counts, int n) {
  for (int i = 0; counts && i < n && i < DK_COUNT; ++i) counts[i] = g_dispatch[i].load(std::memory_order_relaxed);
  return DK_COUNT;
}

const char* rtv_dispatch_name(int id) { return id >= 0 && id < DK_COUNT ? g_dispatch_names[id] : ""; }

int rtv_dispatch_reset(void) {
  for (auto& c : g_dispatch) c.store(0, std::memory_order_relaxed);
  return 0;
}

int rtv_prof_enable(int class_mask) {
  g_prof_mask = (unsigned)class_mask;
  return 0;
}

int rtv_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) {
    g_event_pool.push_back(r.start);
    g_event_pool.push_back(r.stop);
  }
  g_prof.clear();
  for (int c = 0; c < PROF_NCLASS; ++c) {
    g_prof_seen[c] = 0;
    g_prof_seen_work[c] = 0;
  }
  return 0;
}

int rtv_prof_set_stride(int cls, int stride) {
  if (cls < 0 || cls >= PROF_NCLASS || stride < 1) return set_error(-1, "prof_set_stride: class 0..5, stride >= 1");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_stride[cls] = stride;
  return 0;
}

int rtv_prof_bracket_overhead(int n, rtv_stream_t stream_, double* avg_ms) {
  if (n < 1 || n > 4096 || !avg_ms) return set_error(-1, "prof_bracket_overhead: 1 <= n <= 4096, avg_ms != null");
  hipStream_t stream = (hipStream_t)stream_;
  std::vector<hipEvent_t> ev;
  ev.reserve(2 * (size_t)n);
  // every error return destroys the events created so far (they leaked)
  auto fail = [&](const char* what) {
    for (auto& e : ev) (void)hipEventDestroy(e);
    return set_error(-1, what);
  };
  for (int i = 0; i < 2 * n; ++i) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return fail("prof_bracket_overhead: hipEventCreate");
    ev.push_back(e);
  }
  for (int i = 0; i < n; ++i) {
    if (hipEventRecord(ev[2 * i], stream) != hipSuccess || hipEventRecord(ev[2 * i + 1], stream) != hipSuccess)
      return fail("prof_bracket_overhead: hipEventRecord");
  }
  if (hipStreamSynchronize(stream) != hipSuccess) return fail("prof_bracket_overhead: hipStreamSynchronize");
  double tot = 0;
  for (int i = 0; i < n; ++i) {
    float t = 0;
    if (hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]) != hipSuccess) return fail("prof_bracket_overhead: hipEventElapsedTime");
    tot += t;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  *avg_ms = tot / n;
  return 0;
}

int rtv_prof_read_seen(int cls, int64_t* launches, double* work) {
  if (cls < 0 || cls >= PROF_NCLASS) return set_error(-1, "prof_read_seen: class 0..5");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (launches) *launches = g_prof_seen[cls];
  if (work) *work = g_prof_seen_work[cls];
  return 0;
}

int rtv_prof_read(int cls, double* total_ms, int64_t* launches, double* total_work) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double ms = 0, work = 0;
  int64_t n = 0;
  for (auto& r : g_prof) {
    if (r.cls != cls) continue;
    hipError_t e = hipEventSynchronize(r.stop);
    if (e != hipSuccess) return set_error((int)e, "prof_read: hipEventSynchronize");
    float t = 0;
    e = hipEventElapsedTime(&t, r.start, r.stop);
    if (e != hipSuccess) return set_error((int)e, "prof_read: hipEventElapsedTime");
    ms += t;
    work += r.work;
    ++n;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  if (total_work) *total_work = work;
  return 0;
}

int rtv_gemm(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
             const void* bias, int act, const void* gate, int gate_stride, int rows_per_frame, int row_offset,
             const void* residual, int ldr, int dtype, int tile_cfg, rtv_stream_t stream) {
  if (!A || !W || !C) return set_error(-1, "gemm: null operand");
  if (((uintptr_t)A | (uintptr_t)W) & 15) return set_error(-1, "gemm: A/W must be 16-byte aligned");
  if (((uintptr_t)C | (uintptr_t)bias | (uintptr_t)gate | (uintptr_t)residual) & 7)
    return set_error(-1, "gemm: C/bias/gate/residual must be 8-byte aligned");
  if (gate && (gate_stride % 4)) return set_error(-1, "gemm: gate_stride must be a multiple of 4");
  if (act < 0 || act > 2) return set_error(-1, "gemm: unknown activation");
  GemmParams p;
  p.A = (const uint16_t*)A;
  p.W = (const uint16_t*)W;
  p.C = (uint16_t*)C;
  p.lda = lda;
  p.ldw = ldw;
  p.ldc = ldc;
  p.M = M;
  p.N = N;
  p.K = K;
  p.bias = (const uint16_t*)bias;
  p.act = act;
  p.gate = (const uint16_t*)gate;
  p.gate_stride = gate_stride;
  p.rows_per_frame = rows_per_frame;
  p.row_offset = row_offset;
  p.residual = (const uint16_t*)residual;
  p.ldr = ldr;
  p.tiles_m = p.tiles_n = 0;
  return launch_gemm(p, dtype, tile_cfg, (hipStream_t)stream);
}

}  // extern "C"
