// Error reporting, device check and the HIP-event profiling recorder.
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"

namespace mmk {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

int kernel_setup(const void* kern, int threads, int lds_bytes, KernelSetup* out) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, KernelSetup> cache;
  int dev = 0;
  MMK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find({kern, dev});
  if (it == cache.end()) {
    KernelSetup ks{0, 1};
    if (lds_bytes > 64 * 1024) MMK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    MMK_HIP(hipDeviceGetAttribute(&ks.cus, hipDeviceAttributeMultiprocessorCount, dev));
    int occ = 0;
    MMK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, threads, lds_bytes));
    ks.wgs_per_cu = occ > 1 ? occ : 1;
    it = cache.emplace(std::make_pair(kern, dev), ks).first;
  }
  *out = it->second;
  return 0;
}

struct EventPair {
  int id;
  hipEvent_t a, b;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<EventPair> g_events;      // recorded, unresolved
static std::vector<hipEvent_t> g_free;       // recycled events
static int32_t g_count[MMK_K_COUNT];
static double g_ms[MMK_K_COUNT];

static hipEvent_t get_event() {
  if (!g_free.empty()) {
    hipEvent_t e = g_free.back();
    g_free.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

ProfScope::ProfScope(int kernel_id, hipStream_t s) : id(kernel_id), stream(s), slot(nullptr) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  EventPair* p = new EventPair{kernel_id, get_event(), get_event()};
  if (p->a && p->b) {
    (void)hipEventRecord(p->a, s);
    slot = p;
  } else {
    delete p;
  }
}
ProfScope::~ProfScope() {
  if (!slot) return;
  EventPair* p = static_cast<EventPair*>(slot);
  (void)hipEventRecord(p->b, stream);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_events.push_back(*p);
  delete p;
}

ProfEvents::ProfEvents(int kernel_id) : id(kernel_id), start(nullptr), stop(nullptr) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  start = get_event();
  stop = get_event();
  if (!start || !stop) start = stop = nullptr;
}
ProfEvents::~ProfEvents() {
  if (!start) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_events.push_back(EventPair{id, start, stop});
}

static void resolve_locked() {
  for (auto& e : g_events) {
    float ms = 0.f;
    if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      g_count[e.id] += 1;
      g_ms[e.id] += ms;
    }
    g_free.push_back(e.a);
    g_free.push_back(e.b);
  }
  g_events.clear();
}

}  // namespace mmk

using namespace mmk;

extern "C" {

int mmk_abi_version(void) { return MMK_ABI_VERSION; }
const char* mmk_last_error(void) { return g_err.c_str(); }

int mmk_device_check(void) {
  int dev = 0;
  MMK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  MMK_HIP(hipGetDeviceProperties(&prop, dev));
  std::string arch(prop.gcnArchName);
  MMK_REQUIRE(arch.rfind("gfx950", 0) == 0, "this library is built for gfx950 (MI355X) only, found " + arch);
  return 0;
}

int mmk_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (on) {
    resolve_locked();
    for (int k = 0; k < MMK_K_COUNT; ++k) {
      g_count[k] = 0;
      g_ms[k] = 0.0;
    }
  }
  g_prof_on = on != 0;
  return 0;
}

int mmk_profile_read(int32_t* count, double* total_ms) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  resolve_locked();
  for (int k = 0; k < MMK_K_COUNT; ++k) {
    if (count) count[k] = g_count[k];
    if (total_ms) total_ms[k] = g_ms[k];
  }
  return 0;
}

const char* mmk_kernel_name(int id) {
  static const char* names[MMK_K_COUNT] = {
      "match_ids", "pack_rows", "transpose", "sim_stats", "lse_reduce", "loss_combine", "sim_grad", "grad_gemm",
      "grad_finalize", "l2norm", "ijepa_loss_fwd", "ijepa_loss_bwd", "gather_rows", "scatter_rows", "pred_assemble",
      "pred_assemble_bwd", "ema_update", "mask_to_index", "layernorm_fwd", "layernorm_bwd", "activation", "attn_fwd",
      "attn_bwd", "wgrad", "recall_ranks", "clip_fused", "mlp_gemm", "win_attn_fwd", "win_attn_bwd", "clip_bwd_fused"};
  if (id < 0 || id >= MMK_K_COUNT) return "?";
  return names[id];
}

}  // extern "C"
