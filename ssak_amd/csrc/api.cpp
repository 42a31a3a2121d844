// Status plumbing shared by every entry point of libssak_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include <string.h>

#include <mutex>
#include <vector>

#include "kernels.h"

static thread_local char g_err[512] = "";

void ssak_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ssak_version(void) { return 510; }  // round 6: ssak_drop_file_cache, plan_tile 129 gone, conv0 statistics as (mean, rstd) (INTEGRATION.md "ABI 510")
extern "C" const char* ssak_last_error(void) { return g_err; }

// ---- optional per-launch timing (bench.py's roofline leg): HIP events around launches, on the launch's own stream ----
// The first slots are the kernel classes of the train step (kernels.h: PROF_*); every GEMM instantiation registers a slot
// under its own name (as rocprofv3 prints it) at its first launch.  An event pair keeps consecutive kernels from overlapping head to tail, so
// bracketing everything costs a few per cent of a step: benchmarks survey all slots in warm-up steps and bracket one slot
// inside their timed region (ssak_prof_enable(stream, 2 + slot)).
// State is PER STREAM (keyed by device + stream): enabling the timing for one caller's stream neither slows nor records the
// launches of another handle / thread.  The fast path of an untimed process is one relaxed atomic load.
#include <atomic>
#include <map>
namespace {
struct ProfRec {
  hipEvent_t e0, e1;
  int slot;
  double work;
};
struct ProfStream {
  int mode = 0;  // 0 off, 1 every launch, 2 + i only slot i, -1 the slots of `want`
  std::vector<char> want;
  std::vector<ProfRec> recs;
};
struct ProfKey {
  int dev;
  hipStream_t st;
  bool operator<(const ProfKey& o) const { return dev != o.dev ? dev < o.dev : st < o.st; }
};
std::mutex g_prof_mu;
std::map<ProfKey, ProfStream> g_prof_streams;
std::atomic<int> g_prof_active{0};  // streams with mode != 0
std::vector<hipEvent_t> g_event_pool;
ProfKey prof_key(hipStream_t st) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return ProfKey{dev, st};
}
hipEvent_t prof_event() {  // g_prof_mu held
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
struct SlotInfo {
  char name[112];
  int bound;
};
SlotInfo g_slots[PROF_MAX_SLOTS];
int g_nslots = 0;
std::mutex g_slot_mu;
int register_locked(const char* name, int bound) {
  for (int i = 0; i < g_nslots; ++i)
    if (!strcmp(g_slots[i].name, name)) return i;
  if (g_nslots >= PROF_MAX_SLOTS) return PROF_MAX_SLOTS - 1;  // (overflow: share the last slot)
  snprintf(g_slots[g_nslots].name, sizeof(g_slots[g_nslots].name), "%s", name);
  g_slots[g_nslots].bound = bound;
  return g_nslots++;
}
void register_classes() {
  static const struct {
    const char* name;
    int bound;
  } kClasses[PROF_CLASS_SLOTS] = {
      {"attn_fwd_kernel (fused attention forward)", SSAK_BOUND_MFMA},
      {"attn_bwd_dq_kernel + attn_bwd_dkv_kernel (fused attention backward)", SSAK_BOUND_MFMA},
      {"ln_fwd_kernel (residual + dropout + LayerNorm)", SSAK_BOUND_HBM},
      {"ln_bwd_kernel (LayerNorm backward + column partials)", SSAK_BOUND_HBM},
      {"conv0_moments_kernel + conv0_channel_stats_kernel + conv0_mfma_kernel (conv0 + GroupNorm + GELU)", SSAK_BOUND_HBM},
      {"adamw_kernel (clip + AdamW + bf16 shadow)", SSAK_BOUND_HBM},
      {"sumsq_kernel (gradient norm)", SSAK_BOUND_HBM},
      {"ctc_lsm + ctc_lat + ctc_grad (CTC loss + gradient)", SSAK_BOUND_LATENCY},
      {"norm_stats + norm_apply (waveform normalise)", SSAK_BOUND_HBM},
      {"row / element-wise helpers (pack, SpecAugment, casts, column sums, GELU', adds, reductions)", SSAK_BOUND_HBM},
      {"positional-conv weight-norm prepare / backward", SSAK_BOUND_HBM},
      {"softmax fwd / bwd (unfused attention fallback)", SSAK_BOUND_HBM},
      {"posconv_direct_kernel (grouped positional convolution forward / input gradient)", SSAK_BOUND_MFMA},
      {"posconv_wgrad_kernel + posconv_wgrad_sum_kernel (grouped positional convolution weight gradient)", SSAK_BOUND_MFMA},
      {"stft_fft_mel_kernel + logmel_finalize_kernel (Whisper log-mel features)", SSAK_BOUND_HBM},
  };
  if (g_nslots == 0)
    for (int i = 0; i < PROF_CLASS_SLOTS; ++i) register_locked(kClasses[i].name, kClasses[i].bound);
}
}  // namespace

int ssak_prof_register(const char* name, int bound) {
  std::lock_guard<std::mutex> lock(g_slot_mu);
  register_classes();
  return register_locked(name, bound);
}

bool ssak_prof_wanted(int slot, hipStream_t st) {
  if (g_prof_active.load(std::memory_order_relaxed) == 0) return false;
  std::lock_guard<std::mutex> lock(g_prof_mu);
  auto it = g_prof_streams.find(prof_key(st));
  if (it == g_prof_streams.end()) return false;
  const ProfStream& ps = it->second;
  return ps.mode == 1 || ps.mode == slot + 2 || (ps.mode == -1 && slot >= 0 && slot < (int)ps.want.size() && ps.want[slot]);
}

ProfScope::ProfScope(int slot, double work, hipStream_t st) : st_(st), slot_(slot), work_(work), on_(ssak_prof_wanted(slot, st)) {
  if (on_) {
    {
      std::lock_guard<std::mutex> lock(g_prof_mu);
      e0_ = prof_event();
    }
    (void)hipEventRecord(e0_, st_);
  }
}
ProfScope::~ProfScope() {
  if (on_) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    hipEvent_t e1 = prof_event();
    (void)hipEventRecord(e1, st_);
    g_prof_streams[prof_key(st_)].recs.push_back(ProfRec{e0_, e1, slot_, work_});
  }
}

extern "C" int ssak_prof_enable(void* stream, int on) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  ProfStream& ps = g_prof_streams[prof_key((hipStream_t)stream)];
  const int mode = on < 0 ? 0 : on;
  if ((ps.mode != 0) != (mode != 0)) g_prof_active.fetch_add(mode != 0 ? 1 : -1, std::memory_order_relaxed);
  ps.mode = mode;
  return SSAK_OK;
}

// time only the launches of the listed slots (a kernel that serves several products has one slot per product)
extern "C" int ssak_prof_enable_slots(void* stream, const int32_t* slots, int n) {
  SSAK_REQUIRE(slots && n > 0, "prof_enable_slots: empty list");
  std::lock_guard<std::mutex> lock(g_prof_mu);
  ProfStream& ps = g_prof_streams[prof_key((hipStream_t)stream)];
  if (ps.mode == 0) g_prof_active.fetch_add(1, std::memory_order_relaxed);
  ps.mode = -1;
  ps.want.assign(PROF_MAX_SLOTS, 0);
  for (int i = 0; i < n; ++i)
    if (slots[i] >= 0 && slots[i] < PROF_MAX_SLOTS) ps.want[slots[i]] = 1;
  return SSAK_OK;
}

extern "C" int ssak_prof_collect(void* stream, ssak_prof_entry* out, int cap) {
  SSAK_REQUIRE(out && cap >= PROF_MAX_SLOTS, "prof_collect: need room for %d entries", PROF_MAX_SLOTS);
  int n;
  {
    std::lock_guard<std::mutex> lock(g_slot_mu);
    register_classes();
    n = g_nslots;
    for (int i = 0; i < n; ++i) {
      snprintf(out[i].name, sizeof(out[i].name), "%s", g_slots[i].name);
      out[i].bound = g_slots[i].bound;
      out[i].launches = 0;
      out[i].total_ms = 0.0;
      out[i].total_flops = 0.0;
    }
  }
  std::vector<ProfRec> recs;
  {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    recs.swap(g_prof_streams[prof_key((hipStream_t)stream)].recs);
  }
  for (ProfRec& r : recs) {
    SSAK_HIP(hipEventSynchronize(r.e1));
    float ms = 0.f;
    SSAK_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
    out[r.slot].launches += 1;
    out[r.slot].total_ms += ms;
    out[r.slot].total_flops += r.work;
  }
  std::lock_guard<std::mutex> lock(g_prof_mu);
  for (ProfRec& r : recs) {
    g_event_pool.push_back(r.e0);
    g_event_pool.push_back(r.e1);
  }
  return n;
}
