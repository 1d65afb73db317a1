// Error reporting for libsagnn.so: a thread-local message behind sagnn_last_error().
#include "common.h"

#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace sagnn {

std::string& last_error_slot() {
  static thread_local std::string slot;
  return slot;
}

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error_slot() = buf;
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s: %s (hipError %d)", what, hipGetErrorString(e), (int)e);
  last_error_slot() = buf;
  return (int)e;
}

// The arithmetic engine of the GEMM-shaped fusion stages: chosen per CALLING THREAD through the C ABI
// (sagnn_set_engine), never through the environment; the default is the f16 x 2 split.
static thread_local int tl_engine = SAGNN_ENGINE_F16X2;
bool force_valu() { return tl_engine == SAGNN_ENGINE_VALU; }
bool force_f32_mfma() { return tl_engine == SAGNN_ENGINE_F32; }

}  // namespace sagnn

extern "C" int sagnn_set_engine(int engine) {
  if (engine != SAGNN_ENGINE_F16X2 && engine != SAGNN_ENGINE_F32 && engine != SAGNN_ENGINE_VALU)
    return sagnn::fail(SAGNN_ERR_ARG, "sagnn_set_engine: unknown engine %d", engine);
  sagnn::tl_engine = engine;
  return SAGNN_OK;
}
extern "C" int sagnn_get_engine(void) { return sagnn::tl_engine; }

namespace sagnn {

int ensure_dynamic_lds(const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> done;   // (device, kernel) -> limit set there
  int dev = 0;
  SAGNN_HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = done[std::make_pair(dev, kernel)];
  if (bytes <= have) return SAGNN_OK;
  SAGNN_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  have = bytes;
  return SAGNN_OK;
}

int cu_count_current() {
  static std::mutex mu;
  static std::map<int, int> cus;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::lock_guard<std::mutex> lock(mu);
  int& c = cus[dev];
  if (c == 0) {
    int v = 0;
    c = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return c;
}

unsigned int* redo_counter() {
  static std::mutex mu;
  static std::map<int, unsigned int*> ctr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  unsigned int*& c = ctr[dev];
  if (!c) {
    if (hipMalloc((void**)&c, sizeof(unsigned int)) != hipSuccess) {
      c = nullptr;
      return nullptr;
    }
    (void)hipMemset(c, 0, sizeof(unsigned int));
  }
  return c;
}

// ---- profiler ---------------------------------------------------------------------------------
namespace {
struct ProfRecord {
  hipEvent_t start, stop;
  int kind;
  int64_t a, b;
};
struct Profiler {
  std::mutex mu;
  std::vector<ProfRecord> rec;  // pre-created events
  int used = 0;
  bool enabled = false;
};
Profiler& profiler() {
  static Profiler p;
  return p;
}
}  // namespace

ProfileScope::ProfileScope(int kind, hipStream_t stream, int64_t a, int64_t b) : slot_(-1), stream_(stream) {
  Profiler& p = profiler();
  if (!p.enabled) return;
  std::lock_guard<std::mutex> lock(p.mu);
  if (!p.enabled || p.used >= (int)p.rec.size()) return;
  slot_ = p.used++;
  p.rec[slot_].kind = kind;
  p.rec[slot_].a = a;
  p.rec[slot_].b = b;
  (void)hipEventRecord(p.rec[slot_].start, stream_);
}

ProfileScope::~ProfileScope() {
  if (slot_ < 0) return;
  (void)hipEventRecord(profiler().rec[slot_].stop, stream_);
}

}  // namespace sagnn

extern "C" int sagnn_profile_enable(int capacity) {
  auto& p = sagnn::profiler();
  std::lock_guard<std::mutex> lock(p.mu);
  for (auto& r : p.rec) {
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  p.rec.clear();
  p.used = 0;
  p.enabled = false;
  if (capacity <= 0) return SAGNN_OK;
  p.rec.resize(capacity);
  for (auto& r : p.rec) {
    SAGNN_HIP_TRY(hipEventCreate(&r.start));
    SAGNN_HIP_TRY(hipEventCreate(&r.stop));
  }
  p.enabled = true;
  return SAGNN_OK;
}

extern "C" int sagnn_profile_read(float* ms, int32_t* kind, int64_t* units_a, int64_t* units_b,
                                  int cap, int* n_out) {
  auto& p = sagnn::profiler();
  std::lock_guard<std::mutex> lock(p.mu);
  if (!n_out) return sagnn::fail(SAGNN_ERR_NULL, "n_out is NULL");
  const int n = p.used < cap ? p.used : cap;
  for (int i = 0; i < n; ++i) {
    SAGNN_HIP_TRY(hipEventSynchronize(p.rec[i].stop));
    float t = 0.f;
    SAGNN_HIP_TRY(hipEventElapsedTime(&t, p.rec[i].start, p.rec[i].stop));
    if (ms) ms[i] = t;
    if (kind) kind[i] = p.rec[i].kind;
    if (units_a) units_a[i] = p.rec[i].a;
    if (units_b) units_b[i] = p.rec[i].b;
  }
  *n_out = n;
  p.used = 0;
  return SAGNN_OK;
}

extern "C" int sagnn_range_redo_count(int64_t* count, int reset) {
  if (!count) return sagnn::fail(SAGNN_ERR_NULL, "count is NULL");
  unsigned int* c = sagnn::redo_counter();
  if (!c) return sagnn::fail(SAGNN_ERR_NOMEM, "no device counter");
  unsigned int v = 0;
  SAGNN_HIP_TRY(hipDeviceSynchronize());
  SAGNN_HIP_TRY(hipMemcpy(&v, c, sizeof v, hipMemcpyDeviceToHost));
  if (reset) SAGNN_HIP_TRY(hipMemset(c, 0, sizeof v));
  *count = (int64_t)v;
  return SAGNN_OK;
}

extern "C" int sagnn_version(void) { return SAGNN_VERSION; }

extern "C" size_t sagnn_last_error(char* buf, size_t cap) {
  const std::string& s = sagnn::last_error_slot();
  if (buf && cap > 0) {
    const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(buf, s.data(), n);
    buf[n] = '\0';
  }
  return s.size();
}
