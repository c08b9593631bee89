// Error reporting, device buffers and ABI version for libmbhip.
#include <utility>
#include <mutex>
#include "common.h"
#include <cstdlib>

namespace mb {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
  return MB_EHIP;
}

// The variable is parsed ONCE per distinct value: a call compares the environment string with the cached one
// (the common case -- unset -- is one getenv) and looks the key up in the parsed list.  Tests flip the variable between calls, so the
// string is re-read at every call; threads that mutate the environment while a call runs are not supported (as before).
bool diag_str(const char* key, std::string* value) {
  const char* e = getenv("MBHIP_DIAG");
  if (!e || !*e) return false;
  static thread_local std::string cached;
  static thread_local std::vector<std::pair<std::string, std::string>> parsed;
  if (cached != e) {
    cached = e;
    parsed.clear();
    for (const char* p = e; *p;) {
      const char* end = strchr(p, ',');
      const size_t len = end ? (size_t)(end - p) : strlen(p);
      const char* eq = (const char*)memchr(p, '=', len);
      if (len) parsed.emplace_back(eq ? std::string(p, eq - p) : std::string(p, len), eq ? std::string(eq + 1, len - (eq - p) - 1) : std::string("1"));
      p += len + (end ? 1 : 0);
    }
  }
  for (const auto& kv : parsed)
    if (kv.first == key) {
      if (value) *value = kv.second;
      return true;
    }
  return false;
}

int diag_int(const char* key, int absent) {
  std::string v;
  return diag_str(key, &v) ? atoi(v.c_str()) : absent;
}

int env_int(const char* name, int absent) {
  const char* e = getenv(name);
  return e ? atoi(e) : absent;
}

int pool_stream(int i, hipStream_t* out) {
  static std::mutex mu;
  static hipStream_t pool[64][POOL_STREAMS] = {};
  int dev = 0;
  MB_HIP(hipGetDevice(&dev));
  MB_REQUIRE(out && i >= 0 && i < POOL_STREAMS && dev >= 0 && dev < 64, "pool_stream: stream %d of device %d", i, dev);
  std::lock_guard<std::mutex> lock(mu);
  if (!pool[dev][i]) MB_HIP(hipStreamCreateWithFlags(&pool[dev][i], hipStreamNonBlocking));
  *out = pool[dev][i];
  return MB_OK;
}

int DevBuf::alloc(size_t count) {
  release();
  n = count;
  if (count == 0) return MB_OK;
  MB_HIP(hipMalloc((void**)&p, count * sizeof(float)));
  return MB_OK;
}

int DevBuf::upload(const float* h, size_t count) {
  int rc = alloc(count);
  if (rc) return rc;
  if (count) MB_HIP(hipMemcpy(p, h, count * sizeof(float), hipMemcpyHostToDevice));
  return MB_OK;
}

void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  n = 0;
}

}  // namespace mb

extern "C" const char* mb_last_error(void) { return mb::g_err; }
extern "C" int mb_abi_version(void) { return MB_ABI_VERSION; }
extern "C" int mb_diag_lookup(const char* key, char* out, int n) {
  std::string v;
  if (!key || !mb::diag_str(key, &v)) return -1;
  if (out && n > 0) { strncpy(out, v.c_str(), (size_t)n - 1); out[n - 1] = 0; }
  return (int)v.size();
}
