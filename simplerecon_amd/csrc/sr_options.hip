// sr_options.hip -- the library's run-time switches as an explicit, thread-safe option table (r05).
//
// r01-r04 read tuning / ablation switches -- and the two fenced split-precision modes, which change the NUMERICS of every call
// -- with getenv() inside the entry points, several of them per launch; hosts flipped them by mutating the process
// environment (setenv vs getenv is a data race in glibc, leaks to child processes, costs an environ scan per launch).
// Now every switch is an entry of this table: atomically readable (sr_option_get), explicitly settable (sr_option_set), and
// initialised ONCE, at the first access, from the environment variable of the same name (so `SR_WINO_XCD=0 python ...` keeps
// working for tuning runs).  After that first access the environment is never read again.  The table is process-wide: an option
// set while another thread launches affects that thread's next launch (documented in include/simplerecon_hip.h).
#include <atomic>
#include <mutex>
#include <stdlib.h>
#include <string.h>

#include "sr_common.h"

namespace {

struct OptDef { const char* name; int dflt; int kind; };   // kind 0: integer, 1: split-precision mode string
const OptDef kDefs[SR_OPT_COUNT] = {
    {"SR_MLP_SPLIT", 0, 1},       {"SR_WINO_SPLIT", 0, 1},     {"SR_WINO_XCD", 1, 0},        {"SR_WINO_STAGGER", 0, 0},
    {"SR_WINO_WG_PER_CU", 2, 0},  {"SR_WINO_NT", 0, 0},        {"SR_WINO_KSPLIT", 0, 0},     {"SR_CONV_WINO", 1, 0},
    {"SR_CONV_TILE", 0, 0},       {"SR_CONV_KSPLIT", 1, 0},    {"SR_MLP_VEC_STORE", 1, 0},   {"SR_MLP_XCD", 1, 0},
    {"SR_MLP_BWD_VALU", 0, 0},    {"SR_T16_XCD", 1, 0},        {"SR_POOL_BW", 0, 0},         {"SR_POOL_XCD", 1, 0},
    {"SR_PW_NT", 0, 0},           {"SR_PW_KS", 0, 0},          {"SR_PT_CFG", -1, 0},         {"SR_PT_KS", 0, 0},
    {"SR_DOT_LDS", 1, 0},         {"SR_DOT_QUAD", 1, 0},       {"SR_DOT_LDS_G", 0, 0},       {"SR_DOT_LDS_CULL", 1, 0},
    {"SR_DOT_LDS_CAP", 634, 0},   {"SR_GEMM_AUTOTUNE", 1, 0},  {"SR_UPSAMPLE_QUAD", 1, 0},   {"SR_POOL_STREAM", 1, 0},
    {"SR_MLP_RESERVE_CUS", 0, 0},
};
std::atomic<int> g_val[SR_OPT_COUNT];
std::once_flag g_once;

int parse(const OptDef& d, const char* e) {
  if (d.kind == 1) {
    if (!e || !*e || !strcmp(e, "0") || !strcmp(e, "off") || !strcmp(e, "fp32")) return 0;
    if (!strcmp(e, "bf16")) return 1;
    if (!strcmp(e, "f16") || !strcmp(e, "fp16")) return 2;
    return -1;   // unknown: the entry points that honour the mode refuse to run (SR_ERR_INVALID_ARGUMENT)
  }
  return (e && *e) ? atoi(e) : d.dflt;
}

void init() {
  for (int i = 0; i < SR_OPT_COUNT; ++i) {
    const char* e = getenv(kDefs[i].name);
    g_val[i].store(e ? parse(kDefs[i], e) : kDefs[i].dflt, std::memory_order_relaxed);
  }
}

}  // namespace

int sr_opt(int id) {
  std::call_once(g_once, init);
  return g_val[id].load(std::memory_order_relaxed);
}

extern "C" int sr_option_count(void) { return SR_OPT_COUNT; }

extern "C" const char* sr_option_name(int id) { return (id >= 0 && id < SR_OPT_COUNT) ? kDefs[id].name : nullptr; }

extern "C" int sr_option_id(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < SR_OPT_COUNT; ++i)
    if (!strcmp(kDefs[i].name, name)) return i;
  return -1;
}

extern "C" int sr_option_get(int id, int* value) {
  if (id < 0 || id >= SR_OPT_COUNT || !value) return SR_ERR_INVALID_ARGUMENT;
  *value = sr_opt(id);
  return SR_OK;
}

extern "C" int sr_option_set(int id, int value, int* previous) {
  if (id < 0 || id >= SR_OPT_COUNT) return SR_ERR_INVALID_ARGUMENT;
  std::call_once(g_once, init);
  const int old = g_val[id].exchange(value, std::memory_order_relaxed);
  if (previous) *previous = old;
  return SR_OK;
}

extern "C" int sr_option_default(int id, int* value) {
  if (id < 0 || id >= SR_OPT_COUNT || !value) return SR_ERR_INVALID_ARGUMENT;
  *value = kDefs[id].dflt;
  return SR_OK;
}

// CU count of the current device, cached per device index (launch plans of every translation unit ask this; ADVICE r05: a
// process-wide static answered with the first device's count for all of them)
int sr_device_cus() {
  constexpr int kMaxDevices = 64;
  static std::atomic<int> cus[kMaxDevices];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
  int v = cus[dev].load(std::memory_order_relaxed);
  if (v == 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cus[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}
