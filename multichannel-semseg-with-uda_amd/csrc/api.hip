// Version and error reporting of the C ABI.
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void mcdseg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int mcdseg_version(void) { return MCDSEG_VERSION; }
extern "C" const char* mcdseg_last_error(void) { return g_err; }

// ---- the option table (options.h): process-wide atomics -- a backward pass runs on autograd's worker threads, so a thread-local table
// set by the thread that built the graph would not reach the kernels it is meant for
#include <atomic>

#include "options.h"

namespace {
struct OptEntry {
  const char* name;
  int64_t def;
};
const OptEntry kOptions[MCD_OPT_COUNT] = {
#define X(name, def) {#name, def},
    MCD_OPTIONS(X)
#undef X
};
std::atomic<int64_t> g_opt[MCD_OPT_COUNT] = {
#define X(name, def) {def},
    MCD_OPTIONS(X)
#undef X
};
int opt_index(const char* name) {
  if (name == nullptr) return -1;
  if (strncmp(name, "MCDSEG_", 7) == 0) name += 7;  // (the environment variable's spelling is accepted too)
  for (int i = 0; i < MCD_OPT_COUNT; ++i)
    if (strcmp(kOptions[i].name, name) == 0) return i;
  return -1;
}
}  // namespace

int64_t mcd_opt(McdOpt which) { return g_opt[which].load(std::memory_order_relaxed); }

extern "C" int32_t mcdseg_option_count(void) { return MCD_OPT_COUNT; }
extern "C" const char* mcdseg_option_name(int32_t index) { return index >= 0 && index < MCD_OPT_COUNT ? kOptions[index].name : nullptr; }

extern "C" int mcdseg_set_option(const char* name, int64_t value) {
  const int i = opt_index(name);
  MCD_REQUIRE(i >= 0, "set_option: unknown option %s", name ? name : "(null)");
  g_opt[i].store(value, std::memory_order_relaxed);
  return 0;
}

extern "C" int mcdseg_get_option(const char* name, int64_t* value, int64_t* default_value) {
  const int i = opt_index(name);
  MCD_REQUIRE(i >= 0, "get_option: unknown option %s", name ? name : "(null)");
  if (value) *value = g_opt[i].load(std::memory_order_relaxed);
  if (default_value) *default_value = kOptions[i].def;
  return 0;
}
