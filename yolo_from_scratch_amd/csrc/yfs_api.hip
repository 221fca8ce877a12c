// libyfs.so: version, thread-local error reporting, tuning knobs.
#include "yfs_common.h"
#include <atomic>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static thread_local char g_err[512] = {0};

void yfs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

YFS_API int yfs_version(void) { return 300; }

// ---- dispatch query: which conv kernel variant the calling thread's last conv entry point launched ------------------
static thread_local char g_kernel[96] = {0};
void yfs_note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}
YFS_API int yfs_last_kernel(char* buf, size_t n) {
    size_t len = strlen(g_kernel);
    if (buf && n) {
        size_t c = len < n - 1 ? len : n - 1;
        memcpy(buf, g_kernel, c);
        buf[c] = 0;
    }
    return (int)len;
}

YFS_API int yfs_last_error(char* buf, size_t n) {
    size_t len = strlen(g_err);
    if (buf && n) {
        size_t c = len < n - 1 ? len : n - 1;
        memcpy(buf, g_err, c);
        buf[c] = 0;
    }
    return (int)len;
}

// ---- tuning knobs (include/yfs.h: measurement aids; the library's only process-wide mutable state) ------------------
namespace {
std::atomic<int> g_knob[YFS_KNOB_COUNT];
std::once_flag g_knob_once;
const char* const kKnobEnv[YFS_KNOB_COUNT] = {"YFS_IGEMM_TILE", "YFS_WGRAD_TILE", "YFS_WGRAD_SPLITS", "YFS_NMS_FUSED", "YFS_STEM_KERNEL",
                                                   "YFS_IGEMM_PERSIST"};
void knob_init() {   // the environment gives the initial values, read exactly once: a later yfs_set_knob always wins
    std::call_once(g_knob_once, [] {
        for (int k = 0; k < YFS_KNOB_COUNT; ++k) {
            const char* e = getenv(kKnobEnv[k]);
            g_knob[k].store(e ? atoi(e) : 0, std::memory_order_relaxed);
        }
    });
}
}  // namespace

int yfs_knob(int knob) {
    knob_init();
    return (knob >= 0 && knob < YFS_KNOB_COUNT) ? g_knob[knob].load(std::memory_order_relaxed) : 0;
}

YFS_API int yfs_get_knob(int knob) {
    YFS_REQUIRE(knob >= 0 && knob < YFS_KNOB_COUNT, YFS_ERR_INVALID_ARG, "yfs_get_knob: unknown knob %d", knob);
    return yfs_knob(knob);
}

YFS_API int yfs_set_knob(int knob, int value) {
    YFS_REQUIRE(knob >= 0 && knob < YFS_KNOB_COUNT, YFS_ERR_INVALID_ARG, "yfs_set_knob: unknown knob %d", knob);
    knob_init();
    g_knob[knob].store(value, std::memory_order_relaxed);
    return YFS_OK;
}
