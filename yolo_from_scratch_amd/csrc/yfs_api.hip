// libyfs.so: version + thread-local error reporting.
#include "yfs_common.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>

static thread_local char g_err[512] = {0};

void yfs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

YFS_API int yfs_version(void) { return 100; }

YFS_API int yfs_last_error(char* buf, size_t n) {
    size_t len = strlen(g_err);
    if (buf && n) {
        size_t c = len < n - 1 ? len : n - 1;
        memcpy(buf, g_err, c);
        buf[c] = 0;
    }
    return (int)len;
}
