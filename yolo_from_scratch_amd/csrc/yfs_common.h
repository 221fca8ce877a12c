// Shared host/device helpers for libyfs.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/yfs.h"

#define YFS_API extern "C" __attribute__((visibility("default")))

// thread-local error message (yfs_api.hip)
void yfs_set_error(const char* fmt, ...);

#define YFS_REQUIRE(cond, code, ...)          \
    do {                                      \
        if (!(cond)) {                        \
            yfs_set_error(__VA_ARGS__);       \
            return (code);                    \
        }                                     \
    } while (0)

#define YFS_CHECK_LAUNCH(name)                                                    \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            yfs_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
            return YFS_ERR_LAUNCH;                                                \
        }                                                                         \
    } while (0)

static inline int64_t yfs_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device helpers -----------------------------------------------------------------------
#define YFS_WAVE 64

__device__ __forceinline__ float yfs_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;   // lane 0 holds the sum, fixed reduction tree => deterministic
}

// IoU in the op order of iou_xywh_torch (reference yolo_toy_regressor_torch.py:159-182).
// Must stay free of FMA contraction: files that include this for mask decisions are built
// with -ffp-contract=off.
__device__ __forceinline__ float yfs_iou_corners(float px1, float py1, float px2, float py2, float gx1, float gy1,
                                                 float gx2, float gy2) {
    float iw = fminf(px2, gx2) - fmaxf(px1, gx1);
    float ih = fminf(py2, gy2) - fmaxf(py1, gy1);
    iw = iw > 0.0f ? iw : 0.0f;
    ih = ih > 0.0f ? ih : 0.0f;
    float inter = iw * ih;
    float ap = (px2 - px1) * (py2 - py1);
    float ag = (gx2 - gx1) * (gy2 - gy1);
    float uni = ap + ag - inter;
    return uni > 0.0f ? inter / uni : 0.0f;
}

__device__ __forceinline__ float yfs_iou_xywh(float pxc, float pyc, float pw, float ph, float gxc, float gyc, float gw,
                                              float gh) {
    return yfs_iou_corners(pxc - 0.5f * pw, pyc - 0.5f * ph, pxc + 0.5f * pw, pyc + 0.5f * ph, gxc - 0.5f * gw,
                           gyc - 0.5f * gh, gxc + 0.5f * gw, gyc + 0.5f * gh);
}

__device__ __forceinline__ float yfs_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
