// Shared host/device helpers for libyfs.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/yfs.h"

#define YFS_API extern "C" __attribute__((visibility("default")))

// thread-local error message (yfs_api.hip)
void yfs_set_error(const char* fmt, ...);
// records the kernel variant an entry point chose (thread-local; read back with yfs_last_kernel)
void yfs_note_kernel(const char* fmt, ...);
// current value of a tuning knob (yfs_knob_id in include/yfs.h; 0 = automatic choice)
int yfs_knob(int knob);

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

// Stage the global floats [src, src+len) into LDS asynchronously (global -> LDS DMA, `buffer_load ... lds`):
// one wave-instruction moves 1 KiB (lane i lands at chunk base + 16 i), no staging registers, and every
// instruction of the copy is issued before anything is waited for.  A per-thread `load; wait; ds_write` loop
// (what the compiler makes of the obvious copy loop: flat loads with vmcnt(0) after each) pays one HBM round
// trip per 16 bytes and left these HBM-bound kernels 80 % idle (SQ_WAIT_ANY / SQ_WAVE_CYCLES).
// `src` is only 4-byte aligned in general (per-image offsets such as n*21125 floats): the copy starts at the
// enclosing 16-byte boundary and src[i] lands at lds[skew + i] (skew returned).  Bytes past the end are
// written as zeros up to the next 1 KiB boundary, so `lds` must have room for YFS_STAGE_FLOATS(len) floats.
// The caller must execute `yfs_stage_wait()` and a barrier before reading the LDS image.
#define YFS_STAGE_FLOATS(len) ((((len) + 3 + 4) * 4 + 1023) / 1024 * 256)
__device__ __forceinline__ int yfs_stage_floats_dma(const float* __restrict__ src, int len, float* lds, int nwaves) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(src);
    const int skew = (int)((addr >> 2) & 3);
    const unsigned nbytes = (unsigned)(((skew + len + 3) >> 2) << 4);
    const __amdgpu_buffer_rsrc_t r =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(addr - (uintptr_t)skew * 4), 0, (int)nbytes, 0x00020000);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nchunks = (int)((nbytes + 1023u) >> 10);
    unsigned char* lb = reinterpret_cast<unsigned char*>(lds);
    for (int c = wave; c < nchunks; c += nwaves) {
        const unsigned off = (unsigned)c * 1024u + (unsigned)lane * 16u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(lb + c * 1024), 16, off, 0, 0, 0);
    }
    return skew;
}
__device__ __forceinline__ void yfs_stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

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
