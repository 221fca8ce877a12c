// MaxPool2d(2, 2) forward / backward on NHWC bf16 (reference: pool() at yolo_toy_regressor_torch.py:216,
// four call sites in feat).  HBM-bound: 16 B per lane, every byte moved once.  The backward optionally
// applies the LeakyReLU derivative of the producing Conv2D block (its output is the pool's input), which
// saves one full pass over the largest activations of the network.
#include "yfs_common.h"

namespace {
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
constexpr int kThreads = 256;

// idx (optional): per (output pixel, 8-channel group) one uint16 = eight 2-bit window positions (first maximum wins).
// With it the backward needs neither x nor a second argmax: it reads y, dy and 2 bits per element.
__global__ void __launch_bounds__(kThreads) maxpool2_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                                unsigned short* __restrict__ idx, int H, int W,
                                                                int C, long long total /* N*Ho*Wo*C/8 */) {
    const int c8n = C / 8, Ho = H / 2, Wo = W / 2;
    for (long long id = (long long)blockIdx.x * kThreads + threadIdx.x; id < total; id += (long long)gridDim.x * kThreads) {
        const int c8 = (int)(id % c8n);
        long long pix = id / c8n;
        const int wo = (int)(pix % Wo);
        pix /= Wo;
        const int ho = (int)(pix % Ho);
        const long long n = pix / Ho;
        const bf16_t* p = x + (((n * H + 2 * ho) * W + 2 * wo) * (long long)C) + c8 * 8;
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(p + C);
        const bf16x8 c = *reinterpret_cast<const bf16x8*>(p + (long long)W * C);
        const bf16x8 d = *reinterpret_cast<const bf16x8*>(p + (long long)W * C + C);
        bf16x8 o;
        unsigned packed = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float m = (float)a[j];
            bf16_t mv = a[j];
            unsigned bi = 0;
            if ((float)b[j] > m) { m = (float)b[j]; mv = b[j]; bi = 1; }
            if ((float)c[j] > m) { m = (float)c[j]; mv = c[j]; bi = 2; }
            if ((float)d[j] > m) { m = (float)d[j]; mv = d[j]; bi = 3; }
            o[j] = mv;
            packed |= bi << (2 * j);
        }
        *reinterpret_cast<bf16x8*>(y + id * 8) = o;
        if (idx) idx[id] = (unsigned short)packed;
    }
}

// Backward from the saved window positions: dx[window k] = (k == idx) ? dy * (fuse_leaky && !(y > 0) ? slope : 1) : 0.
// y is the pooled output = the value of x at the argmax, so its sign is the LeakyReLU mask of the routed element.
__global__ void __launch_bounds__(kThreads) maxpool2_bwd_idx_kernel(const bf16_t* __restrict__ y, const unsigned short* __restrict__ idx,
                                                                    const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int H,
                                                                    int W, int C, long long total, int fuse_leaky, float slope) {
    const int c8n = C / 8, Ho = H / 2, Wo = W / 2;
    for (long long id = (long long)blockIdx.x * kThreads + threadIdx.x; id < total; id += (long long)gridDim.x * kThreads) {
        const int c8 = (int)(id % c8n);
        long long pix = id / c8n;
        const int wo = (int)(pix % Wo);
        pix /= Wo;
        const int ho = (int)(pix % Ho);
        const long long n = pix / Ho;
        const long long o00 = (((n * H + 2 * ho) * W + 2 * wo) * (long long)C) + c8 * 8;
        const long long offs[4] = {o00, o00 + C, o00 + (long long)W * C, o00 + (long long)W * C + C};
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + id * 8);
        const unsigned packed = idx[id];
        bf16x8 yv;
        if (fuse_leaky) yv = *reinterpret_cast<const bf16x8*>(y + id * 8);
        bf16x8 r[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned bi = (packed >> (2 * j)) & 3u;
            bf16_t gv = g[j];
            if (fuse_leaky && !((float)yv[j] > 0.0f)) gv = (bf16_t)((float)gv * slope);
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k][j] = ((unsigned)k == bi) ? gv : (bf16_t)0.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<bf16x8*>(dx + offs[k]) = r[k];
    }
}

__global__ void __launch_bounds__(kThreads) maxpool2_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                bf16_t* __restrict__ dx, int H, int W, int C, long long total,
                                                                int fuse_leaky, float slope) {
    const int c8n = C / 8, Ho = H / 2, Wo = W / 2;
    for (long long id = (long long)blockIdx.x * kThreads + threadIdx.x; id < total; id += (long long)gridDim.x * kThreads) {
        const int c8 = (int)(id % c8n);
        long long pix = id / c8n;
        const int wo = (int)(pix % Wo);
        pix /= Wo;
        const int ho = (int)(pix % Ho);
        const long long n = pix / Ho;
        const long long o00 = (((n * H + 2 * ho) * W + 2 * wo) * (long long)C) + c8 * 8;
        const long long offs[4] = {o00, o00 + C, o00 + (long long)W * C, o00 + (long long)W * C + C};
        bf16x8 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const bf16x8*>(x + offs[k]);
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + id * 8);
        bf16x8 r[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float m = (float)v[0][j];
            int bi = 0;
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if ((float)v[k][j] > m) { m = (float)v[k][j]; bi = k; }   // first maximum wins (torch / oracle order)
            bf16_t gv = g[j];
            if (fuse_leaky && !(m > 0.0f)) gv = (bf16_t)((float)gv * slope);
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k][j] = (k == bi) ? gv : (bf16_t)0.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<bf16x8*>(dx + offs[k]) = r[k];
    }
}
}  // namespace

static int pool_check(const char* fn, int64_t N, int H, int W, int C) {
    YFS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, YFS_ERR_INVALID_ARG, "%s: non-positive dimension", fn);
    YFS_REQUIRE(H % 2 == 0 && W % 2 == 0, YFS_ERR_UNSUPPORTED, "%s: H=%d, W=%d must be even", fn, H, W);
    YFS_REQUIRE(C % 8 == 0, YFS_ERR_UNSUPPORTED, "%s: C=%d must be a multiple of 8", fn, C);
    return YFS_OK;
}

static int pool_fwd(const char* fn, const void* x, void* y, void* idx, int64_t N, int H, int W, int C, void* stream) {
    YFS_REQUIRE(x && y, YFS_ERR_INVALID_ARG, "%s: null pointer", fn);
    int rc = pool_check(fn, N, H, W, C);
    if (rc) return rc;
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 8);
    long long blk = yfs_cdiv(total, kThreads);
    if (blk > 16384) blk = 16384;
    maxpool2_fwd_kernel<<<dim3((unsigned)blk), dim3(kThreads), 0, (hipStream_t)stream>>>((const bf16_t*)x, (bf16_t*)y,
                                                                                     (unsigned short*)idx, H, W, C, total);
    YFS_CHECK_LAUNCH("maxpool2_fwd_kernel");
    return YFS_OK;
}

YFS_API int yfs_maxpool2_fwd_bf16(const void* x, void* y, int64_t N, int H, int W, int C, void* stream) {
    return pool_fwd("yfs_maxpool2_fwd_bf16", x, y, nullptr, N, H, W, C, stream);
}

YFS_API int yfs_maxpool2_fwd_idx_bf16(const void* x, void* y, void* idx, int64_t N, int H, int W, int C, void* stream) {
    YFS_REQUIRE(idx, YFS_ERR_INVALID_ARG, "yfs_maxpool2_fwd_idx_bf16: null idx");
    return pool_fwd("yfs_maxpool2_fwd_idx_bf16", x, y, idx, N, H, W, C, stream);
}

YFS_API int yfs_maxpool2_bwd_idx_bf16(const void* y, const void* idx, const void* dy, void* dx, int64_t N, int H, int W, int C,
                                      int fuse_leaky, float slope, void* stream) {
    YFS_REQUIRE(y && idx && dy && dx, YFS_ERR_INVALID_ARG, "yfs_maxpool2_bwd_idx_bf16: null pointer");
    int rc = pool_check("yfs_maxpool2_bwd_idx_bf16", N, H, W, C);
    if (rc) return rc;
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 8);
    long long blk = yfs_cdiv(total, kThreads);
    if (blk > 16384) blk = 16384;
    maxpool2_bwd_idx_kernel<<<dim3((unsigned)blk), dim3(kThreads), 0, (hipStream_t)stream>>>(
        (const bf16_t*)y, (const unsigned short*)idx, (const bf16_t*)dy, (bf16_t*)dx, H, W, C, total, fuse_leaky, slope);
    YFS_CHECK_LAUNCH("maxpool2_bwd_idx_kernel");
    return YFS_OK;
}

YFS_API int yfs_maxpool2_bwd_bf16(const void* x, const void* dy, void* dx, int64_t N, int H, int W, int C, int fuse_leaky,
                                  float slope, void* stream) {
    YFS_REQUIRE(x && dy && dx, YFS_ERR_INVALID_ARG, "yfs_maxpool2_bwd_bf16: null pointer");
    int rc = pool_check("yfs_maxpool2_bwd_bf16", N, H, W, C);
    if (rc) return rc;
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 8);
    long long blk = yfs_cdiv(total, kThreads);
    if (blk > 16384) blk = 16384;
    maxpool2_bwd_kernel<<<dim3((unsigned)blk), dim3(kThreads), 0, (hipStream_t)stream>>>((const bf16_t*)x, (const bf16_t*)dy,
                                                                                     (bf16_t*)dx, H, W, C, total, fuse_leaky, slope);
    YFS_CHECK_LAUNCH("maxpool2_bwd_kernel");
    return YFS_OK;
}
