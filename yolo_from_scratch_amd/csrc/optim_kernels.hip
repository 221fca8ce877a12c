// clip_grad_norm_(max_norm) + SGD(momentum, weight_decay) as two passes over flat fp32 arenas
// (reference: yolo_toy_regressor_torch.py:331-335 and :378-381).  HBM-bound; float4 per lane.
#include "yfs_common.h"

namespace {
constexpr int kThreads = 256;
constexpr int kMaxBlocks = 4096;

__global__ void __launch_bounds__(kThreads) sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ partial) {
    __shared__ float wsum[kThreads / 64];
    float acc = 0.f;
    const long long n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long long)gridDim.x * kThreads) {
        const float4 v = g4[i];
        acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0) {
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += kThreads) acc += g[i] * g[i];
    }
    acc = yfs_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ void __launch_bounds__(kThreads) sumsq_finalize_kernel(const float* __restrict__ partial, int count, float* __restrict__ out) {
    __shared__ float wsum[kThreads / 64];
    float acc = 0.f;
    for (int i = threadIdx.x; i < count; i += kThreads) acc += partial[i];
    acc = yfs_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ void __launch_bounds__(kThreads) sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                       long long n, const float* __restrict__ sumsq, float max_norm, float lr,
                                                       float momentum, float wd, int first_step,
                                                       __bf16* __restrict__ p16) {
    float coef = 1.0f;
    if (sumsq && max_norm > 0.0f) {
        const float c = max_norm / (sqrtf(sumsq[0]) + 1e-6f);   // clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max=1)
        coef = c < 1.0f ? c : 1.0f;
    }
    const long long n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* b4 = reinterpret_cast<float4*>(buf);
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long long)gridDim.x * kThreads) {
        float4 pv = p4[i];
        const float4 gv = g4[i];
        float4 bv = first_step ? make_float4(0.f, 0.f, 0.f, 0.f) : b4[i];
        const float d0 = gv.x * coef + wd * pv.x, d1 = gv.y * coef + wd * pv.y, d2 = gv.z * coef + wd * pv.z,
                    d3 = gv.w * coef + wd * pv.w;
        bv.x = first_step ? d0 : momentum * bv.x + d0;
        bv.y = first_step ? d1 : momentum * bv.y + d1;
        bv.z = first_step ? d2 : momentum * bv.z + d2;
        bv.w = first_step ? d3 : momentum * bv.w + d3;
        pv.x -= lr * bv.x; pv.y -= lr * bv.y; pv.z -= lr * bv.z; pv.w -= lr * bv.w;
        b4[i] = bv;
        p4[i] = pv;
        if (p16) {   // bf16 shadow of the updated parameters: what the conv / linear kernels read next step
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
            bf16x4_t o;
            o[0] = (__bf16)pv.x; o[1] = (__bf16)pv.y; o[2] = (__bf16)pv.z; o[3] = (__bf16)pv.w;
            reinterpret_cast<bf16x4_t*>(p16)[i] = o;
        }
    }
    if (blockIdx.x == 0) {
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += kThreads) {
            const float d = g[i] * coef + wd * p[i];
            const float b = first_step ? d : momentum * buf[i] + d;
            buf[i] = b;
            const float pn = p[i] - lr * b;
            p[i] = pn;
            if (p16) p16[i] = (__bf16)pn;
        }
    }
}

int nblocks(long long n) {
    long long b = yfs_cdiv(n >> 2 > 0 ? n >> 2 : 1, kThreads);
    return (int)(b < kMaxBlocks ? b : kMaxBlocks);
}
}  // namespace

YFS_API size_t yfs_sumsq_workspace_bytes(int64_t n) { return n > 0 ? (size_t)nblocks(n) * sizeof(float) : 0; }

YFS_API int yfs_sumsq_f32(const float* g, int64_t n, float* sumsq_out, void* workspace, size_t workspace_bytes, void* stream) {
    YFS_REQUIRE(g && sumsq_out && n > 0, YFS_ERR_INVALID_ARG, "yfs_sumsq_f32: null pointer or n <= 0");
    YFS_REQUIRE((reinterpret_cast<uintptr_t>(g) & 15) == 0, YFS_ERR_INVALID_ARG, "yfs_sumsq_f32: g must be 16-byte aligned");
    YFS_REQUIRE(workspace && workspace_bytes >= yfs_sumsq_workspace_bytes(n), YFS_ERR_WORKSPACE,
                "yfs_sumsq_f32: workspace needs %zu bytes", yfs_sumsq_workspace_bytes(n));
    hipStream_t s = (hipStream_t)stream;
    const int nb = nblocks(n);
    sumsq_kernel<<<dim3(nb), dim3(kThreads), 0, s>>>(g, n, (float*)workspace);
    YFS_CHECK_LAUNCH("sumsq_kernel");
    sumsq_finalize_kernel<<<1, kThreads, 0, s>>>((const float*)workspace, nb, sumsq_out);
    YFS_CHECK_LAUNCH("sumsq_finalize_kernel");
    return YFS_OK;
}

YFS_API int yfs_sgd_step_f32(float* p, const float* g, float* buf, int64_t n, const float* sumsq, float max_norm, float lr,
                             float momentum, float weight_decay, int first_step, void* p_bf16, void* stream) {
    YFS_REQUIRE(p && g && buf && n > 0, YFS_ERR_INVALID_ARG, "yfs_sgd_step_f32: null pointer or n <= 0");
    YFS_REQUIRE(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(buf)) & 15) == 0,
                YFS_ERR_INVALID_ARG, "yfs_sgd_step_f32: p, g and buf must be 16-byte aligned");
    sgd_kernel<<<dim3(nblocks(n)), dim3(kThreads), 0, (hipStream_t)stream>>>(p, g, buf, n, sumsq, max_norm, lr, momentum,
                                                                          weight_decay, first_step, (__bf16*)p_bf16);
    YFS_CHECK_LAUNCH("sgd_kernel");
    return YFS_OK;
}
