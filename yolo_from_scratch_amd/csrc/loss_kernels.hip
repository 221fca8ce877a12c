// Kernel set 2: fused composite YOLOv1 loss (forward + gradient in one launch) and MSE.
// Spec: assets/yolo_loss_function.png, README.md:97-115 of the reference; conventions in SPEC.md §1.
// The loss slots in where the reference calls loss_fn(preds, yb) (yolo_toy_regressor_torch.py:329).
// Built with -ffp-contract=off so the responsible-box choice (argmax IoU) matches the CPU oracle.
#include "yfs_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxD = 40;   // 5B + C supported by the LDS carve

// lds[skew + i] -> dst[i], i in [0,len): 16-byte stores where source and destination phases agree.
__device__ __forceinline__ void unstage_floats(float* __restrict__ dst, int len, const float* lds, int skew, int tid) {
    const int dskew = (int)((reinterpret_cast<uintptr_t>(dst) >> 2) & 3);
    if (dskew == skew) {
        const int head = (4 - skew) & 3;           // elements before the first aligned float4
        const int h = head < len ? head : len;
        if (tid < h) dst[tid] = lds[skew + tid];
        const int nq = (len - h) >> 2;
        float4* d4 = reinterpret_cast<float4*>(dst + h);
        const float4* l4 = reinterpret_cast<const float4*>(lds + skew + h);
        for (int q = tid; q < nq; q += kThreads) d4[q] = l4[q];
        const int done = h + nq * 4;
        if (tid < len - done) dst[done + tid] = lds[skew + done + tid];
    } else {
        for (int i = tid; i < len; i += kThreads) dst[i] = lds[skew + i];
    }
}

__device__ __forceinline__ float ssqrt(float v) {
    float s = sqrtf(fabsf(v) + 1e-6f);
    return v > 0.0f ? s : (v < 0.0f ? -s : 0.0f);
}
__device__ __forceinline__ float dssqrt(float v) { return v != 0.0f ? 0.5f / sqrtf(fabsf(v) + 1e-6f) : 0.0f; }

// One block = 256 consecutive grid cells of the flattened [N*S*S] list; pred and target are pulled
// into LDS with coalesced 16-byte loads, one thread owns one cell, the gradient overwrites the
// cell's pred slots in LDS and leaves with coalesced 16-byte stores: every HBM byte moves once.
__global__ void __launch_bounds__(kThreads) yolov1_loss_kernel(const float* __restrict__ pred,
                                                               const float* __restrict__ target, int64_t ncells,
                                                               int S, int B, int C, float l_coord, float l_noobj,
                                                               float inv_n, float* __restrict__ grad,
                                                               float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float wsum[4][kThreads / 64];
    const int D = 5 * B + C;
    const int tid = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * kThreads;
    const int nc = (int)((ncells - c0) < kThreads ? (ncells - c0) : kThreads);
    float* lp = lds;
    float* lt = lds + YFS_STAGE_FLOATS(kThreads * D);
    const int sp = yfs_stage_floats_dma(pred + c0 * D, nc * D, lp, kThreads / 64);
    const int st = yfs_stage_floats_dma(target + c0 * D, nc * D, lt, kThreads / 64);
    yfs_stage_wait();
    __syncthreads();

    float coord = 0.0f, objl = 0.0f, noobj = 0.0f, clsl = 0.0f;
    if (tid < nc) {
        float* p = lp + sp + tid * D;          // becomes the gradient in place
        const float* t = lt + st + tid * D;
        const int64_t cell = c0 + tid;
        const int ci = (int)(cell % ((int64_t)S * S));
        const float frow = (float)(ci / S), fcol = (float)(ci % S), fS = (float)S;
        if (t[4] > 0.0f) {
            const float gx = (fcol + t[0]) / fS, gy = (frow + t[1]) / fS, gw = t[2], gh = t[3];
            int best = 0;
            float biou = 0.0f;
            for (int j = 0; j < B; ++j) {
                const float* q = p + 5 * j;
                float v = yfs_iou_xywh((fcol + q[0]) / fS, (frow + q[1]) / fS, q[2], q[3], gx, gy, gw, gh);
                if (j == 0 || v > biou) { biou = v; best = j; }
            }
            for (int j = 0; j < B; ++j) {
                float* q = p + 5 * j;
                if (j == best) {
                    const float dx = q[0] - t[0], dy = q[1] - t[1];
                    const float sw = ssqrt(q[2]) - ssqrt(t[2]), sh = ssqrt(q[3]) - ssqrt(t[3]);
                    const float dc = q[4] - biou;
                    coord += l_coord * (dx * dx + dy * dy) + l_coord * (sw * sw + sh * sh);
                    objl += dc * dc;
                    const float g2 = 2.0f * l_coord * sw * dssqrt(q[2]) * inv_n;
                    const float g3 = 2.0f * l_coord * sh * dssqrt(q[3]) * inv_n;
                    q[0] = 2.0f * l_coord * dx * inv_n;
                    q[1] = 2.0f * l_coord * dy * inv_n;
                    q[2] = g2;
                    q[3] = g3;
                    q[4] = 2.0f * dc * inv_n;
                } else {
                    const float c = q[4];
                    noobj += l_noobj * (c * c);
                    q[0] = 0.0f; q[1] = 0.0f; q[2] = 0.0f; q[3] = 0.0f;
                    q[4] = 2.0f * l_noobj * c * inv_n;
                }
            }
            for (int c = 0; c < C; ++c) {
                const float d = p[5 * B + c] - t[5 * B + c];
                clsl += d * d;
                p[5 * B + c] = 2.0f * d * inv_n;
            }
        } else {
            for (int j = 0; j < B; ++j) {
                float* q = p + 5 * j;
                const float c = q[4];
                noobj += l_noobj * (c * c);
                q[0] = 0.0f; q[1] = 0.0f; q[2] = 0.0f; q[3] = 0.0f;
                q[4] = 2.0f * l_noobj * c * inv_n;
            }
            for (int c = 0; c < C; ++c) p[5 * B + c] = 0.0f;
        }
    }
    // block reduction of the four terms: fixed shuffle tree, then fixed order over the 4 waves
    coord = yfs_wave_sum(coord); objl = yfs_wave_sum(objl); noobj = yfs_wave_sum(noobj); clsl = yfs_wave_sum(clsl);
    if ((tid & 63) == 0) {
        wsum[0][tid >> 6] = coord; wsum[1][tid >> 6] = objl; wsum[2][tid >> 6] = noobj; wsum[3][tid >> 6] = clsl;
    }
    __syncthreads();
    if (tid < 4) partial[(int64_t)blockIdx.x * 4 + tid] = (wsum[tid][0] + wsum[tid][1]) + (wsum[tid][2] + wsum[tid][3]);
    if (grad) unstage_floats(grad + c0 * D, nc * D, lp, sp, tid);
}

// loss_out[1..4] = sum over blocks of partial[:, k] * scale; loss_out[0] = their sum.  One block, wave k -> term k.
__global__ void __launch_bounds__(kThreads) loss_finalize_kernel(const float* __restrict__ partial, int64_t nblk,
                                                                 float scale, float* __restrict__ loss_out) {
    __shared__ float term[4];
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.0f;
    for (int64_t i = lane; i < nblk; i += 64) acc += partial[i * 4 + k];
    acc = yfs_wave_sum(acc);
    if (lane == 0) term[k] = acc * scale;
    __syncthreads();
    if (threadIdx.x == 0) {
        loss_out[1] = term[0]; loss_out[2] = term[1]; loss_out[3] = term[2]; loss_out[4] = term[3];
        loss_out[0] = (term[0] + term[1]) + (term[2] + term[3]);
    }
}

// MSE: grid-stride float4 body; grad = 2 (p - t) / n.
__global__ void __launch_bounds__(kThreads) mse_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                       int64_t n, float inv_n, float* __restrict__ grad,
                                                       float* __restrict__ partial) {
    __shared__ float wsum[kThreads / 64];
    float acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const float d = pred[i] - target[i];
        acc += d * d;
        if (grad) grad[i] = 2.0f * d * inv_n;
    }
    acc = yfs_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ void __launch_bounds__(kThreads) sum_finalize_kernel(const float* __restrict__ partial, int64_t count,
                                                                float scale, float* __restrict__ out) {
    __shared__ float wsum[kThreads / 64];
    float acc = 0.0f;
    for (int64_t i = threadIdx.x; i < count; i += kThreads) acc += partial[i];
    acc = yfs_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) * scale;
}

int64_t mse_blocks(int64_t n) {
    int64_t b = yfs_cdiv(n, kThreads);
    return b < 2048 ? b : 2048;
}

}  // namespace

YFS_API size_t yfs_yolov1_loss_workspace_bytes(int64_t N, int S) {
    if (N <= 0 || S <= 0) return 0;
    return (size_t)yfs_cdiv(N * S * S, kThreads) * 4 * sizeof(float);
}

YFS_API int yfs_yolov1_loss_fwd_bwd_f32(const float* pred, const float* target, int64_t N, int S, int B, int C,
                                        float l_coord, float l_noobj, float* loss_out, float* grad, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    YFS_REQUIRE(pred && target && loss_out, YFS_ERR_INVALID_ARG, "yfs_yolov1_loss_fwd_bwd_f32: null pointer");
    YFS_REQUIRE(N > 0 && S > 0 && B > 0 && C >= 0, YFS_ERR_INVALID_ARG,
                "yfs_yolov1_loss_fwd_bwd_f32: bad shape N=%lld S=%d B=%d C=%d", (long long)N, S, B, C);
    const int D = 5 * B + C;
    YFS_REQUIRE(D <= kMaxD, YFS_ERR_UNSUPPORTED, "yfs_yolov1_loss_fwd_bwd_f32: 5B+C=%d exceeds %d", D, kMaxD);
    const size_t need = yfs_yolov1_loss_workspace_bytes(N, S);
    YFS_REQUIRE(workspace && workspace_bytes >= need, YFS_ERR_WORKSPACE,
                "yfs_yolov1_loss_fwd_bwd_f32: workspace needs %zu bytes, got %zu", need, workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    const int64_t ncells = N * S * S;
    const int64_t nblk = yfs_cdiv(ncells, kThreads);
    const size_t lds = 2 * (size_t)YFS_STAGE_FLOATS(kThreads * D) * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)yolov1_loss_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        YFS_REQUIRE(e == hipSuccess, YFS_ERR_LAUNCH, "yfs_yolov1_loss_fwd_bwd_f32: cannot reserve %zu bytes of LDS", lds);
    }
    yolov1_loss_kernel<<<dim3((unsigned)nblk), dim3(kThreads), lds, s>>>(pred, target, ncells, S, B, C, l_coord, l_noobj,
                                                                       1.0f / (float)N, grad, (float*)workspace);
    YFS_CHECK_LAUNCH("yolov1_loss_kernel");
    loss_finalize_kernel<<<1, kThreads, 0, s>>>((const float*)workspace, nblk, 1.0f / (float)N, loss_out);
    YFS_CHECK_LAUNCH("loss_finalize_kernel");
    return YFS_OK;
}

YFS_API size_t yfs_mse_workspace_bytes(int64_t n) { return n > 0 ? (size_t)mse_blocks(n) * sizeof(float) : 0; }

YFS_API int yfs_mse_fwd_bwd_f32(const float* pred, const float* target, int64_t n, float* loss_out, float* grad,
                                void* workspace, size_t workspace_bytes, void* stream) {
    YFS_REQUIRE(pred && target && loss_out && n > 0, YFS_ERR_INVALID_ARG, "yfs_mse_fwd_bwd_f32: null pointer or n <= 0");
    YFS_REQUIRE(workspace && workspace_bytes >= yfs_mse_workspace_bytes(n), YFS_ERR_WORKSPACE,
                "yfs_mse_fwd_bwd_f32: workspace needs %zu bytes", yfs_mse_workspace_bytes(n));
    hipStream_t s = (hipStream_t)stream;
    const int64_t nblk = mse_blocks(n);
    mse_kernel<<<dim3((unsigned)nblk), dim3(kThreads), 0, s>>>(pred, target, n, 1.0f / (float)n, grad, (float*)workspace);
    YFS_CHECK_LAUNCH("mse_kernel");
    sum_finalize_kernel<<<1, kThreads, 0, s>>>((const float*)workspace, nblk, 1.0f / (float)n, loss_out);
    YFS_CHECK_LAUNCH("sum_finalize_kernel");
    return YFS_OK;
}
