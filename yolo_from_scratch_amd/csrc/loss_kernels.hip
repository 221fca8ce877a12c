// Kernel set 2: fused composite YOLOv1 loss (forward + gradient in one launch) and MSE.
// Spec: assets/yolo_loss_function.png, README.md:97-115 of the reference; conventions in SPEC.md §1.
// The loss slots in where the reference calls loss_fn(preds, yb) (yolo_toy_regressor_torch.py:329).
// Built with -ffp-contract=off so the responsible-box choice (argmax IoU) matches the CPU oracle.
#include "yfs_common.h"
#include <atomic>

// Scratch of yfs_yolov1_loss_noalloc_f32 (the entry point for callers that cannot allocate): the per-block partial
// sums of the four loss terms.  One copy per GPU the library runs on, held until the process ends.  Host code never
// uses this name as an address: noalloc_scratch() resolves it per device with hipGetSymbolAddress.
constexpr int64_t kNoAllocMaxBlocks = 65536;                         // x 256 cells = 16,777,216 grid cells
static __device__ __align__(256) float scratch[kNoAllocMaxBlocks * 4];   // 1 MiB

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

// ---- single-box losses of the reference's practice scripts ------------------------------------------------------------
// ciou_loss  practice/yolo_toy_regressor_3_torch.py:121-138 (+ xywh_to_xyxy :106-110, box_iou_xyxy :112-119): one thread
// per box pair; forward in the reference's op order, backward by hand in reverse (alpha is a constant: torch.no_grad).
// torch.maximum / minimum give half of the gradient to each side on exact ties; clamp(min=m) passes it where x >= m.
__device__ __forceinline__ float tie_ge(float a, float b) { return a > b ? 1.0f : (a == b ? 0.5f : 0.0f); }

__global__ void __launch_bounds__(kThreads) ciou_kernel(const float4* __restrict__ pred, const float4* __restrict__ gt, int64_t n,
                                                        float* __restrict__ loss, float* __restrict__ iou_out,
                                                        float* __restrict__ ciou_out, float4* __restrict__ grad) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const float4 pb = pred[i], gb = gt[i];
    const float p1x = pb.x - pb.z / 2.0f, p1y = pb.y - pb.w / 2.0f, p2x = pb.x + pb.z / 2.0f, p2y = pb.y + pb.w / 2.0f;
    const float g1x = gb.x - gb.z / 2.0f, g1y = gb.y - gb.w / 2.0f, g2x = gb.x + gb.z / 2.0f, g2y = gb.y + gb.w / 2.0f;
    const float tlx = fmaxf(p1x, g1x), tly = fmaxf(p1y, g1y), brx = fminf(p2x, g2x), bry = fminf(p2y, g2y);
    const float dwx = brx - tlx, dwy = bry - tly;
    const float iw = dwx > 0.0f ? dwx : 0.0f, ih = dwy > 0.0f ? dwy : 0.0f;
    const float inter = iw * ih;
    const float pwr = p2x - p1x, phr = p2y - p1y, gwr = g2x - g1x, ghr = g2y - g1y;
    const float wa = pwr > 0.0f ? pwr : 0.0f, ha = phr > 0.0f ? phr : 0.0f;
    const float wb = gwr > 0.0f ? gwr : 0.0f, hb = ghr > 0.0f ? ghr : 0.0f;
    const float uni = wa * ha + wb * hb - inter + 1e-9f;
    const float iou = inter / uni;
    const float pcx = (p1x + p2x) / 2.0f, pcy = (p1y + p2y) / 2.0f, gcx = (g1x + g2x) / 2.0f, gcy = (g1y + g2y) / 2.0f;
    const float ddx = pcx - gcx, ddy = pcy - gcy;
    const float cd = ddx * ddx + ddy * ddy;
    const float etlx = fminf(p1x, g1x), etly = fminf(p1y, g1y), ebrx = fmaxf(p2x, g2x), ebry = fmaxf(p2y, g2y);
    const float ex = ebrx - etlx, ey = ebry - etly;
    const float c2 = ex * ex + ey * ey + 1e-9f;
    const float pw = pwr > 1e-9f ? pwr : 1e-9f, ph = phr > 1e-9f ? phr : 1e-9f;
    const float gw = gwr > 1e-9f ? gwr : 1e-9f, gh = ghr > 1e-9f ? ghr : 1e-9f;
    const float kv = 4.0f / (3.14159265358979323846f * 3.14159265358979323846f);
    const float r = pw / ph;
    const float A = atanf(gw / gh) - atanf(r);
    const float v = kv * (A * A);
    const float alpha = v / (1.0f - iou + v + 1e-9f);
    const float ciou = iou - cd / c2 - alpha * v;
    const float one_m = 1.0f - ciou;
    loss[i] = one_m > 0.0f ? one_m : 0.0f;
    if (iou_out) iou_out[i] = iou;
    if (ciou_out) ciou_out[i] = ciou;
    if (!grad) return;
    // ---- backward of sum(loss) ----
    const float dciou = one_m >= 0.0f ? -1.0f : 0.0f;
    float d1x = 0.f, d1y = 0.f, d2x = 0.f, d2y = 0.f;                 // gradients of the pred corners
    // v = kv * A^2, A = const - atan(pw / ph)
    const float dA = -dciou * alpha * kv * 2.0f * A;
    const float dr = -dA / (1.0f + r * r);
    const float dpw = (pwr >= 1e-9f) ? dr / ph : 0.0f;
    const float dph = (phr >= 1e-9f) ? -dr * pw / (ph * ph) : 0.0f;
    d2x += dpw; d1x -= dpw; d2y += dph; d1y -= dph;
    // - cd / c2
    const float dcd = -dciou / c2, dc2 = dciou * cd / (c2 * c2);
    const float dex = 2.0f * ex * dc2, dey = 2.0f * ey * dc2;
    d2x += dex * tie_ge(p2x, g2x); d1x -= dex * tie_ge(g1x, p1x);     // ebr = max(p2, g2), etl = min(p1, g1)
    d2y += dey * tie_ge(p2y, g2y); d1y -= dey * tie_ge(g1y, p1y);
    const float dpcx = 2.0f * ddx * dcd, dpcy = 2.0f * ddy * dcd;
    d1x += 0.5f * dpcx; d2x += 0.5f * dpcx; d1y += 0.5f * dpcy; d2y += 0.5f * dpcy;
    // iou = inter / uni
    const float duni = -dciou * inter / (uni * uni);
    const float dinter = dciou / uni - duni;
    const float dwa = duni * ha * (pwr >= 0.0f ? 1.0f : 0.0f), dha = duni * wa * (phr >= 0.0f ? 1.0f : 0.0f);
    d2x += dwa; d1x -= dwa; d2y += dha; d1y -= dha;
    const float diw = dinter * ih * (dwx >= 0.0f ? 1.0f : 0.0f), dih = dinter * iw * (dwy >= 0.0f ? 1.0f : 0.0f);
    d2x += diw * tie_ge(g2x, p2x); d1x -= diw * tie_ge(p1x, g1x);     // br = min(p2, g2), tl = max(p1, g1)
    d2y += dih * tie_ge(g2y, p2y); d1y -= dih * tie_ge(p1y, g1y);
    grad[i] = make_float4(d1x + d2x, d1y + d2y, 0.5f * (d2x - d1x), 0.5f * (d2y - d1y));
}

// yolo_single_box_loss  practice/yolo_toy_regressor_torch_20251015.py:368-383: 2 (sl1(sig(x), tx) + sl1(sig(y), ty)) +
// 4 (sl1(w_log, log(tw + eps)) + sl1(h_log, log(th + eps))), each smooth_l1 (beta = 1) a MEAN over the n boxes.
__device__ __forceinline__ float sl1(float d) { const float a = fabsf(d); return a < 1.0f ? 0.5f * d * d : a - 0.5f; }
__device__ __forceinline__ float sl1_grad(float d) { return fabsf(d) < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f); }

__global__ void __launch_bounds__(kThreads) single_box_loss_kernel(const float4* __restrict__ raw, const float4* __restrict__ tg,
                                                                   int64_t n, float inv_n, float4* __restrict__ grad,
                                                                   float* __restrict__ partial) {
    __shared__ float wsum[kThreads / 64];
    float acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const float4 rw = raw[i], t = tg[i];
        const float x = yfs_sigmoid(rw.x), y = yfs_sigmoid(rw.y);
        const float dx = x - t.x, dy = y - t.y, dw = rw.z - logf(t.z + 1e-6f), dh = rw.w - logf(t.w + 1e-6f);
        acc += 2.0f * (sl1(dx) + sl1(dy)) + 4.0f * (sl1(dw) + sl1(dh));
        if (grad)
            grad[i] = make_float4(2.0f * sl1_grad(dx) * x * (1.0f - x) * inv_n, 2.0f * sl1_grad(dy) * y * (1.0f - y) * inv_n,
                                  4.0f * sl1_grad(dw) * inv_n, 4.0f * sl1_grad(dh) * inv_n);
    }
    acc = yfs_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
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

// ---- the same loss for callers that cannot allocate ------------------------------------------------------------------
namespace {
constexpr int kMaxDevices = 64;
std::atomic<float*> g_scratch_ptr[kMaxDevices];   // device address of `scratch` on each GPU (every GPU holds its own copy)

// Device address of this GPU's copy of `scratch`; resolved once per device, with that device current.
int noalloc_scratch(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    YFS_REQUIRE(e == hipSuccess && dev >= 0 && dev < kMaxDevices, YFS_ERR_LAUNCH,
                "yfs_yolov1_loss_noalloc_f32: hipGetDevice failed or device index %d out of range", dev);
    // no lock on the call path: the address of a device's copy never changes, so racing first calls on one device resolve
    // the same value and either store is fine
    float* known = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!known) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        YFS_REQUIRE(e == hipSuccess && p, YFS_ERR_LAUNCH, "yfs_yolov1_loss_noalloc_f32: hipGetSymbolAddress failed: %s",
                    hipGetErrorString(e));
        known = (float*)p;
        g_scratch_ptr[dev].store(known, std::memory_order_release);
    }
    *out = known;
    return YFS_OK;
}
}  // namespace

YFS_API int64_t yfs_yolov1_loss_noalloc_max_cells(void) { return kNoAllocMaxBlocks * kThreads; }

YFS_API int yfs_yolov1_loss_noalloc_f32(const float* pred, const float* target, int64_t N, int S, int B, int C, float l_coord,
                                        float l_noobj, float* loss_out, float* grad, void* stream) {
    YFS_REQUIRE(N > 0 && S > 0, YFS_ERR_INVALID_ARG, "yfs_yolov1_loss_noalloc_f32: bad shape N=%lld S=%d", (long long)N, S);
    YFS_REQUIRE(N * S * S <= kNoAllocMaxBlocks * kThreads, YFS_ERR_UNSUPPORTED,
                "yfs_yolov1_loss_noalloc_f32: N*S*S=%lld grid cells exceed the built-in scratch (%lld cells); use "
                "yfs_yolov1_loss_fwd_bwd_f32 with a caller-owned workspace",
                (long long)(N * S * S), (long long)(kNoAllocMaxBlocks * kThreads));
    float* ws = nullptr;
    int rc = noalloc_scratch(&ws);
    if (rc) return rc;
    // Every partial the finalize kernel reads (blocks 0 .. nblk-1) is written by this call's loss kernel first, so
    // nothing in the scratch has to start at zero; what a previous call left behind is never read.
    return yfs_yolov1_loss_fwd_bwd_f32(pred, target, N, S, B, C, l_coord, l_noobj, loss_out, grad, ws,
                                       (size_t)yfs_cdiv(N * S * S, kThreads) * 4 * sizeof(float), stream);
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

YFS_API int yfs_ciou_loss_f32(const float* pred, const float* gt, int64_t n, float* loss, float* iou, float* ciou, float* grad,
                              void* stream) {
    YFS_REQUIRE(pred && gt && loss && n > 0, YFS_ERR_INVALID_ARG, "yfs_ciou_loss_f32: null pointer or n <= 0");
    ciou_kernel<<<dim3((unsigned)yfs_cdiv(n, kThreads)), dim3(kThreads), 0, (hipStream_t)stream>>>(
        (const float4*)pred, (const float4*)gt, n, loss, iou, ciou, (float4*)grad);
    YFS_CHECK_LAUNCH("ciou_kernel");
    return YFS_OK;
}

YFS_API size_t yfs_single_box_loss_workspace_bytes(int64_t n) { return yfs_mse_workspace_bytes(n); }

YFS_API int yfs_single_box_loss_fwd_bwd_f32(const float* raw, const float* targets, int64_t n, float* loss_out, float* grad,
                                            void* workspace, size_t workspace_bytes, void* stream) {
    YFS_REQUIRE(raw && targets && loss_out && n > 0, YFS_ERR_INVALID_ARG, "yfs_single_box_loss_fwd_bwd_f32: null pointer or n <= 0");
    YFS_REQUIRE(workspace && workspace_bytes >= yfs_single_box_loss_workspace_bytes(n), YFS_ERR_WORKSPACE,
                "yfs_single_box_loss_fwd_bwd_f32: workspace needs %zu bytes", yfs_single_box_loss_workspace_bytes(n));
    hipStream_t s = (hipStream_t)stream;
    const int64_t nblk = mse_blocks(n);
    single_box_loss_kernel<<<dim3((unsigned)nblk), dim3(kThreads), 0, s>>>((const float4*)raw, (const float4*)targets, n,
                                                                          1.0f / (float)n, (float4*)grad, (float*)workspace);
    YFS_CHECK_LAUNCH("single_box_loss_kernel");
    sum_finalize_kernel<<<1, kThreads, 0, s>>>((const float*)workspace, nblk, 1.0f / (float)n, loss_out);
    YFS_CHECK_LAUNCH("sum_finalize_kernel");
    return YFS_OK;
}
