// Kernel set 4: YOLOv2 anchor target assignment + ignore-IoU mask.
// No reference code (README.md:55-59 of the reference names the feature); conventions in SPEC.md §4.
// Built with -ffp-contract=off: the masks must equal the CPU oracle's bit for bit.
#include "yfs_common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ void unstage_floats(float* __restrict__ dst, int len, const float* lds, int tid) {
    // lds[i] -> dst[i]; dst is only 4-byte aligned in general (per-image offset), so align the body.
    const int dskew = (int)((reinterpret_cast<uintptr_t>(dst) >> 2) & 3);
    const int head = (4 - dskew) & 3;
    const int h = head < len ? head : len;
    if (tid < h) dst[tid] = lds[tid];
    const int nq = (len - h) >> 2;
    float4* d4 = reinterpret_cast<float4*>(dst + h);
    for (int q = tid; q < nq; q += kThreads) {
        const float* s = lds + h + 4 * q;     // LDS side is unaligned when h != 0: scalar reads
        d4[q] = make_float4(s[0], s[1], s[2], s[3]);
    }
    const int done = h + nq * 4;
    if (tid < len - done) dst[done + tid] = lds[done + tid];
}

// One workgroup per image.  Phase 1: one thread per gt resolves (cell, anchor) and the regression
// targets into LDS.  Phase 2: one thread per (cell, anchor) scans the image's gts in index order —
// the first gt that maps to it wins (deterministic, no atomics) — and takes the max IoU of its
// predicted box over all valid gts for the ignore mask.  Targets leave through LDS with 16-byte stores.
__global__ void __launch_bounds__(kThreads) assign_v2_kernel(const float* __restrict__ gt, int G, int S, int A,
                                                             const float* __restrict__ anchors,
                                                             const float4* __restrict__ pred_boxes, float ignore_thr,
                                                             uint8_t* __restrict__ obj_mask,
                                                             uint8_t* __restrict__ ignore_mask,
                                                             float* __restrict__ targets) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int64_t n = blockIdx.x;
    const int per = S * S * A;
    // carve: per gt 10 floats: slot(int), tx, ty, tw, th, cls, cx, cy, w, h
    float* ginfo = lds;
    int* vlist = reinterpret_cast<int*>(lds + ((10 * G + 3) & ~3));     // indices of the valid gts, ascending
    float* tstage = lds + ((10 * G + 3) & ~3) + ((G + 3) & ~3);
    __shared__ int nvalid;
    const float* g = gt + n * (int64_t)G * 5;
    const float fS = (float)S;
    for (int k = tid; k < G; k += kThreads) {
        const float cx = g[5 * k], cy = g[5 * k + 1], w = g[5 * k + 2], h = g[5 * k + 3];
        float* o = ginfo + 10 * k;
        int slot = -1;
        if (w > 0.0f && h > 0.0f) {
            const float fx = cx * fS, fy = cy * fS;
            int col = (int)floorf(fx), row = (int)floorf(fy);
            col = col < 0 ? 0 : (col > S - 1 ? S - 1 : col);
            row = row < 0 ? 0 : (row > S - 1 ? S - 1 : row);
            const float gw = w * fS, gh = h * fS;
            int best = 0;
            float bi = -1.0f;
            for (int a = 0; a < A; ++a) {
                const float aw = anchors[2 * a], ah = anchors[2 * a + 1];
                const float inter = fminf(gw, aw) * fminf(gh, ah);
                const float uni = gw * gh + aw * ah - inter;
                const float iou = uni > 0.0f ? inter / uni : 0.0f;
                if (iou > bi) { bi = iou; best = a; }
            }
            slot = (row * S + col) * A + best;
            o[1] = fx - (float)col;
            o[2] = fy - (float)row;
            o[3] = logf(gw / anchors[2 * best]);
            o[4] = logf(gh / anchors[2 * best + 1]);
            o[5] = g[5 * k + 4];
        }
        o[0] = __int_as_float(slot);
        o[6] = cx; o[7] = cy; o[8] = w; o[9] = h;
    }
    __syncthreads();
    // The gt list is padded to G rows and typically holds 1-3 real boxes: compact the valid indices (in index
    // order, so "lowest gt index wins" is unchanged) and let the per-slot loops below run over those only.
    if (tid < 64) {
        int cnt = 0;
        for (int base = 0; base < G; base += 64) {
            const int k = base + tid;
            const bool valid = k < G && __float_as_int(ginfo[10 * k]) >= 0;
            const unsigned long long bal = __ballot(valid);
            if (valid) vlist[cnt + __builtin_popcountll(bal & ((1ull << tid) - 1ull))] = k;
            cnt += __builtin_popcountll(bal);
        }
        if (tid == 0) nvalid = cnt;
    }
    __syncthreads();
    const int V = nvalid;
    for (int c = tid; c < per; c += kThreads) {
        int owner = -1;
        for (int v = 0; v < V; ++v) {
            const int k = vlist[v];
            if (__float_as_int(ginfo[10 * k]) == c) { owner = k; break; }
        }
        float* t = tstage + 5 * c;
        if (owner >= 0) {
            const float* o = ginfo + 10 * owner;
            t[0] = o[1]; t[1] = o[2]; t[2] = o[3]; t[3] = o[4]; t[4] = o[5];
        } else {
            t[0] = 0.0f; t[1] = 0.0f; t[2] = 0.0f; t[3] = 0.0f; t[4] = 0.0f;
        }
        uint8_t ign = 0;
        if (pred_boxes) {
            const float4 p = pred_boxes[n * per + c];
            float best = 0.0f;
            for (int vi = 0; vi < V; ++vi) {
                const float* o = ginfo + 10 * vlist[vi];
                const float v = yfs_iou_xywh(p.x, p.y, p.z, p.w, o[6], o[7], o[8], o[9]);
                if (v > best) best = v;
            }
            ign = (uint8_t)((best > ignore_thr) && owner < 0);
        }
        obj_mask[n * per + c] = (uint8_t)(owner >= 0);
        ignore_mask[n * per + c] = ign;
    }
    __syncthreads();
    unstage_floats(targets + n * (int64_t)per * 5, per * 5, tstage, tid);
}


// YOLOv1 grid targets from ground-truth boxes (the step between the reference's
// box_to_xywh_norm, yolo_toy_regressor_torch.py:113-119, and the composite loss).  One workgroup per image; one
// thread per gt slot finds its cell, the lowest gt index that maps to a cell owns it (same rule as the YOLOv2
// assignment), the owner writes (x offset, y offset, w, h, 1) into box slot 0 and the one-hot class.  The image's
// target is zero-filled by the whole workgroup first (16-byte stores), owners write after a barrier.
__global__ void __launch_bounds__(kThreads) encode_targets_v1_kernel(const float* __restrict__ gt, int G, int S, int B, int C,
                                                                    float* __restrict__ target) {
    extern __shared__ int cell_of[];   // [G]: cell index or -1
    const int64_t n = blockIdx.x;
    const int tid = threadIdx.x;
    const int D = 5 * B + C, len = S * S * D;
    float* t = target + n * (int64_t)len;
    {   // zero fill: t is only 4-byte aligned in general
        const int dskew = (int)((reinterpret_cast<uintptr_t>(t) >> 2) & 3);
        const int head = ((4 - dskew) & 3) < len ? ((4 - dskew) & 3) : len;
        if (tid < head) t[tid] = 0.0f;
        const int nq = (len - head) >> 2;
        float4* t4 = reinterpret_cast<float4*>(t + head);
        for (int q = tid; q < nq; q += kThreads) t4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int done = head + nq * 4;
        if (tid < len - done) t[done + tid] = 0.0f;
    }
    const float* g = gt + n * (int64_t)G * 5;
    for (int i = tid; i < G; i += kThreads) {
        int cell = -1;
        if (g[i * 5 + 2] > 0.0f) {   // w <= 0 marks padding
            int col = (int)floorf(g[i * 5 + 0] * (float)S), row = (int)floorf(g[i * 5 + 1] * (float)S);
            col = col < S - 1 ? col : S - 1;
            row = row < S - 1 ? row : S - 1;
            col = col > 0 ? col : 0;
            row = row > 0 ? row : 0;
            cell = row * S + col;
        }
        cell_of[i] = cell;
    }
    __syncthreads();   // also orders the zero fill before the owners' stores (same workgroup, global memory)
    __threadfence_block();
    for (int i = tid; i < G; i += kThreads) {
        const int cell = cell_of[i];
        if (cell < 0) continue;
        bool owner = true;
        for (int j = 0; j < i; ++j) owner = owner && cell_of[j] != cell;
        if (!owner) continue;
        const int row = cell / S, col = cell - row * S;
        float* o = t + (int64_t)cell * D;
        o[0] = g[i * 5 + 0] * (float)S - (float)col;
        o[1] = g[i * 5 + 1] * (float)S - (float)row;
        o[2] = g[i * 5 + 2];
        o[3] = g[i * 5 + 3];
        o[4] = 1.0f;
        const int c = (int)g[i * 5 + 4];
        if (c >= 0 && c < C) o[5 * B + c] = 1.0f;
    }
}

}  // namespace

// k-means anchor clustering, assignment step (README.md:56 "k-means or VOC defaults"; SPEC.md §6): every box
// (w, h) goes to the anchor with the highest shape IoU (both placed at the origin), ties -> lowest anchor.
namespace {
constexpr int kMaxAnchors = 64;
__global__ void __launch_bounds__(kThreads) anchor_assign_kernel(const float2* __restrict__ wh, int64_t n,
                                                                 const float* __restrict__ anchors, int k,
                                                                 int32_t* __restrict__ assign, float* __restrict__ best_iou) {
    __shared__ float a[2 * kMaxAnchors];
    for (int i = threadIdx.x; i < 2 * k; i += kThreads) a[i] = anchors[i];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const float2 b = wh[i];
        int best = 0;
        float bi = -1.0f;
        for (int j = 0; j < k; ++j) {
            const float aw = a[2 * j], ah = a[2 * j + 1];
            const float inter = fminf(b.x, aw) * fminf(b.y, ah);
            const float uni = b.x * b.y + aw * ah - inter;
            const float iou = uni > 0.0f ? inter / uni : 0.0f;
            if (iou > bi) { bi = iou; best = j; }
        }
        assign[i] = best;
        if (best_iou) best_iou[i] = bi;
    }
}
}  // namespace

YFS_API int yfs_anchor_assign_f32(const float* wh, int64_t n, const float* anchors, int k, int32_t* assign, float* best_iou,
                                  void* stream) {
    YFS_REQUIRE(wh && anchors && assign, YFS_ERR_INVALID_ARG, "yfs_anchor_assign_f32: null pointer");
    YFS_REQUIRE(n > 0 && k > 0 && k <= kMaxAnchors, YFS_ERR_INVALID_ARG, "yfs_anchor_assign_f32: bad shape n=%lld k=%d (k <= %d)",
                (long long)n, k, kMaxAnchors);
    YFS_REQUIRE((reinterpret_cast<uintptr_t>(wh) & 7) == 0, YFS_ERR_INVALID_ARG, "yfs_anchor_assign_f32: wh must be 8-byte aligned");
    long long blk = yfs_cdiv(n, kThreads);
    if (blk > 4096) blk = 4096;
    anchor_assign_kernel<<<dim3((unsigned)blk), dim3(kThreads), 0, (hipStream_t)stream>>>((const float2*)wh, n, anchors, k, assign,
                                                                                      best_iou);
    YFS_CHECK_LAUNCH("anchor_assign_kernel");
    return YFS_OK;
}

YFS_API int yfs_assign_targets_v2_f32(const float* gt, int64_t N, int G, int S, int A, const float* anchors,
                                      const float* pred_boxes, float ignore_thr, uint8_t* obj_mask,
                                      uint8_t* ignore_mask, float* targets, void* stream) {
    YFS_REQUIRE(gt && anchors && obj_mask && ignore_mask && targets, YFS_ERR_INVALID_ARG,
                "yfs_assign_targets_v2_f32: null pointer");
    YFS_REQUIRE(N > 0 && G > 0 && S > 0 && A > 0, YFS_ERR_INVALID_ARG,
                "yfs_assign_targets_v2_f32: bad shape N=%lld G=%d S=%d A=%d", (long long)N, G, S, A);
    const size_t lds = ((size_t)((10 * G + 3) & ~3) + (size_t)((G + 3) & ~3) + (size_t)S * S * A * 5 + 8) * sizeof(float);
    YFS_REQUIRE(lds <= 150 * 1024, YFS_ERR_UNSUPPORTED,
                "yfs_assign_targets_v2_f32: S=%d A=%d G=%d needs %zu bytes of LDS (max 153600)", S, A, G, lds);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)assign_v2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        YFS_REQUIRE(e == hipSuccess, YFS_ERR_LAUNCH, "yfs_assign_targets_v2_f32: cannot reserve %zu bytes of LDS", lds);
    }
    assign_v2_kernel<<<dim3((unsigned)N), dim3(kThreads), lds, (hipStream_t)stream>>>(
        gt, G, S, A, anchors, (const float4*)pred_boxes, ignore_thr, obj_mask, ignore_mask, targets);
    YFS_CHECK_LAUNCH("assign_v2_kernel");
    return YFS_OK;
}

YFS_API int yfs_encode_targets_v1_f32(const float* gt, int64_t N, int G, int S, int B, int C, float* target, void* stream) {
    YFS_REQUIRE(gt && target, YFS_ERR_INVALID_ARG, "yfs_encode_targets_v1_f32: null pointer");
    YFS_REQUIRE(N > 0 && G > 0 && G <= 4096 && S > 0 && B > 0 && C > 0, YFS_ERR_INVALID_ARG,
                "yfs_encode_targets_v1_f32: bad shape N=%lld G=%d S=%d B=%d C=%d", (long long)N, G, S, B, C);
    encode_targets_v1_kernel<<<dim3((unsigned)N), dim3(kThreads), (size_t)G * sizeof(int), (hipStream_t)stream>>>(gt, G, S, B, C,
                                                                                                                target);
    YFS_CHECK_LAUNCH("encode_targets_v1_kernel");
    return YFS_OK;
}
