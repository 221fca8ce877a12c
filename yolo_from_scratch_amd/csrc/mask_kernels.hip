// Segmentation masks -> bounding boxes of their 8-connected components -> the largest box as [xc, yc, w, h]
// (reference: mask_bytes_to_boxes yolo_toy_regressor_torch.py:98-111 = skimage label() with its
// default full connectivity + regionprops().bbox in label order, box_to_xywh_norm :113-119, largest-box rule :195-196).
// HBM-bound integer work, bit-exact by construction.
//
// Labelling is a union-find over pixel indices in global memory (one thread per pixel, whole batch per launch):
//   init    label[p] = first pixel of p's horizontal run inside its wave's 64 pixels (two ballots and a count-leading-
//           zeros: runs are linked without a single memory operation), -1 for background
//   merge   only the links that are not implied by others are made: the row link across a 64-pixel boundary, the link
//           to N once per overlap of two runs (skipped when W and NW are both set: W makes it), NW only if N and W are
//           clear, NE only if N and E are clear.  unite() hooks the larger root under the smaller with atomicMin, so the
//           root of a component ends up as its minimum index = its first pixel in raster order — which is exactly
//           the label order of the reference, and gives y_min for free
//   flatten label[p] = root; roots initialise their box record and every 256-pixel block counts its roots
//   bbox    pixels that are extreme among their 8 neighbours update x_min / x_max / y_max of their root with atomics
//   scan    per image: exclusive prefix of the block counts = index of each block's first component
//   emit    roots write their box at (block base + rank inside the block); every block records its bid for "largest":
//           max of (area << 32 | ~first pixel) = largest area, first in label order on ties (np.argmax)
//   final   per image: the largest bid; its box -> [xc/W, yc/H, w/W, h/H] in fp64, rounded once to fp32 as numpy does
#include "yfs_common.h"
#include <limits.h>

namespace {
constexpr int kThreads = 256;

struct MaskArgs {
    const uint8_t* mask;
    int H, W, CH;
    int HW, nblk;                 // pixels per image, 256-pixel blocks per image
    int* label;                   // [N][HW]
    int4* bb;                     // [N][HW]: (x_min, x_max, y_max, -) at root pixels only
    int* blkcnt;                  // [N][nblk]
    unsigned long long* blkbest;  // [N][nblk]
};

__device__ __forceinline__ int ld_coherent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int find_root(const int* L, int a) {
    int p;
    while ((p = ld_coherent(L + a)) != a) a = p;
    return a;
}

// Labels only ever decrease and a stale parent is still a member of the same tree, so plain retries are enough.
__device__ __forceinline__ void unite(int* L, int a, int b) {
    while (true) {
        a = find_root(L, a);
        b = find_root(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(L + a, b);     // hook root a under the smaller root b
        if (old == a) return;
        a = old;                                   // a was hooked by somebody else meanwhile: carry on from its new parent
    }
}

__global__ void __launch_bounds__(kThreads) mask_init_kernel(MaskArgs g) {
    const int n = blockIdx.x / g.nblk, blk = blockIdx.x % g.nblk;
    const int p = blk * kThreads + threadIdx.x, lane = threadIdx.x & 63;
    const bool in = p < g.HW;
    const int x = in ? p % g.W : -1;
    bool fg = false;
    if (in) {
        const uint8_t* m = g.mask + ((size_t)n * g.HW + p) * g.CH;
        for (int c = 0; c < g.CH; ++c) fg |= m[c] != 0;
    }
    const unsigned long long Z = __ballot(!fg), R = __ballot(x == 0);
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const unsigned long long a = Z & below, b = R & (below | (1ull << lane));
    const int sa = a ? 64 - __clzll((long long)a) : 0;      // lane after the last background pixel before this one
    const int sb = b ? 63 - __clzll((long long)b) : 0;      // lane that starts this pixel's image row
    const int s = sa > sb ? sa : sb;
    if (in) g.label[(size_t)n * g.HW + p] = fg ? p - (lane - s) : -1;
}

__global__ void __launch_bounds__(kThreads) mask_merge_kernel(MaskArgs g) {
    const int n = blockIdx.x / g.nblk, blk = blockIdx.x % g.nblk;
    const int p = blk * kThreads + threadIdx.x;
    if (p >= g.HW) return;
    int* L = g.label + (size_t)n * g.HW;
    if (L[p] < 0) return;
    const int W = g.W, y = p / W, x = p - y * W;
    const bool up = y > 0, lf = x > 0, rt = x < W - 1;
    const bool w_ = lf && L[p - 1] >= 0, e_ = rt && L[p + 1] >= 0;
    const bool n_ = up && L[p - W] >= 0;
    const bool nw = up && lf && L[p - W - 1] >= 0, ne = up && rt && L[p - W + 1] >= 0;
    if (w_ && (p & 63) == 0) unite(L, p, p - 1);
    if (n_) {
        if (!(w_ && nw)) unite(L, p, p - W);
    } else {
        if (nw && !w_) unite(L, p, p - W - 1);
        if (ne && !e_) unite(L, p, p - W + 1);
    }
}

__global__ void __launch_bounds__(kThreads) mask_flatten_kernel(MaskArgs g) {
    __shared__ int wcnt[kThreads / 64];
    const int n = blockIdx.x / g.nblk, blk = blockIdx.x % g.nblk;
    const int p = blk * kThreads + threadIdx.x;
    int* L = g.label + (size_t)n * g.HW;
    bool root = false;
    if (p < g.HW && L[p] >= 0) {
        const int r = find_root(L, p);
        root = r == p;
        if (root) g.bb[(size_t)n * g.HW + p] = make_int4(INT_MAX, -1, -1, 0);
        else L[p] = r;
    }
    const unsigned long long m = __ballot(root);
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) g.blkcnt[(size_t)n * g.nblk + blk] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

// Only pixels that are extreme among their own 8 neighbours bid: the leftmost pixel of a component has no set pixel at
// W, NW or SW (likewise E / NE / SE for x_max and S / SW / SE for y_max), and a filled shape has a handful of those,
// not two per row — every bid is an atomic on ONE record per component, and same-address atomics are serviced one
// after another by the L2 (run starts and run ends bidding: 152 us per 64 VOC-sized masks, this rule: see DESIGN.md).
__global__ void __launch_bounds__(kThreads) mask_bbox_kernel(MaskArgs g) {
    const int n = blockIdx.x / g.nblk, blk = blockIdx.x % g.nblk;
    const int p = blk * kThreads + threadIdx.x;
    if (p >= g.HW) return;
    const int* L = g.label + (size_t)n * g.HW;
    const int r = L[p];
    if (r < 0) return;
    const int W = g.W, y = p / W, x = p - y * W;
    const bool up = y > 0, dn = y < g.H - 1, lf = x > 0, rt = x < W - 1;
    const bool w_ = lf && L[p - 1] >= 0, e_ = rt && L[p + 1] >= 0;
    const bool nw = up && lf && L[p - W - 1] >= 0, ne = up && rt && L[p - W + 1] >= 0;
    const bool s_ = dn && L[p + W] >= 0, sw = dn && lf && L[p + W - 1] >= 0, se = dn && rt && L[p + W + 1] >= 0;
    int* b = reinterpret_cast<int*>(g.bb + (size_t)n * g.HW + r);
    if (!(w_ || nw || sw) && x < ld_coherent(b + 0)) atomicMin(b + 0, x);
    if (!(e_ || ne || se) && x > ld_coherent(b + 1)) atomicMax(b + 1, x);
    if (!(s_ || sw || se) && y > ld_coherent(b + 2)) atomicMax(b + 2, y);
}

// one workgroup per image: blkcnt[n][:] -> exclusive prefix in place, count[n] = number of components
__global__ void __launch_bounds__(kThreads) mask_scan_kernel(MaskArgs g, int32_t* __restrict__ count) {
    __shared__ int part[kThreads];
    const int n = blockIdx.x, t = threadIdx.x;
    int* c = g.blkcnt + (size_t)n * g.nblk;
    const int per = (g.nblk + kThreads - 1) / kThreads;
    const int lo = t * per, hi = lo + per < g.nblk ? lo + per : g.nblk;
    int s = 0;
    for (int i = lo; i < hi; ++i) s += c[i];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < kThreads; d <<= 1) {
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = lo; i < hi; ++i) {
        const int v = c[i];
        c[i] = run;
        run += v;
    }
    if (t == kThreads - 1) count[n] = part[t];
}

// Roots write their box at (block base + rank inside the block); the block's bid for "largest" — the maximum of
// (area << 32 | ~first pixel) over its roots: largest area, first in label order on ties — goes to blkbest without
// atomics (one u64 per image would take every root's atomic on a single cache line).
__global__ void __launch_bounds__(kThreads) mask_emit_kernel(MaskArgs g, int4* __restrict__ boxes, int max_boxes) {
    __shared__ int wcnt[kThreads / 64];
    __shared__ unsigned long long wbest[kThreads / 64];
    const int n = blockIdx.x / g.nblk, blk = blockIdx.x % g.nblk;
    const int p = blk * kThreads + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int* L = g.label + (size_t)n * g.HW;
    const bool root = p < g.HW && L[p] == p;
    const unsigned long long m = __ballot(root);
    unsigned long long key = 0ull;
    int4 box = make_int4(0, 0, 0, 0);
    if (root) {
        const int4 b = g.bb[(size_t)n * g.HW + p];
        const int y0 = p / g.W;
        box = make_int4(b.x, y0, b.y, b.z);                                                  // x_min, y_min, x_max, y_max
        const unsigned long long area = (unsigned long long)(b.y - b.x + 1) * (unsigned long long)(b.z - y0 + 1);
        key = (area << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)p);
    }
    unsigned long long kmax = key;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(kmax, o, 64);
        kmax = other > kmax ? other : kmax;
    }
    if (lane == 0) {
        wcnt[wave] = __popcll(m);
        wbest[wave] = kmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long k = wbest[0];
        for (int w = 1; w < kThreads / 64; ++w) k = wbest[w] > k ? wbest[w] : k;
        g.blkbest[(size_t)n * g.nblk + blk] = k;
    }
    if (!root) return;
    int k = g.blkcnt[(size_t)n * g.nblk + blk] + __popcll(m & (lane ? (~0ull >> (64 - lane)) : 0ull));
    for (int w = 0; w < wave; ++w) k += wcnt[w];
    if (k < max_boxes) boxes[(size_t)n * max_boxes + k] = box;
}

// one workgroup per image: the largest of the block bids -> best_box and [xc/W, yc/H, w/W, h/H]
__global__ void __launch_bounds__(kThreads) mask_final_kernel(MaskArgs g, const int32_t* __restrict__ dims,
                                                              int32_t* __restrict__ best_box, float* __restrict__ best_xywh) {
    __shared__ unsigned long long wbest[kThreads / 64];
    const int n = blockIdx.x, t = threadIdx.x;
    const unsigned long long* bids = g.blkbest + (size_t)n * g.nblk;
    unsigned long long kmax = 0ull;
    for (int i = t; i < g.nblk; i += kThreads) kmax = bids[i] > kmax ? bids[i] : kmax;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(kmax, o, 64);
        kmax = other > kmax ? other : kmax;
    }
    if ((t & 63) == 0) wbest[t >> 6] = kmax;
    __syncthreads();
    if (t != 0) return;
    unsigned long long key = wbest[0];
    for (int w = 1; w < kThreads / 64; ++w) key = wbest[w] > key ? wbest[w] : key;
    int4 box = make_int4(-1, -1, -1, -1);
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (key != 0ull) {                                  // every component has an area >= 1
        const int p = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
        const int4 b = g.bb[(size_t)n * g.HW + p];
        box = make_int4(b.x, p / g.W, b.y, b.z);
        const double Hd = dims ? (double)dims[2 * n] : (double)g.H, Wd = dims ? (double)dims[2 * n + 1] : (double)g.W;
        const int w = box.z - box.x + 1, h = box.w - box.y + 1;             // :113-119, same operations in fp64
        const double xc = (double)box.x + (double)(w - 1) / 2.0, yc = (double)box.y + (double)(h - 1) / 2.0;
        out = make_float4((float)(xc / Wd), (float)(yc / Hd), (float)((double)w / Wd), (float)((double)h / Hd));
    }
    if (best_box) reinterpret_cast<int4*>(best_box)[n] = box;
    if (best_xywh) reinterpret_cast<float4*>(best_xywh)[n] = out;
}

struct MaskLayout {
    size_t label, bb, blkcnt, best, total;
    int nblk;
};
MaskLayout mask_layout(int64_t N, int H, int W) {
    MaskLayout l;
    const size_t hw = (size_t)H * W;
    l.nblk = (int)((hw + kThreads - 1) / kThreads);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    l.bb = 0;                                              // int4 records first: 16-byte aligned
    l.label = up(l.bb + (size_t)N * hw * sizeof(int4));
    l.blkcnt = up(l.label + (size_t)N * hw * sizeof(int));
    l.best = up(l.blkcnt + (size_t)N * l.nblk * sizeof(int));
    l.total = up(l.best + (size_t)N * l.nblk * sizeof(unsigned long long));
    return l;
}
}  // namespace

YFS_API size_t yfs_mask_to_boxes_workspace_bytes(int64_t N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    return mask_layout(N, H, W).total;
}

YFS_API int yfs_mask_to_boxes_u8(const uint8_t* mask, int64_t N, int H, int W, int CH, const int32_t* dims, int32_t* boxes,
                                 int max_boxes, int32_t* count, int32_t* best_box, float* best_xywh, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    YFS_REQUIRE(mask && count && workspace, YFS_ERR_INVALID_ARG, "yfs_mask_to_boxes_u8: null pointer");
    YFS_REQUIRE(N > 0 && H > 0 && W > 0 && CH > 0 && CH <= 16 && max_boxes >= 0 && (boxes || max_boxes == 0), YFS_ERR_INVALID_ARG,
                "yfs_mask_to_boxes_u8: bad shape N=%lld H=%d W=%d CH=%d max_boxes=%d", (long long)N, H, W, CH, max_boxes);
    YFS_REQUIRE((long long)H * W <= (1ll << 30), YFS_ERR_UNSUPPORTED, "yfs_mask_to_boxes_u8: %d x %d pixels per mask (max 2^30)", H, W);
    const MaskLayout l = mask_layout(N, H, W);
    YFS_REQUIRE((long long)N * l.nblk < (1ll << 31), YFS_ERR_UNSUPPORTED,
                "yfs_mask_to_boxes_u8: N=%lld masks of %d x %d need more than 2^31 workgroups; split the batch", (long long)N, H, W);
    YFS_REQUIRE(workspace_bytes >= l.total, YFS_ERR_INVALID_ARG, "yfs_mask_to_boxes_u8: workspace of %zu bytes, %zu needed",
                workspace_bytes, l.total);
    const uintptr_t wsp = reinterpret_cast<uintptr_t>(workspace);
    YFS_REQUIRE((wsp & 15) == 0 && (reinterpret_cast<uintptr_t>(boxes) & 15) == 0 && (reinterpret_cast<uintptr_t>(best_box) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(best_xywh) & 15) == 0,
                YFS_ERR_INVALID_ARG, "yfs_mask_to_boxes_u8: workspace, boxes, best_box and best_xywh must be 16-byte aligned");
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    MaskArgs g;
    g.mask = mask;
    g.H = H, g.W = W, g.CH = CH, g.HW = H * W, g.nblk = l.nblk;
    g.bb = reinterpret_cast<int4*>(ws + l.bb);
    g.label = reinterpret_cast<int*>(ws + l.label);
    g.blkcnt = reinterpret_cast<int*>(ws + l.blkcnt);
    g.blkbest = reinterpret_cast<unsigned long long*>(ws + l.best);
    const dim3 grid((unsigned)(N * l.nblk)), block(kThreads);
    hipStream_t s = (hipStream_t)stream;
    mask_init_kernel<<<grid, block, 0, s>>>(g);
    YFS_CHECK_LAUNCH("mask_init_kernel");
    mask_merge_kernel<<<grid, block, 0, s>>>(g);
    YFS_CHECK_LAUNCH("mask_merge_kernel");
    mask_flatten_kernel<<<grid, block, 0, s>>>(g);
    YFS_CHECK_LAUNCH("mask_flatten_kernel");
    mask_bbox_kernel<<<grid, block, 0, s>>>(g);
    YFS_CHECK_LAUNCH("mask_bbox_kernel");
    mask_scan_kernel<<<dim3((unsigned)N), block, 0, s>>>(g, count);
    YFS_CHECK_LAUNCH("mask_scan_kernel");
    mask_emit_kernel<<<grid, block, 0, s>>>(g, reinterpret_cast<int4*>(boxes), max_boxes);
    YFS_CHECK_LAUNCH("mask_emit_kernel");
    mask_final_kernel<<<dim3((unsigned)N), block, 0, s>>>(g, dims, best_box, best_xywh);
    YFS_CHECK_LAUNCH("mask_final_kernel");
    return YFS_OK;
}
