// Kernel set 3: box math, grid/anchor decode, class-wise NMS (gfx950).
// Built with -ffp-contract=off: IoU-vs-threshold decisions must match the CPU oracle bit for bit,
// so every multiply and add rounds separately, in the op order of the reference's iou_xywh_torch
// (yolo_toy_regressor_torch.py:159-182).
#include "yfs_common.h"

namespace {

constexpr int kThreads = 256;

struct Decoded {
    float x, y, w, h, score;
    int cls;
};

// The grid decode is VALU-bound: 25 exponentials per box (20 of them the softmax), ~20 instructions each with the
// correctly rounded expf.  Decoded boxes and scores are compared with the oracle at 1e-5 relative (they go through libm
// on the CPU side anyway, SURVEY.md 7 "hard parts"), and NMS parity is checked on the GPU's own decoded boxes, so the
// decode uses the hardware exponential: exp2(x * log2 e), a few 1e-7 relative for |x| < 20.
__device__ __forceinline__ float dec_exp(float x) { return __expf(x); }
__device__ __forceinline__ float dec_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

// raw (tx,ty,tw,th,to, C logits) at r[0..5+C), read with stride 1 from LDS.
__device__ __forceinline__ Decoded decode_one(const float* r, int C, int row, int col, int S, float aw, float ah) {
    Decoded d;
    const float fS = (float)S;
    d.x = (dec_sigmoid(r[0]) + (float)col) / fS;
    d.y = (dec_sigmoid(r[1]) + (float)row) / fS;
    d.w = aw * dec_exp(r[2]) / fS;
    d.h = ah * dec_exp(r[3]) / fS;
    const float conf = dec_sigmoid(r[4]);
    float m = r[5];
    int best = 0;
    for (int c = 1; c < C; ++c) {
        float v = r[5 + c];
        if (v > m) { m = v; best = c; }
    }
    float sum = 0.0f;
    for (int c = 0; c < C; ++c) sum += dec_exp(r[5 + c] - m);
    d.score = conf * (1.0f / sum);
    d.cls = best;
    return d;
}

// ---------------- paired IoU + deterministic mean -----------------------------------------
__global__ void __launch_bounds__(kThreads) iou_rows_kernel(const float4* __restrict__ pred,
                                                            const float4* __restrict__ gt, int64_t n,
                                                            float* __restrict__ iou_out, float* __restrict__ partial) {
    __shared__ float wsum[kThreads / 64];
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    float v = 0.0f;
    if (i < n) {
        float4 p = pred[i], g = gt[i];
        v = yfs_iou_xywh(p.x, p.y, p.z, p.w, g.x, g.y, g.z, g.w);
        if (iou_out) iou_out[i] = v;
    }
    if (partial) {
        float s = yfs_wave_sum(v);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    }
}

// one block: sums `count` partials in a fixed order, writes out[0] = sum * scale
__global__ void __launch_bounds__(kThreads) finalize_sum_kernel(const float* __restrict__ partial, int64_t count,
                                                                float scale, float* __restrict__ out) {
    __shared__ float wsum[kThreads / 64];
    float acc = 0.0f;
    for (int64_t i = threadIdx.x; i < count; i += kThreads) acc += partial[i];
    float s = yfs_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) * scale;
}

__device__ __forceinline__ float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

__global__ void __launch_bounds__(kThreads) iou_variant_kernel(const float4* __restrict__ pred,
                                                               const float4* __restrict__ gt, int64_t n, int variant,
                                                               float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    float4 p = pred[i], g = gt[i];
    float r;
    if (variant == 0) {
        r = yfs_iou_xywh(p.x, p.y, p.z, p.w, g.x, g.y, g.z, g.w);
    } else if (variant == 1) {   // box_iou_xyxy, practice/yolo_toy_regressor_3_torch.py:112-119
        float px1 = p.x - p.z / 2, py1 = p.y - p.w / 2, px2 = p.x + p.z / 2, py2 = p.y + p.w / 2;
        float gx1 = g.x - g.z / 2, gy1 = g.y - g.w / 2, gx2 = g.x + g.z / 2, gy2 = g.y + g.w / 2;
        float iw = fminf(px2, gx2) - fmaxf(px1, gx1), ih = fminf(py2, gy2) - fmaxf(py1, gy1);
        iw = iw > 0 ? iw : 0; ih = ih > 0 ? ih : 0;
        float inter = iw * ih;
        float aw = px2 - px1, ah = py2 - py1, bw = gx2 - gx1, bh = gy2 - gy1;
        aw = aw > 0 ? aw : 0; ah = ah > 0 ? ah : 0; bw = bw > 0 ? bw : 0; bh = bh > 0 ? bh : 0;
        float uni = aw * ah + bw * bh - inter + 1e-9f;
        r = inter / uni;
    } else {                     // older iou_xywh_torch, practice/yolo_toy_regressor_torch_20251015.py:226-278
        const float eps = 1e-7f;
        float pw = p.z > eps ? p.z : eps, ph = p.w > eps ? p.w : eps;
        float gw = g.z > eps ? g.z : eps, gh = g.w > eps ? g.w : eps;
        float px1 = clamp01(p.x - 0.5f * pw), py1 = clamp01(p.y - 0.5f * ph);
        float px2 = clamp01(p.x + 0.5f * pw), py2 = clamp01(p.y + 0.5f * ph);
        float gx1 = clamp01(g.x - 0.5f * gw), gy1 = clamp01(g.y - 0.5f * gh);
        float gx2 = clamp01(g.x + 0.5f * gw), gy2 = clamp01(g.y + 0.5f * gh);
        float iw = fminf(px2, gx2) - fmaxf(px1, gx1), ih = fminf(py2, gy2) - fmaxf(py1, gy1);
        iw = iw > 0 ? iw : 0; ih = ih > 0 ? ih : 0;
        float inter = iw * ih;
        float aw = px2 - px1, ah = py2 - py1, bw = gx2 - gx1, bh = gy2 - gy1;
        aw = aw > 0 ? aw : 0; ah = ah > 0 ? ah : 0; bw = bw > 0 ? bw : 0; bh = bh > 0 ? bh : 0;
        float uni = aw * ah + bw * bh - inter;
        r = clamp01(inter / (uni + eps));
    }
    out[i] = r;
}

// ---------------- N x M pairwise IoU ------------------------------------------------------
// Each block: 64 rows of `a` (one per lane of a wave, kept in registers as corners) x a strip
// of columns; `b` corners are staged in LDS and read as wave-wide broadcasts.
constexpr int kPairCols = 256;
__global__ void __launch_bounds__(kThreads) pairwise_iou_kernel(const float4* __restrict__ a, int64_t na,
                                                                const float4* __restrict__ b, int64_t nb,
                                                                float* __restrict__ out) {
    __shared__ float4 bc[kPairCols];
    const int64_t j0 = (int64_t)blockIdx.x * kPairCols;
    const int64_t i0 = (int64_t)blockIdx.y * 64;
    {
        int64_t j = j0 + threadIdx.x;
        float4 v = j < nb ? b[j] : make_float4(0, 0, 0, 0);
        bc[threadIdx.x] = make_float4(v.x - 0.5f * v.z, v.y - 0.5f * v.w, v.x + 0.5f * v.z, v.y + 0.5f * v.w);
    }
    __syncthreads();
    // thread t: column group (t & 63) * 4 .. +3, rows (t >> 6) * 16 .. +15
    const int cg = (threadIdx.x & 63) * 4;
    const int rbase = (threadIdx.x >> 6) * 16;
    float4 c0 = bc[cg], c1 = bc[cg + 1], c2 = bc[cg + 2], c3 = bc[cg + 3];
    for (int r = 0; r < 16; ++r) {
        int64_t i = i0 + rbase + r;
        if (i >= na) break;
        float4 p = a[i];
        float px1 = p.x - 0.5f * p.z, py1 = p.y - 0.5f * p.w, px2 = p.x + 0.5f * p.z, py2 = p.y + 0.5f * p.w;
        float4 o;
        o.x = yfs_iou_corners(px1, py1, px2, py2, c0.x, c0.y, c0.z, c0.w);
        o.y = yfs_iou_corners(px1, py1, px2, py2, c1.x, c1.y, c1.z, c1.w);
        o.z = yfs_iou_corners(px1, py1, px2, py2, c2.x, c2.y, c2.z, c2.w);
        o.w = yfs_iou_corners(px1, py1, px2, py2, c3.x, c3.y, c3.z, c3.w);
        int64_t j = j0 + cg;
        float* dst = out + i * nb + j;
        if (j + 3 < nb && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
            *reinterpret_cast<float4*>(dst) = o;
        } else {
            if (j < nb) dst[0] = o.x;
            if (j + 1 < nb) dst[1] = o.y;
            if (j + 2 < nb) dst[2] = o.z;
            if (j + 3 < nb) dst[3] = o.w;
        }
    }
}

// ---------------- single-box decoders ------------------------------------------------------
__global__ void __launch_bounds__(kThreads) decode_head_kernel(const float4* __restrict__ t, int64_t n, float pw,
                                                               float ph, float4* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    float4 v = t[i];
    float w = expf(v.z) * pw, h = expf(v.w) * ph;
    w = w < 1e-4f ? 1e-4f : (w > 1.0f ? 1.0f : w);
    h = h < 1e-4f ? 1e-4f : (h > 1.0f ? 1.0f : h);
    out[i] = make_float4(yfs_sigmoid(v.x), yfs_sigmoid(v.y), w, h);
}

__global__ void __launch_bounds__(kThreads) decode_xywh_kernel(const float4* __restrict__ t, int64_t n,
                                                               float4* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const float eps = 1e-7f, hi = 1.0f - 1e-7f;
    float4 v = t[i];
    float w = yfs_sigmoid(v.z), h = yfs_sigmoid(v.w);
    w = w < eps ? eps : (w > hi ? hi : w);
    h = h < eps ? eps : (h > hi ? hi : h);
    out[i] = make_float4(yfs_sigmoid(v.x), yfs_sigmoid(v.y), w, h);
}

// ---------------- grid decode (stand-alone) -------------------------------------------------
// One block decodes `bpb` consecutive boxes of the flattened [N*S*S*A] box list: the raw floats
// are pulled into LDS with coalesced 16-byte loads (each HBM byte read exactly once), every
// thread then owns one box and reads its 5+C floats at LDS stride D (odd D => conflict free).
__global__ void __launch_bounds__(kThreads) decode_grid_kernel(const float* __restrict__ raw, int64_t nboxes, int S,
                                                               int A, int C, const float* __restrict__ anchors,
                                                               int bpb, float4* __restrict__ boxes,
                                                               float* __restrict__ score, int32_t* __restrict__ cls,
                                                               float* __restrict__ all_scores) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int D = 5 + C;
    const int64_t b0 = (int64_t)blockIdx.x * bpb;
    const int nb = (int)((nboxes - b0) < bpb ? (nboxes - b0) : bpb);
    const int skew = yfs_stage_floats_dma(raw + b0 * D, nb * D, lds, kThreads / 64);
    yfs_stage_wait();
    __syncthreads();
    const int t = threadIdx.x;
    if (t < nb) {
        const int64_t b = b0 + t;
        const int per = S * S * A;
        const int bi = (int)(b % per);
        const int a = bi % A, cell = bi / A;
        const float* r = lds + skew + t * D;
        Decoded d = decode_one(r, C, cell / S, cell % S, S, anchors[2 * a], anchors[2 * a + 1]);
        boxes[b] = make_float4(d.x, d.y, d.w, d.h);
        score[b] = d.score;
        cls[b] = d.cls;
        if (all_scores) {
            const float conf = dec_sigmoid(r[4]);
            float m = r[5 + d.cls];
            float sum = 0.0f;
            for (int c = 0; c < C; ++c) sum += dec_exp(r[5 + c] - m);
            for (int c = 0; c < C; ++c) all_scores[b * C + c] = conf * (dec_exp(r[5 + c] - m) / sum);
        }
    }
}

// YOLOv1 head decode: one thread per cell, pred staged through LDS the same way.
__global__ void __launch_bounds__(kThreads) decode_v1_kernel(const float* __restrict__ pred, int64_t ncells, int S,
                                                             int B, int C, float4* __restrict__ boxes,
                                                             float* __restrict__ score, int32_t* __restrict__ cls) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int D = 5 * B + C;
    const int64_t c0 = (int64_t)blockIdx.x * kThreads;
    const int nc = (int)((ncells - c0) < kThreads ? (ncells - c0) : kThreads);
    const int skew = yfs_stage_floats_dma(pred + c0 * D, nc * D, lds, kThreads / 64);
    yfs_stage_wait();
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= nc) return;
    const int64_t cell = c0 + t;
    const int ci = (int)(cell % (S * S));
    const int row = ci / S, col = ci % S;
    const float* r = lds + skew + t * D;
    int best = 0;
    float m = r[5 * B];
    for (int c = 1; c < C; ++c) {
        float v = r[5 * B + c];
        if (v > m) { m = v; best = c; }
    }
    const float fS = (float)S;
    for (int j = 0; j < B; ++j) {
        const int64_t b = cell * B + j;
        boxes[b] = make_float4((r[5 * j] + (float)col) / fS, (r[5 * j + 1] + (float)row) / fS, r[5 * j + 2],
                               r[5 * j + 3]);
        score[b] = r[5 * j + 4] * m;
        cls[b] = best;
    }
}

// ---------------- class-wise NMS (optionally fused with the grid decode) -----------------------
// One workgroup per image; the image's candidate boxes never leave LDS.
//   A  filter: score >= thr -> append (key, corners, class) to LDS candidate arrays
//      (fused: decode 256-box chunks staged through LDS first, and write boxes/score/cls out)
//   B  rank sort by 64-bit key = (descending score, ascending index): K^2/256 compares per thread,
//      all LDS broadcasts, two barriers
//   C  greedy suppression in 64-row tiles: all threads build the tile's 64 x (W-b) suppression
//      bitmask (row = lane, so the column box is a wave-wide LDS broadcast); wave 0 resolves the
//      tile's diagonal word with register-only readlane steps, then ORs the kept rows' masks into
//      the per-lane `removed` words
//   D  kept indices (score order) written coalesced, -1 padded
struct NmsArgs {
    const float* raw;        // fused: [N, S*S*A, 5+C]
    const float* anchors;    // fused
    int S, A_anch, C;        // fused
    const float4* boxes_in;  // non-fused
    const float* score_in;
    const int32_t* cls_in;
    float4* boxes_out;       // fused outputs
    float* score_out;
    int32_t* cls_out;
    int32_t* keep;
    int32_t* count;
    int A;                   // boxes per image
    int cap;                 // A rounded up to 64
    int npow;                // power of two >= cap (bitonic sort length bound)
    int bpc;                 // fused: boxes per staged chunk
    float score_thr, iou_thr;
};

// 64-bit sort key: [class : 21][~monotone(score) : 32][slot : 11].  Ascending order = class-major, then descending
// score, then ascending slot; slots are handed out in box-index order, so the last field is the index tie-break.
constexpr int kSlotBits = 11;                 // YFS_NMS_MAX_BOXES = 2048
constexpr unsigned long long kSlotMask = (1ull << kSlotBits) - 1ull;
constexpr int kClsShift = 32 + kSlotBits;
constexpr int kQShift = kSlotBits + 6;      // one-sort path (class < 64): cidx = [class-major position:11 | class:6 | box index:11]
__device__ __forceinline__ unsigned score_bits(float s) {
    unsigned u = __float_as_uint(s);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // monotone float -> uint
    return ~u;                                         // ascending = descending score
}
__device__ __forceinline__ unsigned long long make_key(int cls, float s, int slot) {
    return ((unsigned long long)((unsigned)cls & 0x1FFFFFu) << kClsShift) | ((unsigned long long)score_bits(s) << kSlotBits) |
           (unsigned)slot;
}

// yfs_iou_corners for the suppression loop, which is VALU-issue-bound: fminf / fmaxf on values that come back from LDS
// make the compiler canonicalise every operand first (8 extra instructions per pair); the hardware min / max already
// implement minNum / maxNum (the non-NaN operand, signalling NaNs quieted), so they are issued directly.  Same op order,
// same results.
__device__ __forceinline__ float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float nms_pair_iou(const float4 p, const float4 q) {
    float iw = hw_min(p.z, q.z) - hw_max(p.x, q.x);
    float ih = hw_min(p.w, q.w) - hw_max(p.y, q.y);
    iw = iw > 0.0f ? iw : 0.0f;
    ih = ih > 0.0f ? ih : 0.0f;
    const float inter = iw * ih;
    const float ap = (p.z - p.x) * (p.w - p.y);
    const float ag = (q.z - q.x) * (q.w - q.y);
    const float uni = ap + ag - inter;
    return uni > 0.0f ? inter / uni : 0.0f;
}

// In-place ascending bitonic sort of n = 2^k 64-bit keys in LDS by the whole workgroup (one compare-exchange per
// thread and step; n log^2 n / 4 exchanges instead of the n^2 compares of a rank sort).
__device__ __forceinline__ void bitonic_sort(unsigned long long* a, int n, int tid) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n >> 1); t += kThreads) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i | j;
                const unsigned long long x = a[i], y = a[p];
                const bool up = (i & k) == 0;
                if ((x > y) == up) { a[i] = y; a[p] = x; }
            }
            __syncthreads();
        }
    }
}

// The same sort for n = 256 * 2^LR keys with the keys in registers: thread t holds elements t*R .. t*R+R-1 (R = 2^LR),
// so exchange distances below R stay inside a thread, distances R .. 32R are lane exchanges inside a wave (no LDS
// traffic, no barrier) and only the distances 64R and 128R (3 of the 45 steps at n = 512) go through LDS.  The LDS
// version is VALU-issue-bound (index arithmetic, two reads, two predicated writes and a barrier per step for every
// wave); this one issues about half the instructions.  Same result: the keys are unique, so ascending order is unique.
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int lane_mask) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, lane_mask, 64);
    const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), lane_mask, 64);
    return ((unsigned long long)hi << 32) | lo;
}
template <int LR>
__device__ __forceinline__ void bitonic_sort_regs(unsigned long long* a, int tid) {
    constexpr int R = 1 << LR, n = kThreads * R;
    const int base = tid * R;
    unsigned long long v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = a[base + r];
#pragma unroll
    for (int k = 2; k <= n; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < R) {                                  // inside the thread
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (r & j) continue;
                    const bool up = k == n || ((base + r) & k) == 0;
                    const unsigned long long x = v[r], y = v[r | j];
                    if ((x > y) == up) { v[r] = y; v[r | j] = x; }
                }
            } else {
                const bool cross_wave = j >= 64 * R;
                if (cross_wave) {
#pragma unroll
                    for (int r = 0; r < R; ++r) a[base + r] = v[r];
                    __syncthreads();
                }
                const bool lower = (base & j) == 0;          // (j >= R: the same for all r of a thread)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const bool up = k == n || ((base + r) & k) == 0;
                    const unsigned long long y = cross_wave ? a[(base + r) ^ j] : shfl_xor_u64(v[r], j >> LR);
                    const bool take_min = lower == up;
                    const bool y_less = y < v[r];
                    v[r] = (y_less == take_min) ? y : v[r];
                }
                if (cross_wave) __syncthreads();             // partners read before the next cross-wave step overwrites
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) a[base + r] = v[r];
    __syncthreads();
}

#ifdef YFS_NMS_TIMING
__device__ long long yfs_nms_dbg[2][16];
#define NMS_T(k) do { if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 517)) yfs_nms_dbg[blockIdx.x != 0][k] = clock64(); } while (0)
#else
#define NMS_T(k) do { } while (0)
#endif
template <bool FUSED>
__global__ void __launch_bounds__(kThreads) nms_kernel(NmsArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int wcnt[(YFS_NMS_MAX_BOXES + kThreads - 1) / kThreads][kThreads / 64];   // [round or parity][wave]
    __shared__ int ctl[4];
    __shared__ int cstart_s[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n = blockIdx.x;
    const int A = g.A, cap = g.cap, npow = g.npow;
    // carve (all offsets multiples of 16): candidates live at their slot for the whole kernel
    float4* cbox = reinterpret_cast<float4*>(smem);                                           // 16*cap  corners
    int* cidx = reinterpret_cast<int*>(smem + 16 * (size_t)cap);                               // 4*cap   box index
    unsigned long long* key = reinterpret_cast<unsigned long long*>(smem + 20 * (size_t)cap);  // 8*npow
    unsigned char* tail = smem + 20 * (size_t)cap + 8 * (size_t)npow;
    int* kflag = reinterpret_cast<int*>(tail);                                                 // 4*cap
    int* segs = kflag + cap;                                                                   // 4*cap
    unsigned long long* over = reinterpret_cast<unsigned long long*>(tail + 8 * (size_t)cap);  // 8*cap  overlap masks
    float* rawlds = reinterpret_cast<float*>(tail);   // fused: raw-head staging, dead before kflag / segs are used

    NMS_T(0);
    // ---- A: decode / filter, candidates compacted in box-index order (ballot prefix, no atomics) ----
    int K = 0;
    int par = 0;
    int maxcls = 0;            // largest class id among this thread's candidates
    if (tid == 0) ctl[0] = 0;
    auto compact = [&](bool pass, int bi, float4 corners, float score, int cls) {
        const unsigned long long bal = __ballot(pass);
        if (lane == 0) wcnt[par][wave] = __builtin_popcountll(bal);
        __syncthreads();
        int off = K, tot = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) {
            const int cw = wcnt[par][w];
            off += w < wave ? cw : 0;
            tot += cw;
        }
        if (pass) {
            const int slot = off + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
            cbox[slot] = corners;
            cidx[slot] = bi | (int)(((unsigned)cls & 0x1FFFFFu) << kSlotBits);   // [class:21 | box index:11]
            key[slot] = make_key(cls, score, slot);
            maxcls = ((unsigned)cls & 0x1FFFFFu) > (unsigned)maxcls ? (int)((unsigned)cls & 0x1FFFFFu) : maxcls;
        }
        K += tot;
        par ^= 1;
    };
    if (FUSED) {
        const int D = 5 + g.C;
        const float* rimg = g.raw + n * (int64_t)A * D;
        for (int b0 = 0; b0 < A; b0 += g.bpc) {
            const int nb = (A - b0) < g.bpc ? (A - b0) : g.bpc;
            const int skew = yfs_stage_floats_dma(rimg + (int64_t)b0 * D, nb * D, rawlds, kThreads / 64);
            yfs_stage_wait();
            __syncthreads();
            for (int t0 = 0; t0 < nb; t0 += kThreads) {
                const int t = t0 + tid;
                bool pass = false;
                float4 corners = make_float4(0.f, 0.f, 0.f, 0.f);
                float sc = 0.f;
                int cl = 0;
                const int bi = b0 + t;
                if (t < nb) {
                    const int a = bi % g.A_anch, cell = bi / g.A_anch;
                    Decoded d = decode_one(rawlds + skew + t * D, g.C, cell / g.S, cell % g.S, g.S, g.anchors[2 * a],
                                           g.anchors[2 * a + 1]);
                    const int64_t o = n * A + bi;
                    g.boxes_out[o] = make_float4(d.x, d.y, d.w, d.h);
                    g.score_out[o] = d.score;
                    g.cls_out[o] = d.cls;
                    pass = d.score >= g.score_thr;
                    corners = make_float4(d.x - 0.5f * d.w, d.y - 0.5f * d.h, d.x + 0.5f * d.w, d.y + 0.5f * d.h);
                    sc = d.score;
                    cl = d.cls;
                }
                compact(pass, bi, corners, sc, cl);
            }
            __syncthreads();   // everyone is done with rawlds before the next chunk lands in it
        }
    } else {
        // all rounds of 256 boxes at once: scores loaded together, one ballot per round, ONE barrier for the wave counts
        // of every round (instead of load -> ballot -> barrier per round), then the boxes and classes of the passing
        // candidates are fetched together and written to their slots
        constexpr int kMaxA = (YFS_NMS_MAX_BOXES + kThreads - 1) / kThreads;
        float scr[kMaxA];
        unsigned long long bal[kMaxA];
#pragma unroll
        for (int r = 0; r < kMaxA; ++r) {
            const int i = r * kThreads + tid;
            scr[r] = i < A ? g.score_in[n * A + i] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < kMaxA; ++r) {
            const int i = r * kThreads + tid;
            bal[r] = __ballot(i < A && scr[r] >= g.score_thr);
            if (lane == 0) wcnt[r][wave] = __builtin_popcountll(bal[r]);
        }
        __syncthreads();
        int off = 0;
#pragma unroll
        for (int r = 0; r < kMaxA; ++r) {
            if (r * kThreads >= A) break;            // block-uniform
            int mine = off;
#pragma unroll
            for (int w = 0; w < kThreads / 64; ++w) {
                const int cw = wcnt[r][w];
                mine += w < wave ? cw : 0;
                off += cw;
            }
            if ((bal[r] >> lane) & 1ull) {
                const int i = r * kThreads + tid;
                const int64_t o = n * A + i;
                const float4 v = g.boxes_in[o];
                const int cl = g.cls_in[o];
                const int slot = mine + __builtin_popcountll(bal[r] & ((1ull << lane) - 1ull));
                cbox[slot] = make_float4(v.x - 0.5f * v.z, v.y - 0.5f * v.w, v.x + 0.5f * v.z, v.y + 0.5f * v.w);
                cidx[slot] = i | (int)(((unsigned)cl & 0x1FFFFFu) << kSlotBits);   // [class:21 | box index:11]
                key[slot] = make_key(cl, scr[r], slot);
                maxcls = ((unsigned)cl & 0x1FFFFFu) > (unsigned)maxcls ? (int)((unsigned)cl & 0x1FFFFFu) : maxcls;
            }
        }
        K = off;
        __syncthreads();
    }

    // ---- B: sort by (class, score desc, index): every class becomes one contiguous, score-ordered segment ----
    // Class-wise NMS never compares boxes of different classes, so suppression runs per segment: with ~450
    // candidates in 20 classes that is ~5 K IoU evaluations per image instead of the ~100 K of an all-pairs mask.
    // Two orders are needed: class-major for the suppression, global (score, index) for the output.  When every class id
    // is below 64 (any YOLO head) ONE bitonic sort by (score, index) serves both: a stable counting sort by class on top
    // of it (ballot ranks per 64-position chunk, 64 x chunks table of counts) yields the class segments in score order,
    // and the output is an ordered compaction of the kept flags along the sorted array.  Otherwise the keys carry the
    // class in their top bits and a second sort restores the output order (phase D).
    {
        int mc = maxcls;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const int o = __shfl_xor(mc, d, 64);
            mc = o > mc ? o : mc;
        }
        if (lane == 0) atomicMax(&ctl[0], mc);
    }
    int ns = 2;
    while (ns < K) ns <<= 1;                       // K <= cap <= npow
    __syncthreads();
    NMS_T(1);
    const bool onesort = ctl[0] < 64 && cap >= 128;   // (cslot sits behind 64 segment starts inside the 4*cap segs region)
    for (int p = K + tid; p < ns; p += kThreads) key[p] = ~0ull;
    if (onesort)
        for (int p = tid; p < K; p += kThreads) key[p] &= (1ull << kClsShift) - 1ull;
    __syncthreads();
    if (ns == kThreads * 2) bitonic_sort_regs<1>(key, tid);          // block-uniform
    else if (ns == kThreads * 4) bitonic_sort_regs<2>(key, tid);
    else bitonic_sort(key, ns, tid);
    NMS_T(2);
    int nseg = 0;
    unsigned short* cslot = reinterpret_cast<unsigned short*>(segs + 64);   // onesort: slot of the q-th box, class-major
    if (onesort) {
        unsigned short* cnt = reinterpret_cast<unsigned short*>(over);      // [chunks][64]; `over` is free until C1
        const int nchunks = (K + 63) >> 6;
        for (int e = tid; e < nchunks * 64; e += kThreads) cnt[e] = 0;
        __syncthreads();
        constexpr int kMaxR = (YFS_NMS_MAX_BOXES + kThreads - 1) / kThreads;
        int rankw[kMaxR], clsr[kMaxR], slotr[kMaxR];
#pragma unroll
        for (int r = 0; r < kMaxR; ++r) {
            const int p = tid + r * kThreads;              // chunk = wave + 4 r, lane = position inside the chunk
            rankw[r] = 0; clsr[r] = -1; slotr[r] = 0;
            if (r * kThreads < K) {                         // wave-uniform
                const bool valid = p < K;
                if (valid) {
                    slotr[r] = (int)(key[p] & kSlotMask);
                    clsr[r] = (int)((unsigned)cidx[slotr[r]] >> kSlotBits);
                }
                // lanes of the chunk that hold the same class: six ballots (one per class bit) instead of one round per
                // distinct class (~19 rounds of readlane + ballot for 64 boxes in 20 classes)
                unsigned long long m = __ballot(valid);
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    const bool bit = (clsr[r] >> b) & 1;
                    const unsigned long long bal = __ballot(valid && bit);
                    m &= bit ? bal : ~bal;
                }
                if (valid) {
                    rankw[r] = __builtin_popcountll(m & ((1ull << lane) - 1ull));
                    if (rankw[r] == 0) cnt[(wave + 4 * r) * 64 + clsr[r]] = (unsigned short)__builtin_popcountll(m);
                }
            }
        }
        __syncthreads();
        if (tid < 64) {                                     // per class: exclusive prefix over the chunks, total
            int run = 0;
            for (int ch = 0; ch < nchunks; ++ch) {
                const int v = cnt[ch * 64 + tid];
                cnt[ch * 64 + tid] = (unsigned short)run;
                run += v;
            }
            int incl = run;                                 // exclusive scan over the 64 classes (this is wave 0)
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(incl, d, 64);
                if (lane >= d) incl += v;
            }
            const int start = incl - run;
            cstart_s[tid] = start;
            const unsigned long long ne = __ballot(run > 0);
            if (run > 0) segs[__builtin_popcountll(ne & ((1ull << lane) - 1ull))] = start;
            if (tid == 0) ctl[2] = __builtin_popcountll(ne);
        }
        __syncthreads();
        nseg = ctl[2];
#pragma unroll
        for (int r = 0; r < kMaxR; ++r) {
            const int p = tid + r * kThreads;
            if (p < K) {
                const int q = cstart_s[clsr[r]] + cnt[(wave + 4 * r) * 64 + clsr[r]] + rankw[r];
                cslot[q] = (unsigned short)slotr[r];
                cidx[slotr[r]] |= q << kQShift;     // [q:11 | class:6 | box index:11]: phase D finds the keep flag of a slot
            }
        }
        __syncthreads();
    } else {
    // segment starts in ascending order (ordered ballot compaction, as in phase A) -> lengths by subtraction
    for (int p0 = 0; p0 < K; p0 += kThreads) {
        const int p = p0 + tid;
        const bool st = p < K && (p == 0 || (key[p] >> kClsShift) != (key[p - 1] >> kClsShift));
        const unsigned long long bal = __ballot(st);
        if (lane == 0) wcnt[par][wave] = __builtin_popcountll(bal);
        __syncthreads();
        int off = nseg, tot = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) {
            const int cw = wcnt[par][w];
            off += w < wave ? cw : 0;
            tot += cw;
        }
        if (st) segs[off + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = p;
        nseg += tot;
        par ^= 1;
    }
    __syncthreads();
    }
    // slot of the q-th box in class-major order
    auto member = [&](int q) -> int { return onesort ? (int)cslot[q] : (int)(key[q] & kSlotMask); };
    {   // boxes into class-major order, in place through registers: phase C then reads cbox[q] with no indirection (one
        // dependent LDS read less per pair, and the next pair's address is known early enough to prefetch)
        constexpr int kMaxQ = (YFS_NMS_MAX_BOXES + kThreads - 1) / kThreads;
        float4 tmp[kMaxQ];
#pragma unroll
        for (int r = 0; r < kMaxQ; ++r) {
            const int q = tid + r * kThreads;
            if (q < K) tmp[r] = cbox[member(q)];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kMaxQ; ++r) {
            const int q = tid + r * kThreads;
            if (q < K) cbox[q] = tmp[r];
        }
        // (visible to phase C through the barrier that follows the pair-count scan)
    }

    // ---- C: greedy suppression per class segment ----
    // The textbook loop (for every kept box, test all later boxes) is one serial chain per segment and costs the same
    // ~45 VALU per step whether 5 or 64 lanes hold a member: with ~23 members per class most lanes idle.  Instead:
    //   C1  the overlap bitmask of every member: over[j] bit i = (IoU(i, j) > threshold) for the earlier members i < j of
    //       its segment (segments of <= 64 members).  All (i, j) pairs of all segments form one index space; every thread
    //       takes a contiguous run of it (one binary search over prefix sums of pair counts + one triangular-root, then
    //       (i, j) advance incrementally), so the ~5 K IoUs of an image are spread evenly over the 256 threads whatever
    //       the class histogram is; the bits of one j are collected in a register and published with one LDS atomic OR.
    //       The phase is VALU-issue-bound (~50 instructions per pair, 4 cycles each per wave), so balance is what
    //       counts: per-pair work items (search + root per pair) took 36 K cycles, rows t and len-1-t of a segment as
    //       one item (no atomics, but the longest segment sets the trip count of its wave) 20-28 K, this form 18-25 K.
    // Segments longer than 64 keep the wave-per-segment loop.
    NMS_T(3);
    int* ppre = kflag;   // [nseg] exclusive prefix sums of item counts; the region becomes the keep flags after C1
    auto seg_start = [&](int sgi) { return segs[sgi]; };
    auto seg_length = [&](int sgi) { return (sgi + 1 < nseg ? segs[sgi + 1] : K) - segs[sgi]; };
    if (wave == 0) {     // exclusive scan of the pair counts, 64 segments per round; number of long segments
        int carry = 0, nl = 0;
        for (int s0 = 0; s0 < nseg; s0 += 64) {
            const int sgi = s0 + lane;
            int cnt = 0;
            bool lng = false;
            if (sgi < nseg) {
                const int len = seg_length(sgi);
                cnt = len <= 64 ? len * (len - 1) / 2 : 0;
                lng = len > 64;
            }
            nl += __builtin_popcountll(__ballot(lng));
            int incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(incl, d, 64);
                if (lane >= d) incl += v;
            }
            if (sgi < nseg) ppre[sgi] = carry + incl - cnt;
            carry += __shfl(incl, 63, 64);
        }
        if (lane == 0) { ctl[3] = carry; ctl[2] = nl; }   // total number of pairs; long segments (ctl[2] was nseg: consumed)
    }
    for (int p = tid; p < K; p += kThreads) over[p] = 0ull;
    __syncthreads();
    NMS_T(4);
    const int P = ctl[3];
    {
        const int run = (P + kThreads - 1) / kThreads;
        int q = tid * run;
        const int qend = q + run < P ? q + run : P;
        if (q < qend) {
            int lo = 0, hi = nseg;                   // last segment with ppre <= q (empty ones share their successor's value)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (ppre[mid] <= q) lo = mid; else hi = mid;
            }
            const int r = q - ppre[lo];
            int j = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)r)) * 0.5f);
            while (j * (j - 1) / 2 > r) --j;
            while ((j + 1) * j / 2 <= r) ++j;
            int i = r - j * (j - 1) / 2;               // 0 <= i < j < len
            int p0 = seg_start(lo), len = seg_length(lo);
            float4 bq = cbox[p0 + j];
            float4 a = cbox[p0 + i];
            unsigned long long bits = 0ull;          // overlaps found for the current j
            for (;;) {
                const float4 an = cbox[p0 + i + 1];  // next pair of this row (i + 1 <= j: in range), lands during the IoU
                const float v = nms_pair_iou(a, bq);
                if (v > g.iou_thr) bits |= 1ull << i;
                const bool last = ++q == qend;
                a = an;
                if (++i == j || last) {              // leaving this j: publish its bits (other threads may hold part of it)
                    if (bits) atomicOr(&over[p0 + j], bits);
                    bits = 0ull;
                    if (last) break;
                    i = 0;
                    if (++j == len) {                // next segment that has pairs (exists: q < P)
                        do { ++lo; len = seg_length(lo); } while (len < 2 || len > 64);
                        p0 = seg_start(lo);
                        j = 1;
                    }
                    bq = cbox[p0 + j];
                    a = cbox[p0];
                }
            }
        }
    }
    __syncthreads();
    NMS_T(5);
    // keep flags, indexed by class-major position (the same index as `over`)
    for (int sgi = tid; sgi < nseg; sgi += kThreads) {
        const int p0 = seg_start(sgi), len = seg_length(sgi);
        if (len > 64) continue;                      // the wave loop below owns these
        unsigned long long kept = 0;
#pragma unroll 4
        for (int j = 0; j < len; ++j)                // loads only: independent of `kept`, so they are issued ahead
            if ((over[p0 + j] & kept) == 0ull) kept |= 1ull << j;
        for (int j = 0; j < len; ++j) kflag[p0 + j] = (int)((kept >> j) & 1ull);
    }
    if (ctl[2] > 0) {   // long segments (rare): one wave each; member lane + 64 t <-> bit t of this lane's `rem`
        int nlong = 0;
        for (int sgi = 0; sgi < nseg; ++sgi) {
            const int len = seg_length(sgi);
            if (len <= 64) continue;
            if ((nlong++ & (kThreads / 64 - 1)) != wave) continue;
            const int p0 = seg_start(sgi);
            const int T = (len + 63) >> 6;
            for (int j = lane; j < len; j += 64) kflag[p0 + j] = 0;   // (same wave as the writes of 1 below: in order)
            unsigned long long rem = 0;
            for (int i = 0; i < len; ++i) {
                const int ti = i >> 6, own = i & 63;
                const unsigned lo32 = (unsigned)__shfl((int)(unsigned)(rem & 0xffffffffull), own, 64);
                const unsigned hi32 = (unsigned)__shfl((int)(unsigned)(rem >> 32), own, 64);
                const unsigned long long w = ((unsigned long long)hi32 << 32) | lo32;
                if ((w >> ti) & 1ull) continue;
                if (lane == 0) kflag[p0 + i] = 1;
                const float4 pb = cbox[p0 + i];
                for (int t = ti; t < T; ++t) {
                    const int j = lane + 64 * t;
                    if (j > i && j < len) {
                        const float4 q = cbox[p0 + j];
                        const float v = yfs_iou_corners(pb.x, pb.y, pb.z, pb.w, q.x, q.y, q.z, q.w);
                        if (v > g.iou_thr) rem |= 1ull << t;
                    }
                }
            }
        }
    }
    if (tid == 0) ctl[1] = 0;
    __syncthreads();
    NMS_T(6);

    // ---- D: kept boxes in (score desc, index) order ----
    int32_t* kp = g.keep + n * (int64_t)A;
    if (onesort) {
        // the array is already in output order: ordered compaction of the keep flags along it
        int nk = 0;
        for (int p0 = 0; p0 < K; p0 += kThreads) {
            const int p = p0 + tid;
            int slot = 0;
            bool f = false;
            if (p < K) {
                slot = (int)(key[p] & kSlotMask);
                f = kflag[(unsigned)cidx[slot] >> kQShift] != 0;
            }
            const unsigned long long bal = __ballot(f);
            if (lane == 0) wcnt[par][wave] = __builtin_popcountll(bal);
            __syncthreads();
            int off = nk, tot = 0;
#pragma unroll
            for (int w = 0; w < kThreads / 64; ++w) {
                const int cw = wcnt[par][w];
                off += w < wave ? cw : 0;
                tot += cw;
            }
            if (f) kp[off + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = cidx[slot] & (int)kSlotMask;
            nk += tot;
            par ^= 1;
        }
        for (int i = nk + tid; i < A; i += kThreads) kp[i] = -1;
        if (tid == 0) g.count[n] = nk;
        NMS_T(7);
        return;
    }
    // class-major keys: drop the class field and sort once more
    for (int p0 = 0; p0 < ns; p0 += kThreads) {
        const int p = p0 + tid;
        bool f = false;
        if (p < ns) {
            const unsigned long long k = key[p];
            f = p < K && kflag[p] != 0;
            key[p] = f ? (k & ((1ull << kClsShift) - 1ull)) : ~0ull;
        }
        const unsigned long long bal = __ballot(f);
        if (lane == 0 && bal) atomicAdd(&ctl[1], __builtin_popcountll(bal));   // a count: order does not matter
    }
    __syncthreads();
    const int nk = ctl[1];
    bitonic_sort(key, ns, tid);
    for (int i = tid; i < A; i += kThreads) kp[i] = i < nk ? (cidx[key[i] & kSlotMask] & (int)kSlotMask) : -1;
    if (tid == 0) g.count[n] = nk;
}

size_t nms_lds_bytes(int cap, int npow, bool fused, int bpc, int D) {
    size_t head = 20 * (size_t)cap + 8 * (size_t)npow;
    size_t tail = 16 * (size_t)cap;   // keep flags / pair prefix, segment starts, overlap masks
    if (fused) {
        size_t rawb = (size_t)YFS_STAGE_FLOATS(bpc * D) * 4;
        if (rawb > tail) tail = rawb;
    }
    return head + tail;
}

int pow2_at_least(int v) {
    int p = 2;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace

// =============================== C ABI ========================================================
YFS_API size_t yfs_iou_xywh_workspace_bytes(int64_t n) {
    return n > 0 ? (size_t)yfs_cdiv(n, kThreads) * sizeof(float) : 0;
}

YFS_API int yfs_iou_xywh_f32(const float* pred, const float* gt, int64_t n, float* iou_out, float* mean_out,
                             void* workspace, size_t workspace_bytes, void* stream) {
    YFS_REQUIRE(pred && gt && n > 0, YFS_ERR_INVALID_ARG, "yfs_iou_xywh_f32: null input or n <= 0 (n=%lld)", (long long)n);
    YFS_REQUIRE(iou_out || mean_out, YFS_ERR_INVALID_ARG, "yfs_iou_xywh_f32: no output requested");
    YFS_REQUIRE(!mean_out || (workspace && workspace_bytes >= yfs_iou_xywh_workspace_bytes(n)), YFS_ERR_WORKSPACE,
                "yfs_iou_xywh_f32: workspace needs %zu bytes", yfs_iou_xywh_workspace_bytes(n));
    hipStream_t s = (hipStream_t)stream;
    const int64_t nblk = yfs_cdiv(n, kThreads);
    iou_rows_kernel<<<dim3((unsigned)nblk), dim3(kThreads), 0, s>>>((const float4*)pred, (const float4*)gt, n, iou_out,
                                                                    mean_out ? (float*)workspace : nullptr);
    YFS_CHECK_LAUNCH("iou_rows_kernel");
    if (mean_out) {
        finalize_sum_kernel<<<1, kThreads, 0, s>>>((const float*)workspace, nblk, 1.0f / (float)n, mean_out);
        YFS_CHECK_LAUNCH("finalize_sum_kernel");
    }
    return YFS_OK;
}

YFS_API int yfs_iou_variant_f32(const float* pred, const float* gt, int64_t n, int variant, float* iou_out,
                                void* stream) {
    YFS_REQUIRE(pred && gt && iou_out && n > 0, YFS_ERR_INVALID_ARG, "yfs_iou_variant_f32: null pointer or n <= 0");
    YFS_REQUIRE(variant >= 0 && variant <= 2, YFS_ERR_INVALID_ARG, "yfs_iou_variant_f32: variant %d not in 0..2", variant);
    iou_variant_kernel<<<dim3((unsigned)yfs_cdiv(n, kThreads)), dim3(kThreads), 0, (hipStream_t)stream>>>(
        (const float4*)pred, (const float4*)gt, n, variant, iou_out);
    YFS_CHECK_LAUNCH("iou_variant_kernel");
    return YFS_OK;
}

YFS_API int yfs_pairwise_iou_f32(const float* a, int64_t na, const float* b, int64_t nb, float* out, void* stream) {
    YFS_REQUIRE(a && b && out && na > 0 && nb > 0, YFS_ERR_INVALID_ARG, "yfs_pairwise_iou_f32: null pointer or empty input");
    dim3 grid((unsigned)yfs_cdiv(nb, kPairCols), (unsigned)yfs_cdiv(na, 64));
    YFS_REQUIRE(grid.y <= 65535, YFS_ERR_UNSUPPORTED, "yfs_pairwise_iou_f32: na=%lld too large", (long long)na);
    pairwise_iou_kernel<<<grid, dim3(kThreads), 0, (hipStream_t)stream>>>((const float4*)a, na, (const float4*)b, nb, out);
    YFS_CHECK_LAUNCH("pairwise_iou_kernel");
    return YFS_OK;
}

YFS_API int yfs_decode_head_f32(const float* t, int64_t n, float prior_w, float prior_h, float* out, void* stream) {
    YFS_REQUIRE(t && out && n > 0, YFS_ERR_INVALID_ARG, "yfs_decode_head_f32: null pointer or n <= 0");
    decode_head_kernel<<<dim3((unsigned)yfs_cdiv(n, kThreads)), dim3(kThreads), 0, (hipStream_t)stream>>>(
        (const float4*)t, n, prior_w, prior_h, (float4*)out);
    YFS_CHECK_LAUNCH("decode_head_kernel");
    return YFS_OK;
}

YFS_API int yfs_decode_xywh_f32(const float* t, int64_t n, float* out, void* stream) {
    YFS_REQUIRE(t && out && n > 0, YFS_ERR_INVALID_ARG, "yfs_decode_xywh_f32: null pointer or n <= 0");
    decode_xywh_kernel<<<dim3((unsigned)yfs_cdiv(n, kThreads)), dim3(kThreads), 0, (hipStream_t)stream>>>(
        (const float4*)t, n, (float4*)out);
    YFS_CHECK_LAUNCH("decode_xywh_kernel");
    return YFS_OK;
}

static int pick_boxes_per_block(int D) {
    int bpb = kThreads;
    while ((size_t)bpb * D * 4 > 48 * 1024 && bpb > 32) bpb >>= 1;
    return bpb;
}

YFS_API int yfs_decode_grid_f32(const float* raw, int64_t N, int S, int A, int C, const float* anchors, float* boxes,
                                float* score, int32_t* cls, float* all_scores, void* stream) {
    YFS_REQUIRE(raw && anchors && boxes && score && cls, YFS_ERR_INVALID_ARG, "yfs_decode_grid_f32: null pointer");
    YFS_REQUIRE(N > 0 && S > 0 && A > 0 && C > 0 && C <= 1024, YFS_ERR_INVALID_ARG,
                "yfs_decode_grid_f32: bad shape N=%lld S=%d A=%d C=%d", (long long)N, S, A, C);
    const int D = 5 + C;
    const int bpb = pick_boxes_per_block(D);
    const int64_t nboxes = N * S * S * A;
    const size_t lds = (size_t)YFS_STAGE_FLOATS(bpb * D) * 4;
    decode_grid_kernel<<<dim3((unsigned)yfs_cdiv(nboxes, bpb)), dim3(kThreads), lds, (hipStream_t)stream>>>(
        raw, nboxes, S, A, C, anchors, bpb, (float4*)boxes, score, cls, all_scores);
    YFS_CHECK_LAUNCH("decode_grid_kernel");
    return YFS_OK;
}

YFS_API int yfs_decode_v1_f32(const float* pred, int64_t N, int S, int B, int C, float* boxes, float* score,
                              int32_t* cls, void* stream) {
    YFS_REQUIRE(pred && boxes && score && cls, YFS_ERR_INVALID_ARG, "yfs_decode_v1_f32: null pointer");
    YFS_REQUIRE(N > 0 && S > 0 && B > 0 && C > 0 && (5 * B + C) <= 40, YFS_ERR_INVALID_ARG,
                "yfs_decode_v1_f32: bad shape N=%lld S=%d B=%d C=%d (5B+C <= 40)", (long long)N, S, B, C);
    const int D = 5 * B + C;
    const int64_t ncells = N * S * S;
    const size_t lds = (size_t)YFS_STAGE_FLOATS(kThreads * D) * 4;
    decode_v1_kernel<<<dim3((unsigned)yfs_cdiv(ncells, kThreads)), dim3(kThreads), lds, (hipStream_t)stream>>>(
        pred, ncells, S, B, C, (float4*)boxes, score, cls);
    YFS_CHECK_LAUNCH("decode_v1_kernel");
    return YFS_OK;
}

YFS_API int yfs_nms_classwise_f32(const float* boxes, const float* score, const int32_t* cls, int64_t N, int A,
                                  float score_thr, float iou_thr, int32_t* keep, int32_t* count, void* stream) {
    YFS_REQUIRE(boxes && score && cls && keep && count, YFS_ERR_INVALID_ARG, "yfs_nms_classwise_f32: null pointer");
    YFS_REQUIRE(N > 0 && A > 0, YFS_ERR_INVALID_ARG, "yfs_nms_classwise_f32: bad shape N=%lld A=%d", (long long)N, A);
    YFS_REQUIRE(A <= YFS_NMS_MAX_BOXES, YFS_ERR_UNSUPPORTED, "yfs_nms_classwise_f32: A=%d exceeds %d boxes per image", A,
                YFS_NMS_MAX_BOXES);
    NmsArgs g = {};
    g.boxes_in = (const float4*)boxes; g.score_in = score; g.cls_in = cls;
    g.keep = keep; g.count = count; g.A = A; g.cap = (A + 63) & ~63; g.npow = pow2_at_least(g.cap);
    g.score_thr = score_thr; g.iou_thr = iou_thr;
    const size_t lds = nms_lds_bytes(g.cap, g.npow, false, 0, 0);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)nms_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        YFS_REQUIRE(e == hipSuccess, YFS_ERR_LAUNCH, "yfs_nms_classwise_f32: cannot reserve %zu bytes of LDS", lds);
    }
    nms_kernel<false><<<dim3((unsigned)N), dim3(kThreads), lds, (hipStream_t)stream>>>(g);
    YFS_CHECK_LAUNCH("nms_kernel");
    return YFS_OK;
}

YFS_API int yfs_decode_nms_f32(const float* raw, int64_t N, int S, int A, int C, const float* anchors, float score_thr,
                               float iou_thr, float* boxes, float* score, int32_t* cls, int32_t* keep, int32_t* count,
                               void* stream) {
    YFS_REQUIRE(raw && anchors && boxes && score && cls && keep && count, YFS_ERR_INVALID_ARG, "yfs_decode_nms_f32: null pointer");
    YFS_REQUIRE(N > 0 && S > 0 && A > 0 && C > 0 && C <= 1024, YFS_ERR_INVALID_ARG,
                "yfs_decode_nms_f32: bad shape N=%lld S=%d A=%d C=%d", (long long)N, S, A, C);
    const int64_t per = (int64_t)S * S * A;
    YFS_REQUIRE(per <= YFS_NMS_MAX_BOXES, YFS_ERR_UNSUPPORTED, "yfs_decode_nms_f32: %lld boxes per image exceeds %d",
                (long long)per, YFS_NMS_MAX_BOXES);
    // Default: two launches.  The decode is VALU-bound (25 expf per box) and perfectly parallel over boxes, while inside
    // the per-image NMS workgroup it could only proceed in LDS-sized chunks of 128 boxes, each paying the staging latency
    // on the image's critical path (7 chunks = 36 % of the kernel).  Decoded boxes / scores / classes are outputs of this
    // entry point anyway, so the only extra traffic is the NMS kernel reading the 3.4 KB of scores per image back
    // (+ the boxes that pass).  Measured on config 4: 0.102 -> two launches; YFS_NMS_FUSED=1 keeps the one-launch kernel.
    const bool fused = yfs_knob(YFS_KNOB_NMS_FUSED) != 0;
    if (!fused) {
        int rc = yfs_decode_grid_f32(raw, N, S, A, C, anchors, boxes, score, cls, nullptr, stream);
        if (rc) return rc;
        return yfs_nms_classwise_f32(boxes, score, cls, N, (int)per, score_thr, iou_thr, keep, count, stream);
    }
    NmsArgs g = {};
    g.raw = raw; g.anchors = anchors; g.S = S; g.A_anch = A; g.C = C;
    g.boxes_out = (float4*)boxes; g.score_out = score; g.cls_out = cls;
    g.keep = keep; g.count = count; g.A = (int)per; g.cap = ((int)per + 63) & ~63; g.npow = pow2_at_least(g.cap);
    // 128-box staging chunks: at S=13, A=5, C=20 the workgroup then needs 39 KiB of LDS and four fit on a CU, so a
    // batch of 1024 images is one resident round instead of 1.33 rounds of three per CU
    g.bpc = pick_boxes_per_block(5 + C);
    if (g.bpc > 128) g.bpc = 128;
    g.score_thr = score_thr; g.iou_thr = iou_thr;
    const size_t lds = nms_lds_bytes(g.cap, g.npow, true, g.bpc, 5 + C);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)nms_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        YFS_REQUIRE(e == hipSuccess, YFS_ERR_LAUNCH, "yfs_decode_nms_f32: cannot reserve %zu bytes of LDS", lds);
    }
    nms_kernel<true><<<dim3((unsigned)N), dim3(kThreads), lds, (hipStream_t)stream>>>(g);
    YFS_CHECK_LAUNCH("nms_kernel<fused>");
    return YFS_OK;
}

#ifdef YFS_NMS_TIMING
YFS_API int yfs_debug_nms_times(long long* host32) {
    return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(yfs_nms_dbg), sizeof(long long) * 32);
}
#endif
