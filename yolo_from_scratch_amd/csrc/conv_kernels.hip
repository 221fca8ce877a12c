// Kernel set 1: Conv2D block (Conv2d k/s, pad k/2, bias + LeakyReLU) forward / dgrad / wgrad as
// bf16 MFMA implicit GEMMs for gfx950.  Replaces the 24 Conv2D(...) call sites of the reference
// backbone (yolo_toy_regressor_torch.py:206-211, :218-254) and their autograd.
//
// Layout: activations NHWC bf16, weights [Cout][KH][KW][Cin] bf16 (reduction index contiguous),
// fp32 accumulation in v_mfma_f32_16x16x32_bf16.
//
//   forward / dgrad ("igemm"): out[pixel][n] = sum_{tap,c} in[gather(pixel,tap)][c] * wt[n][tap][c]
//     MFMA roles: A operand = weight rows (n), B operand = gathered pixels, so every lane ends up with
//     4 consecutive output channels of one pixel.  Operands go global -> LDS directly (buffer_load ... lds,
//     16 B per lane; conv padding and tile tails are out-of-range offsets that the hardware zero-fills), LDS
//     rows of 128 B XOR-swizzled by (row & 7) on the SOURCE side so the fragment ds_read_b128 is conflict
//     free; the epilogue works on the accumulators in place: bias (or folded-BN scale/shift) / LeakyReLU / the
//     activation-derivative mask run in fp32, one v_permlane16_swap per register pairs two fragments so that the
//     bf16 result leaves as whole 16 B per lane (epi_swap8; only the pool-fused form goes through an fp32 LDS tile).
//     igemm_kernel: BM x BN x 64 tiles, 4 or 8 waves, 2-stage ring; igemm8_kernel: 256 (224) x 256 tiles,
//     8 waves in two staggered groups, half-tiles staged two K-steps ahead, one persistent workgroup per CU that
//     issues the next tile's first loads from inside the epilogue (the long-K 3x3 layers).
//   wgrad: dw[co][tap][ci] = sum_pixel dz[pixel][co] * x[gather(pixel,tap)][ci] — both operands have the
//     reduction index (pixel) as the slow axis, so the LDS tiles keep [pixel][channel] rows and the
//     fragments are read with ds_read_b64_tr_b16 (hardware transpose); split-K over pixels with fp32
//     partial slabs and a fixed-order reduction (deterministic).
#include <algorithm>
#include <atomic>
#include <type_traits>
#include "yfs_common.h"
#include <cstdlib>

namespace {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// Raw buffer descriptor over a whole tensor: loads whose byte offset is >= num_records return zeros in
// hardware, which is how conv padding, tile tails and strided-dgrad holes are produced without branches
// (a predicated global_load forces the compiler to wait for it right away; these stay in flight).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
}

constexpr int kThreads = 256;
constexpr int BK = 64;   // reduction depth per LDS tile (128 B rows)

enum { EPI_BIAS_LEAKY = 0, EPI_MASK = 1, EPI_NONE = 2, EPI_POOL = 3, EPI_PARTIAL = 4, EPI_UNPOOL = 5 };

// n / d for 0 <= n < 2^31 with a host-prepared multiplier (Granlund-Montgomery round-up form).
struct FastDiv {
    unsigned mul, shift, d;
};
FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    f.d = d;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    f.shift = l;
    f.mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) { return (__umulhi(n, f.mul) + n) >> f.shift; }

struct IgemmArgs {
    const bf16_t* x;      // [N,H,W,C]
    const bf16_t* w;      // [NOUT][taps][C]
    const float* bias;    // [NOUT]            (EPI_BIAS_LEAKY, EPI_POOL)
    const float* scale;   // [NOUT] or null    (folded BatchNorm: y = leaky(acc * scale + bias); null = 1, the reference's block)
    const bf16_t* act;    // [M][NOUT]         (EPI_MASK: multiply by act > 0 ? 1 : slope)
    unsigned short* pool_idx;   // EPI_POOL: [M/4][NOUT/8] eight 2-bit window positions each; out is then [M/4][NOUT]
                                // EPI_UNPOOL (read): [M][NOUT/8], the record of the 2x2 pool whose OUTPUT the rows are; out is then
                                // [N][2*OH][2*OW][NOUT]: EPI_MASK's value at the recorded position of each window, zeros elsewhere
    bf16_t* out;          // [M][NOUT]
    int N, H, W, C;
    int OH, OW, NOUT;
    unsigned xbytes, wbytes;
    int KH, KW;           // taps = KH*KW
    int PS;               // elements between adjacent pixels of x (== C except for the space-to-depth stem)
    int s_mul, pad;       // src = o*s_mul - pad + k
    int M;                // N*OH*OW
    int mtiles, ntiles;
    float slope;
    // SUB (dgrad of a stride-s conv): the output pixels are split into s*s parity classes (ih % s, iw % s).  Inside
    // a class only every s-th tap per axis touches a real dz pixel, and those taps form an ordinary stride-1
    // gather over the class's own (OH/s x OW/s) grid, so each class is a dense implicit GEMM with ~1/s^2 of the
    // taps instead of a dilated gather that multiplies zeros 1 - 1/s^2 of the time.
    FastDiv div_hw, div_w;     // igemm8_kernel: row -> (n, oh, ow) without integer divisions (the pooled grid for EPI_POOL)
    int sub_s;                 // s
    int cls_ph[4], cls_pw[4];  // class parities, heaviest class (most taps) first
    int cls_end[4];            // running end of each class's logical block range (m-tiles of the class x ntiles)
    // split-K (igemm8_kernel<EPI_PARTIAL>): the grid is ksplit copies of the tile grid, copy s reduces K-steps
    // [s * ksteps_per_split, ...) and writes its fp32 tile to partial[s][M][NOUT]; splitk_epilogue_kernel sums the copies
    // in index order and applies the epilogue.  ksplit = 1 everywhere else.
    float* partial;
    size_t partial_cap;        // bytes the caller's workspace holds (0 = none: no split-K)
    int ksplit, ksteps_per_split;
    int want_epi;              // the epilogue the split-K reduction applies (EPI_BIAS_LEAKY / EPI_MASK / EPI_NONE)
};

// XCD-aware, bijective remap: blocks that share blockIdx % 8 share an XCD (and its L2); give each XCD a
// contiguous run of logical tiles so the NT blocks that re-read one pixel tile hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

__device__ __forceinline__ unsigned swz(int row, int chunk) { return (unsigned)(row * 128 + ((chunk ^ (row & 7)) << 4)); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- epilogue straight from the MFMA accumulators (igemm_kernel and igemm8_kernel, every form but the pool-fused one) ----
// A lane of a 16x16 output fragment holds 4 consecutive channels of one pixel.  One v_permlane16_swap per register trades
// them with the lane 16 further on (the next 4 channels of the same pixel) against that lane's values of a second
// fragment: afterwards every lane owns 8 consecutive channels of one pixel, i.e. one 16 B store, 64 B runs per pixel row
// and wave instruction.  x = the fragment that lanes with even lane>>4 keep, y = the one lanes with odd lane>>4 keep;
// v[0..3] / v[4..7] = the lower / upper 4 channels of the lane's run.
// The earlier form carried the fp32 tile through LDS (16 writes, a barrier, 8 read -> arithmetic -> store sweeps that the
// row guard kept in separate basic blocks, twice per tile): ~11 K cycles per 224x256 tile whether 32 or 896 workgroups
// were on the chip (tools/probes/igemm8_epilogue_alone.py), i.e. its own latency chain, not the memory system; this one
// takes ~8 K (forward) and ~10.5 K instead of ~16.5 K (masked dgrad).  Same arithmetic per element (fma, compare,
// multiply, one rounding): results are bit-identical.
__device__ __forceinline__ void epi_swap8(const f32x4& x, const f32x4& y, float (&v)[8]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // (scalar copies first: __builtin_bit_cast applied to x[r] itself reads element 0 for every r with this clang)
        const float xr = x[r], yr = y[r];
        const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, xr), __builtin_bit_cast(unsigned, yr), false, false);
        v[r] = __builtin_bit_cast(float, (unsigned)sw[0]);
        v[4 + r] = __builtin_bit_cast(float, (unsigned)sw[1]);
    }
}
// 8 channels of one output pixel after the layer's epilogue arithmetic, rounded once; av = the activation the derivative
// mask is taken from (EPI_MASK, EPI_UNPOOL only)
template <int EPI>
__device__ __forceinline__ bf16x8 epi_value8(const IgemmArgs& g, float (&v)[8], const float (&sv)[8], const float (&bv)[8],
                                             const bf16x8& av) {
    if (EPI == EPI_BIAS_LEAKY) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float tv = __builtin_fmaf(v[j], sv[j], bv[j]);
            v[j] = tv > 0.0f ? tv : tv * g.slope;
        }
    } else if (EPI == EPI_MASK || EPI == EPI_UNPOOL) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)av[j] > 0.0f ? v[j] : v[j] * g.slope;
    }
    bf16x8 ov;
#pragma unroll
    for (int j = 0; j < 8; ++j) ov[j] = (bf16_t)v[j];
    return ov;
}
// o = element offset of the 8 channels in out (and in act)
template <int EPI>
__device__ __forceinline__ void epi_emit8(const IgemmArgs& g, long long o, float (&v)[8], const float (&sv)[8], const float (&bv)[8],
                                          const bf16x8& av) {
    *reinterpret_cast<bf16x8*>(g.out + o) = epi_value8<EPI>(g, v, sv, bv, av);
}
// EPI_UNPOOL = EPI_MASK + the backward of the 2x2 max pool in front of the layer (a dgrad whose input is a pool's output):
// the masked value goes to the window position the pool recorded, zeros to the other three (r[k]: k = dy * 2 + dx).
// Bit-identical to EPI_MASK followed by maxpool2_bwd_idx_kernel.
__device__ __forceinline__ void epi_route4(const bf16x8& gv, unsigned packed, bf16x8 (&r)[4]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned bi = (packed >> (2 * j)) & 3u;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k][j] = (unsigned)k == bi ? gv[j] : (bf16_t)0.0f;
    }
}
// every lane gets the value of the lane 8 further on in its group of 16 (wrapping): DPP row_ror:8
__device__ __forceinline__ unsigned epi_ror8(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, false); }
__device__ __forceinline__ long long epi_ror8(long long x) {
    const unsigned lo = epi_ror8((unsigned)x), hi = epi_ror8((unsigned)((unsigned long long)x >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ bf16x8 epi_ror8(const bf16x8& x) {
    const u32x4 in = __builtin_bit_cast(u32x4, x);
    u32x4 out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned t = in[i];
        out[i] = epi_ror8(t);
    }
    return __builtin_bit_cast(bf16x8, out);
}
__device__ __forceinline__ bf16x8 epi_sel(bool c, const bf16x8& a, const bf16x8& b) {
    const u32x4 ua = __builtin_bit_cast(u32x4, a), ub = __builtin_bit_cast(u32x4, b);
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned x = ua[i], y = ub[i];
        o[i] = c ? x : y;
    }
    return __builtin_bit_cast(bf16x8, o);
}

// Epilogue row slabs: one pass when the fp32 tile (BM rows of BN*4+16 bytes) fits in ~72 KiB (the 128-wide tiles),
// else the smallest power-of-two split whose slab fits inside the operand ring.
template <int BM, int BN, int STAGES>
constexpr int igemm_epi_passes() {
    const size_t ring = (size_t)STAGES * (BM + BN) * 128, row = (size_t)BN * 4 + 16;
    if ((size_t)BM * row <= 72 * 1024) return 1;
    int p = 2;
    while ((size_t)(BM / p) * row > ring) p *= 2;
    return p;
}
template <int BM, int BN, int STAGES>
constexpr size_t igemm_lds_bytes() {
    const size_t ring = (size_t)STAGES * (BM + BN) * 128;
    const size_t epi = (size_t)(BM / igemm_epi_passes<BM, BN, STAGES>()) * (BN * 4 + 16);
    return ring > epi ? ring : epi;
}

// BM pixels x BN channels per workgroup, WM x WN waves (wave tile BM/WM x BN/WN), STAGES-deep LDS ring.
//   <128,128,2,2,2>: 4 waves, 64 KiB ring, 2 workgroups per CU (the other workgroup covers the barrier waits)
//   <256,128,4,2,3>: 8 waves, 144 KiB ring, 1 workgroup per CU: 1/3 less L2->LDS traffic per FLOP (the 128x128
//                    tile needs ~31 TB/s from L2 for 2 PFLOP/s, i.e. all of it), loads get two K-steps to land
//                    (counted vmcnt: the newest tile stays in flight across the barrier)
template <int BM, int BN, int WM, int WN, int STAGES, int EPI, bool SUB>
__global__ void __launch_bounds__(64 * WM * WN, 2) igemm_kernel(IgemmArgs g) {
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int FP = BM / WM / 16;   // pixel fragments per wave
    constexpr int FC = BN / WN / 16;   // channel fragments per wave
    constexpr int RA = (BM / 8 + NW - 1) / NW;   // LDS-DMA instructions per wave per tile, pixel operand (8 rows each)
    constexpr bool RAGGED_A = (BM / 8) % NW != 0;   // BM = 224: 28 row groups over 8 waves, the last wave stages none
    constexpr int RB = BN / 8 / NW;    // ... weight operand
    constexpr int TILE_BYTES = (BM + BN) * 128;
    constexpr int EPI_STRIDE = BN * 4 + 16;
    static_assert(RA >= 1 && RB >= 1 && STAGES >= 2 && STAGES <= 3 && BM % 8 == 0 && (BM / WM) % 16 == 0 && (!RAGGED_A || STAGES == 2),
                  "unsupported igemm configuration");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int logical = xcd_remap(blockIdx.x, SUB ? g.cls_end[g.sub_s * g.sub_s - 1] : g.mtiles * g.ntiles);
    // SUB: which parity class this block belongs to, and that class's pixel grid
    int ph = 0, pw = 0, OHc = g.OH, OWc = g.OW, Mc = g.M;
    if (SUB) {
        int cls = 0;
        while (logical >= g.cls_end[cls]) ++cls;
        if (cls > 0) logical -= g.cls_end[cls - 1];
        ph = g.cls_ph[cls];
        pw = g.cls_pw[cls];
        OHc = (g.OH - ph + g.sub_s - 1) / g.sub_s;
        OWc = (g.OW - pw + g.sub_s - 1) / g.sub_s;
        Mc = g.N * OHc * OWc;
    }
    const int nt = logical % g.ntiles, mt = logical / g.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- per-thread staging coordinates -------------------------------------------------------
    // Operands go global -> LDS directly (buffer_load ... lds): one wave-instruction writes 64 x 16 B = 8 rows
    // of 128 B, lane l at base + 16*l.  The XOR swizzle therefore sits on the SOURCE side: lane l (row l>>3 of
    // the group, LDS slot l&7) fetches global chunk (l&7) ^ (l>>3), which is where swz() expects it.
    const int chunk = (lane & 7) ^ (lane >> 3);      // global 16-byte chunk this lane fetches
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(g.x, g.xbytes), rw = make_rsrc(g.w, g.wbytes);
    int bh[RA], bw[RA];
    unsigned rowbyte[RA];   // byte offset of (n, bh, bw, chunk) — wraps mod 2^32 for negative bh/bw, fixed up by + tap offset
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + (wave * RA + i) * 8 + (lane >> 3);
        if (m < Mc) {
            int n, oh, ow;
            if (EPI == EPI_POOL) {
                // pool-fused forward: rows are enumerated window-major, m = 4 * (pooled pixel) + (dy * 2 + dx), so
                // the four pixels of every 2x2 pooling window are four consecutive rows of one tile
                const int q = m >> 2, sub = m & 3, PW = OWc >> 1, PHW = (OHc >> 1) * PW;
                n = q / PHW;
                const int r = q - n * PHW;
                const int py = r / PW, px = r - py * PW;
                oh = 2 * py + (sub >> 1);
                ow = 2 * px + (sub & 1);
            } else {
                n = m / (OHc * OWc);
                const int rem = m - n * (OHc * OWc);
                oh = rem / OWc;
                ow = rem - oh * OWc;
            }
            bh[i] = SUB ? oh : oh * g.s_mul - g.pad;   // SUB: class-grid coordinates; the tap supplies the shift
            bw[i] = SUB ? ow : ow * g.s_mul - g.pad;
            rowbyte[i] = (unsigned)(((long long)(n * g.H + bh[i]) * g.W + bw[i]) * g.PS * 2) + chunk * 16;
        } else {
            bh[i] = -(1 << 28);
            bw[i] = -(1 << 28);
            rowbyte[i] = 0;
        }
    }
    const unsigned ktot = (unsigned)(g.KH * g.KW * g.C);
    const unsigned wrow = (unsigned)(n0 + wave * RB * 8 + (lane >> 3)) * ktot * 2 + chunk * 16;

    const int cchunks = g.C / BK;
    // SUB: taps kh0, kh0 + s, ... (the ones with (ih - pad + kh) divisible by s for this class); source row =
    // class row + (ph - pad + kh) / s, an exact division
    const int kh0 = SUB ? ((g.pad - ph) % g.sub_s + g.sub_s) % g.sub_s : 0;
    const int kw0 = SUB ? ((g.pad - pw) % g.sub_s + g.sub_s) % g.sub_s : 0;
    const int kstep = SUB ? g.sub_s : 1;
    const int nkh = kh0 < g.KH ? (g.KH - kh0 + kstep - 1) / kstep : 0;
    const int nkw = kw0 < g.KW ? (g.KW - kw0 + kstep - 1) / kstep : 0;
    const int nit = nkh * nkw * cchunks;

    int kh = kh0, kw = kw0, cc = 0;   // position of the NEXT tile to stage
    typedef __attribute__((address_space(3))) void* lds_ptr;

    // Source offsets of this thread's rows for the current tap; recomputed (bounds test included) only when the tap
    // changes, i.e. once per C/64 K-steps: the steps in between just add 128 bytes.  Rows outside the image keep a
    // sentinel that stays beyond the descriptor's range for any channel offset (tensors are < 0xFFF00000 bytes).
    unsigned xoff[RA];
    constexpr unsigned kOutside = 0xFFF00000u;
    auto stage_tile = [&](int it, int buf) {
        unsigned char* base = smem + buf * TILE_BYTES;
        if (cc == 0) {   // block-uniform
            const int dh = SUB ? (ph - g.pad + kh) / g.sub_s : kh, dw = SUB ? (pw - g.pad + kw) / g.sub_s : kw;
            const unsigned tapoff = (unsigned)(((dh * g.W + dw) * g.PS) * 2);
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int sh = bh[i] + dh, sw = bw[i] + dw;
                const bool ok = (unsigned)sh < (unsigned)g.H && (unsigned)sw < (unsigned)g.W;
                xoff[i] = ok ? rowbyte[i] + tapoff : kOutside;
            }
        }
        const unsigned cco = (unsigned)(cc * (BK * 2));
        const unsigned wk = SUB ? (unsigned)(((kh * g.KW + kw) * g.C + cc * BK) * 2) : (unsigned)it * (BK * 2);
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            if (RAGGED_A && (wave_u * RA + i) * 8 >= BM) continue;   // wave-uniform: no row group here
            const unsigned off = xoff[i] + cco;   // (a typed local: expressions passed straight to the builtin drop the kernel)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(base + (wave_u * RA + i) * 1024), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const unsigned off = wrow + (unsigned)(8 * i) * ktot * 2 + wk;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(base + BM * 128 + (wave_u * RB + i) * 1024), 16, off, 0, 0, 0);
        }
        if (++cc == cchunks) {
            cc = 0;
            kw += kstep;
            if (kw >= g.KW) { kw = kw0; kh += kstep; }
        }
    };

    f32x4 acc[FC][FP];
#pragma unroll
    for (int a = 0; a < FC; ++a)
#pragma unroll
        for (int b = 0; b < FP; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st)
        if (st < nit) stage_tile(st, st);

    const int frow = lane & 15, fk = lane >> 4;
    int cur = 0, nxt = STAGES - 1;   // ring slots of tile `it` and of tile `it + STAGES - 1`
    for (int it = 0; it < nit; ++it) {
        // Tile `it` must have landed.  LDS-DMA completion is tracked by vmcnt only (in issue order), so with a
        // 3-deep ring the RA+RB loads of tile it+1 may stay in flight; __syncthreads() alone waits for nothing.
        if (STAGES == 3 && it + 1 < nit) wait_vmcnt<RA + RB>();
        else wait_vmcnt<0>();
        // after this barrier every wave has finished reading the slot that is restaged next (it was tile it-1's)
        __syncthreads();
        if (it + STAGES - 1 < nit) stage_tile(it + STAGES - 1, nxt);
        const unsigned char* pb = smem + cur * TILE_BYTES;
        const unsigned char* wb = pb + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 wf[FC], pf[FP];
#pragma unroll
            for (int a = 0; a < FC; ++a)
                wf[a] = *reinterpret_cast<const bf16x8*>(wb + swz(wn * (BN / WN) + a * 16 + frow, ks * 4 + fk));
#pragma unroll
            for (int b = 0; b < FP; ++b)
                pf[b] = *reinterpret_cast<const bf16x8*>(pb + swz(wm * (BM / WM) + b * 16 + frow, ks * 4 + fk));
#pragma unroll
            for (int a = 0; a < FC; ++a)
#pragma unroll
                for (int b = 0; b < FP; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], pf[b], acc[a][b], 0, 0, 0);
        }
        cur = cur + 1 == STAGES ? 0 : cur + 1;
        nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
    }
    // ---- epilogue ---------------------------------------------------------------------------------------------------
    // acc[a][b][r] = out[pixel = wm*BM/WM + b*16 + (lane&15)][chan = wn*BN/WN + a*16 + (lane>>4)*4 + r]
    if constexpr (EPI != EPI_POOL) {
        // straight from the accumulators (see epi_swap8): channel fragments a, a + 1 pair up, lanes with odd fk take a + 1
        static_assert(FC % 2 == 0, "the direct epilogue pairs channel fragments");
        const int par = fk & 1, half = fk >> 1;
        auto run = [&](auto guard_tag) {
            // GUARD = the tile hangs over the end of the GEMM's rows: every row is tested.  Whole tiles run in one basic
            // block; the masked form then sends all its activation loads first (one in flight per fragment pair, what the
            // compiler schedules by itself, left the epilogue waiting on each)
            constexpr bool GUARD = decltype(guard_tag)::value;
            constexpr bool MASKED = EPI == EPI_MASK || EPI == EPI_UNPOOL;
            constexpr bool PRELOAD = MASKED && !GUARD && FP * FC / 2 <= 16;
            long long orow[FP];   // this lane's row of fragment b: element offset of its first channel in out, < 0 = no such row
            long long o00[EPI == EPI_UNPOOL ? FP : 1];   // EPI_UNPOOL: ... of the row's 2x2 window in the full-resolution out
#pragma unroll
            for (int b = 0; b < FP; ++b) {
                int m = m0 + wm * (BM / WM) + b * 16 + frow;
                const bool ok = !GUARD || m < Mc;
                if (EPI == EPI_UNPOOL) {
                    const int mm = ok ? m : 0;
                    const int n = mm / (g.OH * g.OW);
                    const int rem = mm - n * (g.OH * g.OW);
                    const int oh = rem / g.OW, ow = rem - oh * g.OW;
                    o00[b] = ((long long)(n * 2 * g.OH + 2 * oh) * (2 * g.OW) + 2 * ow) * g.NOUT;
                }
                if (SUB && ok) {   // class-grid pixel -> pixel of the full output
                    const int n = m / (OHc * OWc);
                    const int rem = m - n * (OHc * OWc);
                    const int oh = rem / OWc, ow = rem - oh * OWc;
                    m = (n * g.OH + oh * g.sub_s + ph) * g.OW + ow * g.sub_s + pw;
                }
                orow[b] = ok ? (long long)m * g.NOUT : -1;
            }
            bf16x8 avs[FC / 2][FP];
            unsigned short pks[FC / 2][FP];
            if (PRELOAD) {
#pragma unroll
                for (int a2 = 0; a2 < FC / 2; ++a2)
#pragma unroll
                    for (int b = 0; b < FP; ++b) {
                        const long long o = orow[b] + n0 + wn * (BN / WN) + a2 * 32 + par * 16 + half * 8;
                        avs[a2][b] = *reinterpret_cast<const bf16x8*>(g.act + o);
                        if (EPI == EPI_UNPOOL) pks[a2][b] = g.pool_idx[o >> 3];
                    }
            }
            const long long down = (long long)(2 * g.OW) * g.NOUT;   // EPI_UNPOOL: one full-resolution row further down
            if constexpr (FC % 4 == 0 && EPI == EPI_UNPOOL) {
                // 128 B runs: two fragment pairs (64 channels) per step.  After the swap a store instruction would cover 16
                // rows x 64 B; lanes of rows 0-7 trade their upper 32 channels for the lower 32 of rows 8-15 (DPP
                // row_ror:8), so that one instruction writes rows 0-7 and the next rows 8-15, 128 B = whole L2 lines each.
                // Only this store-bound form (four output pixels per GEMM row) gains from it: feat.4 169 -> 128 us, feat.9
                // 98 -> 79; the ordinary forward / dgrad epilogues of the 1x1 layers measured 0-1 % slower with it (same
                // box, L4 / L6 / L9 / L17 / L20: 627-635 -> 636-637 us) and keep the 64 B runs.
                const bool lo8 = frow < 8;
                long long prow[FP], p00[EPI == EPI_UNPOOL ? FP : 1];   // the partner lane's rows
#pragma unroll
                for (int b = 0; b < FP; ++b) {
                    prow[b] = epi_ror8(orow[b]);
                    if (EPI == EPI_UNPOOL) p00[b] = epi_ror8(o00[b]);
                }
#pragma unroll
                for (int a4 = 0; a4 < FC / 4; ++a4) {
                    const int cA = n0 + wn * (BN / WN) + a4 * 64 + par * 16 + half * 8, cB = cA + 32;
                    float bvA[8], svA[8], bvB[8], svB[8];
                    if (EPI == EPI_BIAS_LEAKY) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            bvA[j] = g.bias[cA + j];
                            svA[j] = g.scale ? g.scale[cA + j] : 1.0f;
                            bvB[j] = g.bias[cB + j];
                            svB[j] = g.scale ? g.scale[cB + j] : 1.0f;
                        }
                    }
#pragma unroll
                    for (int b = 0; b < FP; ++b) {
                        float vA[8], vB[8];
                        epi_swap8(acc[4 * a4][b], acc[4 * a4 + 1][b], vA);
                        epi_swap8(acc[4 * a4 + 2][b], acc[4 * a4 + 3][b], vB);
                        const bool own = !GUARD || orow[b] >= 0, partner = !GUARD || prow[b] >= 0;
                        bf16x8 avA = {}, avB = {};
                        unsigned pkA = 0, pkB = 0;
                        if (MASKED) {
                            if (PRELOAD) {
                                avA = avs[2 * a4][b];
                                avB = avs[2 * a4 + 1][b];
                                if (EPI == EPI_UNPOOL) { pkA = pks[2 * a4][b]; pkB = pks[2 * a4 + 1][b]; }
                            } else if (own) {
                                avA = *reinterpret_cast<const bf16x8*>(g.act + orow[b] + cA);
                                avB = *reinterpret_cast<const bf16x8*>(g.act + orow[b] + cB);
                                if (EPI == EPI_UNPOOL) { pkA = g.pool_idx[(orow[b] + cA) >> 3]; pkB = g.pool_idx[(orow[b] + cB) >> 3]; }
                            }
                        }
                        const bf16x8 ovA = epi_value8<EPI>(g, vA, svA, bvA, avA), ovB = epi_value8<EPI>(g, vB, svB, bvB, avB);
                        // first store: rows 0-7 (their own lanes at cA, the partners at cB); second store: rows 8-15
                        const bool st1 = lo8 ? own : partner, st2 = lo8 ? partner : own;
                        if constexpr (EPI == EPI_UNPOOL) {
                            bf16x8 rA[4], rB[4];
                            epi_route4(ovA, pkA, rA);
                            epi_route4(ovB, pkB, rB);
                            const long long o1 = lo8 ? o00[b] + cA : p00[b] + cB, o2 = lo8 ? p00[b] + cA : o00[b] + cB;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const bf16x8 recv = epi_ror8(epi_sel(lo8, rB[k], rA[k]));
                                const long long ok = (k & 1) * g.NOUT + (k >> 1) * down;
                                if (st1) *reinterpret_cast<bf16x8*>(g.out + o1 + ok) = epi_sel(lo8, rA[k], recv);
                                if (st2) *reinterpret_cast<bf16x8*>(g.out + o2 + ok) = epi_sel(lo8, recv, rB[k]);
                            }
                        } else {
                            const bf16x8 recv = epi_ror8(epi_sel(lo8, ovB, ovA));
                            const long long o1 = lo8 ? orow[b] + cA : prow[b] + cB, o2 = lo8 ? prow[b] + cA : orow[b] + cB;
                            if (st1) *reinterpret_cast<bf16x8*>(g.out + o1) = epi_sel(lo8, ovA, recv);
                            if (st2) *reinterpret_cast<bf16x8*>(g.out + o2) = epi_sel(lo8, recv, ovB);
                        }
                    }
                }
            } else {
#pragma unroll
            for (int a2 = 0; a2 < FC / 2; ++a2) {
                const int c = n0 + wn * (BN / WN) + a2 * 32 + par * 16 + half * 8;
                float bv[8], sv[8];
                if (EPI == EPI_BIAS_LEAKY) {
                    // per-channel scale of a folded BatchNorm: 1 when absent — fma(v, 1, b) == v + b bit for bit, so the
                    // reference's bias-only block needs no second code path
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        bv[j] = g.bias[c + j];
                        sv[j] = g.scale ? g.scale[c + j] : 1.0f;
                    }
                }
#pragma unroll
                for (int b = 0; b < FP; ++b) {
                    float v[8];
                    epi_swap8(acc[2 * a2][b], acc[2 * a2 + 1][b], v);
                    if (GUARD && orow[b] < 0) continue;
                    const long long o = orow[b] + c;
                    bf16x8 av = {};
                    if (MASKED) av = PRELOAD ? avs[a2][b] : *reinterpret_cast<const bf16x8*>(g.act + o);
                    const bf16x8 ov = epi_value8<EPI>(g, v, sv, bv, av);
                    if constexpr (EPI == EPI_UNPOOL) {
                        bf16x8 r[4];
                        epi_route4(ov, PRELOAD ? pks[a2][b] : g.pool_idx[o >> 3], r);
#pragma unroll
                        for (int k = 0; k < 4; ++k) *reinterpret_cast<bf16x8*>(g.out + o00[b] + c + (k & 1) * g.NOUT + (k >> 1) * down) = r[k];
                    } else {
                        *reinterpret_cast<bf16x8*>(g.out + o) = ov;
                    }
                }
            }
            }
        };
        if (m0 + BM <= Mc) run(std::false_type()); else run(std::true_type());
        return;
    }
    __syncthreads();   // all waves done with the ring before the epilogue reuses it

    // ---- pool-fused epilogue: fp32 tile through LDS (a 2x2 window spans four pixel rows of the tile) -------------------
    // Large tiles go through the LDS in EPASS row slabs of PR rows so that the staging fits in the ring.
    constexpr int EPASS = igemm_epi_passes<BM, BN, STAGES>();
    constexpr int PR = BM / EPASS;
    constexpr int TPR = BN / 8;            // items per window row (8 channels each)
    static_assert(PR % 16 == 0, "epilogue slab must hold whole fragments");
#pragma unroll
    for (int pass = 0; pass < EPASS; ++pass) {
#pragma unroll
        for (int a = 0; a < FC; ++a)
#pragma unroll
            for (int b = 0; b < FP; ++b) {
                const int prow = wm * (BM / WM) + b * 16;
                if (EPASS > 1 && prow / PR != pass) continue;
                const int ch = wn * (BN / WN) + a * 16 + fk * 4;
                *reinterpret_cast<f32x4*>(smem + (prow - pass * PR + frow) * EPI_STRIDE + ch * 4) = acc[a][b];
            }
        __syncthreads();
        // bias + LeakyReLU + one rounding to bf16 for the four pixels of a window, then the 2x2 max over the
        // ROUNDED values with "first maximum wins" — bit-identical to conv forward followed by the pool kernel;
        // only the pooled tensor and the 2-bit positions leave the CU (the full-resolution activation has no
        // reader: pool backward works from the pooled output and the positions).
        for (int item = tid; item < (PR / 4) * TPR; item += NT) {
            const int w = item / TPR, c8 = item - w * TPR;
            const int q = ((m0 + pass * PR) >> 2) + w;
            if (q * 4 >= Mc) continue;
            // per-channel scale of a folded BatchNorm: 1 when absent — fma(v, 1, b) == v + b bit for bit, so the
            // reference's bias-only block needs no second code path
            float bvp[8], svp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bvp[j] = g.bias[n0 + c8 * 8 + j];
                svp[j] = g.scale ? g.scale[n0 + c8 * 8 + j] : 1.0f;
            }
            bf16x8 best;
            unsigned packed = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned char* rp = smem + (4 * w + k) * EPI_STRIDE + c8 * 32;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(rp);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(rp + 16);
                const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t = __builtin_fmaf(v[j], svp[j], bvp[j]);
                    const bf16_t yv = (bf16_t)(t > 0.0f ? t : t * g.slope);
                    if (k == 0 || (float)yv > (float)best[j]) {
                        best[j] = yv;
                        packed = (packed & ~(3u << (2 * j))) | ((unsigned)k << (2 * j));
                    }
                }
            }
            *reinterpret_cast<bf16x8*>(g.out + (long long)q * g.NOUT + n0 + c8 * 8) = best;
            g.pool_idx[(long long)q * (g.NOUT / 8) + (n0 >> 3) + c8] = (unsigned short)packed;
        }
        if (pass + 1 < EPASS) __syncthreads();
    }
}

template <int BM, int BN, int WM, int WN, int STAGES, int EPI, bool SUB>
int launch_igemm(IgemmArgs& g, hipStream_t s, const char* name) {
    g.mtiles = (int)yfs_cdiv(g.M, BM);
    g.ntiles = g.NOUT / BN;
    long long nblk = (long long)g.mtiles * g.ntiles;
    if (SUB) {
        // parity classes ordered by tap count, heaviest first, so the long blocks are not left for the tail
        const int sb = g.sub_s;
        YFS_REQUIRE(sb == 2, YFS_ERR_UNSUPPORTED, "%s: strided dgrad supports stride 2 only (got %d)", name, sb);
        int order[4] = {0, 1, 2, 3}, taps[4];
        for (int c = 0; c < 4; ++c) {
            const int ph = c / sb, pw = c % sb;
            const int kh0 = ((g.pad - ph) % sb + sb) % sb, kw0 = ((g.pad - pw) % sb + sb) % sb;
            taps[c] = (kh0 < g.KH ? (g.KH - kh0 + sb - 1) / sb : 0) * (kw0 < g.KW ? (g.KW - kw0 + sb - 1) / sb : 0);
        }
        std::stable_sort(order, order + 4, [&](int a, int b) { return taps[a] > taps[b]; });
        nblk = 0;
        for (int k = 0; k < 4; ++k) {
            const int ph = order[k] / sb, pw = order[k] % sb;
            const long long Mc = (long long)g.N * ((g.OH - ph + sb - 1) / sb) * ((g.OW - pw + sb - 1) / sb);
            nblk += yfs_cdiv(Mc, BM) * g.ntiles;   // an empty class (OH or OW == 1) contributes no blocks
            g.cls_ph[k] = ph; g.cls_pw[k] = pw; g.cls_end[k] = (int)nblk;
        }
    }
    const size_t lds = igemm_lds_bytes<BM, BN, STAGES>();
    auto kern = igemm_kernel<BM, BN, WM, WN, STAGES, EPI, SUB>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        YFS_REQUIRE(e == hipSuccess, YFS_ERR_LAUNCH, "%s: cannot reserve %zu bytes of LDS", name, lds);
    }
    kern<<<dim3((unsigned)nblk), dim3(64 * WM * WN), lds, s>>>(g);
    YFS_CHECK_LAUNCH(name);
    yfs_note_kernel("igemm_kernel<%d,%d,epi%d%s>", BM, BN, EPI, SUB ? ",sub" : "");
    return YFS_OK;
}

// ---- 256 x 256 x 64 tile, 8 waves, staggered wave groups ("8-phase" schedule) -----------------------------------
// The 2-stage loops above pay one vmcnt(0) + barrier hand-off per K-step with every wave in the same state: all
// loading, then all multiplying.  Here the tile is cut into four 16 KiB half-tiles per K-step (pixel rows 0-127 /
// 128-255 = P0 / P1, weight rows W0 / W1; two LDS buffers of four half-tiles = 128 KiB), a K-step into four phases,
// two 64 x 32 output quadrants of every wave each (wave = 2 x 2 quadrants: 128 pixels x 64 channels, 128 accumulator
// registers), and the two wave rows run one barrier apart: while waves 0-3 multiply (32 MFMAs between two barriers),
// waves 4-7 — their partners on the same four SIMDs — read fragments and issue the next LDS-DMA, and vice versa, so
// the matrix pipe of every SIMD always has one wave feeding it.
//   phase A: read W0, P0, W1 fragments | stage P1 of tile t+1           | MFMA quadrants (P0, W0), (P0, W1)
//   phase B: read P1                   | stage W0, P0, W1 of tile t+2   | MFMA (P1, W1), (P1, W0)   + the only vmcnt wait
// (Round 2 started with four phases of 16 MFMAs, one half-tile staged per phase; merging them pairwise halves the
// barriers per K-step and measured 1.6-3 % faster on the same box for every layer that runs this kernel.)
// A half-tile is restaged one phase after its fragments were read, which is why every wave retires its reads with
// lgkmcnt(0) BEFORE the phase's first barrier, i.e. the loads run two K-steps ahead with two buffers.  vmcnt(6) in
// phase B leaves the three half-tiles just issued in flight (2 LDS-DMA instructions per wave and half-tile) and
// retires everything older = all of tile t+1, which is first read two barriers later.  Barriers are raw s_barrier (a
// __syncthreads() would drain the DMA queue).
constexpr int B8_HALF = 128 * 128;     // one half-tile: 128 rows x 128 B
constexpr int B8_BUF = 4 * B8_HALF;    // [P0 | P1 | W0 | W1]
constexpr int B8_LDS = 2 * B8_BUF;
constexpr int B8_LDS_IG = B8_LDS + 512 * 32;   // igemm8_kernel: + 8 words of staging state per thread
constexpr int B8_EPI_STRIDE = 256 * 4 + 16;   // epilogue: four slabs of <= 64 rows of the fp32 tile through the ring

// FP1 = pixel fragments per wave in the SECOND pixel half-tile: 4 -> 256-row tile, 3 -> 224-row tile (P1 holds 96 rows,
// phases 3 and 4 run 12 MFMAs).  Every GEMM M of this network is 64 * (7 * 2^k)^2, so 224-row tiles have no ragged edge
// and their counts (224, 448, 896, ...) fill 256 CUs to 87.5-100 % where 196 / 392 / 784 tiles of 256 rows fill 76.6 %.
#ifdef YFS_IG8_TIMING
__device__ long long yfs_ig8_dbg[8];
// timeline of every workgroup of the launch (first 8192): [block][0..4] = the section clocks, [block][5] = HW_ID | XCC_ID << 32,
// so that the host can line the workgroups of one CU up and see what lies between one's end and the next one's start
__device__ long long yfs_ig8_tl[8192 * 6];
#define IG8_T(k, slot) do { if (threadIdx.x == 0) { const long long c_ = clock64(); if ((slot) == 17) yfs_ig8_dbg[k] = c_; \
    if ((slot) < 8192) { yfs_ig8_tl[(slot) * 6 + (k)] = c_; if ((k) == 0) yfs_ig8_tl[(slot) * 6 + 5] = \
        (long long)(unsigned)__builtin_amdgcn_s_getreg(63492) | ((long long)(unsigned)__builtin_amdgcn_s_getreg(63508) << 32); } } } while (0)
#else
#define IG8_T(k, slot) do { } while (0)
#endif
// FW1 = channel fragments per wave in the SECOND weight half-tile: 2 -> 256-channel tile, 1 -> 192-channel tile (W1 holds
// 64 rows, its MFMA blocks are half as long): feat.2's dgrad writes 192 channels.
// Persistent form: the plain forward launches (bias + LeakyReLU) only.
//  - not the dgrad forms: they share the chip with the weight-gradient stream and, in a data-parallel step, with RCCL's
//    kernels, and a static assignment of tiles to workgroups cannot give way: with 16-64 workgroups of another queue
//    holding CUs for milliseconds, feat.7's 1792 tiles took 643-667 us persistent against 454-543 us as one workgroup per
//    tile, which the hardware hands to whichever CU is free (tools/probes/persist_under_contention.py); inside the
//    two-stream training step persistent dgrads measured +-0.02 ms of 10.98.  The forward pass has the chip to itself;
//  - not the pool-fused epilogue (it parks the tile in the LDS the next tile's loads would land in);
//  - not the 256 x 256 tile (128 accumulators + the epilogue's preloaded operands + the next tile's staging state do not fit
//    in 256 registers: the compiler spills, and a scratch access next to the counted vmcnt waits would miscount them);
//  - not the split-K form (its grids are at most one workgroup per CU by construction, and with the K-range state carried
//    around a tile loop the compiler keeps it in vector registers: feat.26 forward 71 -> 77 us).
// YFS_KNOB_IGEMM_PERSIST = 1 turns it off for the forward launches too.
template <int EPI, int FP1, int FW1>
constexpr bool igemm8_persistent() { return EPI == EPI_BIAS_LEAKY && !(FP1 == 4 && FW1 == 2); }

template <int EPI, int FP1, int FW1 = 2>
__global__ void __launch_bounds__(512, 2) igemm8_kernel(IgemmArgs g) {
    constexpr int BM = 128 + FP1 * 32, BN = 128 + FW1 * 64;
    constexpr bool PERSIST = igemm8_persistent<EPI, FP1, FW1>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int Mc = g.M;
    // Persistent (igemm8_persistent): the grid is one workgroup per CU and workgroup w takes tiles w,
    // w + G, w + 2G, ... (virtual block ids: the XCD-aware order is that of a grid of one workgroup per tile).  The epilogue
    // touches no LDS, so the next tile's coordinates are computed and its first two K-steps are on their way into the ring
    // BEFORE the finished tile's epilogue runs: neither the workgroup turn-over (~1.7 K cycles) nor the cold first loads
    // (4-7 K cycles from entry to the first MFMA, tools/probes/igemm8_turnover.py) sit between two main loops.
    const int tiles_mn = g.mtiles * g.ntiles, total = tiles_mn * (EPI == EPI_PARTIAL ? g.ksplit : 1);

    // ---- staging coordinates: 4 pixel rows and 4 weight rows per thread (2 per half-tile) -----------------------------
    const int chunk = (lane & 7) ^ (lane >> 3);   // source-side XOR swizzle (see igemm_kernel)
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(g.x, g.xbytes), rw = make_rsrc(g.w, g.wbytes);
    // per-row staging state lives in LDS behind the ring (8 words per thread; read back once per tap): in registers it is
    // what pushed the two-phase K-step over 256 VGPRs.  st[j] = byte offset of row j's tap-(0,0) source pixel,
    // st[4 + j] = (bh << 16) | (bw & 0xFFFF), that pixel's row / column (both within +-16 K: dispatch_igemm checks)
    unsigned* const st = reinterpret_cast<unsigned*>(smem + B8_LDS) + tid * 8;
    const unsigned ktot = (unsigned)(g.KH * g.KW * g.C);
    const int cchunks = g.C / BK;
    const int ktotal = g.KH * g.KW * cchunks;
    // the tile the staging state describes: rows m0.., channels n0.., K-steps [ks0, kend) of the layer's ktotal (all of them
    // unless EPI_PARTIAL); ts = (kh, kw, cc) = the K-step staged next; xoff = this thread's source offsets for that tap
    // (recomputed when the tap changes)
    int m0 = 0, n0 = 0, split = 0, kend = 0, nit = 0;
    unsigned wrow = 0;
    int kh = 0, kw = 0, cc = 0, ts = 0;
    unsigned xoff[4];
    constexpr unsigned kOutside = 0xFFF00000u;
    auto tap_offsets = [&]() {
        const unsigned tapoff = (unsigned)(((kh * g.W + kw) * g.PS) * 2);
        const u32x4 rb = *reinterpret_cast<const u32x4*>(st), hw = *reinterpret_cast<const u32x4*>(st + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sh = ((int)hw[j] >> 16) + kh, sw = (int)(short)(hw[j] & 0xFFFFu) + kw;
            const bool ok = (unsigned)sh < (unsigned)g.H && (unsigned)sw < (unsigned)g.W;
            xoff[j] = ok ? rb[j] + tapoff : kOutside;
        }
    };
    auto advance = [&]() {
        ++ts;
        if (++cc == cchunks) {
            cc = 0;
            if (++kw >= g.KW) { kw = 0; ++kh; }
            tap_offsets();
        }
    };
    auto stage_P = [&](int h, int buf) {
        // wave-uniform part goes in the scalar offset.  (readfirstlane: in some instantiations the compiler keeps the K-step
        // state in vector registers once it is carried around the tile loop, and would wrap every LDS-DMA in a waterfall
        // loop over this operand: 2200 -> 3080 cycles per K-step)
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(cc * (BK * 2));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned off = xoff[h * 2 + i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(smem + buf * B8_BUF + h * B8_HALF + (wave * 2 + i) * 1024), 16,
                                                     off, so, 0, 0);
        }
    };
    auto stage_W = [&](int h, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)(h * 128 + i * 8) * ktot + (unsigned)ts * BK) * 2));
            const unsigned off = wrow;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + buf * B8_BUF + (2 + h) * B8_HALF + (wave * 2 + i) * 1024),
                                                     16, off, so, 0, 0);
        }
    };

    f32x4 acc[2][2][2][4];   // [pixel half][weight half][channel fragment][pixel fragment]
#pragma unroll
    for (int i = 0; i < 32; ++i) (&acc[0][0][0][0])[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = lane >> 4;
    bf16x8 pf[4][2], w0f[2][2], w1f[2][2];
    auto read_P = [&](const unsigned char* hb, auto nf_tag) {   // nf_tag: fragments per wave in this half (4 or FP1)
        constexpr int NF = decltype(nf_tag)::value;
#pragma unroll
        for (int b = 0; b < NF; ++b)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                pf[b][ks] = *reinterpret_cast<const bf16x8*>(hb + swz(wm * (NF * 16) + b * 16 + frow, ks * 4 + fk));
    };
    auto read_W = [&](const unsigned char* hb, bf16x8 (&wf)[2][2], auto na_tag) {   // na_tag: channel fragments per wave (2 or FW1)
        constexpr int NA = decltype(na_tag)::value;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wf[a][ks] = *reinterpret_cast<const bf16x8*>(hb + swz(wn * (NA * 16) + a * 16 + frow, ks * 4 + fk));
    };
    auto mfma16 = [&](f32x4 (&c)[2][4], const bf16x8 (&wf)[2][2], auto nf_tag, auto na_tag) {
        constexpr int NF = decltype(nf_tag)::value, NA = decltype(na_tag)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NF; ++b) c[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a][ks], pf[b][ks], c[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto lgkm0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // Staging state of virtual block vb, and its first K-step (tile 0) plus W0 / P0 / W1 of its second on their way: the
    // steady state at the top of a K-step once tile 0 has landed.  The weight half-tiles of tile 0 need no pixel
    // coordinates: they are on their way while those are computed
    auto begin_tile = [&](int vb) {
        IG8_T(0, vb);
        auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };   // (the divisions run on the vector ALU)
        int logical = uni(xcd_remap(vb, total));
        split = 0;
        if (EPI == EPI_PARTIAL) {   // K-split outermost: the blocks of one split share its slice of the weights in L2
            split = uni(logical / tiles_mn);
            logical -= split * tiles_mn;
        }
        const int nt = uni(logical % g.ntiles), mt = uni(logical / g.ntiles);
        m0 = mt * BM;
        n0 = nt * BN;
        wrow = (unsigned)(n0 + wave * 16 + (lane >> 3)) * ktot * 2 + chunk * 16;
        const int ks0 = EPI == EPI_PARTIAL ? split * g.ksteps_per_split : 0;
        kend = EPI == EPI_PARTIAL ? (ks0 + g.ksteps_per_split < ktotal ? ks0 + g.ksteps_per_split : ktotal) : ktotal;
        nit = kend - ks0;
        kh = kw = cc = 0;
        ts = ks0;
        if (EPI == EPI_PARTIAL) {
            const int tap0 = uni(ks0 / cchunks);
            cc = uni(ks0 - tap0 * cchunks);
            kh = uni(tap0 / g.KW);
            kw = uni(tap0 - kh * g.KW);
        }
        stage_W(0, 0); stage_W(1, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // j = half * 2 + i: tile row half*128 + wave*16 + i*8 + (lane >> 3)
            const int hr = wave * 16 + (j & 1) * 8 + (lane >> 3);   // row inside the half-tile
            const int m = m0 + (j >> 1) * 128 + hr;
            if (m < Mc && (j < 2 || hr < FP1 * 32)) {
                int n, oh, ow;
                // (host-prepared multipliers: four rows x two divisions per thread were ~280 instructions of every tile's prologue)
                if (EPI == EPI_POOL) {   // window-major row order (see igemm_kernel); div_hw / div_w divide by the pooled grid
                    const int q = m >> 2, sub = m & 3;
                    n = (int)fdiv((unsigned)q, g.div_hw);
                    const int r = q - n * (int)g.div_hw.d;
                    const int py = (int)fdiv((unsigned)r, g.div_w), px = r - py * (int)g.div_w.d;
                    oh = 2 * py + (sub >> 1);
                    ow = 2 * px + (sub & 1);
                } else {
                    n = (int)fdiv((unsigned)m, g.div_hw);
                    const int rem = m - n * (int)g.div_hw.d;
                    oh = (int)fdiv((unsigned)rem, g.div_w);
                    ow = rem - oh * (int)g.div_w.d;
                }
                const int bh = oh * g.s_mul - g.pad, bw = ow * g.s_mul - g.pad;
                st[4 + j] = ((unsigned)bh << 16) | ((unsigned)bw & 0xFFFFu);
                st[j] = (unsigned)(((long long)(n * g.H + bh) * g.W + bw) * g.PS * 2) + chunk * 16;
            } else {
                st[4 + j] = 0xC000C000u;   // (-16384, -16384): outside for every tap
                st[j] = 0;
            }
        }
        tap_offsets();
        stage_P(0, 0); stage_P(1, 0);
        if (nit > 1) {
            advance();
            stage_W(0, 1); stage_P(0, 1); stage_W(1, 1);
        }
    };
    int vb = (int)blockIdx.x;
    begin_tile(vb);
    if (nit > 1) wait_vmcnt<6>(); else wait_vmcnt<0>();

    const std::integral_constant<int, 4> F4;
    const std::integral_constant<int, FP1> F1;
    const std::integral_constant<int, 2> A2;
    const std::integral_constant<int, FW1> A1;
    auto tile = [&](int t, auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        const unsigned char* tb = smem + BUF * B8_BUF;
        // phase A
        read_W(tb + 2 * B8_HALF, w0f, A2);
        read_P(tb, F4);
        read_W(tb + 3 * B8_HALF, w1f, A1);
        if (t + 1 < nit) stage_P(1, BUF ^ 1);
        lgkm0();
        bar();
        mfma16(acc[0][0], w0f, F4, A2);
        mfma16(acc[0][1], w1f, F4, A1);
        bar();
        // phase B
        read_P(tb + B8_HALF, F1);
        advance();
        if (ts < kend) {
            stage_W(0, BUF);
            stage_P(0, BUF);
            stage_W(1, BUF);
            wait_vmcnt<6>();
        } else {
            wait_vmcnt<0>();
        }
        lgkm0();
        bar();
        mfma16(acc[1][1], w1f, F1, A1);
        mfma16(acc[1][0], w0f, F1, A2);
        bar();
    };
    for (;;) {   // one pass per tile of this workgroup
    bar();
    if (wm == 1) bar();   // waves 4-7 run one barrier behind waves 0-3 from here on
    IG8_T(1, vb);
    int t = 0;
    for (; t + 1 < nit; t += 2) {
        tile(t, std::integral_constant<int, 0>());
        tile(t + 1, std::integral_constant<int, 1>());
    }
    if (t < nit) tile(t, std::integral_constant<int, 0>());
    if (wm == 0) bar();   // catch up with waves 4-7: every wave is past its last fragment read after this
    IG8_T(2, vb);
    // the finished tile, and whether another follows (its prologue goes out from inside the epilogue)
    const int cm0 = m0, cn0 = n0, csplit = split;
    [[maybe_unused]] const int cvb = vb;   // (section clocks of the timing build)
    vb += (int)gridDim.x;
    const bool more = PERSIST && vb < total;

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    // acc[ph][wh][a][b][r] = out[pixel ph*128 + wm*NF*16 + b*16 + (lane&15)][chan wh*128 + wn*NA*16 + a*16 + (lane>>4)*4 + r]
    // with NF = 4 pixel fragments per wave in the first pixel half and FP1 in the second, NA = 2 channel fragments per
    // wave in the first weight half and FW1 in the second.
    if constexpr (EPI != EPI_POOL) {
        // straight from the accumulators (see epi_swap8)
        const int par = fk & 1, half = fk >> 1;
        // GUARD = the tile hangs over the end of the GEMM's rows: every row is tested.  Whole tiles (all of them at this
        // network's sizes) run without tests in one basic block, so the stores of one fragment pair overlap the arithmetic
        // of the next.  Everything the epilogue LOADS (bias / scale rows; in the masked form of a whole tile all 14-16
        // activation vectors of the wave: one in flight per fragment pair, what the compiler schedules by itself, left it
        // waiting ~700 cycles on each) is requested first, then the next tile's prologue goes out, then the arithmetic and
        // the stores run: vmcnt retires in order, so a load requested after the prologue's DMA would wait for all of it.
        auto run = [&](auto guard_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value;
        constexpr bool PRELOAD = EPI == EPI_MASK && !GUARD;
        // after the swap: NA == 2 pairs the two channel fragments of one pixel fragment (lanes with odd fk own fragment
        // a = 1); NA == 1 pairs pixel fragments b, b + 1 of the only channel fragment (lanes with odd fk own rows of b + 1)
        int cb[2], cc8[2];   // first channel of the wave's slice of weight half wh; of this lane's 8-channel run in it
        float bvs[2][8], svs[2][8];
        bf16x8 avs[2][2][4];
#pragma unroll
        for (int wh = 0; wh < 2; ++wh) {
            const int NA = wh == 1 ? FW1 : 2;
            cb[wh] = cn0 + wh * 128 + wn * (NA * 16);
            cc8[wh] = cb[wh] + (NA == 2 ? par * 16 : 0) + half * 8;
            if (EPI == EPI_BIAS_LEAKY) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    bvs[wh][j] = g.bias[cc8[wh] + j];
                    svs[wh][j] = g.scale ? g.scale[cc8[wh] + j] : 1.0f;
                }
            }
            if (PRELOAD) {
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    const int NF = ph == 1 ? FP1 : 4;
                    const int r0 = cm0 + ph * 128 + wm * (NF * 16) + frow;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        if (b >= NF || (NA == 1 && ((b & 1) || b + 1 >= NF))) continue;
                        const int m = r0 + (NA == 2 ? b : b + par) * 16;
                        avs[wh][ph][b] = *reinterpret_cast<const bf16x8*>(g.act + (long long)m * g.NOUT + cc8[wh]);
                    }
                }
            }
        }
        if constexpr (PERSIST) {
            if (more) {
                __builtin_amdgcn_sched_barrier(0);
                begin_tile(vb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int wh = 0; wh < 2; ++wh) {
            const int NA = wh == 1 ? FW1 : 2;
            const int cbase = cb[wh], c = cc8[wh];
            auto emit = [&](int m, float (&v)[8], const bf16x8& pre) {   // the lane's 8 channels of output row m
                if (GUARD && m >= Mc) return;
                const long long o = (long long)m * g.NOUT + c;
                bf16x8 av = {};
                if (EPI == EPI_MASK) av = PRELOAD ? pre : *reinterpret_cast<const bf16x8*>(g.act + o);
                epi_emit8<EPI>(g, o, v, svs[wh], bvs[wh], av);
            };
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const int NF = ph == 1 ? FP1 : 4;
                const int r0 = cm0 + ph * 128 + wm * (NF * 16) + frow;
                if (EPI == EPI_PARTIAL) {   // the raw fp32 sums of this block's K range; the epilogue runs in the reduction
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            if (a >= NA || b >= NF) continue;
                            const int m = r0 + b * 16;
                            if (!GUARD || m < Mc)
                                *reinterpret_cast<f32x4*>(g.partial + ((long long)csplit * Mc + m) * g.NOUT + cbase + a * 16 + fk * 4) =
                                    acc[ph][wh][a][b];
                        }
                } else if (NA == 2) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        if (b >= NF) continue;
                        float v[8];
                        epi_swap8(acc[ph][wh][0][b], acc[ph][wh][1][b], v);
                        emit(r0 + b * 16, v, avs[wh][ph][b]);
                    }
                } else {
#pragma unroll
                    for (int b = 0; b + 1 < NF; b += 2) {
                        float v[8];
                        epi_swap8(acc[ph][wh][0][b], acc[ph][wh][0][b + 1], v);
                        emit(r0 + (b + par) * 16, v, avs[wh][ph][b]);
                    }
                    if (NF & 1) {   // (224 x 192 tile: a third pixel fragment without a partner leaves as 8 B per lane)
                        const int m = r0 + (NF - 1) * 16, c4 = cbase + fk * 4;
                        if (!GUARD || m < Mc) {
                            const long long o = (long long)m * g.NOUT + c4;
                            f32x4 v = acc[ph][wh][0][NF - 1];
                            if (EPI == EPI_BIAS_LEAKY) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const float tv = __builtin_fmaf(v[j], g.scale ? g.scale[c4 + j] : 1.0f, g.bias[c4 + j]);
                                    v[j] = tv > 0.0f ? tv : tv * g.slope;
                                }
                            } else if (EPI == EPI_MASK) {
                                const bf16x4 av = *reinterpret_cast<const bf16x4*>(g.act + o);
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] = (float)av[j] > 0.0f ? v[j] : v[j] * g.slope;
                            }
                            bf16x4 ov;
#pragma unroll
                            for (int j = 0; j < 4; ++j) ov[j] = (bf16_t)v[j];
                            *reinterpret_cast<bf16x4*>(g.out + o) = ov;
                        }
                    }
                }
            }
        }
        };
        if (cm0 + BM <= Mc) run(std::false_type()); else run(std::true_type());
    } else {
        // EPI_POOL: the fp32 tile through LDS (the 2x2 windows span four pixel rows of the tile = four lanes of a fragment
        // column, and the window's argmax goes out beside the maximum).  Two passes (pixel halves); in each, both wave rows
        // park their slab (<= 64 rows of the fp32 tile) side by side in LDS (2 x 65 KiB: the ring and the staging state
        // behind it are dead by now), one barrier, then one item per thread and slab: (window, 8 channels); see
        // igemm_kernel for the pooling rule
        constexpr int TPR = BN / 8;
        constexpr int SLAB = 64 * B8_EPI_STRIDE;
        static_assert(2 * SLAB <= B8_LDS_IG, "epilogue slabs must fit in the kernel's LDS");
        const int w = tid / TPR, c8 = tid - w * TPR;
        float bvp[8], svp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bvp[j] = g.bias[n0 + c8 * 8 + j];
            svp[j] = g.scale ? g.scale[n0 + c8 * 8 + j] : 1.0f;
        }
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            constexpr int kNF[2] = {4, FP1};
            const int NF = kNF[ph];
            const int PR = NF * 16;                                       // rows per slab
            const int rbase = ph * 128;                                   // first tile row of slab 0; slab 1 follows at + PR
            unsigned char* const myslab = smem + wm * SLAB;
#pragma unroll
            for (int wh = 0; wh < 2; ++wh)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        if (b >= NF || (wh == 1 && a >= FW1)) continue;
                        const int ch = wh * 128 + wn * (wh == 1 ? FW1 * 16 : 32) + a * 16 + fk * 4;
                        *reinterpret_cast<f32x4*>(myslab + (b * 16 + frow) * B8_EPI_STRIDE + ch * 4) = acc[ph][wh][a][b];
                    }
            __syncthreads();
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const int q = ((m0 + rbase + sl * PR) >> 2) + w;
                if (w < PR / 4 && q * 4 < Mc) {
                    bf16x8 best;
                    unsigned packed = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned char* rp = smem + sl * SLAB + (4 * w + k) * B8_EPI_STRIDE + c8 * 32;
                        const f32x4 v0 = *reinterpret_cast<const f32x4*>(rp);
                        const f32x4 v1 = *reinterpret_cast<const f32x4*>(rp + 16);
                        const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float tv = __builtin_fmaf(v[j], svp[j], bvp[j]);
                            const bf16_t yv = (bf16_t)(tv > 0.0f ? tv : tv * g.slope);
                            if (k == 0 || (float)yv > (float)best[j]) {
                                best[j] = yv;
                                packed = (packed & ~(3u << (2 * j))) | ((unsigned)k << (2 * j));
                            }
                        }
                    }
                    *reinterpret_cast<bf16x8*>(g.out + (long long)q * g.NOUT + n0 + c8 * 8) = best;
                    g.pool_idx[(long long)q * (g.NOUT / 8) + (n0 >> 3) + c8] = (unsigned short)packed;
                }
            }
            if (ph == 0) __syncthreads();
        }
    }
    IG8_T(3, cvb);
    // Everything of this tile has left and the next tile's first K-step has landed after this wait (~1 K cycles after the
    // last store was issued; the second K-step's half-tiles were requested a whole epilogue ago).  The builtin, not the
    // inline-asm wait, and on the way out of the last tile too: the compiler has to KNOW that nothing of the epilogue is
    // outstanding at the top of the tile loop, on every path its data flow merges there.  Otherwise it protects the
    // fragment registers, which the epilogue's loads reused, with vmcnt(0) waits of its own at the top of every K-step
    // and the loop loses its two K-steps of load run-ahead (85 K -> 90 K cycles per feat.7 tile).
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0); expcnt and lgkmcnt untouched
    IG8_T(4, cvb);
    if (!more) break;
#pragma unroll
    for (int i = 0; i < 32; ++i) (&acc[0][0][0][0])[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

#ifdef YFS_IG8_TIMING
}  // namespace
extern "C" __attribute__((visibility("default"))) int yfs_debug_ig8_times(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(yfs_ig8_dbg), 8 * sizeof(long long));
}
extern "C" __attribute__((visibility("default"))) int yfs_debug_ig8_timeline(long long* out, int nblocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(yfs_ig8_tl), (size_t)(nblocks < 8192 ? nblocks : 8192) * 6 * sizeof(long long));
}
namespace {
#endif
// CUs of the current device (a multiple of 8 on every CDNA part: the XCD-aware tile order relies on workgroup w and
// workgroup w + cu_count() landing on the same XCD, see xcd_remap)
int cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        n -= n % 8;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

template <int EPI, int FP1, int FW1 = 2>
int launch_igemm8(IgemmArgs& g, hipStream_t s, const char* name) {
    g.mtiles = (int)yfs_cdiv(g.M, 128 + FP1 * 32);
    g.ntiles = g.NOUT / (128 + FW1 * 64);
    g.div_hw = make_fastdiv(EPI == EPI_POOL ? (unsigned)((g.OH / 2) * (g.OW / 2)) : (unsigned)(g.OH * g.OW));
    g.div_w = make_fastdiv(EPI == EPI_POOL ? (unsigned)(g.OW / 2) : (unsigned)g.OW);
    auto kern = igemm8_kernel<EPI, FP1, FW1>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, B8_LDS_IG);
    YFS_REQUIRE(e == hipSuccess, YFS_ERR_LAUNCH, "%s: cannot reserve %d bytes of LDS", name, B8_LDS_IG);
    const int copies = EPI == EPI_PARTIAL ? g.ksplit : 1;
    int grid = g.mtiles * g.ntiles * copies;
    if (igemm8_persistent<EPI, FP1, FW1>() && yfs_knob(YFS_KNOB_IGEMM_PERSIST) == 0) {   // one workgroup per CU walks the tiles (the kernel strides by gridDim.x)
        const int cus = cu_count();
        YFS_REQUIRE(cus > 0, YFS_ERR_LAUNCH, "%s: cannot read the device's CU count", name);
        if (grid > cus) grid = cus;
    }
    kern<<<dim3((unsigned)grid), dim3(512), B8_LDS_IG, s>>>(g);
    YFS_CHECK_LAUNCH(name);
    yfs_note_kernel("igemm8_kernel<%d,%d,epi%d>", 128 + FP1 * 32, 128 + FW1 * 64, EPI);
    return YFS_OK;
}

// out[m][c] = epilogue(sum_s partial[s][m][c]), s in index order (fixed: deterministic), eight channels per thread.
// Same epilogue arithmetic as the in-kernel one: fma(v, scale, bias) + LeakyReLU, or the activation-derivative mask.
__global__ void __launch_bounds__(kThreads) splitk_epilogue_kernel(IgemmArgs g, long long n8) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n8) return;
    const long long o = i * 8;
    const long long mn = (long long)g.M * g.NOUT;
    float v[8];
    {
        const f32x4 a = *reinterpret_cast<const f32x4*>(g.partial + o), b = *reinterpret_cast<const f32x4*>(g.partial + o + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
    for (int sidx = 1; sidx < g.ksplit; ++sidx) {
        const float* p = g.partial + sidx * mn + o;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
        v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3]; v[4] += b[0]; v[5] += b[1]; v[6] += b[2]; v[7] += b[3];
    }
    const int c = (int)(o % g.NOUT);
    if (g.want_epi == EPI_BIAS_LEAKY) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float tv = __builtin_fmaf(v[j], g.scale ? g.scale[c + j] : 1.0f, g.bias[c + j]);
            v[j] = tv > 0.0f ? tv : tv * g.slope;
        }
    } else if (g.want_epi == EPI_MASK) {
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(g.act + o);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)av[j] > 0.0f ? v[j] : v[j] * g.slope;
    }
    bf16x8 ov;
#pragma unroll
    for (int j = 0; j < 8; ++j) ov[j] = (bf16_t)v[j];
    *reinterpret_cast<bf16x8*>(g.out + o) = ov;
}

// Split-K plan of the 8-phase kernel for GEMMs with few pixels and a long reduction (feat.25-27: 3136 pixels x 1024
// channels, K = 9216): 56 tiles of 224 x 256 cannot fill 256 CUs and the small tiles that can re-read the 19 MB of
// weights once per 64 pixels (L2-bound at ~0.5 PFLOP/s).  Four K-splits of the big tile put 224 workgroups on the chip,
// each streaming a quarter of the weights; the fp32 partial tiles (12.8 MB per split) are summed by
// splitk_epilogue_kernel.  Returns the split count (1 = do not split).
int igemm8_ksplit(const IgemmArgs& g, int rows, int force) {
    const int ksteps = g.KH * g.KW * g.C / BK;
    const long long tiles = yfs_cdiv(g.M, rows) * (g.NOUT / 256);
    int ks = 1;
    if (force == 84) ks = ksteps < 4 ? ksteps : 4;                       // tests: split whatever the shape
    else if (tiles <= 128 && ksteps >= 64) ks = (int)std::min<long long>(256 / tiles, ksteps / 16);
    if (ks < 2) return 1;
    const int per = (int)yfs_cdiv(ksteps, ks);
    return (int)yfs_cdiv(ksteps, per);
}
size_t igemm8_partial_bytes(const IgemmArgs& g, int ks) { return ks > 1 ? (size_t)ks * g.M * g.NOUT * sizeof(float) : 0; }

int launch_igemm8_splitk(IgemmArgs& g, int ks, int want_epi, hipStream_t s, const char* name) {
    const int ksteps = g.KH * g.KW * g.C / BK;
    g.ksteps_per_split = (int)yfs_cdiv(ksteps, ks);
    g.ksplit = (int)yfs_cdiv(ksteps, g.ksteps_per_split);
    g.want_epi = want_epi;
    const int rc = g.M % 224 == 0 ? launch_igemm8<EPI_PARTIAL, 3>(g, s, name) : launch_igemm8<EPI_PARTIAL, 4>(g, s, name);
    if (rc) return rc;
    const long long n8 = (long long)g.M * g.NOUT / 8;
    splitk_epilogue_kernel<<<dim3((unsigned)yfs_cdiv(n8, kThreads)), dim3(kThreads), 0, s>>>(g, n8);
    YFS_CHECK_LAUNCH(name);
    yfs_note_kernel("igemm8_kernel<%d,256,epi4,ksplit%d>+splitk_epilogue<epi%d>", g.M % 224 == 0 ? 224 : 256, g.ksplit, want_epi);
    return YFS_OK;
}

// Tile choice (measured on MI355X, batch 64, tools/conv_bench.py).  Default: 128x128 / 4 waves / 2-stage 64-deep
// ring at two workgroups per CU (L7 ~1000, L18 ~1070 TFLOP/s).  Ablations show why it sits at ~40 % of the MFMA
// peak: with the global->LDS staging removed the same loop runs at 1340-1560 TFLOP/s, with staging that fetches
// nothing (all lanes out of range) at 1050-1240: a 128x128 tile streams 32 KiB per 2.1 MFLOP, ~31 B/clk/CU at
// this rate, which is what the vector-memory -> LDS path sustains per CU (MI355X_MICROARCH.md: 66-73 GB/s per CU).
// 256x256 / 8 waves / one workgroup per CU halves the bytes per FLOP and reaches 1090-1130 TFLOP/s where its tile
// count quantises well on 256 CUs (196, 392, 3136 tiles), but loses when it does not (98, 784) or when K is short
// (four-pass epilogue); it is chosen by the round-count model below.  Tried and rejected: 256x128 / 8 waves /
// 3 stages (875-990), 32-deep rings at four workgroups per CU (790-880), 32-deep 4/5-stage rings (780-870),
// 256x128 / 4 waves / 32-deep (810-980), issuing the loads between the MFMAs instead of in a burst (-6 %),
// 256x256 / 4 waves of 128x128 at one wave per SIMD with all 256 AGPRs as accumulators (the vendor GEMM's shape;
// compiles without spills but the compiler's schedule leaves the single wave exposed: 730-890), 256x192 / 8 waves for
// the 192-channel outputs (stem + pool 584 vs 571 us, feat.2 dgrad 759 vs 738 us: no gain), a 3-deep ring for grids of
// at most one workgroup per CU (feat.25-27, 200 tiles: 101 vs 96 us — a lone workgroup is not latency-bound either: its
// 32 KiB per K-step at ~31 B/clk is 1060 of the 1390 cycles a step takes, the same per-CU staging limit as with two).
// First fragment reads of a K-step issued BEFORE the next tile's global -> LDS loads (so the matrix pipe starts one LDS
// latency earlier after the barrier): 12.70 -> 12.88 ms per step — the loads landing that much later costs more.
// A weight-major tile order (an XCD's run of tiles shares weight tiles instead of pixel tiles) for the layers whose weights
// outweigh their activations (feat.21-27): 12.85 -> 12.88 ms per step, everywhere: 12.97 (rejected).
// 128x192 serves the 192-channel outputs (pixel tile read once instead of three times with 128x64).  The YFS_KNOB_IGEMM_TILE knob (128|256|2562|224|64128|8) forces a variant for benchmarking and tests.
template <int EPI, bool SUB>
int dispatch_igemm(IgemmArgs& g, hipStream_t s, const char* name) {
    const int force = yfs_knob(YFS_KNOB_IGEMM_TILE);
    if constexpr (EPI == EPI_UNPOOL) {
        // dgrad of the layer behind a pool, writing that pool's input gradient (feat.4 / 9 / 20, all 1x1): store-bound (four
        // output pixels per GEMM row), on the small kernel only
        static_assert(!SUB, "EPI_UNPOOL is a stride-1 epilogue");
        if (g.NOUT % 192 == 0 && g.NOUT % 128 != 0) return launch_igemm<128, 192, 2, 2, 2, EPI, false>(g, s, name);
        if (g.NOUT % 128 != 0) return launch_igemm<128, 64, 2, 2, 2, EPI, false>(g, s, name);
        if (force == 128 || (force == 0 && g.KH * g.KW > 1)) return launch_igemm<128, 128, 2, 2, 2, EPI, false>(g, s, name);
        return launch_igemm<64, 128, 2, 2, 2, EPI, false>(g, s, name);
    } else {
    if constexpr (EPI == EPI_MASK && !SUB) {
        // feat.2's dgrad (192 output channels, K = 2304): the 8-phase kernel with a 192-channel tile (knob 192 forces
        // it, knob 128 the small kernel)
        const bool fits = g.NOUT == 192 && g.H < 16000 && g.W < 16000 && g.OH * g.s_mul < 16000 && g.OW * g.s_mul < 16000;
        if (fits && (force == 192 || (force == 0 && g.KH * g.KW * g.C >= 1152 && yfs_cdiv(g.M, 256) >= 200)))
            return g.M % 256 == 0 || g.M % 224 != 0 ? launch_igemm8<EPI, 4, 1>(g, s, name) : launch_igemm8<EPI, 3, 1>(g, s, name);
    }
    if (g.NOUT % 192 == 0 && g.NOUT % 128 != 0 && force != 64) return launch_igemm<128, 192, 2, 2, 2, EPI, SUB>(g, s, name);
    if (g.NOUT % 128 != 0) return launch_igemm<128, 64, 2, 2, 2, EPI, SUB>(g, s, name);
    // (igemm8_kernel packs a row's source coordinates into 16 bits each)
    const bool ok8 = g.NOUT % 256 == 0 && g.H < 16000 && g.W < 16000 && g.OH * g.s_mul < 16000 && g.OW * g.s_mul < 16000;
    if constexpr (!SUB && EPI != EPI_POOL) {
        if ((force == 0 || force == 84) && ok8 && g.partial_cap > 0) {
            const int ks = igemm8_ksplit(g, g.M % 224 == 0 ? 224 : 256, force);
            if (ks > 1 && igemm8_partial_bytes(g, ks) <= g.partial_cap) return launch_igemm8_splitk(g, ks, EPI, s, name);
        }
    }
    if constexpr (!SUB) {
        if (force == 8 && ok8) return launch_igemm8<EPI, 4>(g, s, name);
        if (force == 87 && ok8) return launch_igemm8<EPI, 3>(g, s, name);
    }
    if (force == 256) return launch_igemm<256, 128, 4, 2, 3, EPI, SUB>(g, s, name);
    if (force == 64128) return launch_igemm<64, 128, 2, 2, 2, EPI, SUB>(g, s, name);
    if constexpr (!SUB) {
        // 8-phase kernel (igemm8_kernel): measured on the 3x3 layers of the backbone, batch 64 — forward L7 418 -> 397 us,
        // L18 403 -> 381, L21 100 -> 94, L24 190 -> 174; dgrad L7 455 -> 426, L18 397 -> 359, L24 206 -> 185, L10 107 -> 102;
        // conv + pool L2 / L7 / L18 1495 -> 1414 in total.  Its main loop runs a 64-deep K-step in ~1.17 us (the rate
        // the chip's clock under MFMA load allows), but every round of 256 tiles pays ~12 us of prologue, epilogue burst
        // and workgroup turnover, so it needs long K and a grid that fills the CUs: K >= 1152 (1728 for the pool-fused
        // form) and at least 200 tiles.  224 rows when that needs fewer or shorter rounds.
        const int kdepth = g.KH * g.KW * g.C;
        // 1x1 layers with K >= 512 whose 224-row tiles fill whole rounds (feat.9-17 forward: 224 / 448 tiles, feat.17 / 20 /
        // 22 dgrad): same box, 29 -> 24 us (feat.9 forward), 49 -> 41.5 (feat.17 forward), 48.5 -> 45 (feat.17 dgrad),
        // 28.7 -> 24.4 (feat.20 dgrad); the pixel tile is read once instead of once per 128 channels.  K = 256 (feat.6, and
        // feat.9's dgrad) is slower on it and stays on the small tiles.
        const int kmin = EPI == EPI_POOL ? 1728 : (g.KH * g.KW == 1 ? 512 : 1152);
        if (force == 0 && ok8 && kdepth >= kmin) {   // (1152: feat.5 forward 158 -> 142 us)
            const long long t224 = yfs_cdiv(g.M, 224) * (g.NOUT / 256), t256 = yfs_cdiv(g.M, 256) * (g.NOUT / 256);
            const double per = kdepth / 64 * 1.17;
            const double c224 = (double)yfs_cdiv(t224, 256) * (12.0 + per * 0.95), c256 = (double)yfs_cdiv(t256, 256) * (12.0 + per);
            if (t224 >= 200) return c224 <= c256 ? launch_igemm8<EPI, 3>(g, s, name) : launch_igemm8<EPI, 4>(g, s, name);
        }
    }
    bool big = force == 2562 || force == 224, rows224 = force == 224;
    if (force == 0 && !SUB) {
        // rounds x time per round: 128x128 runs 512 tiles per round (2 per CU) in 2u, 256x256 runs 256 tiles per
        // round in 4u / 1.18 (its measured per-CU advantage); take the big tile when that wins by >= 10 %.
        // 224x256 is the same kernel with 7 instead of 8 pixel fragments per wave: every M of this network is
        // 64 * (7 * 2^k)^2, so 224-row tiles have no ragged edge and their counts (224, 448, 896, ...) fill 256 CUs to
        // 87.5 % where 196 / 392 tiles of 256 rows fill 76.6 %.
        const long long t128 = yfs_cdiv(g.M, 128) * (g.NOUT / 128), t256 = yfs_cdiv(g.M, 256) * yfs_cdiv(g.NOUT, 256);
        const long long t224 = yfs_cdiv(g.M, 224) * yfs_cdiv(g.NOUT, 256);
        const double c128 = (double)yfs_cdiv(t128, 512) * 2.0, c256 = (double)yfs_cdiv(t256, 256) * (4.0 / 1.18);
        const double c224 = (double)yfs_cdiv(t224, 256) * (4.0 / 1.18) * (224.0 / 256.0);
        rows224 = c224 < c256;
        big = (rows224 ? c224 : c256) < 0.9 * c128 && g.KH * g.KW * g.C >= 1728;   // (shorter K: the epilogue passes eat the gain; 224 rows at K >= 1152 or 512: no change)
    }
    if constexpr (SUB) {
        // stride-2 dgrad (feat.25): four parity classes of 3136 pixels each; as 224 x 256 tiles they are 224 workgroups, one
        // per CU in a single round, against 800 tiles of 128 x 128 in 1.6 rounds of two per CU: 126 -> 107 us (same box)
        if (force == 0 && g.NOUT % 256 == 0 && g.KH * g.KW * g.C >= 4608) {
            const int sb = g.sub_s;
            long long t224 = 0;
            for (int c = 0; c < sb * sb; ++c)
                t224 += yfs_cdiv((long long)g.N * ((g.OH - c / sb + sb - 1) / sb) * ((g.OW - c % sb + sb - 1) / sb), 224) * (g.NOUT / 256);
            if (t224 >= 128 && t224 <= 256) { big = true; rows224 = true; }
        }
    }
    if (big && g.NOUT % 256 == 0)
        return rows224 ? launch_igemm<224, 256, 2, 4, 2, EPI, SUB>(g, s, name) : launch_igemm<256, 256, 2, 4, 2, EPI, SUB>(g, s, name);
    // fewer 128x128 tiles than CUs (feat.25-27: 200 tiles): 64-row tiles put a workgroup on every CU (392 tiles; M = 49 * 64
    // exactly): 96 -> 87 us per launch.  Not for the strided dgrad (slower there) nor for larger grids (L9/L17/L20: slower).
    if (force == 0 && !SUB && yfs_cdiv(g.M, 128) * (g.NOUT / 128) <= 256 && yfs_cdiv(g.M, 64) * (g.NOUT / 128) > 256)
        return launch_igemm<64, 128, 2, 2, 2, EPI, SUB>(g, s, name);
    // 1x1 dgrad (HBM-bound: reads dz and the mask source, writes dx): three 48 KiB workgroups per CU keep more loads in
    // flight than two 64 KiB ones - same box, batch 64: L4 52.7 -> 47.3 us, L6 91.6 -> 87.2, L9 39.5 -> 36.0, L17 49.8 -> 47.3,
    // L20 29.4 -> 28.2 (the 1x1 forward launches are slower with this tile and stay on 128x128)
    if (force == 0 && !SUB && EPI == EPI_MASK && g.KH * g.KW == 1) return launch_igemm<64, 128, 2, 2, 2, EPI, SUB>(g, s, name);
    return launch_igemm<128, 128, 2, 2, 2, EPI, SUB>(g, s, name);
    }
}

int check_conv_shape(const char* fn, int64_t N, int H, int W, int Cin, int Cout, int K, int stride) {
    YFS_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, YFS_ERR_INVALID_ARG, "%s: non-positive dimension", fn);
    YFS_REQUIRE(K == 1 || K == 3 || K == 5 || K == 7, YFS_ERR_UNSUPPORTED, "%s: kernel size %d not in {1,3,5,7}", fn, K);
    YFS_REQUIRE(stride == 1 || stride == 2, YFS_ERR_UNSUPPORTED, "%s: stride %d not in {1,2}", fn, stride);
    YFS_REQUIRE(Cin % 64 == 0, YFS_ERR_UNSUPPORTED, "%s: Cin=%d must be a multiple of 64 (use yfs_im2col_bf16 for the stem)", fn, Cin);
    YFS_REQUIRE(Cout % 64 == 0, YFS_ERR_UNSUPPORTED, "%s: Cout=%d must be a multiple of 64", fn, Cout);
    const int pad = K / 2;
    const int64_t Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
    YFS_REQUIRE(N * Ho * Wo * Cout * 2 < 0xFFF00000ll && N * H * W * Cin * 2 < 0xFFF00000ll, YFS_ERR_UNSUPPORTED,
                "%s: an activation tensor exceeds the 4 GiB range of one buffer descriptor (split the batch)", fn);
    return YFS_OK;
}

}  // namespace

namespace {
void fwd_args(IgemmArgs& g, const void* x, const void* w, const float* scale, const float* shift, void* y, int64_t N, int H, int W,
              int Cin, int Cout, int K, int stride, float slope) {
    const int pad = K / 2;
    g.x = (const bf16_t*)x; g.w = (const bf16_t*)w; g.bias = shift; g.scale = scale; g.out = (bf16_t*)y;
    g.N = (int)N; g.H = H; g.W = W; g.C = Cin;
    g.OH = (H + 2 * pad - K) / stride + 1; g.OW = (W + 2 * pad - K) / stride + 1; g.NOUT = Cout;
    g.KH = g.KW = K; g.PS = Cin; g.s_mul = stride; g.pad = pad; g.slope = slope;
    g.M = (int)(N * g.OH * g.OW);
    g.xbytes = (unsigned)(N * H * W * Cin * 2); g.wbytes = (unsigned)((int64_t)Cout * K * K * Cin * 2);
    g.ksplit = 1;
}
void dgrad_args(IgemmArgs& g, const void* dz, const void* wt, const void* x_act, void* dx, int64_t N, int H, int W, int Cin,
                int Cout, int K, int stride, float slope) {
    const int pad = K / 2;
    g.x = (const bf16_t*)dz; g.w = (const bf16_t*)wt; g.act = (const bf16_t*)x_act; g.out = (bf16_t*)dx;
    g.N = (int)N;
    g.H = (H + 2 * pad - K) / stride + 1; g.W = (W + 2 * pad - K) / stride + 1; g.C = Cout;   // gathered tensor = dz
    g.OH = H; g.OW = W; g.NOUT = Cin;
    g.KH = g.KW = K; g.PS = Cout; g.s_mul = 1; g.pad = K - 1 - pad; g.sub_s = stride; g.slope = slope;
    g.M = (int)(N * H * W);
    g.xbytes = (unsigned)(N * g.H * g.W * Cout * 2); g.wbytes = (unsigned)((int64_t)Cout * K * K * Cin * 2);
    g.ksplit = 1;
}
// bytes of fp32 partial tiles the split-K form would use for this GEMM (0: it does not split)
size_t igemm_workspace_bytes(const IgemmArgs& g, bool sub) {
    const bool ok8 = g.NOUT % 256 == 0 && g.H < 16000 && g.W < 16000 && g.OH * g.s_mul < 16000 && g.OW * g.s_mul < 16000;
    const int force = yfs_knob(YFS_KNOB_IGEMM_TILE);
    if (sub || !ok8 || !(force == 0 || force == 84)) return 0;
    return igemm8_partial_bytes(g, igemm8_ksplit(g, g.M % 224 == 0 ? 224 : 256, force));
}
}  // namespace

YFS_API size_t yfs_conv2d_fwd_workspace_bytes(int64_t N, int H, int W, int Cin, int Cout, int K, int stride) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || K <= 0 || stride <= 0) return 0;
    IgemmArgs g = {};
    fwd_args(g, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, Cout, K, stride, 0.0f);
    return igemm_workspace_bytes(g, false);
}

YFS_API size_t yfs_conv2d_dgrad_workspace_bytes(int64_t N, int H, int W, int Cin, int Cout, int K, int stride) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || K <= 0 || stride <= 0) return 0;
    IgemmArgs g = {};
    dgrad_args(g, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, Cout, K, stride, 0.0f);
    return igemm_workspace_bytes(g, stride != 1);
}

YFS_API int yfs_conv2d_fwd_ws_bf16(const void* x, const void* w, const float* scale, const float* shift, void* y, int64_t N,
                                   int H, int W, int Cin, int Cout, int K, int stride, float slope, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    YFS_REQUIRE(x && w && shift && y, YFS_ERR_INVALID_ARG, "yfs_conv2d_fwd_bf16: null pointer");
    int rc = check_conv_shape("yfs_conv2d_fwd_bf16", N, H, W, Cin, Cout, K, stride);
    if (rc) return rc;
    YFS_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, YFS_ERR_WORKSPACE, "yfs_conv2d_fwd_ws_bf16: workspace must be 16-byte aligned");
    IgemmArgs g = {};
    fwd_args(g, x, w, scale, shift, y, N, H, W, Cin, Cout, K, stride, slope);
    g.partial = (float*)workspace; g.partial_cap = workspace ? workspace_bytes : 0;
    return dispatch_igemm<EPI_BIAS_LEAKY, false>(g, (hipStream_t)stream, "yfs_conv2d_fwd_bf16");
}

YFS_API int yfs_conv2d_fwd_affine_bf16(const void* x, const void* w, const float* scale, const float* shift, void* y, int64_t N,
                                       int H, int W, int Cin, int Cout, int K, int stride, float slope, void* stream) {
    return yfs_conv2d_fwd_ws_bf16(x, w, scale, shift, y, N, H, W, Cin, Cout, K, stride, slope, nullptr, 0, stream);
}

YFS_API int yfs_conv2d_fwd_bf16(const void* x, const void* w, const float* bias, void* y, int64_t N, int H, int W,
                                int Cin, int Cout, int K, int stride, float slope, void* stream) {
    YFS_REQUIRE(x && w && bias && y, YFS_ERR_INVALID_ARG, "yfs_conv2d_fwd_bf16: null pointer");
    return yfs_conv2d_fwd_affine_bf16(x, w, nullptr, bias, y, N, H, W, Cin, Cout, K, stride, slope, stream);
}

YFS_API int yfs_conv2d_fwd_pool_bf16(const void* x, const void* w, const float* bias, void* y_pooled, void* pool_idx, int64_t N,
                                     int H, int W, int Cin, int Cout, int K, int stride, float slope, void* stream) {
    YFS_REQUIRE(x && w && bias && y_pooled && pool_idx, YFS_ERR_INVALID_ARG, "yfs_conv2d_fwd_pool_bf16: null pointer");
    int rc = check_conv_shape("yfs_conv2d_fwd_pool_bf16", N, H, W, Cin, Cout, K, stride);
    if (rc) return rc;
    const int pad = K / 2;
    IgemmArgs g = {};
    g.x = (const bf16_t*)x; g.w = (const bf16_t*)w; g.bias = bias; g.out = (bf16_t*)y_pooled; g.pool_idx = (unsigned short*)pool_idx;
    g.N = (int)N; g.H = H; g.W = W; g.C = Cin;
    g.OH = (H + 2 * pad - K) / stride + 1; g.OW = (W + 2 * pad - K) / stride + 1; g.NOUT = Cout;
    YFS_REQUIRE(g.OH % 2 == 0 && g.OW % 2 == 0, YFS_ERR_UNSUPPORTED,
                "yfs_conv2d_fwd_pool_bf16: conv output %dx%d must be even for the fused 2x2 pool", g.OH, g.OW);
    g.KH = g.KW = K; g.PS = Cin; g.s_mul = stride; g.pad = pad; g.slope = slope;
    g.M = (int)(N * g.OH * g.OW);
    g.xbytes = (unsigned)(N * H * W * Cin * 2); g.wbytes = (unsigned)((int64_t)Cout * K * K * Cin * 2);
    g.ksplit = 1;
    return dispatch_igemm<EPI_POOL, false>(g, (hipStream_t)stream, "yfs_conv2d_fwd_pool_bf16");
}

YFS_API int yfs_conv2d_dgrad_bf16(const void* dz, const void* wt, const void* x_act, void* dx, int64_t N, int H, int W,
                                  int Cin, int Cout, int K, int stride, float slope, void* stream) {
    return yfs_conv2d_dgrad_ws_bf16(dz, wt, x_act, dx, N, H, W, Cin, Cout, K, stride, slope, nullptr, 0, stream);
}

YFS_API int yfs_conv2d_dgrad_ws_bf16(const void* dz, const void* wt, const void* x_act, void* dx, int64_t N, int H, int W,
                                     int Cin, int Cout, int K, int stride, float slope, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    YFS_REQUIRE(dz && wt && dx, YFS_ERR_INVALID_ARG, "yfs_conv2d_dgrad_bf16: null pointer");
    int rc = check_conv_shape("yfs_conv2d_dgrad_bf16", N, H, W, Cout, Cin, K, stride);   // roles swap: reduce over Cout
    if (rc) return rc;
    YFS_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, YFS_ERR_WORKSPACE, "yfs_conv2d_dgrad_ws_bf16: workspace must be 16-byte aligned");
    IgemmArgs g = {};
    dgrad_args(g, dz, wt, x_act, dx, N, H, W, Cin, Cout, K, stride, slope);
    g.partial = (float*)workspace; g.partial_cap = workspace ? workspace_bytes : 0;
    hipStream_t s = (hipStream_t)stream;
    if (stride == 1) {
        return x_act ? dispatch_igemm<EPI_MASK, false>(g, s, "yfs_conv2d_dgrad_bf16")
                     : dispatch_igemm<EPI_NONE, false>(g, s, "yfs_conv2d_dgrad_bf16");
    }
    return x_act ? dispatch_igemm<EPI_MASK, true>(g, s, "yfs_conv2d_dgrad_bf16")
                 : dispatch_igemm<EPI_NONE, true>(g, s, "yfs_conv2d_dgrad_bf16");
}

YFS_API int yfs_conv2d_dgrad_unpool_bf16(const void* dz, const void* wt, const void* y_pooled, const void* pool_idx, void* dx_full,
                                         int64_t N, int H, int W, int Cin, int Cout, int K, float slope, void* stream) {
    YFS_REQUIRE(dz && wt && y_pooled && pool_idx && dx_full, YFS_ERR_INVALID_ARG, "yfs_conv2d_dgrad_unpool_bf16: null pointer");
    int rc = check_conv_shape("yfs_conv2d_dgrad_unpool_bf16", N, H, W, Cout, Cin, K, 1);   // roles swap: reduce over Cout
    if (rc) return rc;
    YFS_REQUIRE(N * 4 * H * W * Cin * 2 < 0xFFF00000ll, YFS_ERR_UNSUPPORTED,
                "yfs_conv2d_dgrad_unpool_bf16: the full-resolution gradient exceeds 4 GiB (split the batch)");
    IgemmArgs g = {};
    dgrad_args(g, dz, wt, y_pooled, dx_full, N, H, W, Cin, Cout, K, 1, slope);
    g.pool_idx = (unsigned short*)pool_idx;   // read only in this form
    return dispatch_igemm<EPI_UNPOOL, false>(g, (hipStream_t)stream, "yfs_conv2d_dgrad_unpool_bf16");
}

// =============================== wgrad ==========================================================
namespace {

struct WgradArgs {
    const bf16_t* dz;   // [M][Cout]   (POOLZ: the POOLED gradient [M/4][Cout])
    const bf16_t* x;    // [N,H,W,Cin]
    const unsigned short* zidx;   // POOLZ: [M/4][Cout/8] eight 2-bit window positions each (the pool's argmax record)
    const bf16_t* zy;             // POOLZ: pooled activation [M/4][Cout] for the LeakyReLU derivative, or null
    float slope;
    unsigned zidx_bytes;
    float* partial;     // [splits][Cout][taps*Cin]
    float* dbpart;      // [splits][taps*citiles][2][Cout] partial column sums of dz (bias gradient), or null
    int H, W, Cin, OH, OW, Cout, KH, KW, PS, stride, pad, M;   // PS: elements between adjacent pixels of x
    // A ci tile wider than 64 normally covers contiguous channels; for the stem's row-pair form (two kernel rows
    // of 64 contiguous elements each per "tap") every group of 64 channels sits xgroup_bytes further on and one tap
    // advances kh_mul input rows.
    unsigned xgroup_bytes;   // 128 = contiguous
    int kh_mul;              // 1 normally
    // SGD epilogue (wgrad_kernel<..., SGD = true>): the tile of dw never leaves the CU — it is the gradient of the
    // optimiser step applied to the same tile of the fp32 master weight, momentum buffer and bf16 shadow in place.
    float* sgd_p;
    float* sgd_m;
    bf16_t* sgd_p16;          // may be null
    const float* sgd_sumsq;   // total squared gradient norm of the whole model (device scalar), or null = no clipping
    float sgd_max_norm, sgd_lr, sgd_momentum, sgd_wd, sgd_gscale;
    int sgd_first;
    unsigned xbytes, zbytes;
    int cotiles, citiles, splits, ktiles, ktiles_per_split;
    FastDiv div_ohw, div_ow;
};

// LDS image of one operand tile: [64 pixels][W channels] bf16, row-major (CPP = W/8 16-byte chunks per pixel
// row), filled by buffer_load ... lds (lane-linear, so a wave-instruction fetches whole pixel rows from HBM) and
// read with ds_read_b64_tr_b16.  A 32-lane half of a transposed read touches pixels {8g+4h+q : g=0,1; q=0..3},
// 32 bytes each; the chunk index is XORed with a function of the pixel so those 8 x 32 B land on disjoint banks:
//   256-B rows (CPP=16): all rows alias -> 4-bit XOR ((p&3)<<2 | (p>>2)&3)
//   128-B / 384-B rows (CPP=8, 24): rows alias in pairs -> XOR of the 32-B pair index with (p>>1)&1 | ((p>>3)&1)<<1
// The XOR is an involution inside an aligned group of 8 (or 16) chunks, applied on the SOURCE side when staging.
template <int CPP>
__device__ __forceinline__ int wg_swz(int pixel, int c) {
    if (CPP == 16) return c ^ (((pixel & 3) << 2) | ((pixel >> 2) & 3));
    return c ^ ((((pixel >> 1) & 1) | (((pixel >> 3) & 1) << 1)) << 1);
}
template <int CPP>
__device__ __forceinline__ unsigned wg_addr(int pixel, int ch /* channel inside the tile, multiple of 4 */) {
    return (unsigned)(pixel * (CPP * 16) + wg_swz<CPP>(pixel, ch >> 3) * 16 + (ch & 7) * 2);
}

// BX = ci tile (MFMA rows), BZ = co tile (MFMA cols).  4 waves as 2 (ci) x 2 (co).
// POOLZ: the block is followed by a 2x2 max pool and nothing else reads its dz (the stem): instead of reading a
// dz tensor that pool backward would have to write first (3/4 zeros), the dz tile is built in LDS from the pooled
// gradient, the pool's 2-bit argmax record and the pooled activation (LeakyReLU derivative).  The reduction then
// enumerates output pixels window-major (m = 4 * window + position) so that one window is four consecutive rows.
template <int BX, int BZ, bool POOLZ, bool SGD = false>
__global__ void __launch_bounds__(kThreads, 2) wgrad_kernel(WgradArgs g) {
    constexpr int FX = BX / 32, FZ = BZ / 32;          // fragments per wave
    constexpr int CX = BX / 8, CZ = BZ / 8;            // 16-byte chunks per pixel row
    constexpr int LX = CX / 4, LZ = CZ / 4;            // LDS-DMA instructions per wave per tile
    constexpr int XBYTES = BX * 128, ZBYTES = BZ * 128;   // 64 pixels * channels * 2 B
    constexpr int TILE_BYTES = XBYTES + ZBYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wx = wave >> 1, wz = wave & 1;
    const int taps = g.KH * g.KW;
    const int tiles = g.cotiles * g.citiles * taps;
    const int logical = xcd_remap(blockIdx.x, tiles * g.splits);
    // consecutive logical ids sweep (tap, ci tile) for a fixed (split, co tile): they re-read one dz slab
    const int inner = g.citiles * taps;
    const int tci = logical % inner;
    const int rest = logical / inner;
    const int cot = rest % g.cotiles, split = rest / g.cotiles;
    const int tap = tci / g.citiles, cit = tci % g.citiles;
    const int kh = tap / g.KW, kw = tap % g.KW;
    const int co0 = cot * BZ, ci0 = cit * BX;

    const int kt0 = split * g.ktiles_per_split;
    int kt1 = kt0 + g.ktiles_per_split;
    if (kt1 > g.ktiles) kt1 = g.ktiles;

    const __amdgpu_buffer_rsrc_t srx = make_rsrc(g.x, g.xbytes), srz = make_rsrc(g.dz, g.zbytes);

    // Per-lane source of each of this wave's LDS-DMA instructions.  A wave fills 16 pixels of every 64-pixel
    // k-tile (pixels wave*16 .. +15), CX/CZ lanes per pixel.
    //   dz: byte offset = zoff[i] + kt * (64 * Cout * 2): linear in kt, and pixels past M fall off the end of the
    //       descriptor (zeros) by themselves.
    //   x:  the pixel -> (n, oh, ow) -> input offset map costs two divisions and the padding test; doing that for
    //       every lane of every load made the loop VALU-bound (140 VALU per 32 MFMA).  Instead lane l computes it
    //       ONCE for pixel wave*16 + (l&15) of k-tile kt + (l>>4), i.e. for four k-tiles at a time, and the loads
    //       fetch their pixel's value with ds_bpermute and add their channel-chunk offset.  Invalid pixels carry
    //       kPixOOB, which stays past the end of the descriptor after that add (tensors are
    //       < 0xFFF00000 bytes, which also keeps the dz offset of a tail tile from wrapping).
    constexpr unsigned kPixOOB = 0xFFFF0000u;
    unsigned zoff[LZ], xchb[LX];
    int xsrc[LX];
#pragma unroll
    for (int i = 0; i < LZ; ++i) {
        const int l = (wave * LZ + i) * 64 + lane;
        const int pix = l / CZ;
        zoff[i] = ((unsigned)pix * (unsigned)g.Cout + (unsigned)(co0 + wg_swz<CZ>(pix, l % CZ) * 8)) * 2u;
    }
#pragma unroll
    for (int i = 0; i < LX; ++i) {
        const int l = (wave * LX + i) * 64 + lane;
        const int pix = l / CX;
        const int cs = wg_swz<CX>(pix, l % CX);
        xchb[i] = (unsigned)((cs & 7) * 16) + (unsigned)(cs >> 3) * g.xgroup_bytes;
        xsrc[i] = (pix - wave * 16) * 4;   // bpermute byte address of the lane holding that pixel (first k-tile of a group)
    }
    const unsigned tapoff = (unsigned)(((kh * g.kh_mul - g.pad) * g.W + (kw - g.pad)) * g.PS + ci0) * 2u;   // wraps; added mod 2^32
    unsigned pixcache = 0;
    auto pixel_offsets = [&](int ktg) -> unsigned {
        const int m = (ktg + (lane >> 4)) * 64 + wave * 16 + (lane & 15);
        unsigned n, oh, ow;
        if (POOLZ) {   // div_ohw / div_ow divide by the POOLED grid here
            const unsigned q = (unsigned)m >> 2, sub = (unsigned)m & 3u;
            n = fdiv(q, g.div_ohw);
            const unsigned rem = q - n * g.div_ohw.d;
            const unsigned py = fdiv(rem, g.div_ow);
            oh = 2 * py + (sub >> 1);
            ow = 2 * (rem - py * g.div_ow.d) + (sub & 1);
        } else {
            n = fdiv((unsigned)m, g.div_ohw);
            const unsigned rem = (unsigned)m - n * g.div_ohw.d;
            oh = fdiv(rem, g.div_ow);
            ow = rem - oh * g.div_ow.d;
        }
        const int ih = (int)oh * g.stride, iw = (int)ow * g.stride;
        const bool ok = m < g.M && (unsigned)(ih - g.pad + kh * g.kh_mul) < (unsigned)g.H &&
                        (unsigned)(iw - g.pad + kw) < (unsigned)g.W;
        return ok ? (unsigned)((((int)n * g.H + ih) * g.W + iw) * g.PS) * 2u + tapoff : kPixOOB;
    };

    // LDS: [x tile | dz tile] x 2 ring slots
    auto stage_tile = [&](int kt, int buf) {
        unsigned char* base = smem + buf * TILE_BYTES;
        const int t = (kt - kt0) & 3;
        if (t == 0) pixcache = pixel_offsets(kt);
        const unsigned zk = (unsigned)kt * (unsigned)(64 * 2) * (unsigned)g.Cout;
        if (!POOLZ) {
#pragma unroll
            for (int i = 0; i < LZ; ++i) {
                // (a typed local: with a type-dependent offset expression the host pass re-checks the builtin at
                //  instantiation, fails silently and drops the kernel's launch stub)
                const unsigned off = zoff[i] + zk;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srz, (lds_ptr)(base + XBYTES + (wave_u * LZ + i) * 1024), 16, off, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < LX; ++i) {
            const int src = xsrc[i] + t * 64;
            const unsigned off = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)pixcache) + xchb[i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srx, (lds_ptr)(base + (wave_u * LX + i) * 1024), 16, off, 0, 0, 0);
        }
    };

    // POOLZ: a 64-row k-tile is 16 windows.  One item = (window, 4-channel half chunk): 8 B of pooled gradient and one
    // byte of window positions (+ 8 B of pooled activation when the LeakyReLU derivative is still to be applied),
    // exactly CZ/8 items per thread.  Items are fetched into registers TWO tiles ahead (descriptor loads: windows past
    // the end read zeros) and expanded into the four LDS rows of the next tile's dz slot inside the current tile's
    // MFMA block, so the VALU work runs under the matrix pipe instead of in a phase of its own.  The expansion works
    // on bf16 PAIRS: E holds the one-hot window position of both halves, (E >> k) & 0x00010001 is a {0,1} pair, and a
    // packed 16-bit multiply keeps or clears each half (16 VALU per pair for all four rows; element-wise selects cost
    // ~10x that and dominated the kernel).
    constexpr int NZH = CZ / 8;
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 zcur[NZH], znxt[NZH], zynxt[NZH];
    unsigned zpcur[NZH], zpnxt[NZH], zgoff[NZH], zioff[NZH], zlds[NZH][4];
    const __amdgpu_buffer_rsrc_t sri = make_rsrc(POOLZ ? (const void*)g.zidx : (const void*)g.dz, POOLZ ? g.zidx_bytes : 0u);
    const __amdgpu_buffer_rsrc_t sry = make_rsrc((POOLZ && g.zy) ? g.zy : g.dz, g.zbytes);
    const bool zleaky = POOLZ && g.zy != nullptr;
    if (POOLZ) {
#pragma unroll
        for (int i = 0; i < NZH; ++i) {
            const int h = tid + i * kThreads;
            const int w = h / (2 * CZ), rem = h % (2 * CZ), c = rem >> 1, half = rem & 1;
            zgoff[i] = ((unsigned)w * (unsigned)g.Cout + (unsigned)co0 + (unsigned)(c * 8 + half * 4)) * 2u;
            zioff[i] = ((unsigned)w * (unsigned)(g.Cout >> 3) + (unsigned)(co0 >> 3) + (unsigned)c) * 2u + (unsigned)half;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pixel = 4 * w + k;
                zlds[i][k] = (unsigned)(XBYTES + pixel * (CZ * 16) + wg_swz<CZ>(pixel, c) * 16 + half * 8);
            }
        }
    }
    auto zload_next = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NZH; ++i) {
            const unsigned off = zgoff[i] + (unsigned)kt * 32u * (unsigned)g.Cout;          // 16 windows x Cout x 2 B
            const unsigned ioff = zioff[i] + (unsigned)kt * 32u * (unsigned)(g.Cout >> 3);
            znxt[i] = __builtin_amdgcn_raw_buffer_load_b64(srz, off, 0, 0);
            if (zleaky) zynxt[i] = __builtin_amdgcn_raw_buffer_load_b64(sry, off, 0, 0);
            zpnxt[i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(sri, ioff, 0, 0);
        }
    };
    // znxt -> zcur, with the LeakyReLU derivative applied when asked for (same rounding as pool backward:
    // bf16(float(g) * slope) where the activation is <= 0)
    auto zrotate = [&]() {
#pragma unroll
        for (int i = 0; i < NZH; ++i) {
            zcur[i] = znxt[i];
            zpcur[i] = zpnxt[i];
        }
        if (zleaky) {
#pragma unroll
            for (int i = 0; i < NZH; ++i)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const unsigned gw = zcur[i][d], yw = zynxt[i][d];
                    const float ylo = __uint_as_float(yw << 16), yhi = __uint_as_float(yw & 0xFFFF0000u);
                    const bf16_t slo = (bf16_t)(__uint_as_float(gw << 16) * g.slope);
                    const bf16_t shi = (bf16_t)(__uint_as_float(gw & 0xFFFF0000u) * g.slope);
                    const unsigned lo16 = ylo > 0.0f ? (gw & 0xFFFFu) : (unsigned)__builtin_bit_cast(unsigned short, slo);
                    const unsigned hi16 = yhi > 0.0f ? (gw >> 16) : (unsigned)__builtin_bit_cast(unsigned short, shi);
                    zcur[i][d] = lo16 | (hi16 << 16);
                }
        }
    };
    auto zexpand = [&](int buf) {   // branch-free: meant to sit inside the MFMA basic block
        unsigned char* tb = smem + buf * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < NZH; ++i) {
            u32x2 r[4];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const unsigned gw = zcur[i][d];
                const unsigned lo = (zpcur[i] >> (4 * d)) & 3u, hi = (zpcur[i] >> (4 * d + 2)) & 3u;
                const unsigned E = (1u << lo) | (0x10000u << hi);
                const u16x2 gv2 = __builtin_bit_cast(u16x2, gw);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned e = (E >> k) & 0x00010001u;
                    const u16x2 prod = gv2 * __builtin_bit_cast(u16x2, e);
                    r[k][d] = __builtin_bit_cast(unsigned, prod);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x2*>(tb + zlds[i][k]) = r[k];
        }
    };

    f32x4 acc[FX][FZ];
#pragma unroll
    for (int a = 0; a < FX; ++a)
#pragma unroll
        for (int b = 0; b < FZ; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // transposed-read lane geometry: lane (g4 = lane>>4, i = lane&15) supplies the address of pixel
    // 8*g4 + 4*h + (i>>2), channels 4*(i&3)..+3 of the fragment's 16, and receives channel i of pixels
    // 8*g4 + 4*h + 0..3 — exactly elements j = 4h..4h+3 of the MFMA operand A[row i][k = 8*g4 + j].
    const int g4 = lane >> 4, li = lane & 15;
    const int tr_pix = 8 * g4 + (li >> 2), tr_ch = 4 * (li & 3);

    // Bias gradient db[co] = sum over pixels of dz: the dz fragments are already in registers, so the column sum
    // is one more MFMA against an all-ones operand.  The `inner` workgroups that share a dz slab take every
    // inner-th k-tile each (and the two wave rows one 32-pixel half each), so the extra MFMAs are spread evenly
    // (+25 % / inner); every (workgroup, wave row) writes its own partial row, reduced in fixed order afterwards.
    const bool do_db = g.dbpart != nullptr;
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;
    f32x4 accdb[FZ];
#pragma unroll
    for (int b = 0; b < FZ; ++b) accdb[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int dbc = kt0 % inner;

    // one 64-pixel k-tile; three straight-line copies (MODE is a compile-time tag) so that the column-sum MFMAs
    // do not put a branch inside the read/MFMA schedule
    auto tile_mfma = [&](const unsigned char* xb, int nbuf, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const unsigned char* zb = xb + XBYTES;
        if (POOLZ) zexpand(nbuf);   // next tile's dz rows; independent of everything below, scheduled among the MFMAs
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xf[FX], zf[FZ];
#pragma unroll
            for (int a = 0; a < FX; ++a) {
                const int ch = wx * (BX / 2) + a * 16 + tr_ch;
                bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (bf16x4 __attribute__((address_space(3)))*)(xb + wg_addr<CX>(ks * 32 + tr_pix, ch)));
                bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (bf16x4 __attribute__((address_space(3)))*)(xb + wg_addr<CX>(ks * 32 + tr_pix + 4, ch)));
                xf[a] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int b = 0; b < FZ; ++b) {
                const int ch = wz * (BZ / 2) + b * 16 + tr_ch;
                bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (bf16x4 __attribute__((address_space(3)))*)(zb + wg_addr<CZ>(ks * 32 + tr_pix, ch)));
                bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (bf16x4 __attribute__((address_space(3)))*)(zb + wg_addr<CZ>(ks * 32 + tr_pix + 4, ch)));
                zf[b] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int a = 0; a < FX; ++a)
#pragma unroll
                for (int b = 0; b < FZ; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[a], zf[b], acc[a][b], 0, 0, 0);
            if (MODE == 1 + ks) {
#pragma unroll
                for (int b = 0; b < FZ; ++b) accdb[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, zf[b], accdb[b], 0, 0, 0);
            }
        }
    };

    if (kt0 < kt1) {
        stage_tile(kt0, 0);
        if (POOLZ) zload_next(kt0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (POOLZ) {
        zrotate();
        zexpand(0);
        zload_next(kt0 + 1);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int nbuf = ((kt - kt0) + 1) & 1;
        if (POOLZ) {   // registers: tile kt+1 becomes current, tile kt+2 starts loading (past the end: zeros / unused)
            zrotate();
            zload_next(kt + 2);
        }
        if (kt + 1 < kt1) stage_tile(kt + 1, nbuf);
        const int mode = (do_db && dbc == tci) ? 1 + wx : 0;   // 0: no column sum; 1 / 2: this wave sums k-half 0 / 1
        dbc = dbc + 1 == inner ? 0 : dbc + 1;
        const unsigned char* xb = smem + ((kt - kt0) & 1) * TILE_BYTES;
        if (mode == 0) tile_mfma(xb, nbuf, std::integral_constant<int, 0>());
        else if (mode == 1) tile_mfma(xb, nbuf, std::integral_constant<int, 1>());
        else tile_mfma(xb, nbuf, std::integral_constant<int, 2>());
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA of the next tile has landed (vmcnt tracks it)
        __syncthreads();
    }

    // acc[a][b][r] = dw[co = co0 + wz*BZ/2 + b*16 + (lane&15)][tap][ci = ci0 + wx*BX/2 + a*16 + (lane>>4)*4 + r]
    const long long kw_tot = (long long)taps * g.Cin;
    if constexpr (SGD) {
        // fp32 tile -> LDS ([BZ rows][BX floats], 16-byte slots XOR-swizzled by the row so that neither the fragment
        // writes nor the row reads conflict; exactly the 64 KiB of the operand ring, free after the last barrier),
        // then every thread streams float4 columns of whole rows: p, m read and written once, coalesced.
        static_assert(BX == 128 && BZ == 128, "SGD epilogue is instantiated for the 128x128 tile");
#pragma unroll
        for (int a = 0; a < FX; ++a)
#pragma unroll
            for (int b = 0; b < FZ; ++b) {
                const int row = wz * (BZ / 2) + b * 16 + li;
                const int c4 = (wx * (BX / 2) + a * 16 + g4 * 4) >> 2;
                *reinterpret_cast<f32x4*>(smem + row * (BX * 4) + ((c4 ^ (row & 31)) << 4)) = acc[a][b];
            }
        __syncthreads();
        float coef = g.sgd_gscale;
        if (g.sgd_sumsq && g.sgd_max_norm > 0.0f) {
            const float c = g.sgd_max_norm / (sqrtf(g.sgd_sumsq[0]) + 1e-6f);   // clip_grad_norm_ (as sgd_kernel)
            coef *= c < 1.0f ? c : 1.0f;
        }
        const int c4 = tid & 31, r0 = tid >> 5;
        constexpr int UN = 4;
#pragma unroll 1
        for (int sw = 0; sw < BZ / 8; sw += UN) {
            f32x4 pv[UN], mv[UN], gv[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {   // all loads of the batch in flight before the first use
                const int row = (sw + u) * 8 + r0;
                const long long e = (long long)(co0 + row) * kw_tot + ci0 + c4 * 4;
                pv[u] = *reinterpret_cast<const f32x4*>(g.sgd_p + e);
                if (!g.sgd_first) mv[u] = *reinterpret_cast<const f32x4*>(g.sgd_m + e);
                gv[u] = *reinterpret_cast<const f32x4*>(smem + row * (BX * 4) + ((c4 ^ (row & 31)) << 4));
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int row = (sw + u) * 8 + r0;
                const long long e = (long long)(co0 + row) * kw_tot + ci0 + c4 * 4;
                f32x4 bo, po;
                bf16x4 ps;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = gv[u][r] * coef + g.sgd_wd * pv[u][r];
                    bo[r] = g.sgd_first ? d : g.sgd_momentum * mv[u][r] + d;
                    po[r] = pv[u][r] - g.sgd_lr * bo[r];
                    ps[r] = (bf16_t)po[r];
                }
                *reinterpret_cast<f32x4*>(g.sgd_m + e) = bo;
                *reinterpret_cast<f32x4*>(g.sgd_p + e) = po;
                if (g.sgd_p16) *reinterpret_cast<bf16x4*>(g.sgd_p16 + e) = ps;
            }
        }
        return;
    }
    float* pbase = g.partial + (long long)split * g.Cout * kw_tot;
#pragma unroll
    for (int a = 0; a < FX; ++a)
#pragma unroll
        for (int b = 0; b < FZ; ++b) {
            const int co = co0 + wz * (BZ / 2) + b * 16 + li;
            const int ci = ci0 + wx * (BX / 2) + a * 16 + g4 * 4;
            *reinterpret_cast<f32x4*>(pbase + (long long)co * kw_tot + (long long)tap * g.Cin + ci) = acc[a][b];
        }
    if (do_db && g4 == 0) {   // all 16 rows of accdb are equal: row 0 (lanes 0-15, register 0) is written
        float* drow = g.dbpart + ((long long)(split * inner + tci) * 2 + wx) * g.Cout;
#pragma unroll
        for (int b = 0; b < FZ; ++b) drow[co0 + wz * (BZ / 2) + b * 16 + li] = accdb[b][0];
    }
}

// ---- weight gradient on the 8-phase schedule (see igemm8_kernel) ------------------------------------------------------
// Tile = (FX * 64 input channels) x 256 output channels of one tap, 8 waves, 64-pixel K-steps.  Half-tiles are LDS
// images [64 pixels][128 channels] (X0 / X1 = the two halves of the ci tile, Z0 / Z1 of the co tile; for the 192-channel
// ci tile the last 32 channel slots of each X image are never read), filled by LDS-DMA, four pixel rows per wave-
// instruction, and read with ds_read_b64_tr_b16.  MFMA roles: A = dz fragment (co), B = x fragment (ci), so a lane ends
// up with 4 consecutive co of one ci; the epilogue transposes through LDS and writes ci-contiguous rows, i.e. the same
// [split][co][tap][ci] slabs as wgrad_kernel (same reduction kernel).
// LDS: X half-tiles of both buffers in the first 64 KiB ([buf][half]), Z half-tiles in the second, so that one address
// register per fragment reaches every half-tile of its operand through the 16-bit immediate offset.
// The LDS-DMA is issued from inline asm here: hipcc puts s_waitcnt vmcnt(0) in front of every ds_read_b64_tr_b16 that
// follows an LDS-DMA it knows about (it cannot tell that they touch different half-tiles), which would drain the
// two-K-steps-deep prefetch at every phase; an asm statement is invisible to that bookkeeping, and completion is
// counted by hand (wait_vmcnt) exactly as in igemm8_kernel.
__device__ __forceinline__ void dma16_asm(__amdgpu_buffer_rsrc_t rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
#ifdef YFS_IG8_TIMING   // (the section clocks make the compiler keep these two wave-uniform values in vector registers)
        : "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane((int)soff))
#else
        : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
#endif
        : "memory");
}

template <int FX>
__global__ void __launch_bounds__(512, 2) wgrad8_kernel(WgradArgs g) {
    constexpr int XT = FX * 64, XH = FX * 32, XCH = XH / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    IG8_T(0, (int)blockIdx.x);   // (tools/probes/wgrad8_sections.py)
    const int taps = g.KH * g.KW;
    const int inner = g.citiles * taps;
    const int logical = xcd_remap(blockIdx.x, g.cotiles * inner * g.splits);
    const int tci = logical % inner, rest = logical / inner;
    const int cot = rest % g.cotiles, split = rest / g.cotiles;
    const int tap = tci / g.citiles, cit = tci % g.citiles;
    const int kh = tap / g.KW, kw = tap % g.KW;
    const int co0 = cot * 256, ci0 = cit * XT;
    const int kt0 = split * g.ktiles_per_split;
    const int kt1 = kt0 + g.ktiles_per_split < g.ktiles ? kt0 + g.ktiles_per_split : g.ktiles;
    const int nit = kt1 - kt0;

    const __amdgpu_buffer_rsrc_t srx = make_rsrc(g.x, g.xbytes), srz = make_rsrc(g.dz, g.zbytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;   // LDS byte address of the ring (M0 of the LDS-DMA)
    // staging: wave w fills pixels w*8 .. w*8+7 of every half-tile, instruction i the four pixels (w*2+i)*4 + (lane>>4),
    // lane&15 = 16-byte LDS slot, which receives global chunk wg_swz(pixel, slot) (source-side swizzle)
    constexpr unsigned kPixOOB = 0xFFFF0000u;
    unsigned zoff[2], xch[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pixel = (wave * 2 + i) * 4 + (lane >> 4);
        const int cs = wg_swz<16>(pixel, lane & 15);
        zoff[i] = ((unsigned)pixel * (unsigned)g.Cout + (unsigned)(co0 + cs * 8)) * 2u;
        xch[i] = cs < XCH ? (unsigned)cs * 16u : kPixOOB;   // (192-channel tile: slots 12-15 of a row fetch nothing)
    }
    // lane l keeps the source offset of pixel w*8 + (l&7) of k-tile ktg + (l>>3): eight k-tiles per refresh
    unsigned pixcache = 0;
    auto pixel_offsets = [&](int ktg) -> unsigned {
        const int m = (ktg + (lane >> 3)) * 64 + wave * 8 + (lane & 7);
        const unsigned n = fdiv((unsigned)m, g.div_ohw);
        const unsigned rem = (unsigned)m - n * g.div_ohw.d;
        const unsigned oh = fdiv(rem, g.div_ow);
        const unsigned ow = rem - oh * g.div_ow.d;
        const int ih = (int)oh * g.stride - g.pad + kh * g.kh_mul, iw = (int)ow * g.stride - g.pad + kw;
        const bool ok = m < g.M && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
        return ok ? (unsigned)((((int)n * g.H + ih) * g.W + iw) * g.PS) * 2u : kPixOOB;
    };
    // X source offsets of k-tile k for this lane's two LDS-DMA instructions (the same for both half-tiles): fetched from
    // the lanes that computed them with ds_bpermute.  Done one phase before the first use, so that the LDS round trip
    // of the permute is not between a phase's fragment reads and its DMA issue.
    unsigned xo[2];
    auto x_offsets = [&](int k) {   // pixcache must hold k's group of eight
        const int grp = ((k - kt0) & 7) * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // (i * 4 + (lane >> 4)) * 4 = bpermute address of the lane that holds this pixel's offset
            const unsigned pv = (unsigned)__builtin_amdgcn_ds_bpermute((i * 4 + (lane >> 4)) * 4 + grp, (int)pixcache);
            const unsigned sum = pv + xch[i];
            xo[i] = (FX == 4 || xch[i] != kPixOOB) ? sum : kPixOOB;   // (OOB + OOB would wrap)
        }
    };
    auto stage_X = [&](int h, int buf) {
        const unsigned so = (unsigned)(ci0 + h * XH) * 2u;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            dma16_asm(srx, lds0 + (unsigned)((buf * 2 + h) * B8_HALF + (wave * 2 + i) * 1024), xo[i], so);
    };
    auto stage_Z = [&](int h, int buf, int k) {
        const unsigned zk = (unsigned)k * (unsigned)(64 * 2) * (unsigned)g.Cout;   // in the vector offset: rows past M must
#pragma unroll                                                                      // fall off the descriptor's end
        for (int i = 0; i < 2; ++i) {
            const unsigned off = zoff[i] + zk;
            dma16_asm(srz, lds0 + (unsigned)(4 * B8_HALF + (buf * 2 + h) * B8_HALF + (wave * 2 + i) * 1024), off, (unsigned)(h * 256));
        }
    };

    f32x4 acc[2][2][2][FX];   // [ci half][co half][co fragment][ci fragment]
#pragma unroll
    for (int i = 0; i < 8 * FX; ++i) (&acc[0][0][0][0])[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 accdb[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) (&accdb[0][0])[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_db = g.dbpart != nullptr;
    int dbc = kt0 % inner;

    // transposed-read geometry (see wgrad_kernel)
    const int g4 = lane >> 4, li = lane & 15;
    const int tr_pix = 8 * g4 + (li >> 2), tr_ch = 4 * (li & 3);
    typedef bf16x4 __attribute__((address_space(3)))* tr_ptr;
    bf16x8 xf[FX][2], z0f[2][2], z1f[2][2];
    // one address per (fragment, pixel group 0-3 / 4-7); K-half, half-tile and buffer are immediate offsets
    const unsigned char* xa[FX][2];
    const unsigned char* za[2][2];
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) {
#pragma unroll
        for (int b = 0; b < FX; ++b) xa[b][hl] = smem + wg_addr<16>(tr_pix + 4 * hl, wm * (FX * 16) + b * 16 + tr_ch);
#pragma unroll
        for (int a = 0; a < 2; ++a) za[a][hl] = smem + 4 * B8_HALF + wg_addr<16>(tr_pix + 4 * hl, wn * 32 + a * 16 + tr_ch);
    }
    auto read_frag = [&](const unsigned char* lo_addr, const unsigned char* hi_addr, int off) -> bf16x8 {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tr_ptr)(lo_addr + off));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tr_ptr)(hi_addr + off));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto read_X = [&](int hoff) {   // hoff = (buf * 2 + half) * B8_HALF; a K-half is 32 pixel rows of 256 B further on
#pragma unroll
        for (int b = 0; b < FX; ++b)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xf[b][ks] = read_frag(xa[b][0], xa[b][1], hoff + ks * 8192);
    };
    auto read_Z = [&](int hoff, bf16x8 (&zf)[2][2]) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) zf[a][ks] = read_frag(za[a][0], za[a][1], hoff + ks * 8192);
    };
    auto mfma_q = [&](f32x4 (&c)[2][FX], const bf16x8 (&zf)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < FX; ++b) c[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf[a][ks], xf[b][ks], c[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // bias gradient: column sums of dz as MFMAs against ones; the `inner` workgroups that share a dz slab take every
    // inner-th k-tile each and the two wave rows one 32-pixel half each (as wgrad_kernel)
    auto mfma_db = [&](f32x4 (&c)[2], const bf16x8 (&zf)[2][2]) {
        // the all-ones operand is rebuilt here (one K-step in `inner` uses it): four registers not held across the loop
        unsigned one2 = 0x3F803F80u;
        asm volatile("" : "+v"(one2));
        const bf16x8 ones = __builtin_bit_cast(bf16x8, (u32x4){one2, one2, one2, one2});
#pragma unroll
        for (int a = 0; a < 2; ++a)
            c[a] = wm == 0 ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf[a][0], ones, c[a], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf[a][1], ones, c[a], 0, 0, 0);
    };
    auto lgkm0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // two phases per K-step (as igemm8_kernel), 32 MFMAs each:
    //   A: read Z0, X0, Z1 | stage X1 of tile t+1            | MFMA (X0, Z0), (X0, Z1) (+ the bias-gradient MFMAs)
    //   B: read X1         | stage Z0, X0, Z1 of tile t+2, vmcnt | MFMA (X1, Z1), (X1, Z0)
    // every half-tile is restaged one phase after its last read, which the lgkmcnt(0) before the barrier has retired.
    // xo holds the X source offsets of tile t+2 from phase A on: phase B stages X0 of t+2 with them and phase A of the
    // next tile X1 of (its) t+1 = the same tile.
    pixcache = pixel_offsets(kt0);
    x_offsets(kt0);
    stage_Z(0, 0, kt0); stage_X(0, 0); stage_Z(1, 0, kt0); stage_X(1, 0);
    if (nit > 1) {
        x_offsets(kt0 + 1);
        stage_Z(0, 1, kt0 + 1); stage_X(0, 1); stage_Z(1, 1, kt0 + 1);   // (X1 of tile 1: phase A of tile 0)
        wait_vmcnt<6>();
    } else {
        wait_vmcnt<0>();
    }
    bar();
    if (wm == 1) bar();
    IG8_T(1, (int)blockIdx.x);

    auto tile = [&](int t, auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        const bool dbmode = do_db && dbc == tci;
        dbc = dbc + 1 == inner ? 0 : dbc + 1;
        const int k1 = kt0 + t + 1, k2 = kt0 + t + 2;
        // phase A
        read_Z((BUF * 2) * B8_HALF, z0f);
        read_X((BUF * 2) * B8_HALF);
        read_Z((BUF * 2 + 1) * B8_HALF, z1f);
        if (k1 < kt1) stage_X(1, BUF ^ 1);
        if (k2 < kt1) {
            if (((k2 - kt0) & 7) == 0) pixcache = pixel_offsets(k2);
            x_offsets(k2);
        }
        lgkm0();
        bar();
        mfma_q(acc[0][0], z0f);
        mfma_q(acc[0][1], z1f);
        if (dbmode) {
            mfma_db(accdb[0], z0f);
            mfma_db(accdb[1], z1f);
        }
        bar();
        // phase B
        read_X((BUF * 2 + 1) * B8_HALF);
        if (k2 < kt1) {
            stage_Z(0, BUF, k2);
            stage_X(0, BUF);
            stage_Z(1, BUF, k2);
            wait_vmcnt<6>();
        } else {
            wait_vmcnt<0>();
        }
        lgkm0();
        bar();
        mfma_q(acc[1][1], z1f);
        mfma_q(acc[1][0], z0f);
        bar();
    };
    int t = 0;
    for (; t + 1 < nit; t += 2) {
        tile(t, std::integral_constant<int, 0>());
        tile(t + 1, std::integral_constant<int, 1>());
    }
    if (t < nit) tile(t, std::integral_constant<int, 0>());
    if (wm == 0) bar();
    IG8_T(2, (int)blockIdx.x);

    // ---- epilogue: transpose through LDS, four slabs of 64 co rows; rows leave as whole ci-contiguous runs --------------
    // acc[xh][zh][a][b][r] = dw[co0 + zh*128 + wn*32 + a*16 + (lane>>4)*4 + r][tap][ci0 + xh*XH + wm*FX*16 + b*16 + (lane&15)]
    constexpr int ESTRIDE = XT * 4 + 16;   // bytes per staged co row (the +16 keeps the 4-row groups of a store on distinct banks)
    const long long kw_tot = (long long)taps * g.Cin;
    float* pbase = g.partial + (long long)split * g.Cout * kw_tot;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        if ((wn >> 1) == (pass & 1)) {
#pragma unroll
            for (int xh = 0; xh < 2; ++xh)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < FX; ++b) {
                        const int row = (wn & 1) * 32 + a * 16 + g4 * 4;
                        const int ci = xh * XH + wm * (FX * 16) + b * 16 + li;
                        const f32x4 v = acc[xh][pass >> 1][a][b];
#pragma unroll
                        for (int r = 0; r < 4; ++r) *reinterpret_cast<float*>(smem + (row + r) * ESTRIDE + ci * 4) = v[r];
                    }
        }
        __syncthreads();
        constexpr int C4 = XT / 4;   // float4 per row
        for (int item = tid; item < 64 * C4; item += 512) {
            const int row = item / C4, c4 = item - row * C4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * ESTRIDE + c4 * 16);
            const int co = co0 + pass * 64 + row;
            *reinterpret_cast<f32x4*>(pbase + (long long)co * kw_tot + (long long)tap * g.Cin + ci0 + c4 * 4) = v;
        }
        if (pass < 3) __syncthreads();
    }
    if (do_db && li == 0) {   // every column of accdb holds the same sums: column 0 (lanes 0, 16, 32, 48) writes
        float* drow = g.dbpart + ((long long)(split * inner + tci) * 2 + wm) * g.Cout;
#pragma unroll
        for (int zh = 0; zh < 2; ++zh)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) drow[co0 + zh * 128 + wn * 32 + a * 16 + g4 * 4 + r] = accdb[zh][a][r];
    }
    IG8_T(3, (int)blockIdx.x);
#ifdef YFS_IG8_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IG8_T(4, (int)blockIdx.x);
#endif
}

// dw[i] = (accumulate ? dw[i] : 0) + sum_s partial[s][i].  G groups of threads share one float4 column: group r
// sums slabs r, r+G, ... and the G sums are combined through LDS in fixed order (deterministic).  Small outputs
// with many slabs (1x1 layers: 32K floats x 224 slabs) would otherwise run on a few dozen serial threads.
// db[c] = (accumulate ? db[c] : 0) + sum of the wgrad kernel's partial column-sum rows.  One block per 16
// channels; 16 row groups (rows r, r+16, ...) per channel are summed independently and combined in fixed order.
__device__ __forceinline__ void colsum_reduce_block(const float* __restrict__ partial, int nsplit, int C, float* __restrict__ db,
                                                    int accumulate, int blk, float (*red)[16 + 1]) {
    const int cl = threadIdx.x & 15, r = threadIdx.x >> 4;
    const int c = blk * 16 + cl;
    float s = 0.f;
    if (c < C) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // four loads in flight; combined in fixed order
        int k = r;
        for (; k + 48 < nsplit; k += 64) {
            s0 += partial[(long long)k * C + c];
            s1 += partial[(long long)(k + 16) * C + c];
            s2 += partial[(long long)(k + 32) * C + c];
            s3 += partial[(long long)(k + 48) * C + c];
        }
        for (; k < nsplit; k += 16) s0 += partial[(long long)k * C + c];
        s = (s0 + s1) + (s2 + s3);
    }
    red[r][cl] = s;
    __syncthreads();
    if (r == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        db[c] = accumulate ? db[c] + t : t;
    }
}

// dw[i] = (accumulate ? dw[i] : 0) + sum_s partial[s][i].  G groups of threads share one float4 column: group r
// sums slabs r, r+G, ... and the G sums are combined through LDS in fixed order (deterministic).  Small outputs
// with many slabs (1x1 layers: 32K floats x 224 slabs) would otherwise run on a few dozen serial threads.
// The blocks past `nblk_main` reduce the bias-gradient rows (colsum_reduce_block) in the same launch: both reductions
// are a few microseconds of work and a launch each was as long as the work itself, 24 times per step.
template <int G>
__global__ void __launch_bounds__(kThreads) splitk_reduce_kernel(const float4* __restrict__ partial, int splits,
                                                                 long long n4, float4* __restrict__ dw,
                                                                 int accumulate, int nblk_main,
                                                                 const float* __restrict__ dbpart, int db_splits, int C,
                                                                 float* __restrict__ db) {
    constexpr int COLS = kThreads / G;
    __shared__ float4 red[G][COLS];
    __shared__ float cred[16][16 + 1];
    if ((int)blockIdx.x >= nblk_main) {
        colsum_reduce_block(dbpart, db_splits, C, db, accumulate, (int)blockIdx.x - nblk_main, cred);
        return;
    }
    const int cl = threadIdx.x % COLS, r = threadIdx.x / COLS;
    for (long long i0 = (long long)blockIdx.x * COLS; i0 < n4; i0 += (long long)nblk_main * COLS) {
        const long long i = i0 + cl;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n4)
            for (int k = r; k < splits; k += G) {
                const float4 v = partial[(long long)k * n4 + i];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        if (G > 1) {
            red[r][cl] = s;
            __syncthreads();
            if (r == 0) {
#pragma unroll
                for (int k = 1; k < G; ++k) {
                    const float4 v = red[k][cl];
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
            }
            __syncthreads();
        }
        if (r == 0 && i < n4) {
            if (accumulate) {
                const float4 o = dw[i];
                s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
            }
            dw[i] = s;
        }
    }
}

struct WgradPlan {
    int bx, bz, cotiles, citiles, splits, ktiles, ktiles_per_split, db_splits;
    bool use8;   // wgrad8_kernel (256-wide co tile, 256- or 192-wide ci tile, one workgroup per CU)
    size_t partial_bytes, db_bytes;
};

// taps = KH*KW.  Tile widths: 128 where the channel count allows, 192 for the two 192-channel tensors of the
// backbone (feat.2's input, feat.0's output), else 64.  Only the instantiated (bx, bz) pairs are produced.
WgradPlan plan_wgrad(int64_t M, int Cin, int Cout, int taps, bool wide192 = false, bool plain = true) {
    WgradPlan p;
    p.use8 = false;
    const int force_tile8 = yfs_knob(YFS_KNOB_WGRAD_TILE);
    if (plain && !wide192 && Cout % 256 == 0 && (Cin % 256 == 0 || Cin == 192) &&
        (force_tile8 == 8 || (force_tile8 == 0 && taps > 1))) {
        // 8-phase kernel: one workgroup per CU, so the grid runs in rounds of 256; same cost model as below
        p.use8 = true;
        p.bx = Cin % 256 == 0 ? 256 : 192;
        p.bz = 256;
        p.citiles = Cin / p.bx;
        p.cotiles = Cout / 256;
        p.ktiles = (int)yfs_cdiv(M, 64);
        const int tiles = p.citiles * p.cotiles * taps;
        const int max_splits = p.ktiles / 8 > 0 ? p.ktiles / 8 : 1;
        // (measured against wgrad_kernel: L7 561 -> 399 us, L10 147 -> 115, L18 587 -> 399, L21 164 -> 119, L24 312 -> 231,
        //  L25 / L26 112 -> 86, L2 (192-channel ci tile) 830 -> 751; the 1x1 layers are HBM-bound and stay on wgrad_kernel
        //  unless forced)
        const double t_tile = p.bx == 256 ? 1.38 : 1.25;                      // us per 64-pixel k-tile
        const double t_slab = (double)Cout * taps * Cin * 8.0 / 4.0e6;        // us per partial copy (write + read at 4 TB/s)
        double best = 1e30;
        int best_per = p.ktiles;
        for (int sp = 1; sp <= max_splits && sp <= 1024; ++sp) {
            const int per = (int)yfs_cdiv(p.ktiles, sp);
            const int real = (int)yfs_cdiv(p.ktiles, per);
            const long long rounds = yfs_cdiv((long long)tiles * real, 256);
            const double t = (double)rounds * (per * t_tile + 8.0) + (real > 1 ? real * t_slab : 0.0);
            if (t < best) { best = t; best_per = per; }
        }
        const int fs = yfs_knob(YFS_KNOB_WGRAD_SPLITS);
        if (fs > 0) best_per = (int)yfs_cdiv(p.ktiles, fs < p.ktiles ? fs : p.ktiles);
        p.ktiles_per_split = best_per;
        p.splits = (int)yfs_cdiv(p.ktiles, p.ktiles_per_split);
        p.partial_bytes = (size_t)p.splits * Cout * taps * Cin * sizeof(float);
        p.db_splits = p.splits * p.citiles * taps * 2;
        p.db_bytes = (size_t)p.db_splits * Cout * sizeof(float);
        return p;
    }
    p.bx = (Cin % 128 == 0) ? 128 : ((Cin % 192 == 0 && Cout % 128 == 0) ? 192 : 64);
    p.bz = (Cout % 128 == 0) ? 128 : ((Cout % 192 == 0 && (p.bx == 64 || wide192)) ? 192 : 64);
    const int force_tile = yfs_knob(YFS_KNOB_WGRAD_TILE);   // sweeps
    if (force_tile == 64 && !wide192) p.bx = p.bz = 64;
    if (force_tile == 12864 && !wide192 && p.bx == 128 && p.bz == 128) p.bz = 64;
    p.citiles = Cin / p.bx;
    p.cotiles = Cout / p.bz;
    p.ktiles = (int)yfs_cdiv(M, 64);
    const int tiles = p.citiles * p.cotiles * taps;
    // Split-K factor: the grid runs in rounds of 512 workgroups (2 per CU), so a block count just above a multiple
    // of 512 wastes most of a round (L7: 72 tiles x 15 splits = 1080 blocks ran at 70 %).  Pick the factor that
    // minimises a simple cost model: rounds x (k-tiles per split x time per k-tile + fixed prologue/epilogue)
    // + the fp32 slab traffic of writing and re-reading `splits` partial copies.  At least 8 k-tiles per split.
    const int max_splits = p.ktiles / 8 > 0 ? p.ktiles / 8 : 1;
    const double t_tile = 1.08 * (double)(p.bx * p.bz) / (128.0 * 128.0);            // us per 64-pixel k-tile
    const double t_slab = (double)Cout * taps * Cin * 8.0 / 4.0e6;                    // us per partial copy (w + r at 4 TB/s)
    double best = 1e30;
    int best_per = p.ktiles;
    for (int sp = 1; sp <= max_splits && sp <= 1024; ++sp) {
        const int per = (int)yfs_cdiv(p.ktiles, sp);
        const int real = (int)yfs_cdiv(p.ktiles, per);
        const long long rounds = yfs_cdiv((long long)tiles * real, 512);
        const double t = (double)rounds * (per * t_tile + 3.0) + real * t_slab;
        if (t < best) { best = t; best_per = per; }
    }
    const int force_splits = yfs_knob(YFS_KNOB_WGRAD_SPLITS);   // sweeps
    if (force_splits > 0) best_per = (int)yfs_cdiv(p.ktiles, force_splits < p.ktiles ? force_splits : p.ktiles);
    p.ktiles_per_split = best_per;
    p.splits = (int)yfs_cdiv(p.ktiles, p.ktiles_per_split);
    p.partial_bytes = (size_t)p.splits * Cout * taps * Cin * sizeof(float);
    p.db_splits = p.splits * p.citiles * taps * 2;   // one partial row per (split, tap, ci tile, wave row)
    p.db_bytes = (size_t)p.db_splits * Cout * sizeof(float);
    return p;
}

template <int FX>
int launch_wgrad8(WgradArgs& g, const WgradPlan& p, hipStream_t s) {
    auto kern = wgrad8_kernel<FX>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, B8_LDS);
    YFS_REQUIRE(e == hipSuccess, YFS_ERR_LAUNCH, "yfs_conv2d_wgrad_bf16: cannot reserve %d bytes of LDS", B8_LDS);
    const int nblk = p.cotiles * p.citiles * g.KH * g.KW * p.splits;
    kern<<<dim3((unsigned)nblk), dim3(512), B8_LDS, s>>>(g);
    YFS_CHECK_LAUNCH("wgrad8_kernel");
    yfs_note_kernel("wgrad8_kernel<%d,256,splits%d>", FX * 64, p.splits);
    return YFS_OK;
}

template <int BX, int BZ, bool POOLZ = false, bool SGD = false>
int launch_wgrad(WgradArgs& g, const WgradPlan& p, hipStream_t s) {
    const size_t lds = 2 * (size_t)(BX + BZ) * 128;
    auto kern = wgrad_kernel<BX, BZ, POOLZ, SGD>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        YFS_REQUIRE(e == hipSuccess, YFS_ERR_LAUNCH, "yfs_conv2d_wgrad_bf16: cannot reserve %zu bytes of LDS", lds);
    }
    const int nblk = p.cotiles * p.citiles * g.KH * g.KW * p.splits;
    kern<<<dim3((unsigned)nblk), dim3(kThreads), lds, s>>>(g);
    YFS_CHECK_LAUNCH("wgrad_kernel");
    yfs_note_kernel("wgrad_kernel<%d,%d%s%s,splits%d>", BX, BZ, POOLZ ? ",poolz" : "", SGD ? ",sgd" : "", p.splits);
    return YFS_OK;
}

}  // namespace

YFS_API size_t yfs_conv2d_wgrad_workspace_bytes(int64_t N, int H, int W, int Cin, int Cout, int K, int stride) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || K <= 0 || stride <= 0) return 0;
    const int pad = K / 2;
    const int64_t M = N * ((H + 2 * pad - K) / stride + 1) * ((W + 2 * pad - K) / stride + 1);
    WgradPlan p = plan_wgrad(M, Cin, Cout, K * K);
    return p.partial_bytes + p.db_bytes;
}

namespace {
// shared by the regular and the stem entry points.  x: pixels PS elements apart, Cin reduced channels per tap.
int wgrad_common(const char* fn, const void* x, const void* dz, float* dw, float* db, int64_t N, int H, int W, int PS, int Cin,
                 int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int accumulate, void* workspace,
                 size_t workspace_bytes, hipStream_t s, const void* pool_idx = nullptr, const void* y_pooled = nullptr,
                 float slope = 0.0f, unsigned xgroup_bytes = 128, int kh_mul = 1) {
    const int64_t M = N * OH * OW;
    const int taps = KH * KW;
    const bool plain = pool_idx == nullptr && xgroup_bytes == 128 && kh_mul == 1 && PS == Cin;
    WgradPlan p = plan_wgrad(M, Cin, Cout, taps, pool_idx != nullptr, plain);
    YFS_REQUIRE(workspace && workspace_bytes >= p.partial_bytes + p.db_bytes, YFS_ERR_WORKSPACE,
                "%s: workspace needs %zu bytes, got %zu", fn, p.partial_bytes + p.db_bytes, workspace_bytes);
    YFS_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, YFS_ERR_WORKSPACE, "%s: workspace must be 16-byte aligned", fn);
    WgradArgs g = {};
    g.dz = (const bf16_t*)dz; g.x = (const bf16_t*)x; g.partial = (float*)workspace;
    // One split and nothing to add to: the single partial slab IS the gradient (same [Cout][taps][Cin] layout), so the
    // kernel writes it straight into dw and the reduction pass below only sums the bias-gradient rows (feat.24-27:
    // 37.7 MB written once instead of written, read and written again).
    const bool direct = p.splits == 1 && !accumulate;
    if (direct) g.partial = dw;
    g.dbpart = db ? (float*)((unsigned char*)workspace + p.partial_bytes) : nullptr;
    g.H = H; g.W = W; g.Cin = Cin; g.OH = OH; g.OW = OW; g.Cout = Cout; g.KH = KH; g.KW = KW; g.PS = PS; g.stride = stride; g.pad = pad;
    g.M = (int)M;
    g.xgroup_bytes = xgroup_bytes; g.kh_mul = kh_mul;
    g.xbytes = (unsigned)(N * H * W * PS * 2); g.zbytes = (unsigned)(M * Cout * 2);
    g.cotiles = p.cotiles; g.citiles = p.citiles; g.splits = p.splits; g.ktiles = p.ktiles;
    g.ktiles_per_split = p.ktiles_per_split;
    g.div_ohw = make_fastdiv((unsigned)(OH * OW)); g.div_ow = make_fastdiv((unsigned)OW);
    int rc;
    if (p.use8) {
        rc = p.bx == 256 ? launch_wgrad8<4>(g, p, s) : launch_wgrad8<3>(g, p, s);
    } else
    if (pool_idx) {   // dz given as (pooled gradient, argmax record, pooled activation): see wgrad_kernel POOLZ
        YFS_REQUIRE(p.bx == 128 && OH % 2 == 0 && OW % 2 == 0, YFS_ERR_UNSUPPORTED,
                    "%s: the pooled-gradient form needs a 128-wide reduction tile and an even %dx%d output", fn, OH, OW);
        g.zidx = (const unsigned short*)pool_idx; g.zy = (const bf16_t*)y_pooled; g.slope = slope;
        g.zbytes = (unsigned)(M / 4 * Cout * 2); g.zidx_bytes = (unsigned)(M / 4 * (Cout / 8) * 2);
        g.div_ohw = make_fastdiv((unsigned)((OH / 2) * (OW / 2))); g.div_ow = make_fastdiv((unsigned)(OW / 2));
        if (p.bz == 192) rc = launch_wgrad<128, 192, true>(g, p, s);
        else if (p.bz == 128) rc = launch_wgrad<128, 128, true>(g, p, s);
        else rc = launch_wgrad<128, 64, true>(g, p, s);
    } else
    if (p.bx == 128 && p.bz == 128) rc = launch_wgrad<128, 128>(g, p, s);
    else if (p.bx == 128 && p.bz == 64) rc = launch_wgrad<128, 64>(g, p, s);
    else if (p.bx == 64 && p.bz == 128) rc = launch_wgrad<64, 128>(g, p, s);
    else if (p.bx == 192 && p.bz == 128) rc = launch_wgrad<192, 128>(g, p, s);
    else if (p.bx == 64 && p.bz == 192) rc = launch_wgrad<64, 192>(g, p, s);
    else rc = launch_wgrad<64, 64>(g, p, s);
    if (rc) return rc;
    const long long n4 = (long long)Cout * taps * Cin / 4;
    // enough thread groups per column that the reduce has >= ~64K threads in flight even for small outputs
    const int G = (p.splits >= 32 && n4 * 4 < 65536 * 4) ? 16 : ((p.splits >= 8 && n4 < 65536 * 4) ? 4 : 1);
    long long rblk = yfs_cdiv(n4, kThreads / G);
    if (rblk > 4096) rblk = 4096;
    if (direct) rblk = 0;
    const float4* part4 = (const float4*)workspace;
    const float* dbp = db ? (const float*)((unsigned char*)workspace + p.partial_bytes) : nullptr;
    const int cblk = db ? (int)yfs_cdiv(Cout, 16) : 0;
    if (rblk + cblk == 0) return YFS_OK;
    const dim3 grid((unsigned)(rblk + cblk));
    if (G == 16)
        splitk_reduce_kernel<16><<<grid, dim3(kThreads), 0, s>>>(part4, p.splits, n4, (float4*)dw, accumulate, (int)rblk, dbp,
                                                                 p.db_splits, Cout, db);
    else if (G == 4)
        splitk_reduce_kernel<4><<<grid, dim3(kThreads), 0, s>>>(part4, p.splits, n4, (float4*)dw, accumulate, (int)rblk, dbp,
                                                                p.db_splits, Cout, db);
    else
        splitk_reduce_kernel<1><<<grid, dim3(kThreads), 0, s>>>(part4, p.splits, n4, (float4*)dw, accumulate, (int)rblk, dbp,
                                                                p.db_splits, Cout, db);
    YFS_CHECK_LAUNCH("splitk_reduce_kernel");
    return YFS_OK;
}
}  // namespace

YFS_API int yfs_conv2d_wgrad_bf16(const void* x, const void* dz, float* dw, float* db, int64_t N, int H, int W, int Cin,
                                  int Cout, int K, int stride, int accumulate, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    YFS_REQUIRE(x && dz && dw, YFS_ERR_INVALID_ARG, "yfs_conv2d_wgrad_bf16: null pointer");
    int rc = check_conv_shape("yfs_conv2d_wgrad_bf16", N, H, W, Cin, Cout, K, stride);
    if (rc) return rc;
    const int pad = K / 2;
    const int OH = (H + 2 * pad - K) / stride + 1, OW = (W + 2 * pad - K) / stride + 1;
    return wgrad_common("yfs_conv2d_wgrad_bf16", x, dz, dw, db, N, H, W, Cin, Cin, OH, OW, Cout, K, K, stride, pad, accumulate,
                        workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- Linear layer: weight gradient fused with the optimiser step ---------------------------------------------------
// dW[o][i] = sum_b dy[b][o] * x[b][i] is the 1x1 "convolution" wgrad over B pixels; with the SGD epilogue the
// 4 B/parameter gradient is never written, summed or re-read (fc.1: 822 MB each).
YFS_API int yfs_linear_wgrad_sgd_bf16(const void* x, const void* dy, int64_t B, int In, int Out, float* p, float* m, void* p_bf16,
                                      const float* sumsq, float max_norm, float lr, float momentum, float weight_decay,
                                      int first_step, float grad_scale, void* stream) {
    YFS_REQUIRE(x && dy && p && m, YFS_ERR_INVALID_ARG, "yfs_linear_wgrad_sgd_bf16: null pointer");
    YFS_REQUIRE(B > 0 && In > 0 && Out > 0 && In % 128 == 0 && Out % 128 == 0, YFS_ERR_UNSUPPORTED,
                "yfs_linear_wgrad_sgd_bf16: In=%d and Out=%d must be positive multiples of 128 (B=%lld)", In, Out, (long long)B);
    YFS_REQUIRE(B * In * 2 < 0xFFF00000ll && B * Out * 2 < 0xFFF00000ll, YFS_ERR_UNSUPPORTED,
                "yfs_linear_wgrad_sgd_bf16: an operand exceeds the 4 GiB range of one buffer descriptor");
    WgradArgs g = {};
    g.dz = (const bf16_t*)dy; g.x = (const bf16_t*)x;
    g.H = 1; g.W = 1; g.Cin = In; g.OH = 1; g.OW = 1; g.Cout = Out; g.KH = 1; g.KW = 1; g.PS = In; g.stride = 1; g.pad = 0;
    g.M = (int)B;
    g.xgroup_bytes = 128; g.kh_mul = 1;
    g.xbytes = (unsigned)(B * In * 2); g.zbytes = (unsigned)(B * Out * 2);
    WgradPlan pl = {};
    pl.bx = pl.bz = 128; pl.citiles = In / 128; pl.cotiles = Out / 128; pl.ktiles = (int)yfs_cdiv(B, 64);
    pl.splits = 1; pl.ktiles_per_split = pl.ktiles;
    g.cotiles = pl.cotiles; g.citiles = pl.citiles; g.splits = 1; g.ktiles = pl.ktiles; g.ktiles_per_split = pl.ktiles;
    g.div_ohw = make_fastdiv(1u); g.div_ow = make_fastdiv(1u);
    g.sgd_p = p; g.sgd_m = m; g.sgd_p16 = (bf16_t*)p_bf16; g.sgd_sumsq = sumsq; g.sgd_max_norm = max_norm; g.sgd_lr = lr;
    g.sgd_momentum = momentum; g.sgd_wd = weight_decay; g.sgd_gscale = grad_scale; g.sgd_first = first_step;
    return launch_wgrad<128, 128, false, true>(g, pl, (hipStream_t)stream);
}

// ---- stem: Conv2D(3, Cout, 7, s=2) as a 4x4 stride-1 convolution over the 2x2 space-to-depth image ------------
// xs [N,Hs,Ws,16] bf16 (12 real channels + 4 zero), Hs = Ws = (H+6+1)/2 rounded up to even; one kernel row of
// the 4x4 window is 4 pixels x 16 channels = 64 contiguous elements = exactly one 128-byte K-chunk, so the
// implicit GEMM runs with KH=4, KW=1, C=64, pixel stride 16 and never materialises an im2col matrix.
namespace {
constexpr int kStemC = 16, kStemKH = 4;
static int stem_ps() { return kStemC; }
int stem_check(const char* fn, int64_t N, int Hs, int Ws, int OH, int OW, int Cout) {
    YFS_REQUIRE(N > 0 && Hs > 0 && Ws > 0 && OH > 0 && OW > 0 && Cout > 0, YFS_ERR_INVALID_ARG, "%s: non-positive dimension", fn);
    YFS_REQUIRE(OH + 3 <= Hs && OW + 4 <= Ws, YFS_ERR_INVALID_ARG,
                "%s: space-to-depth image %dx%d too small for a %dx%d output (needs OH+3 rows, OW+4 columns)", fn, Hs, Ws, OH, OW);
    YFS_REQUIRE(Cout % 64 == 0, YFS_ERR_UNSUPPORTED, "%s: Cout=%d must be a multiple of 64", fn, Cout);
    YFS_REQUIRE(N * OH * OW * Cout * 2 < 0xFFF00000ll && N * Hs * Ws * kStemC * 2 < 0xFFF00000ll, YFS_ERR_UNSUPPORTED,
                "%s: tensor exceeds the 4 GiB range of one buffer descriptor", fn);
    return YFS_OK;
}

template <typename TIn>
__global__ void __launch_bounds__(kThreads) stem_s2d_kernel(const TIn* __restrict__ x, long long sn, long long sc, long long sh,
                                                            long long sw, bf16_t* __restrict__ out, int H, int W, int Hs, int Ws,
                                                            long long total) {
    for (long long id = (long long)blockIdx.x * kThreads + threadIdx.x; id < total; id += (long long)gridDim.x * kThreads) {
        const int j = (int)(id % Ws);
        const long long t = id / Ws;
        const int i = (int)(t % Hs);
        const long long n = t / Hs;
        bf16x8 lo = {0, 0, 0, 0, 0, 0, 0, 0}, hi = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {          // q = dy*2 + dx
            const int ih = 2 * i + (q >> 1) - 3, iw = 2 * j + (q & 1) - 3;
            if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
                const TIn* p = x + n * sn + ih * sh + iw * sw;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const bf16_t v = (bf16_t)(float)p[c * sc];
                    const int e = q * 3 + c;
                    if (e < 8) lo[e] = v; else hi[e - 8] = v;
                }
            }
        }
        bf16x8* o = reinterpret_cast<bf16x8*>(out + id * kStemC);
        o[0] = lo;
        o[1] = hi;
    }
}

// ---- stem forward + 2x2 max pool, weight-stationary -------------------------------------------------------------
// The stem GEMM has K = 256: four K-steps per tile, so a tiled kernel spends its time filling its pipeline and in the
// epilogue (igemm_kernel<128,192,..,EPI_POOL>: 0.30 PFLOP/s).  This kernel turns the loop nest around:
//   * the weights never move: wave (cg, rp) keeps the weight fragments of its CB*16 output channels for the whole K (CB x 8
//     B fragments = 96 registers at Cout = 192) for the life of the kernel;
//   * the input is staged as RAW rows of the space-to-depth image, not as per-pixel windows: a strip of 4 output rows
//     needs 7 input rows = one contiguous piece of xs (7 * Ws * 32 B, 50 KiB at 448^2) which the 8 waves copy with
//     plain linear LDS-DMA, double-buffered, one barrier per strip.  The 4 x 16 channels of a kernel row are 128
//     contiguous bytes starting at the pixel, and adjacent pixels are 32 B apart, so the B fragment of 16 adjacent
//     output pixels is a ds_read_b128 at lane stride 32 B straight from the raw row (conflict-free: the 16-lane groups
//     of ds_read_b128 cover 16 distinct 16-B slots of the 256-B bank row) - the overlapping windows are never copied;
//   * a wave's tile is 2 output rows x 32 columns x its channels, with the PIXELS as the A operand: a lane then holds one
//     channel for 4 adjacent columns, the two rows of a pool window are two accumulators of that lane, so the whole
//     window is lane-local: bias + LeakyReLU + rounding + first-maximum run in registers with no cross-lane traffic,
//     and only the pooled tensor and the 2-bit positions are stored.
// Two waves share a SIMD and free-run inside a strip, so one wave's epilogue VALU overlaps the other's MFMAs.
// Same K order as the tiled kernel (kernel row by kernel row, 32 elements per MFMA): bit-identical results.
struct StemPoolArgs {
    const bf16_t* xs;           // [N,Hs,Ws,16]
    const bf16_t* w2;           // [Cout][4][64]
    const float* bias;          // [Cout]
    bf16_t* out;                // [N,OH/2,OW/2,Cout]
    unsigned short* pool_idx;   // [N*OH/2*OW/2][Cout/8]
    int N, Hs, Ws, OH, OW, Cout;
    unsigned xbytes, obytes, ibytes;
    float slope;
    int nstrips;                // N * OH/4
    int pieces;                 // 1-KiB LDS-DMA pieces per strip
    int buf_bytes;              // pieces * 1024
};

#ifdef YFS_STEM_TIMING
__device__ long long yfs_stem_dbg[64];
#define STEM_T(v) const long long v = clock64()
#else
#define STEM_T(v) do { } while (0)
#endif
__device__ __forceinline__ float hw_maxf(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// LMAX: 0 <= slope <= 1, so LeakyReLU(t) = max(t, t * slope) (one instruction less per value than compare + select;
// the hardware max returns the non-NaN operand and NaN for two NaNs, like the select form does for t = NaN)
constexpr int kStemOutRow = 96, kStemIdxOff = 16 * kStemOutRow, kStemPatch = 2048;   // per-wave LDS patch (CB = 3)

template <int CB, bool LMAX>
__global__ void __launch_bounds__(512, 1) stem_pool_kernel(StemPoolArgs g) {
    static_assert(CB == 3, "the output patch layout assumes 48 channels per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 3, rp = wave >> 2;           // channel group, row pair of the strip
    const int frow = lane & 15, fk = lane >> 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(g.xs, g.xbytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;

    bf16x8 wf[CB][8];
    float bvl[CB];
#pragma unroll
    for (int blk = 0; blk < CB; ++blk) {
        const int ch = (cg * CB + blk) * 16;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
            wf[blk][kk] = *reinterpret_cast<const bf16x8*>(g.w2 + (size_t)(ch + frow) * 256 + kk * 32 + fk * 8);
        bvl[blk] = g.bias[ch + frow];
    }

    const int rowbytes = g.Ws * 32;
    const int spi = g.OH >> 2, PH = g.OH >> 1, PW = g.OW >> 1, ncb = g.OW >> 5, c8 = g.Cout >> 3;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(g.out, g.obytes), ri = make_rsrc(g.pool_idx, g.ibytes);
    // lane (frow, fk) owns channel frow of a block for the pooled pixels fk*2, fk*2+1 of the 8 in a 16-column block
    // the wave's LDS patch behind the two strip buffers: pooled bf16 tile [16 pixels][CB*16 channels] (96-byte rows at
    // CB = 3, i.e. 16-byte piece c = pixel * 6 + column lies at c * 16), then the position words, one 16-byte row per pixel
    unsigned char* const ws = smem + 2 * g.buf_bytes + wave * kStemPatch;
    unsigned char* const wv = ws + (fk * 2) * kStemOutRow + frow * 2;
    unsigned char* const wi = ws + kStemIdxOff + (fk * 2) * 16 + (frow >> 3) * 2;
    // store offsets of the pieces a lane carries out (constants of the lane; the tile's position is a scalar offset;
    // lanes with nothing to store carry an out-of-range offset)
    const unsigned ovoff0 = (unsigned)((lane / 6) * g.Cout) * 2u + (unsigned)(lane % 6) * 16u;
    const unsigned ovoff1 = lane < 32 ? (unsigned)(((64 + lane) / 6) * g.Cout) * 2u + (unsigned)((64 + lane) % 6) * 16u : 0xFFFFFF00u;
    const unsigned ivoff = lane < 16 ? (unsigned)(lane * c8) * 2u : 0xFFFFFF00u;
    auto issue = [&](int s, int buf) {
        const int n = s / spi, oh0 = (s - n * spi) * 4;
        const unsigned src = (unsigned)((n * g.Hs + oh0) * g.Ws) * 32u + (unsigned)lane * 16u;
        for (int p = wave; p < g.pieces; p += 8)
            dma16_asm(rx, lds0 + (unsigned)(buf * g.buf_bytes + p * 1024), src + (unsigned)p * 1024u, 0u);
    };

    int s = blockIdx.x, buf = 0;
    if (s < g.nstrips) issue(s, 0);
    for (; s < g.nstrips; s += gridDim.x) {
#ifdef YFS_STEM_TIMING
        const long long tb0 = clock64();
#endif
        wait_vmcnt<0>();                    // this wave's pieces of strip s have landed (and its stores have left)
        __builtin_amdgcn_s_barrier();
#ifdef YFS_STEM_TIMING
        if (blockIdx.x == 7 && lane == 0) yfs_stem_dbg[wave * 4 + 3] += clock64() - tb0;
#endif       // ... everyone's have, and everyone is done reading the other buffer
        if (s + (int)gridDim.x < g.nstrips) issue(s + gridDim.x, buf ^ 1);
        const int n = s / spi, oh0 = (s - n * spi) * 4;
        const long long qrow = ((long long)n * PH + (oh0 >> 1) + rp) * PW;
        const unsigned char* sb = smem + buf * g.buf_bytes + (2 * rp) * rowbytes + frow * 32 + fk * 16;
#pragma unroll 1
        for (int cb = 0; cb < ncb; ++cb) {
            f32x4 acc[2][2][CB];
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int blk = 0; blk < CB; ++blk) acc[dy][hf][blk] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const unsigned char* tb = sb + cb * 1024;
            // the two waves of a SIMD share its issue slots by priority, then age: alternate the priority tile by tile so
            // that neither half of the workgroup is the permanent loser (it would keep the other half waiting at the
            // strip barrier: 10 K of 40 K cycles per strip, tools/probes/stem_phases.py)
            if ((cb ^ rp) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            STEM_T(t0);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int kh = kk >> 1, h = kk & 1;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const bf16x8 pf = *reinterpret_cast<const bf16x8*>(tb + (dy + kh) * rowbytes + hf * 512 + h * 64);
#pragma unroll
                        for (int blk = 0; blk < CB; ++blk)
                            acc[dy][hf][blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, wf[blk][kk], acc[dy][hf][blk], 0, 0, 0);
                    }
            }
            STEM_T(t1);
            // epilogue.  MFMA roles: A = pixels, B = weights, so lane (frow, fk) holds channel frow of the block for the
            // pixel columns fk*4 .. fk*4+3 of the 16-column block: both columns of a pool window are registers of one
            // lane (r = 0,1 and r = 2,3) and both rows are the two accumulators - the whole window is lane-local.
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                for (int blk = 0; blk < CB; ++blk) {
                    // stage by stage over the 8 values of the block (not value by value): with two waves per SIMD nothing
                    // else hides the latency of a dependent VALU chain
                    float a[2][4], t[2][4], u[2][4];
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int r = 0; r < 4; ++r) t[dy][r] = acc[dy][hf][blk][r] + bvl[blk];
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int r = 0; r < 4; ++r) u[dy][r] = t[dy][r] * g.slope;
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int r = 0; r < 4; ++r) t[dy][r] = LMAX ? hw_maxf(t[dy][r], u[dy][r]) : (t[dy][r] > 0.0f ? t[dy][r] : u[dy][r]);
                    unsigned pk[2][2];
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {       // one v_cvt_pk_bf16_f32 per pair
                            const f32x2 pr = {t[dy][r], t[dy][r + 1]};
                            pk[dy][r >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2));
                        }
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {
                            a[dy][r] = __builtin_bit_cast(float, pk[dy][r >> 1] << 16);
                            a[dy][r + 1] = __builtin_bit_cast(float, pk[dy][r >> 1] & 0xFFFF0000u);
                        }
                    unsigned bits = 0;
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        // first maximum in window order (dy, dx) = (0,0) (0,1) (1,0) (1,1), as a two-level tree: the
                        // first maximum of each row, then the second row only if strictly greater
                        const bool m0 = a[0][2 * w + 1] > a[0][2 * w], m1 = a[1][2 * w + 1] > a[1][2 * w];
                        const float v0 = m0 ? a[0][2 * w + 1] : a[0][2 * w], v1 = m1 ? a[1][2 * w + 1] : a[1][2 * w];
                        const bool m2 = v1 > v0;
                        const float b = m2 ? v1 : v0;
                        const unsigned id = m2 ? (m1 ? 3u : 2u) : (m0 ? 1u : 0u);
                        *reinterpret_cast<unsigned short*>(wv + (hf * 8 + w) * kStemOutRow + blk * 32) =
                            (unsigned short)(__builtin_bit_cast(unsigned, b) >> 16);
                        bits |= id << (16 * w);
                    }
                    // 2-bit positions of 8 adjacent channels -> one u16 per pooled pixel: OR over the 8 lanes
                    bits <<= 2 * (frow & 7);
                    bits |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)bits, 0xB1, 0xF, 0xF, false);    // lane ^ 1
                    bits |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)bits, 0x4E, 0xF, 0xF, false);    // lane ^ 2
                    bits |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)bits, 0x141, 0xF, 0xF, false);   // mirror inside 8 lanes
                    // (all 8 lanes of the group hold the same word now and write it to the same place: no branch)
                    *reinterpret_cast<unsigned short*>(wi + (hf * 8) * 16 + blk * 4) = (unsigned short)bits;
                    *reinterpret_cast<unsigned short*>(wi + (hf * 8 + 1) * 16 + blk * 4) = (unsigned short)(bits >> 16);
                }
            }
            // the wave's pooled tile (16 pixels x CB*16 channels) and its position words leave through the wave's own LDS
            // patch as whole 16-byte pieces: 3 stores per tile instead of 36 two-byte ones (the TA was 69 % busy)
            {
                const unsigned so = (unsigned)((qrow + cb * 16) * g.Cout + cg * CB * 16) * 2u;
                const u32x4 v0 = *reinterpret_cast<const u32x4*>(ws + lane * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v0, ro, ovoff0, so, 0);
                const u32x4 v1 = *reinterpret_cast<const u32x4*>(ws + 1024 + (lane & 31) * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v1, ro, ovoff1, so, 0);
                const u32x4 vi = *reinterpret_cast<const u32x4*>(ws + kStemIdxOff + (lane & 15) * 16);
                const unsigned si = (unsigned)((qrow + cb * 16) * c8 + cg * CB * 2) * 2u;
                __builtin_amdgcn_raw_buffer_store_b96((__attribute__((ext_vector_type(3))) unsigned){vi[0], vi[1], vi[2]}, ri, ivoff, si, 0);
            }
#ifdef YFS_STEM_TIMING
            {
                STEM_T(t2);
                if (blockIdx.x == 7 && lane == 0) {
                    yfs_stem_dbg[wave * 4 + 0] += t1 - t0;
                    yfs_stem_dbg[wave * 4 + 1] += t2 - t1;
                    yfs_stem_dbg[wave * 4 + 2] += 1;
                }
            }
#endif
        }
        buf ^= 1;
    }
}

#ifdef YFS_STEM_TIMING
}  // namespace
extern "C" __attribute__((visibility("default"))) int yfs_debug_stem_times(long long* out, int reset) {
    if (reset) { long long z[64] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(yfs_stem_dbg), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(yfs_stem_dbg), 64 * sizeof(long long));
}
namespace {
#endif
int launch_stem_pool(const StemPoolArgs& a, hipStream_t stream) {
    StemPoolArgs g = a;
    const int strip_bytes = 7 * g.Ws * 32;
    g.pieces = (strip_bytes + 1023) / 1024;
    g.buf_bytes = g.pieces * 1024;
    g.nstrips = g.N * (g.OH / 4);
    const size_t lds = 2 * (size_t)g.buf_bytes + 8 * kStemPatch;
    auto kern = (g.slope >= 0.0f && g.slope <= 1.0f) ? stem_pool_kernel<3, true> : stem_pool_kernel<3, false>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    YFS_REQUIRE(e == hipSuccess, YFS_ERR_LAUNCH, "yfs_conv2d_stem_fwd_pool_bf16: cannot reserve %zu bytes of LDS", lds);
    // persistent: one workgroup per CU walks the strips (3584 strips at batch 64 = 14 per CU)
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    const int grid = g.nstrips < cus ? g.nstrips : cus;
    kern<<<dim3((unsigned)grid), dim3(512), lds, stream>>>(g);
    YFS_CHECK_LAUNCH("stem_pool_kernel");
    return YFS_OK;
}
}  // namespace

YFS_API int yfs_stem_space_to_depth_bf16(const void* x, int x_is_f32, int64_t stride_n, int64_t stride_c, int64_t stride_h,
                                         int64_t stride_w, void* xs, int64_t N, int H, int W, int Hs, int Ws, void* stream) {
    YFS_REQUIRE(x && xs, YFS_ERR_INVALID_ARG, "yfs_stem_space_to_depth_bf16: null pointer");
    YFS_REQUIRE(N > 0 && H > 0 && W > 0 && 2 * Hs >= H + 6 && 2 * Ws >= W + 6, YFS_ERR_INVALID_ARG,
                "yfs_stem_space_to_depth_bf16: Hs, Ws must cover the image padded by 3 (2*Hs >= H+6)");
    const long long total = (long long)N * Hs * Ws;
    long long blk = yfs_cdiv(total, kThreads);
    if (blk > 65536) blk = 65536;
    hipStream_t s = (hipStream_t)stream;
    if (x_is_f32)
        stem_s2d_kernel<float><<<dim3((unsigned)blk), dim3(kThreads), 0, s>>>((const float*)x, stride_n, stride_c, stride_h, stride_w,
                                                                          (bf16_t*)xs, H, W, Hs, Ws, total);
    else
        stem_s2d_kernel<bf16_t><<<dim3((unsigned)blk), dim3(kThreads), 0, s>>>((const bf16_t*)x, stride_n, stride_c, stride_h,
                                                                           stride_w, (bf16_t*)xs, H, W, Hs, Ws, total);
    YFS_CHECK_LAUNCH("stem_s2d_kernel");
    return YFS_OK;
}

YFS_API int yfs_conv2d_stem_fwd_bf16(const void* xs, const void* w2, const float* bias, void* y, int64_t N, int Hs, int Ws, int OH,
                                     int OW, int Cout, float slope, void* stream) {
    YFS_REQUIRE(xs && w2 && bias && y, YFS_ERR_INVALID_ARG, "yfs_conv2d_stem_fwd_bf16: null pointer");
    int rc = stem_check("yfs_conv2d_stem_fwd_bf16", N, Hs, Ws, OH, OW, Cout);
    if (rc) return rc;
    IgemmArgs g = {};
    g.x = (const bf16_t*)xs; g.w = (const bf16_t*)w2; g.bias = bias; g.out = (bf16_t*)y;
    g.N = (int)N; g.H = Hs; g.W = Ws; g.C = 64; g.PS = kStemC; g.OH = OH; g.OW = OW; g.NOUT = Cout;
    g.KH = kStemKH; g.KW = 1; g.s_mul = 1; g.pad = 0; g.slope = slope;
    g.M = (int)(N * OH * OW);
    g.xbytes = (unsigned)(N * Hs * Ws * kStemC * 2); g.wbytes = (unsigned)((int64_t)Cout * kStemKH * 64 * 2);
    return dispatch_igemm<EPI_BIAS_LEAKY, false>(g, (hipStream_t)stream, "yfs_conv2d_stem_fwd_bf16");
}

YFS_API int yfs_conv2d_stem_fwd_pool_bf16(const void* xs, const void* w2, const float* bias, void* y_pooled, void* pool_idx,
                                          int64_t N, int Hs, int Ws, int OH, int OW, int Cout, float slope, void* stream) {
    YFS_REQUIRE(xs && w2 && bias && y_pooled && pool_idx, YFS_ERR_INVALID_ARG, "yfs_conv2d_stem_fwd_pool_bf16: null pointer");
    int rc = stem_check("yfs_conv2d_stem_fwd_pool_bf16", N, Hs, Ws, OH, OW, Cout);
    if (rc) return rc;
    YFS_REQUIRE(OH % 2 == 0 && OW % 2 == 0, YFS_ERR_UNSUPPORTED,
                "yfs_conv2d_stem_fwd_pool_bf16: conv output %dx%d must be even for the fused 2x2 pool", OH, OW);
    // weight-stationary kernel where its shape assumptions hold (the reference's 448^2 stem does), else the tiled kernel
    if (yfs_knob(YFS_KNOB_STEM_KERNEL) != 1 && Cout == 192 && OH % 4 == 0 && OW % 32 == 0 &&
        2 * (size_t)((7 * Ws * 32 + 1023) / 1024) * 1024 + 8 * 2048 <= 160 * 1024) {
        StemPoolArgs a = {};
        a.xs = (const bf16_t*)xs; a.w2 = (const bf16_t*)w2; a.bias = bias; a.out = (bf16_t*)y_pooled;
        a.pool_idx = (unsigned short*)pool_idx;
        a.N = (int)N; a.Hs = Hs; a.Ws = Ws; a.OH = OH; a.OW = OW; a.Cout = Cout; a.slope = slope;
        a.xbytes = (unsigned)(N * Hs * Ws * kStemC * 2);
        a.obytes = (unsigned)(N * (OH / 2) * (OW / 2) * Cout * 2);
        a.ibytes = (unsigned)(N * (OH / 2) * (OW / 2) * (Cout / 8) * 2);
        return launch_stem_pool(a, (hipStream_t)stream);
    }
    IgemmArgs g = {};
    g.x = (const bf16_t*)xs; g.w = (const bf16_t*)w2; g.bias = bias; g.out = (bf16_t*)y_pooled; g.pool_idx = (unsigned short*)pool_idx;
    g.N = (int)N; g.H = Hs; g.W = Ws; g.C = 64; g.PS = stem_ps(); g.OH = OH; g.OW = OW; g.NOUT = Cout;
    g.KH = kStemKH; g.KW = 1; g.s_mul = 1; g.pad = 0; g.slope = slope;
    g.M = (int)(N * OH * OW);
    g.xbytes = (unsigned)(N * Hs * Ws * stem_ps() * 2); g.wbytes = (unsigned)((int64_t)Cout * kStemKH * 64 * 2);
    return dispatch_igemm<EPI_POOL, false>(g, (hipStream_t)stream, "yfs_conv2d_stem_fwd_pool_bf16");
}

YFS_API size_t yfs_conv2d_stem_wgrad_workspace_bytes(int64_t N, int OH, int OW, int Cout) {
    if (N <= 0 || OH <= 0 || OW <= 0 || Cout <= 0) return 0;
    const WgradPlan p = plan_wgrad(N * OH * OW, 64, Cout, kStemKH);
    const WgradPlan q = plan_wgrad(N * OH * OW, 128, Cout, kStemKH / 2, true);   // the pooled (row-pair) form
    const size_t a = p.partial_bytes + p.db_bytes, b = q.partial_bytes + q.db_bytes;
    return a > b ? a : b;
}

YFS_API int yfs_conv2d_stem_wgrad_bf16(const void* xs, const void* dz, float* dw2, float* db, int64_t N, int Hs, int Ws, int OH,
                                       int OW, int Cout, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    YFS_REQUIRE(xs && dz && dw2, YFS_ERR_INVALID_ARG, "yfs_conv2d_stem_wgrad_bf16: null pointer");
    int rc = stem_check("yfs_conv2d_stem_wgrad_bf16", N, Hs, Ws, OH, OW, Cout);
    if (rc) return rc;
    return wgrad_common("yfs_conv2d_stem_wgrad_bf16", xs, dz, dw2, db, N, Hs, Ws, kStemC, 64, OH, OW, Cout, kStemKH, 1, 1, 0,
                        accumulate, workspace, workspace_bytes, (hipStream_t)stream);
}

YFS_API int yfs_conv2d_stem_wgrad_pool_bf16(const void* xs, const void* dy_pooled, const void* pool_idx, const void* y_pooled,
                                            float* dw2, float* db, int64_t N, int Hs, int Ws, int OH, int OW, int Cout, float slope,
                                            int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    YFS_REQUIRE(xs && dy_pooled && pool_idx && dw2, YFS_ERR_INVALID_ARG, "yfs_conv2d_stem_wgrad_pool_bf16: null pointer");
    int rc = stem_check("yfs_conv2d_stem_wgrad_pool_bf16", N, Hs, Ws, OH, OW, Cout);
    if (rc) return rc;
    // two kernel rows per "tap": the 128-wide reduction tile is rows (2t, 2t+1) x 64 contiguous elements, so the dz
    // tile built in LDS feeds twice the MFMA work of the one-row form (dw2 [Cout][4][64] == [Cout][2][128])
    return wgrad_common("yfs_conv2d_stem_wgrad_pool_bf16", xs, dy_pooled, dw2, db, N, Hs, Ws, stem_ps(), 128, OH, OW, Cout,
                        kStemKH / 2, 1, 1, 0, accumulate, workspace, workspace_bytes, (hipStream_t)stream, pool_idx, y_pooled,
                        slope, (unsigned)(Ws * stem_ps() * 2), 2);
}

// =============================== helpers ========================================================
namespace {

__global__ void __launch_bounds__(kThreads) leaky_bwd_kernel(const bf16x8* __restrict__ dy, const bf16x8* __restrict__ y,
                                                             bf16x8* __restrict__ dz, long long n8, float slope) {
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n8; i += (long long)gridDim.x * kThreads) {
        const bf16x8 a = dy[i], b = y[i];
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (float)b[j] > 0.0f ? a[j] : (bf16_t)((float)a[j] * slope);
        dz[i] = o;
    }
}

// fp32 [Cout][T][Cin] -> bf16 same layout (w16) and bf16 [Cin][T][Cout] with taps reversed (wt16).
// 64 x 64 (co x ci) tile per block for one tap, transposed through LDS.
__global__ void __launch_bounds__(kThreads) weight_prep_kernel(const float* __restrict__ w, bf16_t* __restrict__ w16,
                                                               bf16_t* __restrict__ wt16, int Cout, int T, int Cin) {
    __shared__ float tile[64][64 + 1];
    const int tap = blockIdx.z, co0 = blockIdx.y * 64, ci0 = blockIdx.x * 64;
    const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r4 + 4 * i;
        const long long o = ((long long)(co0 + r) * T + tap) * Cin + ci0 + c;
        const float v = w[o];
        tile[r][c] = v;
        if (w16) w16[o] = (bf16_t)v;
    }
    __syncthreads();
    if (wt16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r4 + 4 * i;   // ci
            wt16[((long long)(ci0 + r) * T + (T - 1 - tap)) * Cout + co0 + c] = (bf16_t)tile[c][r];
        }
    }
}

// All layers of a backbone in ONE launch (23 separate launches were ~12 us each, mostly fixed cost): block ->
// (layer, tap, co tile, ci tile) through a prefix table in the kernel arguments.  Source is the fp32 master
// weight or its bf16 shadow.
constexpr int kPrepMax = 32;
struct PrepBatchArgs {
    const void* w[kPrepMax];
    bf16_t* w16[kPrepMax];
    bf16_t* wt16[kPrepMax];
    int Cout[kPrepMax], T[kPrepMax], Cin[kPrepMax], src_bf16[kPrepMax];
    int tile_end[kPrepMax];
    int count;
};
__global__ void __launch_bounds__(kThreads) weight_prep_batch_kernel(PrepBatchArgs g) {
    __shared__ float tile[64][64 + 1];
    int layer = 0, b = blockIdx.x;
    while (b >= g.tile_end[layer]) ++layer;
    if (layer > 0) b -= g.tile_end[layer - 1];
    const int Cout = g.Cout[layer], T = g.T[layer], Cin = g.Cin[layer];
    const int cit = Cin / 64, cot = Cout / 64;
    const int ci0 = (b % cit) * 64, co0 = ((b / cit) % cot) * 64, tap = b / (cit * cot);
    const float* wf = (const float*)g.w[layer];
    const bf16_t* wh = (const bf16_t*)g.w[layer];
    bf16_t* w16 = g.w16[layer];
    bf16_t* wt16 = g.wt16[layer];
    const bool src16 = g.src_bf16[layer] != 0;
    if (src16 && !w16) {
        // the training path: the source is the optimiser's bf16 shadow and only the transposed copy is wanted.
        // 16-byte global accesses on both sides (the element-wise form below moves 2 bytes per lane: 2.1 TB/s)
        if (!wt16) return;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int piece = threadIdx.x + kThreads * k, r = piece >> 3, c8 = piece & 7;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(wh + ((long long)(co0 + r) * T + tap) * Cin + ci0 + c8 * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) tile[r][c8 * 8 + j] = (float)v[j];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int piece = threadIdx.x + kThreads * k, r = piece >> 3, c8 = piece & 7;   // r = ci, c8 = co chunk
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)tile[c8 * 8 + j][r];
            *reinterpret_cast<bf16x8*>(wt16 + ((long long)(ci0 + r) * T + (T - 1 - tap)) * Cout + co0 + c8 * 8) = o;
        }
        return;
    }
    const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r4 + 4 * i;
        const long long o = ((long long)(co0 + r) * T + tap) * Cin + ci0 + c;
        const float v = src16 ? (float)wh[o] : wf[o];
        tile[r][c] = v;
        if (w16) w16[o] = (bf16_t)v;
    }
    __syncthreads();
    if (wt16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r4 + 4 * i;   // ci
            wt16[((long long)(ci0 + r) * T + (T - 1 - tap)) * Cout + co0 + c] = (bf16_t)tile[c][r];
        }
    }
}

// im2col for the Cin=3 stem: one thread per (pixel, 8-wide K chunk).
template <typename TIn>
__global__ void __launch_bounds__(kThreads) im2col_kernel(const TIn* __restrict__ x, bf16_t* __restrict__ cols, int H, int W,
                                                          int Cin, int K, int stride, int pad, int OH, int OW, int Kpad,
                                                          long long total_chunks) {
    const int cpr = Kpad / 8;
    for (long long id = (long long)blockIdx.x * kThreads + threadIdx.x; id < total_chunks;
         id += (long long)gridDim.x * kThreads) {
        const long long m = id / cpr;
        const int kc = (int)(id - m * cpr) * 8;
        const int n = (int)(m / ((long long)OH * OW));
        const int rem = (int)(m - (long long)n * OH * OW);
        const int oh = rem / OW, ow = rem - oh * OW;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kc + j;
            float v = 0.f;
            if (k < K * K * Cin) {
                const int tap = k / Cin, ci = k - tap * Cin;
                const int ih = oh * stride - pad + tap / K, iw = ow * stride - pad + tap % K;
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                    v = (float)x[(((long long)n * H + ih) * W + iw) * Cin + ci];
            }
            o[j] = (bf16_t)v;
        }
        *reinterpret_cast<bf16x8*>(cols + m * Kpad + kc) = o;
    }
}

}  // namespace

YFS_API int yfs_leaky_bwd_bf16(const void* dy, const void* y, void* dz, int64_t n, float slope, void* stream) {
    YFS_REQUIRE(dy && y && dz && n > 0, YFS_ERR_INVALID_ARG, "yfs_leaky_bwd_bf16: null pointer or n <= 0");
    YFS_REQUIRE(n % 8 == 0, YFS_ERR_UNSUPPORTED, "yfs_leaky_bwd_bf16: n=%lld must be a multiple of 8", (long long)n);
    long long blk = yfs_cdiv(n / 8, kThreads);
    if (blk > 4096) blk = 4096;
    leaky_bwd_kernel<<<dim3((unsigned)blk), dim3(kThreads), 0, (hipStream_t)stream>>>((const bf16x8*)dy, (const bf16x8*)y,
                                                                                   (bf16x8*)dz, n / 8, slope);
    YFS_CHECK_LAUNCH("leaky_bwd_kernel");
    return YFS_OK;
}

YFS_API int yfs_weight_prep_bf16(const float* w, void* w_bf16, void* wt_bf16, int Cout, int K, int Cin, void* stream) {
    YFS_REQUIRE(w && (w_bf16 || wt_bf16), YFS_ERR_INVALID_ARG, "yfs_weight_prep_bf16: null pointer");
    YFS_REQUIRE(Cout > 0 && Cin > 0 && K > 0 && Cout % 64 == 0 && Cin % 64 == 0, YFS_ERR_UNSUPPORTED,
                "yfs_weight_prep_bf16: Cout=%d, Cin=%d must be positive multiples of 64", Cout, Cin);
    weight_prep_kernel<<<dim3((unsigned)(Cin / 64), (unsigned)(Cout / 64), (unsigned)(K * K)), dim3(kThreads), 0,
                         (hipStream_t)stream>>>(w, (bf16_t*)w_bf16, (bf16_t*)wt_bf16, Cout, K * K, Cin);
    YFS_CHECK_LAUNCH("weight_prep_kernel");
    return YFS_OK;
}

YFS_API int yfs_weight_prep_batch_bf16(const yfs_weight_desc* descs, int count, void* stream) {
    YFS_REQUIRE(descs && count > 0 && count <= kPrepMax, YFS_ERR_INVALID_ARG,
                "yfs_weight_prep_batch_bf16: need 1..%d descriptors, got %d", kPrepMax, count);
    PrepBatchArgs g = {};
    long long blocks = 0;
    for (int i = 0; i < count; ++i) {
        const yfs_weight_desc& d = descs[i];
        YFS_REQUIRE(d.w && (d.w_bf16 || d.wt_bf16), YFS_ERR_INVALID_ARG, "yfs_weight_prep_batch_bf16: null pointer in descriptor %d", i);
        YFS_REQUIRE(d.Cout > 0 && d.Cin > 0 && d.K > 0 && d.Cout % 64 == 0 && d.Cin % 64 == 0, YFS_ERR_UNSUPPORTED,
                    "yfs_weight_prep_batch_bf16: descriptor %d: Cout=%d, Cin=%d must be positive multiples of 64", i, d.Cout, d.Cin);
        g.w[i] = d.w; g.w16[i] = (bf16_t*)d.w_bf16; g.wt16[i] = (bf16_t*)d.wt_bf16;
        g.Cout[i] = d.Cout; g.T[i] = d.K * d.K; g.Cin[i] = d.Cin; g.src_bf16[i] = d.w_is_bf16;
        blocks += (long long)(d.Cin / 64) * (d.Cout / 64) * d.K * d.K;
        YFS_REQUIRE(blocks < (1ll << 31), YFS_ERR_UNSUPPORTED, "yfs_weight_prep_batch_bf16: too many tiles");
        g.tile_end[i] = (int)blocks;
    }
    g.count = count;
    weight_prep_batch_kernel<<<dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream>>>(g);
    YFS_CHECK_LAUNCH("weight_prep_batch_kernel");
    return YFS_OK;
}

YFS_API int yfs_im2col_bf16(const void* x, int x_is_f32, void* cols, int64_t N, int H, int W, int Cin, int K, int stride,
                            int Kpad, void* stream) {
    YFS_REQUIRE(x && cols, YFS_ERR_INVALID_ARG, "yfs_im2col_bf16: null pointer");
    YFS_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && K > 0 && stride > 0, YFS_ERR_INVALID_ARG, "yfs_im2col_bf16: bad shape");
    YFS_REQUIRE(Kpad % 8 == 0 && Kpad >= K * K * Cin, YFS_ERR_INVALID_ARG,
                "yfs_im2col_bf16: Kpad=%d must be a multiple of 8 and >= K*K*Cin=%d", Kpad, K * K * Cin);
    const int pad = K / 2;
    const int OH = (H + 2 * pad - K) / stride + 1, OW = (W + 2 * pad - K) / stride + 1;
    const long long total = (long long)N * OH * OW * (Kpad / 8);
    long long blk = yfs_cdiv(total, kThreads);
    if (blk > 65536) blk = 65536;
    hipStream_t s = (hipStream_t)stream;
    if (x_is_f32)
        im2col_kernel<float><<<dim3((unsigned)blk), dim3(kThreads), 0, s>>>((const float*)x, (bf16_t*)cols, H, W, Cin, K, stride,
                                                                        pad, OH, OW, Kpad, total);
    else
        im2col_kernel<bf16_t><<<dim3((unsigned)blk), dim3(kThreads), 0, s>>>((const bf16_t*)x, (bf16_t*)cols, H, W, Cin, K,
                                                                         stride, pad, OH, OW, Kpad, total);
    YFS_CHECK_LAUNCH("im2col_kernel");
    return YFS_OK;
}
