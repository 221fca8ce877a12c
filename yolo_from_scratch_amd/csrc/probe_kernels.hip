// Measurement aid (no reference counterpart): the bf16 MFMA rate this chip sustains at the clock it actually
// holds under load, printed by bench.py next to the vendor peak ("report the measured ceiling
// next to the vendor one").  Register-only: 16 independent accumulator chains of v_mfma_f32_16x16x32_bf16 per
// wave, 8 waves per CU x 4 blocks per CU, no memory traffic inside the loop.
#include "yfs_common.h"

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void __launch_bounds__(256) mfma_probe_kernel(int iters, float* __restrict__ out) {
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)(float)((threadIdx.x + j) & 3);
        b[j] = (__bf16)(float)((threadIdx.x * 3 + j) & 1);
    }
    f32x4 acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) s += acc[k];
    if (s[0] + s[1] + s[2] + s[3] == -1.0f) out[0] = s[0];   // never true: keeps the chain alive
}
// The same loop with operands that change from one MFMA to the next (eight pseudo-random fragments per side, so the
// operand buses and multipliers toggle the way they do on real data) and, optionally, on the 32x32x16 shape (half the
// operand register reads per FLOP).  What clock and rate does the matrix pipe alone hold then?
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __forceinline__ unsigned probe_hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
template <int SHAPE>
__global__ void __launch_bounds__(256) mfma_probe_toggle_kernel(int iters, int toggling, float* __restrict__ out) {
    bf16x8 a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned h = probe_hash((threadIdx.x * 8 + k) * 16 + j + 1), h2 = probe_hash(h + 12345);
            const int ka = toggling ? k : 0;
            // bf16 with random sign, random mantissa, exponent 126 or 127: values in +-[0.5, 2)
            const unsigned short ua = (unsigned short)(((probe_hash(h ^ ka) & 0x80FFu) | 0x3F00u));
            const unsigned short ub = (unsigned short)(((probe_hash(h2 ^ ka) & 0x80FFu) | 0x3F00u));
            a[k][j] = __builtin_bit_cast(__bf16, toggling ? ua : (unsigned short)0x3F80);
            b[k][j] = __builtin_bit_cast(__bf16, toggling ? ub : (unsigned short)0x3F80);
        }
    float sum = 0.f;
    if (SHAPE == 0) {
        f32x4 acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k & 7], b[(k * 3 + 1) & 7], acc[k], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    } else {
        f32x16 acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[(k * 3 + 1) & 7], acc[k], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[k][r];
    }
    if (sum == -1.2345f) out[0] = sum;   // never true in practice: keeps the chains alive
}
}  // namespace

YFS_API int yfs_probe_mfma_bf16_ex(int shape, int toggling, int iters, int blocks, double* flop_out, void* scratch, void* stream) {
    YFS_REQUIRE((shape == 0 || shape == 1) && iters > 0 && blocks > 0 && flop_out && scratch, YFS_ERR_INVALID_ARG,
                "yfs_probe_mfma_bf16_ex: bad argument");
    if (shape == 0)
        mfma_probe_toggle_kernel<0><<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(iters, toggling, (float*)scratch);
    else
        mfma_probe_toggle_kernel<1><<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(iters, toggling, (float*)scratch);
    YFS_CHECK_LAUNCH("mfma_probe_toggle_kernel");
    // per wave and iteration: 16 x (2*16*16*32) = 8 x (2*32*32*16) FLOP
    *flop_out = (double)blocks * 4.0 * (double)iters * 16.0 * (2.0 * 16 * 16 * 32);
    return YFS_OK;
}

YFS_API int yfs_probe_mfma_bf16(int iters, int blocks, double* flop_out, void* scratch, void* stream) {
    YFS_REQUIRE(iters > 0 && blocks > 0 && flop_out && scratch, YFS_ERR_INVALID_ARG, "yfs_probe_mfma_bf16: bad argument");
    mfma_probe_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(iters, (float*)scratch);
    YFS_CHECK_LAUNCH("mfma_probe_kernel");
    *flop_out = (double)blocks * 4.0 * (double)iters * 16.0 * (2.0 * 16 * 16 * 32);
    return YFS_OK;
}
