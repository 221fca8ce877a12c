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
}  // namespace

YFS_API int yfs_probe_mfma_bf16(int iters, int blocks, double* flop_out, void* scratch, void* stream) {
    YFS_REQUIRE(iters > 0 && blocks > 0 && flop_out && scratch, YFS_ERR_INVALID_ARG, "yfs_probe_mfma_bf16: bad argument");
    mfma_probe_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(iters, (float*)scratch);
    YFS_CHECK_LAUNCH("mfma_probe_kernel");
    *flop_out = (double)blocks * 4.0 * (double)iters * 16.0 * (2.0 * 16 * 16 * 32);
    return YFS_OK;
}
