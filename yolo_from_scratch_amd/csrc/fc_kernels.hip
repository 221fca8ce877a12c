// Epilogue of the first fully connected layer, fused: split-K partial sums + bias + LeakyReLU + Dropout in one
// pass, and its backward (dropout mask x LeakyReLU derivative + bias gradient) in one pass.
// Reference: yolo_toy_regressor_torch.py:258-263  nn.Linear(50176, 4096) -> nn.LeakyReLU(0.1) -> nn.Dropout(0.2).
// The GEMM itself is a plain library GEMM (hipBLASLt); this replaces the five element-wise launches after it
// (sum over split-K slabs, bias add, bf16 cast, LeakyReLU, dropout) and the three before its backward.
// HBM/latency-bound: [64][4096] is 0.5 MB in bf16.
#include "yfs_common.h"

namespace {
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
constexpr int kThreads = 256;

// Philox4x32-10 (Salmon et al., SC'11): counter = (element group, 0, 0, 0), key = seed.
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned h0 = (unsigned)(p0 >> 32), l0 = (unsigned)p0, h1 = (unsigned)(p1 >> 32), l1 = (unsigned)p1;
    c[0] = h1 ^ c[1] ^ k0;
    c[1] = l1;
    c[2] = h0 ^ c[3] ^ k1;
    c[3] = l0;
}
__device__ __forceinline__ void philox4(unsigned long long seed, unsigned long long ctr, unsigned (&out)[4]) {
    unsigned c[4] = {(unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u};
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = c[i];
}

// one thread = 8 consecutive outputs of one row
__global__ void __launch_bounds__(kThreads) fc_epilogue_fwd_kernel(const float* __restrict__ partial, int splits,
                                                                   const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                                   unsigned char* __restrict__ keep, long long M, int N,
                                                                   float slope, float p_drop, unsigned long long seed) {
    const int n8 = N >> 3;
    const long long total = M * n8;
    const float scale = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (long long id = (long long)blockIdx.x * kThreads + threadIdx.x; id < total; id += (long long)gridDim.x * kThreads) {
        const int c8 = (int)(id % n8);
        const long long e0 = id * 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bias ? bias[c8 * 8 + j] : 0.0f;
        for (int s = 0; s < splits; ++s) {   // fixed order: deterministic
            const float4 a = *reinterpret_cast<const float4*>(partial + (long long)s * M * N + e0);
            const float4 b = *reinterpret_cast<const float4*>(partial + (long long)s * M * N + e0 + 4);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
            v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
        unsigned r[8];
        if (p_drop > 0.0f) {
            unsigned r0[4], r1[4];
            philox4(seed, (unsigned long long)id * 2, r0);
            philox4(seed, (unsigned long long)id * 2 + 1, r1);
#pragma unroll
            for (int j = 0; j < 4; ++j) { r[j] = r0[j]; r[4 + j] = r1[j]; }
        }
        bf16x8 o;
        unsigned long long kb = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = v[j] > 0.0f ? v[j] : v[j] * slope;
            bool k = true;
            if (p_drop > 0.0f) {
                k = (float)(r[j] >> 8) * (1.0f / 16777216.0f) >= p_drop;   // uniform in [0,1) from the top 24 bits
                t = k ? t * scale : 0.0f;
            }
            o[j] = (bf16_t)t;
            kb |= (unsigned long long)(k ? 1u : 0u) << (8 * j);
        }
        *reinterpret_cast<bf16x8*>(y + e0) = o;
        if (keep) *reinterpret_cast<unsigned long long*>(keep + e0) = kb;
    }
}

// dz = dy * keep / (1 - p) * (y > 0 ? 1 : slope); db[n] = sum_m dz[m][n] (fp32 over the ROUNDED dz, fixed order).
// Block = 32 column groups (8 columns each) x 8 row slices; grid = N / 256.
__global__ void __launch_bounds__(kThreads) fc_epilogue_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y,
                                                                   const unsigned char* __restrict__ keep,
                                                                   bf16_t* __restrict__ dz, float* __restrict__ db, long long M,
                                                                   int N, float slope, float p_drop, int accumulate) {
    __shared__ float red[8][256 + 8];
    const int cg = threadIdx.x & 31, rs = threadIdx.x >> 5;
    const int col = blockIdx.x * 256 + cg * 8;
    const float scale = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        for (long long m = rs; m < M; m += 8) {
            const long long e0 = m * N + col;
            const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + e0);
            const bf16x8 yv = *reinterpret_cast<const bf16x8*>(y + e0);
            unsigned long long kb = ~0ull;
            if (keep) kb = *reinterpret_cast<const unsigned long long*>(keep + e0);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = (float)g[j];
                if (p_drop > 0.0f) t = ((kb >> (8 * j)) & 1ull) ? t * scale : 0.0f;
                if (!((float)yv[j] > 0.0f)) t *= slope;
                o[j] = (bf16_t)t;
                acc[j] += (float)o[j];
            }
            *reinterpret_cast<bf16x8*>(dz + e0) = o;
        }
    }
    if (db) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[rs][cg * 8 + j] = acc[j];
        __syncthreads();
        const int c = blockIdx.x * 256 + threadIdx.x;
        if (c < N) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
            db[c] = accumulate ? db[c] + t : t;
        }
    }
}
}  // namespace

YFS_API int yfs_fc_epilogue_fwd_bf16(const float* partial, int splits, const float* bias, void* y, void* keep, int64_t M, int N,
                                     float slope, float p_drop, uint64_t seed, void* stream) {
    YFS_REQUIRE(partial && y, YFS_ERR_INVALID_ARG, "yfs_fc_epilogue_fwd_bf16: null pointer");
    YFS_REQUIRE(M > 0 && N > 0 && N % 8 == 0 && splits >= 1, YFS_ERR_INVALID_ARG,
                "yfs_fc_epilogue_fwd_bf16: need M > 0, N a positive multiple of 8, splits >= 1 (got M=%lld N=%d splits=%d)",
                (long long)M, N, splits);
    YFS_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, YFS_ERR_INVALID_ARG, "yfs_fc_epilogue_fwd_bf16: p_drop %g not in [0, 1)", p_drop);
    YFS_REQUIRE(p_drop == 0.0f || keep, YFS_ERR_INVALID_ARG, "yfs_fc_epilogue_fwd_bf16: dropout needs the keep-mask output");
    long long blk = yfs_cdiv(M * (N / 8), kThreads);
    if (blk > 4096) blk = 4096;
    fc_epilogue_fwd_kernel<<<dim3((unsigned)blk), dim3(kThreads), 0, (hipStream_t)stream>>>(
        partial, splits, bias, (bf16_t*)y, (unsigned char*)keep, M, N, slope, p_drop, seed);
    YFS_CHECK_LAUNCH("fc_epilogue_fwd_kernel");
    return YFS_OK;
}

YFS_API int yfs_fc_epilogue_bwd_bf16(const void* dy, const void* y, const void* keep, void* dz, float* db, int64_t M, int N,
                                     float slope, float p_drop, int accumulate, void* stream) {
    YFS_REQUIRE(dy && y && dz, YFS_ERR_INVALID_ARG, "yfs_fc_epilogue_bwd_bf16: null pointer");
    YFS_REQUIRE(M > 0 && N > 0 && N % 8 == 0, YFS_ERR_INVALID_ARG,
                "yfs_fc_epilogue_bwd_bf16: need M > 0 and N a positive multiple of 8 (got M=%lld N=%d)", (long long)M, N);
    YFS_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, YFS_ERR_INVALID_ARG, "yfs_fc_epilogue_bwd_bf16: p_drop %g not in [0, 1)", p_drop);
    YFS_REQUIRE(p_drop == 0.0f || keep, YFS_ERR_INVALID_ARG, "yfs_fc_epilogue_bwd_bf16: dropout needs the keep mask");
    fc_epilogue_bwd_kernel<<<dim3((unsigned)yfs_cdiv(N, 256)), dim3(kThreads), 0, (hipStream_t)stream>>>(
        (const bf16_t*)dy, (const bf16_t*)y, (const unsigned char*)keep, (bf16_t*)dz, db, M, N, slope, p_drop, accumulate);
    YFS_CHECK_LAUNCH("fc_epilogue_bwd_kernel");
    return YFS_OK;
}
