// Probe: does `buffer_load_dwordx4 ... lds` write zeros for out-of-range lanes?  (needed for branch-free conv padding
// with direct-to-LDS staging).  Build: hipcc --offload-arch=gfx950 -O2 -o lds_dma_oob lds_dma_oob.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k(const unsigned* src, unsigned nbytes, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) lds[i] = 0xDEADBEEFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    unsigned off = (threadIdx.x % 3 == 1) ? 0xFFFFFFFFu : threadIdx.x * 16;   // every third lane out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
    // second instruction, wave-uniform LDS base + 1024
    unsigned off2 = (threadIdx.x & 1) ? 0xFFFFFFFFu : (63 - threadIdx.x) * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 256), 16, off2, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(64 * 4);
    for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
    unsigned *d, *o;
    hipMalloc(&d, 1024); hipMalloc(&o, 2048);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    k<<<1, 64, 2048>>>(d, 1024, o);
    std::vector<unsigned> r(512);
    hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        bool oob = (l % 3 == 1);
        for (int j = 0; j < 4; ++j) {
            unsigned want = oob ? 0u : 1000 + l * 4 + j;
            if (r[l * 4 + j] != want) { if (bad < 8) printf("A lane %d j %d got %u (0x%x) want %u\n", l, j, r[l*4+j], r[l*4+j], want); ++bad; }
        }
        bool oob2 = l & 1;
        for (int j = 0; j < 4; ++j) {
            unsigned want = oob2 ? 0u : 1000 + (63 - l) * 4 + j;
            if (r[256 + l * 4 + j] != want) { if (bad < 16) printf("B lane %d j %d got %u (0x%x) want %u\n", l, j, r[256+l*4+j], r[256+l*4+j], want); ++bad; }
        }
    }
    printf("lds_dma_oob: %s (%d mismatches)\n", bad ? "FAIL" : "OK: out-of-range lanes write zeros, lane i lands at base + 16*i", bad);
    return bad != 0;
}
