// Probe: what does an out-of-range lane of buffer_load_dwordx4 ... lds write to LDS?  (conv_wgrad_ls_kernel relies on zeros.)
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/probes/oob_lds_probe.hip -o /tmp/oob && /tmp/oob
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const unsigned* p, unsigned* out) {
    extern __shared__ unsigned char smem[];
    unsigned* s32 = reinterpret_cast<unsigned*>(smem);
    for (int i = threadIdx.x; i < 512; i += 64) s32[i] = 0xAAAAAAAAu;
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x80000000, 0x00020000);
    int vo = threadIdx.x * 16;
    if (threadIdx.x & 1) vo = (int)0x80000000;                   // odd lanes: out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem), 16, vo, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = s32[i];
}
int main() {
    unsigned *d, *o, h[256], src[256];
    for (int i = 0; i < 256; ++i) src[i] = 0x1000 + i;
    hipMalloc(&d, 1024); hipMalloc(&o, 1024);
    hipMemcpy(d, src, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o);
    hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
    int zeros = 0, stale = 0, good = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            unsigned v = h[l * 4 + j];
            if (l & 1) { zeros += v == 0; stale += v == 0xAAAAAAAAu; } else good += v == (unsigned)(0x1000 + l * 4 + j);
        }
    printf("in-range dwords correct %d/128; out-of-range lanes: zero %d/128, stale %d/128 (first: %08x)\n", good, zeros, stale, h[4]);
    return 0;
}
