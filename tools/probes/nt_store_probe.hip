// nt_store_probe.hip -- does an inline-asm "global_store_dwordx4 v[ptr], v[data], off nt" write what and where it should?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_asm(unsigned* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i * 4 + 3 < n) {
        u32x4 w = {(unsigned)i * 4u, (unsigned)i * 4u + 1u, (unsigned)i * 4u + 2u, (unsigned)i * 4u + 3u};
        void* p = out + (size_t)i * 4;
        asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(w) : "memory");
    }
}
__global__ void k_builtin(unsigned* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i * 4 + 3 < n) {
        u32x4 w = {(unsigned)i * 4u, (unsigned)i * 4u + 1u, (unsigned)i * 4u + 2u, (unsigned)i * 4u + 3u};
        __builtin_nontemporal_store(w, (u32x4*)(out + (size_t)i * 4));
    }
}
int main() {
    const int n = 1 << 22;
    unsigned* d; hipMalloc(&d, n * 4);
    std::vector<unsigned> h(n);
    for (int which = 0; which < 2; ++which) {
        hipMemset(d, 0xFF, n * 4);
        if (which == 0) k_asm<<<n / 4 / 256, 256>>>(d, n); else k_builtin<<<n / 4 / 256, 256>>>(d, n);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        long bad = 0; for (int i = 0; i < n; ++i) bad += h[i] != (unsigned)i;
        printf("%s: %ld of %d words wrong (first words: %u %u %u %u)\n", which == 0 ? "inline asm nt store" : "__builtin_nontemporal_store", bad, n, h[0], h[1], h[2], h[3]);
    }
    return 0;
}
