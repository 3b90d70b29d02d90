// Probe (measurement aid, GPU box only): which (lane, element) does ds_read_b64_tr_b16 deliver where?
// LDS holds short i at index i; lane l passes byte address offs[l]; prints, per lane, the 4 shorts received.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/ds_read_tr_probe.hip -o /tmp/trp && /tmp/trp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void k(const int* offs, short* out) {
    extern __shared__ __attribute__((aligned(16))) short lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (short)i;
    __syncthreads();
    short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) short4v*)((__attribute__((address_space(3))) char*)lds + offs[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    int* d_off; short* d_out;
    hipMalloc(&d_off, 64 * 4); hipMalloc(&d_out, 64 * 4 * 2);
    for (int e = 0; e < 3; ++e) {
        std::vector<int> off(64);
        for (int l = 0; l < 64; ++l) {
            if (e == 0) off[l] = l * 8;                                  // lane l -> shorts 4l..4l+3
            else if (e == 1) off[l] = (l & 15) * 128 + (l >> 4) * 8;     // 16 rows of 128 B, lane group picks an 8-B column
            else off[l] = (l & 3) * 128 + (l >> 2) * 8;                  // 4 rows of 128 B, 16 columns of 8 B
        }
        hipMemcpy(d_off, off.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 16384, 0, d_off, d_out);
        std::vector<short> out(256);
        hipMemcpy(out.data(), d_out, 512, hipMemcpyDeviceToHost);
        printf("experiment %d (value = short index; own address holds 4*lane.. in experiment 0)\n", e);
        for (int l = 0; l < 64; ++l)
            printf("lane %2d addr %5d -> %5d %5d %5d %5d\n", l, off[l], out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
    }
    return 0;
}
