// fp8_mfma_probe.hip -- stand-alone probe (no torch, no library) of the gfx950 block-scaled fp8 MFMA that DESIGN.md section 7 item 1
// plans to put the split-operand product's cross terms on:
//     v_mfma_scale_f32_16x16x128_f8f6f4   (A, B: 32 OCP e4m3 bytes per lane; one E8M0 scale byte per lane and operand)
// Questions: (1) the lane -> (row, k) map of the 32 bytes; (2) which byte of the scale register `opsel` picks and that the scale is
// 2^(e - 127) applied per LANE (= per 32-element block); (3) v_cvt_pk_fp8_f32's rounding and byte placement; (4) the issue rate of a
// register-only loop next to v_mfma_f32_16x16x32_bf16 (the guide says 2x).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/fp8_probe tools/probes/fp8_mfma_probe.hip && /tmp/fp8_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));

static float e4m3_to_f32(uint8_t b) {                      // OCP e4m3fn: bias 7, no infinities, 0x7F / 0xFF = NaN
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 0) v = ldexpf((float)m, -9);                   // subnormal: m * 2^-3 * 2^-6
    else if (e == 15 && m == 7) v = NAN;
    else v = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}

__global__ void mfma_once(const int* a, const int* b, const int* sa, const int* sb, float* c, int opsel_a, int opsel_b) {
    const int l = threadIdx.x;
    v8i av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = a[l * 8 + j]; bv[j] = b[l * 8 + j]; }
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    // opsel must be an immediate: the four values are spelled out
    if (opsel_a == 0 && opsel_b == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, sa[l], 0, sb[l]);
    else if (opsel_a == 1 && opsel_b == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 1, sa[l], 0, sb[l]);
    else if (opsel_a == 2 && opsel_b == 3) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 2, sa[l], 3, sb[l]);
    else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 3, sa[l], 1, sb[l]);
    for (int r = 0; r < 4; ++r) c[l * 4 + r] = acc[r];
}

__global__ void cvt_probe(const float* x, int* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i * 4 + 3 < n) {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(x[i * 4], x[i * 4 + 1], w, false);     // bytes 0, 1
        w = __builtin_amdgcn_cvt_pk_fp8_f32(x[i * 4 + 2], x[i * 4 + 3], w, true);  // bytes 2, 3
        out[i] = w;
    }
}

template <int KIND>
__global__ void __launch_bounds__(256) rate_loop(float* out, int iters, int seed) {
    const int l = threadIdx.x;
    v4f acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
    v8i a, b;
    for (int j = 0; j < 8; ++j) { a[j] = 0x38383838 ^ (l * 0x01010101 & 0x07070707) ^ seed; b[j] = 0x30303030 ^ ((l + j) & 7); }
    v8s ab = *(v8s*)&a, bb = *(v8s*)&b;
    const int sc = 0x7f7f7f7f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, sc, 0, sc);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(__attribute__((ext_vector_type(8))) __bf16*)&ab,
                                                                 *(__attribute__((ext_vector_type(8))) __bf16*)&bb, acc[i], 0, 0, 0);
        }
        a[0] ^= it; ab[0] ^= (short)it;                        // operands change a little every trip
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + l] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main() {
    srand(7);
    std::vector<uint8_t> A(16 * 128), B(16 * 128);          // A[row][k], B[col][k]: finite e4m3 values of modest size
    auto rnd = []() { uint8_t b; do { b = (uint8_t)(rand() & 0xFF); } while ((b & 0x7F) >= 0x58 || ((b >> 3) & 15) < 4); return b; };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    std::vector<uint8_t> SA(64 * 4), SB(64 * 4);            // four candidate scale bytes per lane and operand
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { SA[l * 4 + j] = (uint8_t)(124 + (l * 3 + j * 5) % 7); SB[l * 4 + j] = (uint8_t)(125 + (l + j * 2) % 5); }
    // candidate lane maps: byte j (0..31) of lane l holds K index ...
    const char* mapname[3] = {"k = 32 (l >> 4) + j", "k = 4 (j >> 2 ... interleaved 16-byte halves: k = 64 (j >> 4) + 16 (l >> 4) + (j & 15)", "k = 4 j + (l >> 4)"};
    auto kmap = [](int which, int l, int j) {
        const int g = l >> 4;
        if (which == 0) return 32 * g + j;
        if (which == 1) return 64 * (j >> 4) + 16 * g + (j & 15);
        return 4 * j + g;
    };
    int *da, *db, *dsa, *dsb; float* dc;
    CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 64 * 4)); CK(hipMalloc(&dsb, 64 * 4)); CK(hipMalloc(&dc, 64 * 4 * 4));
    CK(hipMemcpy(dsa, SA.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, SB.data(), 256, hipMemcpyHostToDevice));
    const int opsels[4][2] = {{0, 0}, {1, 0}, {2, 3}, {3, 1}};
    // scale sets: 0 = every byte 127 (2^0: isolates the data layout), 1 = A's scales vary by lane and byte, 2 = B's, 3 = both
    for (int sset = 0; sset < 4; ++sset) {
        std::vector<uint8_t> sa(SA), sb(SB);
        if (!(sset & 1)) for (auto& v : sa) v = 127;
        if (!(sset & 2)) for (auto& v : sb) v = 127;
        CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
        for (int which = 0; which < 3; ++which) {
            std::vector<uint8_t> ra(64 * 32), rb(64 * 32);
            for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) { ra[l * 32 + j] = A[(l & 15) * 128 + kmap(which, l, j)]; rb[l * 32 + j] = B[(l & 15) * 128 + kmap(which, l, j)]; }
            CK(hipMemcpy(da, ra.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, rb.data(), 64 * 32, hipMemcpyHostToDevice));
            for (int t = 0; t < (sset ? 4 : 1); ++t) {
                const int oa = opsels[t][0], ob = opsels[t][1];
                mfma_once<<<1, 64>>>(da, db, dsa, dsb, dc, oa, ob);
                CK(hipDeviceSynchronize());
                float C[256];
                CK(hipMemcpy(C, dc, sizeof(C), hipMemcpyDeviceToHost));
                // hypotheses for the result layout: 0: C[row = 4 (l >> 4) + r][col = l & 15]; 1: transposed (row = l & 15, col = 4 (l >> 4) + r)
                for (int lay = 0; lay < 2; ++lay) {
                    double worst = 0.0, mag = 0.0;
                    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
                        const int row = lay == 0 ? 4 * (l >> 4) + r : (l & 15), col = lay == 0 ? (l & 15) : 4 * (l >> 4) + r;
                        double ref = 0.0;
                        // hardware K order (found by the one-byte experiments below): byte j of lane group g is
                        // k_hw = 64 (j >> 4) + 16 g + (j & 15) -- two 16-byte halves per lane, like two 16x16x64 steps -- and the E8M0
                        // scale of the 32-element block b = k_hw >> 5 of row r is byte `opsel` of lane 16 b + r
                        for (int g = 0; g < 4; ++g)
                            for (int j = 0; j < 32; ++j) {
                                const int khw = 64 * (j >> 4) + 16 * g + (j & 15), blk = khw >> 5;
                                const double sc = ldexp(1.0, (int)sa[(16 * blk + row) * 4 + oa] - 127) * ldexp(1.0, (int)sb[(16 * blk + col) * 4 + ob] - 127);
                                const int k = kmap(which, 16 * g, j);
                                ref += sc * (double)e4m3_to_f32(A[row * 128 + k]) * (double)e4m3_to_f32(B[col * 128 + k]);
                            }
                        worst = fmax(worst, fabs(ref - (double)C[l * 4 + r])); mag = fmax(mag, fabs(ref));
                    }
                    printf("scales %d  k-map %d  opsel (%d, %d)  C layout %d: max |C - ref| = %.3e of max |ref| %.3e  %s\n", sset, which, oa, ob, lay,
                           worst, mag, worst <= 2e-4 * mag ? "MATCH" : "differs");
                }
                if (sset == 0 && which == 0) printf("   C[lane 0][0..3] = %g %g %g %g, C[lane 17][0..3] = %g %g %g %g\n", C[0], C[1], C[2], C[3], C[68], C[69], C[70], C[71]);
            }
        }
    }
    // ---- one-hot discovery of the operand layout (uniform scales 2^0): A has a single 1.0 at (lane la, byte ja)
    {
        std::vector<uint8_t> ones(256, 127);
        CK(hipMemcpy(dsa, ones.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, ones.data(), 256, hipMemcpyHostToDevice));
        const int las[5] = {0, 5, 17, 37, 63}, jas[5] = {0, 1, 5, 16, 31};
        for (int t = 0; t < 5; ++t) {
            const int la = las[t], ja = jas[t];
            std::vector<uint8_t> ra(64 * 32, 0), rb(64 * 32, 0x38);
            ra[la * 32 + ja] = 0x38;
            float C[256];
            auto run = [&]() { hipMemcpy(da, ra.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, rb.data(), 2048, hipMemcpyHostToDevice);
                               mfma_once<<<1, 64>>>(da, db, dsa, dsb, dc, 0, 0); hipDeviceSynchronize(); hipMemcpy(C, dc, sizeof(C), hipMemcpyDeviceToHost); };
            run();
            printf("A one-hot (lane %d, byte %d), B all ones: nonzero C at (lane:reg)", la, ja);
            int nz = 0;
            for (int i = 0; i < 256; ++i) if (C[i] != 0.f) { if (nz < 20) printf(" %d:%d=%g", i / 4, i % 4, C[i]); ++nz; }
            printf("  [%d nonzero]\n", nz);
            // which lane group of B pairs with it
            for (int g = 0; g < 4; ++g) {
                std::fill(rb.begin(), rb.end(), 0);
                for (int l = 16 * g; l < 16 * g + 16; ++l) for (int j = 0; j < 32; ++j) rb[l * 32 + j] = 0x38;
                run();
                int n2 = 0; for (int i = 0; i < 256; ++i) n2 += C[i] != 0.f;
                printf("    B nonzero only in lanes %d..%d: %d nonzero C\n", 16 * g, 16 * g + 15, n2);
            }
            // which byte of B pairs with it
            printf("    B nonzero only in byte j (all lanes): nonzero C for j =");
            for (int jb = 0; jb < 32; ++jb) {
                std::fill(rb.begin(), rb.end(), 0);
                for (int l = 0; l < 64; ++l) rb[l * 32 + jb] = 0x38;
                run();
                int n2 = 0; for (int i = 0; i < 256; ++i) n2 += C[i] != 0.f;
                if (n2) printf(" %d(%d)", jb, n2);
            }
            printf("\n");
        }
    }
    // ---- whose scale byte applies to a lane's 32-element block?  A one-hot (value 1.0) at (la, byte 3), B all ones, every scale
    //      2^0 except ONE byte (lane ls, byte bs) of the A-scale registers = 2^1: the C row doubles iff that byte is the one used
    {
        const int las[4] = {0, 21, 42, 63};
        for (int PB = 3; PB < 32; PB += 16)
        for (int opa = 0; opa < 4; opa += 3) {
            for (int t = 0; t < 4; ++t) {
                const int la = las[t];
                std::vector<uint8_t> ra(64 * 32, 0), rb(64 * 32, 0x38), ones(256, 127);
                ra[la * 32 + PB] = 0x38;
                hipMemcpy(da, ra.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, rb.data(), 2048, hipMemcpyHostToDevice);
                hipMemcpy(dsb, ones.data(), 256, hipMemcpyHostToDevice);
                printf("A one-hot in lane %d byte %d (row %d, lane group %d), opsel_a %d: the result doubles when the 2^1 byte is (lane:byte)", la, PB, la & 15, la >> 4, opa);
                for (int ls = 0; ls < 64; ++ls) for (int bs = 0; bs < 4; ++bs) {
                    std::vector<uint8_t> sa(256, 127);
                    sa[ls * 4 + bs] = 128;
                    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice);
                    mfma_once<<<1, 64>>>(da, db, dsa, dsb, dc, opa, opa == 0 ? 0 : 1);
                    hipDeviceSynchronize();
                    float C[256];
                    hipMemcpy(C, dc, sizeof(C), hipMemcpyDeviceToHost);
                    float mx = 0.f; for (int i = 0; i < 256; ++i) mx = fmaxf(mx, C[i]);
                    if (mx != 1.0f) printf(" %d:%d(x%g)", ls, bs, mx);
                }
                printf("\n");
            }
        }
    }
    // conversion: v_cvt_pk_fp8_f32 against round-to-nearest-even onto the e4m3 grid (saturating)
    {
        const int n = 4096;
        std::vector<float> x(n);
        for (int i = 0; i < n; ++i) x[i] = ldexpf((float)(rand() % 2001 - 1000) / 1000.0f, rand() % 14 - 8);
        float* dx; int* dw;
        CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dw, n));
        CK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice));
        cvt_probe<<<n / 4 / 64, 64>>>(dx, dw, n);
        CK(hipDeviceSynchronize());
        std::vector<uint8_t> w(n);
        CK(hipMemcpy(w.data(), dw, n, hipMemcpyDeviceToHost));
        double worst_rel = 0.0; int bad = 0;
        for (int i = 0; i < n; ++i) {
            const float got = e4m3_to_f32(w[i]);
            // nearest e4m3 value by brute force
            float best = 0.f; double bd = 1e30;
            for (int b = 0; b < 256; ++b) { const float v = e4m3_to_f32((uint8_t)b); if (v == v && fabs((double)v - x[i]) < bd) { bd = fabs((double)v - x[i]); best = v; } }
            if (fabs((double)got - x[i]) > bd * (1 + 1e-6) + 1e-30) ++bad;
            if (x[i] != 0.f && fabsf(x[i]) >= 0.015625f) worst_rel = fmax(worst_rel, fabs((double)got - x[i]) / fabs((double)x[i]));
            (void)best;
        }
        printf("v_cvt_pk_fp8_f32: %d of %d values are not the nearest e4m3 value; worst relative error in the normal range %.4f (2^-4 = 0.0625)\n", bad, n, worst_rel);
    }
    // issue rate
    {
        float* dout; CK(hipMalloc(&dout, 1024 * 256 * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int blocks = 1024, iters = 20000;
        for (int kind = 0; kind < 2; ++kind) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (kind == 0) rate_loop<0><<<blocks, 256>>>(dout, iters, rep); else rate_loop<1><<<blocks, 256>>>(dout, iters, rep);
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double flop = (double)blocks * 4 * iters * 8 * 2.0 * 16 * 16 * (kind == 0 ? 128 : 32);
                if (rep == 1) printf("%s register-only loop: %.1f TF/s (%.2f ms)\n", kind == 0 ? "v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3)" : "v_mfma_f32_16x16x32_bf16", flop / ms / 1e9, ms);
            }
        }
    }
    return 0;
}
