// hipemu.h -- DEVELOPER / TEST TOOL ONLY.  A tiny CPU simulator of the HIP execution model
// (workgroups, 64-lane wavefronts, LDS, __syncthreads, wave shuffles, gfx950 MFMA lane layouts)
// used to debug the index logic of visualbert_amd/csrc/*.hip in a container without a GPU.
//
// It is NOT part of the product: visualbert_amd/ never loads the emulator library, there is no
// CPU fallback path, and every number in bench.py / profiles/ comes from the real gfx950 build.
// The emulator build is selected by -DVB_EMU and produces tests/hipemu/libvisualbert_emu.so,
// which only `VB_EMU=1 pytest -m gpu` (see tests/conftest.py) will load.
//
// Model: one OS thread runs one workgroup at a time; the workgroup's threads are ucontext fibers
// scheduled round-robin and switched only at synchronisation points (__syncthreads, wave-level
// collectives).  Wave collectives are rendezvous points across the 64 lanes of a wave; a lane that
// exits early while its wave-mates wait is reported as a deadlock (on hardware: undefined values).
#pragma once
#include <ucontext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0

namespace hipemu {

constexpr int kMaxThreads = 1024;
constexpr int kWave = 64;
constexpr size_t kStackBytes = 256 * 1024;
constexpr size_t kSlotBytes = 128;

struct WaveX {
    int count;
    unsigned gen;
    int nlanes;
    alignas(16) unsigned char buf[2][kWave][kSlotBytes];
};

struct Fiber {
    ucontext_t ctx;
    unsigned char* stack;
    dim3 tid;
    int lin, lane, wave;
    bool done;
    volatile unsigned* wait_gen;   // blocked while *wait_gen == wait_val
    unsigned wait_val;
    unsigned wave_ops;             // number of wave collectives issued (buffer parity)
};

struct Block {
    dim3 bid, bdim, gdim;
    unsigned char* smem;
    int nthreads;
    Fiber* fibers;
    ucontext_t sched;
    int cur;
    int bar_count;
    volatile unsigned bar_gen;
    WaveX waves[kMaxThreads / kWave];
    const std::function<void()>* body;
};

Block*& tls_block();
void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);

inline Block* blk() { return tls_block(); }
inline Fiber* cur() { Block* b = tls_block(); return &b->fibers[b->cur]; }

inline void yield_to_sched() {
    Block* b = tls_block();
    Fiber* f = &b->fibers[b->cur];
    swapcontext(&f->ctx, &b->sched);
}

inline void wait_on(volatile unsigned* gen, unsigned val) {
    Fiber* f = cur();
    f->wait_gen = gen;
    f->wait_val = val;
    while (*gen == val) yield_to_sched();
    f->wait_gen = nullptr;
}

inline void syncthreads() {
    Block* b = blk();
    unsigned g = b->bar_gen;
    if (++b->bar_count == b->nthreads) {
        b->bar_count = 0;
        b->bar_gen = g + 1;
    } else {
        wait_on(&b->bar_gen, g);
    }
}

// deposit `bytes` of this lane's data, wait for the whole wave, return the 64-slot buffer
inline const unsigned char (*wave_exchange(const void* mine, size_t bytes))[kSlotBytes] {
    if (bytes > kSlotBytes) { fprintf(stderr, "hipemu: slot overflow\n"); abort(); }
    Block* b = blk();
    Fiber* f = cur();
    WaveX& w = b->waves[f->wave];
    int par = f->wave_ops & 1;
    f->wave_ops++;
    memcpy(w.buf[par][f->lane], mine, bytes);
    unsigned g = w.gen;
    if (++w.count == w.nlanes) {
        w.count = 0;
        w.gen = g + 1;
    } else {
        wait_on(&w.gen, g);
    }
    return w.buf[par];
}

template <typename T>
inline T shfl(T v, int src_lane) {
    auto buf = wave_exchange(&v, sizeof(T));
    T r;
    memcpy(&r, buf[src_lane & 63], sizeof(T));
    return r;
}
template <typename T>
inline T shfl_xor(T v, int mask) { return shfl(v, cur()->lane ^ mask); }
template <typename T>
inline T shfl_down(T v, int d) { int l = cur()->lane + d; return shfl(v, l > 63 ? cur()->lane : l); }

// ---- gfx950 MFMA lane layouts (cdna_hip_programming.md section 3) -------------------------
// 16x16x32 (bf16 in, f32 acc): A lane l holds A[i = l&15][k = (l>>4)*8 + j], j = 0..7
//                               B lane l holds B[k = (l>>4)*8 + j][n = l&15]
//                               C/D lane l reg r: row = (l>>4)*4 + r, col = l&15
struct MmaSlot { float a[8]; float b[8]; };
inline void mma_16x16x32(const float a[8], const float b[8], float c[4]) {
    MmaSlot s;
    for (int j = 0; j < 8; ++j) { s.a[j] = a[j]; s.b[j] = b[j]; }
    auto buf = wave_exchange(&s, sizeof(s));
    int lane = cur()->lane;
    int col = lane & 15, g = lane >> 4;
    for (int r = 0; r < 4; ++r) {
        int row = g * 4 + r;
        float acc = c[r];
        for (int gk = 0; gk < 4; ++gk) {
            const MmaSlot* sa = (const MmaSlot*)buf[row + 16 * gk];
            const MmaSlot* sb = (const MmaSlot*)buf[col + 16 * gk];
            for (int j = 0; j < 8; ++j) acc = fmaf(sa->a[j], sb->b[j], acc);
        }
        c[r] = acc;
    }
}

// ---- v_mfma_scale_f32_16x16x128_f8f6f4, both operands OCP e4m3 (layout probed on gfx950: tools/probes/fp8_mfma_probe.hip) ----
//   operand lane l holds row (A) / column (B) l & 15 and 32 bytes j = 0..31 with K index k = 64 (j >> 4) + 16 (l >> 4) + (j & 15)
//   scale register: byte 0 of lane 16 b + r = E8M0 scale (2^(s - 127)) of K block b = k >> 5 of row / column r
//   C/D as every 16x16 f32 MFMA: lane l reg i -> row 4 (l >> 4) + i, column l & 15
inline float e4m3_to_float(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x;
    if (e == 15 && m == 7) x = __builtin_nanf("");
    else if (e == 0) x = ldexpf((float)m, -9);
    else x = ldexpf(1.0f + (float)m / 8.0f, e - 7);
    return s ? -x : x;
}
struct MmaF8Slot { unsigned char a[32]; unsigned char b[32]; int sa, sb; };
inline void mma_scale_16x16x128_f8(const unsigned char a[32], const unsigned char b[32], int sa, int sb, float c[4]) {
    MmaF8Slot s;
    memcpy(s.a, a, 32); memcpy(s.b, b, 32); s.sa = sa; s.sb = sb;
    auto buf = wave_exchange(&s, sizeof(s));
    const int lane = cur()->lane, col = lane & 15, g = lane >> 4;
    for (int i = 0; i < 4; ++i) {
        const int row = g * 4 + i;
        double acc = 0.0;
        for (int q = 0; q < 4; ++q) {                       // the lane group that holds k = 64 (j >> 4) + 16 q + (j & 15)
            const MmaF8Slot* la = (const MmaF8Slot*)buf[row + 16 * q];
            const MmaF8Slot* lb = (const MmaF8Slot*)buf[col + 16 * q];
            for (int j = 0; j < 32; ++j) {
                const int k = 64 * (j >> 4) + 16 * q + (j & 15), blk = k >> 5;
                const int ea = ((const MmaF8Slot*)buf[16 * blk + row])->sa & 255, eb = ((const MmaF8Slot*)buf[16 * blk + col])->sb & 255;
                acc += (double)ldexpf(e4m3_to_float(la->a[j]), ea - 127) * (double)ldexpf(e4m3_to_float(lb->b[j]), eb - 127);
            }
        }
        c[i] = (float)((double)c[i] + acc);
    }
}
// round to nearest even onto the e4m3 grid (|x| <= 448; what v_cvt_pk_fp8_f32 does for in-range inputs)
inline unsigned char float_to_e4m3(float x) {
    const unsigned char sgn = x < 0.f || (x == 0.f && __builtin_signbit(x)) ? 0x80 : 0;
    float a = fabsf(x);
    if (!(a == a)) return sgn | 0x7f;
    if (a > 448.f) a = 448.f;
    if (a == 0.f) return sgn;
    int e;
    frexpf(a, &e);                                          // a = f 2^e, f in [0.5, 1)
    int ee = e - 1;                                         // a = (1 + m/8) 2^ee
    if (ee < -6) ee = -6;                                   // subnormal spacing 2^-9
    const float q = ldexpf(a, 3 - ee);                      // in units of 2^(ee - 3)
    float r = nearbyintf(q);                                // RNE (default rounding mode)
    int mant = (int)r, ex = ee + 7;
    if (ee == -6 && mant < 8) return sgn | (unsigned char)mant;          // subnormal (exponent field 0)
    if (mant == 16) { mant = 8; ex += 1; }
    return sgn | (unsigned char)((ex << 3) | (mant - 8));
}

inline float atomic_add_f32(float* p, float v) {
    unsigned* up = (unsigned*)p;
    unsigned old = __atomic_load_n(up, __ATOMIC_RELAXED);
    for (;;) {
        float f; memcpy(&f, &old, 4);
        float n = f + v; unsigned nu; memcpy(&nu, &n, 4);
        if (__atomic_compare_exchange_n(up, &old, nu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return f;
    }
}

}  // namespace hipemu

#define threadIdx (::hipemu::cur()->tid)
#define blockIdx (::hipemu::blk()->bid)
#define blockDim (::hipemu::blk()->bdim)
#define gridDim (::hipemu::blk()->gdim)
#define __syncthreads() ::hipemu::syncthreads()
#define __shfl_xor(v, m) ::hipemu::shfl_xor((v), (m))
#define __shfl_down(v, d) ::hipemu::shfl_down((v), (d))
#define __shfl(v, l) ::hipemu::shfl((v), (l))
inline float atomicAdd(float* p, float v) { return ::hipemu::atomic_add_f32(p, v); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
inline hipError_t hipMemcpyAsyncD2D(void* d, const void* s, size_t n, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
