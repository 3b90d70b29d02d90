// vb_rt.h -- device-runtime vocabulary shared by every kernel file.
//
// Product build (default): HIP for gfx950 only.  Wave = 64 lanes, MFMA builtins, LDS.
// -DVB_EMU: the same kernel sources compiled as host C++ against tests/hipemu (developer-only
//           kernel-logic simulator; never shipped, never loaded by the visualbert_amd package).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <type_traits>

#ifdef VB_EMU
#include "hipemu.h"
#define VB_KERNEL static void
#define VB_DEVICE static inline
#define VB_LAUNCH_BOUNDS(n)
#define VB_LAUNCH_BOUNDS2(n, w)
#define VB_DYN_SMEM(name) unsigned char* name = ::hipemu::blk()->smem
#define VB_LAUNCH(kernel, grid, block, smem, stream, ...) \
    ::hipemu::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })
#else
#include <hip/hip_runtime.h>
#define VB_KERNEL __global__ void
#define VB_DEVICE static __device__ __forceinline__
#define VB_LAUNCH_BOUNDS(n) __launch_bounds__(n)
#define VB_LAUNCH_BOUNDS2(n, w) __launch_bounds__(n, w)   /* w = minimum waves per SIMD */
// all LDS lives in the dynamic region, 16-byte aligned base (cdna guide, Guideline 17)
#define VB_DYN_SMEM(name)                                                            \
    extern __shared__ __attribute__((aligned(16))) unsigned char vb_dyn_smem_raw[];  \
    unsigned char* name = vb_dyn_smem_raw
#define VB_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), (smem), (stream), __VA_ARGS__)
#endif

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------
// Element-type traits: T in {float, bf16}.  A "fragment" is always 8 consecutive K elements
// per lane (16 B for bf16, 32 B for fp32) so that bf16 and fp32 kernels share ALL index maths.
// ------------------------------------------------------------------------------------------
template <typename T> struct VecOf;
template <> struct VecOf<float> { typedef f32x8 v8; typedef f32x4 v4; };
template <> struct VecOf<bf16> { typedef bf16x8 v8; typedef bf16x4 v4; };

VB_DEVICE float to_f32(float x) { return x; }
VB_DEVICE float to_f32(bf16 x) { return (float)x; }
VB_DEVICE void cvt_f32(float x, float& o) { o = x; }
VB_DEVICE void cvt_f32(float x, bf16& o) { o = (bf16)x; }                 // RNE; v_cvt_pk_bf16_f32 on gfx950
template <typename T> VB_DEVICE T from_f32(float x) { T o; cvt_f32(x, o); return o; }

// 8 consecutive elements <-> 8 floats (16-byte vector accesses; p must be 16-byte aligned)
VB_DEVICE void load8(float (&v)[8], const bf16* p) {
    bf16x8 x = *(const bf16x8*)p;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)x[j];
}
VB_DEVICE void load8(float (&v)[8], const float* p) {
    f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
}
VB_DEVICE void store8(bf16* p, const float (&v)[8]) {
    bf16x8 x;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (bf16)v[j];
    *(bf16x8*)p = x;
}
VB_DEVICE void store8(float* p, const float (&v)[8]) {
    *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
    *(f32x4*)(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

// Streaming ("nt") forms of store8 for results that this kernel will not read again and that are far larger than an XCD's 4 MB L2: the
// GEMM epilogues.  A written-back C tile otherwise sits in L2 as dirty lines and evicts the operand panels the K loops of the same
// XCD are streaming; with nt the lines are marked for early eviction.  Measured at M = 167,936 (profiles/r04_gemm_store_variants_b1024.txt):
// QKV forward 654 -> 556 us, attention-out 225 -> 185 us -- most of the way to the same kernels with the stores predicated off (543 / 180).
// (inline asm, not __builtin_nontemporal_store: when such a store sits in one arm of an if / else whose other arm also stores to the same
//  address -- the epilogue's ablation switch -- the optimiser merges the two and drops the nontemporal flag: the plain-epilogue kernels
//  came out with 0 of 16 stores marked.)
VB_DEVICE void vb_store16_nt(void* p, const u32x4& w) {
#ifdef VB_EMU
    *(u32x4*)p = w;
#else
    // s_nop 1: gfx940-family store-data hazard -- a VALU write to the data registers of a > 8-byte store within 2 wait states of its
    // issue corrupts the store (the compiler pads its own stores; it cannot see inside an asm statement: without the nop every GEMM test
    // failed with garbage results)
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(p), "v"(w) : "memory");
#endif
}
VB_DEVICE void store8_nt(bf16* p, const float (&v)[8]) {
    bf16x8 x;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (bf16)v[j];
    vb_store16_nt(p, *(const u32x4*)&x);
}
VB_DEVICE void store8_nt(float* p, const float (&v)[8]) {
    const f32x4 a = f32x4{v[0], v[1], v[2], v[3]}, b = f32x4{v[4], v[5], v[6], v[7]};
    vb_store16_nt(p, *(const u32x4*)&a);
    vb_store16_nt(p + 4, *(const u32x4*)&b);
}
// split-operand images (VB_BF16X3): x -> hi = bf16(x) (RNE) at p[...] and lo = bf16(x - hi) at p[half + ...] -- bit for bit what
// vb_split_bf16 writes, so a producer kernel can emit the image of its fp32 result itself (one extra 4-byte-per-element store
// instead of a separate read-4 / write-4 pass)
VB_DEVICE void store_split8(bf16* p, long half, const float (&v)[8]) {
    bf16x8 h;
    float lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { h[j] = (bf16)v[j]; lo[j] = v[j] - (float)h[j]; }
    *(bf16x8*)p = h;
    store8(p + half, lo);
}
VB_DEVICE void store_split4(bf16* p, long half, const f32x4& v) {
    bf16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = (bf16)v[j]; l[j] = (bf16)(v[j] - (float)h[j]); }
    *(bf16x4*)p = h;
    *(bf16x4*)(p + half) = l;
}

// ------------------------------------------------------------------------------------------
// MFMA: one 16x16 output fragment, K step of 32.
//   A operand: lane l holds A[i = l&15][k = (l>>4)*8 + j], j = 0..7
//   B operand: lane l holds B[k = (l>>4)*8 + j][n = l&15]
//   C/D      : lane l, reg r -> row = (l>>4)*4 + r, col = l&15
// bf16: v_mfma_f32_16x16x32_bf16.  fp32: eight chained v_mfma_f32_16x16x4_f32 -- instruction j
// consumes element j of both fragments, i.e. its hardware k index (l>>4) stands for the true
// k = (l>>4)*8 + j; the pairing of A and B elements is identical to the bf16 form, and the result
// is an exact fp32 fma chain (guide section 3 "FP32-input MFMA").
// ------------------------------------------------------------------------------------------
#ifdef VB_EMU
VB_DEVICE f32x4 vb_mma(bf16x8 a, bf16x8 b, f32x4 c) {
    float fa[8], fb[8], fc[4];
    for (int j = 0; j < 8; ++j) { fa[j] = (float)a[j]; fb[j] = (float)b[j]; }
    for (int r = 0; r < 4; ++r) fc[r] = c[r];
    ::hipemu::mma_16x16x32(fa, fb, fc);
    f32x4 o; for (int r = 0; r < 4; ++r) o[r] = fc[r];
    return o;
}
VB_DEVICE f32x4 vb_mma(f32x8 a, f32x8 b, f32x4 c) {
    float fa[8], fb[8], fc[4];
    for (int j = 0; j < 8; ++j) { fa[j] = a[j]; fb[j] = b[j]; }
    for (int r = 0; r < 4; ++r) fc[r] = c[r];
    ::hipemu::mma_16x16x32(fa, fb, fc);
    f32x4 o; for (int r = 0; r < 4; ++r) o[r] = fc[r];
    return o;
}
#else
VB_DEVICE f32x4 vb_mma(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
VB_DEVICE f32x4 vb_mma(f32x8 a, f32x8 b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
    return c;
}
#endif

// ------------------------------------------------------------------------------------------
// Block-scaled fp8 MFMA, K step of 128: v_mfma_scale_f32_16x16x128_f8f6f4 with both operands OCP e4m3 (the pipe the two cross terms
// of the split-operand product can ride: DESIGN.md section 7 (1); nothing in the product uses it yet).  None of this is in the guides;
// probed on the device (tools/probes/fp8_mfma_probe.hip, profiles/r04_fp8_mfma_probe.txt):
//   A / B operand: lane l holds row / column l & 15 and 32 bytes, byte j <-> k = 64 (j >> 4) + 16 (l >> 4) + (j & 15)
//   scale_a / scale_b: byte 0 of LANE 16 b + r is the E8M0 scale 2^(s - 127) of K block b = k >> 5 of row / column r
//   C / D: lane l reg i -> row 4 (l >> 4) + i, column l & 15 (as vb_mma)
// vb_cvt4_fp8: four floats -> four e4m3 bytes (round to nearest even; callers scale into |x| <= 448 first)
// ------------------------------------------------------------------------------------------
typedef int i32x8 __attribute__((ext_vector_type(8)));
#ifdef VB_EMU
VB_DEVICE f32x4 vb_mma_f8(i32x8 a, i32x8 b, f32x4 c, int scale_a, int scale_b) {
    float fc[4];
    for (int r = 0; r < 4; ++r) fc[r] = c[r];
    ::hipemu::mma_scale_16x16x128_f8((const unsigned char*)&a, (const unsigned char*)&b, scale_a, scale_b, fc);
    f32x4 o; for (int r = 0; r < 4; ++r) o[r] = fc[r];
    return o;
}
// OA / OB: which BYTE of the scale registers holds this fragment's scale (an immediate of the instruction): four fragments share a register
template <int OA, int OB> VB_DEVICE f32x4 vb_mma_f8_op(i32x8 a, i32x8 b, f32x4 c, int scale_a, int scale_b) {
    return vb_mma_f8(a, b, c, (int)((unsigned)scale_a >> (8 * OA)), (int)((unsigned)scale_b >> (8 * OB)));
}
VB_DEVICE uint32_t vb_cvt4_fp8(float x0, float x1, float x2, float x3) {
    return (uint32_t)::hipemu::float_to_e4m3(x0) | ((uint32_t)::hipemu::float_to_e4m3(x1) << 8) |
           ((uint32_t)::hipemu::float_to_e4m3(x2) << 16) | ((uint32_t)::hipemu::float_to_e4m3(x3) << 24);
}
#else
VB_DEVICE f32x4 vb_mma_f8(i32x8 a, i32x8 b, f32x4 c, int scale_a, int scale_b) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
}
template <int OA, int OB> VB_DEVICE f32x4 vb_mma_f8_op(i32x8 a, i32x8 b, f32x4 c, int scale_a, int scale_b) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OA, scale_a, OB, scale_b);
}
VB_DEVICE uint32_t vb_cvt4_fp8(float x0, float x1, float x2, float x3) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(x0, x1, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(x2, x3, w, true);
    return (uint32_t)w;
}
#endif

// ------------------------------------------------------------------------------------------
// Direct global -> LDS copy, 16 bytes per lane (global_load_lds_dwordx4): the source address is per
// lane, the destination is `lds_wave_base + lane * 16` with a WAVE-UNIFORM base (it travels in M0).
// Data is visible to LDS readers after the wave's vmcnt drains and a workgroup barrier
// (hipcc's __syncthreads() emits the vmcnt(0) while such a load is in flight).
// ------------------------------------------------------------------------------------------
#ifdef VB_EMU
VB_DEVICE void vb_glds16(const void* gsrc, unsigned char* lds_wave_base) {
    memcpy(lds_wave_base + ::hipemu::cur()->lane * 16, gsrc, 16);
}
VB_DEVICE void vb_atomic_add_noret(float* p, float v) { ::hipemu::atomic_add_f32(p, v); }
VB_DEVICE void vb_lds_add(float* p, float v) { ::hipemu::atomic_add_f32(p, v); }
#else
VB_DEVICE void vb_glds16(const void* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
VB_DEVICE void vb_atomic_add_noret(float* p, float v) { unsafeAtomicAdd(p, v); }   // global_atomic_add_f32, no return
// fp32 add into LDS (ds_add_f32): p must point into the workgroup's LDS
VB_DEVICE void vb_lds_add(float* p, float v) {
    __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)p, v, 0, 0, false);
}
#endif

// The same copy through a BUFFER descriptor (buffer_load_dwordx4 ... offen lds): address = descriptor base (4 SGPRs) + per-lane
// 32-bit byte offset (VGPR) + wave-uniform 32-bit byte offset (SGPR).  A K tile advances the SGPR offset, so issuing a copy
// costs no vector arithmetic at all -- the flat form above needs a 64-bit per-lane add unless the compiler happens to match
// its scalar-base addressing mode.  Raw buffer (stride 0) over the whole 32-bit range: callers keep offsets below 2^32.
#ifdef VB_EMU
struct vb_buf { const unsigned char* base; };
VB_DEVICE vb_buf vb_make_buf(const void* p) { return vb_buf{(const unsigned char*)p}; }
VB_DEVICE void vb_glds16_buf(vb_buf b, unsigned voff, unsigned soff, unsigned char* lds_wave_base) {
    memcpy(lds_wave_base + ::hipemu::cur()->lane * 16, b.base + voff + soff, 16);
}
VB_DEVICE void vb_glds16_buf_nt(vb_buf b, unsigned voff, unsigned soff, unsigned char* lds_wave_base) { vb_glds16_buf(b, voff, soff, lds_wave_base); }
#else
typedef __amdgpu_buffer_rsrc_t vb_buf;
VB_DEVICE vb_buf vb_make_buf(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, -1, 0x00020000);      // gfx9-family raw buffer, 2^32 - 1 records
}
VB_DEVICE void vb_glds16_buf(vb_buf b, unsigned voff, unsigned soff, unsigned char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}
// the same copy marked streaming (aux bit 1 = nt): developer experiment -- activations read once per column-tile pass need not displace the weights
VB_DEVICE void vb_glds16_buf_nt(vb_buf b, unsigned voff, unsigned soff, unsigned char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 2);
}
#endif

// 16 bytes per lane through the same descriptor into REGISTERS (buffer_load_dwordx4 ... offen)
#ifdef VB_EMU
VB_DEVICE u32x4 vb_buf_load16(vb_buf b, unsigned voff, unsigned soff) { u32x4 v; memcpy(&v, b.base + voff + soff, 16); return v; }
#else
VB_DEVICE u32x4 vb_buf_load16(vb_buf b, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(b, (int)voff, (int)soff, 0));
}
#endif

// 16 bytes per lane through a descriptor with the sc1 (write-through / L1-bypassing) policy: the hand-off form for tens of KB per
// workgroup (guide, price list "publish-large": an agent-scope release after plain stores writes back the whole XCD L2's dirty lines,
// 8.2 us per 64 KB publish; write-through stores + a drained vmcnt need no release, 3.0 us) -- and sc1 loads on the consumer side
// need no acquire (they are served by L2 / the fabric, never by this CU's possibly stale L1)
#ifdef VB_EMU
VB_DEVICE void vb_buf_store16_sc1(vb_buf b, unsigned voff, unsigned soff, const u32x4& v) { memcpy((unsigned char*)b.base + voff + soff, &v, 16); }
VB_DEVICE u32x4 vb_buf_load16_sc1(vb_buf b, unsigned voff, unsigned soff) { u32x4 v; memcpy(&v, b.base + voff + soff, 16); return v; }
#else
VB_DEVICE void vb_buf_store16_sc1(vb_buf b, unsigned voff, unsigned soff, const u32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, b, (int)voff, (int)soff, 16);
}
VB_DEVICE u32x4 vb_buf_load16_sc1(vb_buf b, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(b, (int)voff, (int)soff, 16));
}
#endif

// counted wait for outstanding vector-memory operations (LDS-direct copies included) + raw workgroup
// barrier: lets the newest K tile(s) stay in flight across the barrier (guide: "Pipelining across
// barriers").  N must be an immediate.
#ifdef VB_EMU
template <int N> VB_DEVICE void vb_wait_vmcnt() {}
VB_DEVICE void vb_wait_vmcnt0_visible() {}
VB_DEVICE void vb_raw_barrier() { __syncthreads(); }
#else
template <int N> VB_DEVICE void vb_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// vmcnt(0) the COMPILER can see (builtin, gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15): its scoreboard is empty afterwards, so it
// does not re-wait -- conservatively, with vmcnt(0), i.e. for every store issued since -- at later uses of values loaded before it
VB_DEVICE void vb_wait_vmcnt0_visible() { __builtin_amdgcn_s_waitcnt(0x0F70); }
VB_DEVICE void vb_raw_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#endif

#ifdef VB_EMU
VB_DEVICE void vb_wait_lgkmcnt0() {}
VB_DEVICE void vb_sched_fence() {}
VB_DEVICE void vb_phase_barrier() { __syncthreads(); }
template <int P> VB_DEVICE void vb_setprio() {}
#else
VB_DEVICE void vb_wait_lgkmcnt0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// nothing may be scheduled across this point (LLVM sched_barrier 0)
VB_DEVICE void vb_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// workgroup barrier that the instruction scheduler may not move code across (the GEMM kernels' phase boundaries)
VB_DEVICE void vb_phase_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
// wave priority for the instruction arbiter of its SIMD (s_setprio): raised around MFMA blocks
template <int P> VB_DEVICE void vb_setprio() { __builtin_amdgcn_s_setprio(P); }
#endif

// The four-wave 256x256 GEMM (gemm.hip, nt_kernel 100) keeps 256 accumulator registers in the AGPR half of the file and
// interleaves its K loop by hand: MFMA with the accumulator pinned ("+a"), ds_read_b128 with an immediate offset, and the
// settling gap MFMA results need before ordinary code reads them.  The simulator executes the same steps synchronously.
#ifdef VB_EMU
#define VB_BIG_MMA(acc, a, b) acc = vb_mma(a, b, acc)
template <int OFF> VB_DEVICE void big_read(bf16x8& d, const unsigned char* smem, unsigned off) { d = *(const bf16x8*)(smem + off + OFF); }
VB_DEVICE unsigned big_lds_base(const unsigned char*) { return 0u; }
VB_DEVICE void big_settle() {}
#else
#define VB_BIG_MMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
template <int OFF> VB_DEVICE void big_read(bf16x8& d, const unsigned char*, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "DS offsets are 16 bits");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
VB_DEVICE unsigned big_lds_base(const unsigned char* smem) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)smem;
}
VB_DEVICE void big_settle() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }   // MFMA results / AGPR writes visible to what follows
#endif

// ds_read_b64_tr_b16 (gfx950): within each 16-lane group, lane s passes the address of 4 contiguous 16-bit values
// M[s][0..3]; lane l receives { M[(l>>2) + 4 j][l & 3] : j = 0..3 }  (measured: profiles/r01_ds_read_tr_probe.txt).
// With lane s pointing at T[k0 + (s>>2)][r0 + 4 (s&3) ..] of a K-major tile T[k][row] -- a [4 k][16 row] block --
// lane l gets T[k0 + j][r0 + l], j = 0..3: four consecutive k of ONE row, i.e. half of a 16x16x32 MFMA operand
// fragment, from a tile that was never transposed in memory.
#ifdef VB_EMU
VB_DEVICE bf16x4 vb_lds_read_tr(const unsigned char* p) {
    uint64_t mine;
    memcpy(&mine, p, 8);
    auto slots = ::hipemu::wave_exchange(&mine, 8);
    const int l = ::hipemu::cur()->lane, g16 = l & ~15, d = (l & 15) >> 2, c = l & 3;
    bf16x4 out;
    for (int j = 0; j < 4; ++j) { bf16 v; memcpy(&v, slots[g16 + d + 4 * j] + c * 2, 2); out[j] = v; }
    return out;
}
#else
VB_DEVICE bf16x4 vb_lds_read_tr(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
}
#endif

// The two transposing reads of one operand fragment (k + 0..3 -> lo, k + 4..7 -> hi, 2 KB apart) at p + OFF, as INLINE ASM.
// Why not the builtin above: hipcc 7.2 puts `s_waitcnt vmcnt(0)` in front of every builtin transposing read that follows an
// LDS-direct copy (it cannot tell the copy's destination from the read's source; plain ds_read_b128 loads do not get that
// wait) -- in a pipelined K loop that drains the whole copy stream twice per K tile.  The asm form is invisible to that
// pass AND to its lgkmcnt bookkeeping: the caller must execute `s_waitcnt lgkmcnt(0)` (vb_raw_barrier does) before the
// first use of lo / hi, and must not combine or move them before that point.
#ifdef VB_EMU
template <int OFF> VB_DEVICE void vb_lds_read_tr_pair(bf16x4& lo, bf16x4& hi, const unsigned char* p) {
    lo = vb_lds_read_tr(p + OFF);
    hi = vb_lds_read_tr(p + OFF + 2048);
}
#else
template <int OFF> VB_DEVICE void vb_lds_read_tr_pair(bf16x4& lo, bf16x4& hi, const unsigned char* p) {
    static_assert(OFF >= 0 && OFF + 2048 < 65536, "DS instruction offsets are 16 bits");
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)p;
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                 : "=&v"(lo), "=&v"(hi) : "v"(a), "n"(OFF), "n"(OFF + 2048));
}
#endif
VB_DEVICE bf16x8 vb_join(bf16x4 lo, bf16x4 hi) { return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; }

// compile-time loop: f(std::integral_constant<int, I>()) for I = 0 .. N - 1 (indices usable as template arguments / asm immediates)
template <int I, int N, typename F> VB_DEVICE void vb_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>()); vb_static_for<I + 1, N>(f); }
}

// x(lane) + x(lane ^ 32): the two 32-lane halves of a wave exchanged with ONE VALU instruction (v_permlane32_swap,
// gfx950) instead of an LDS-crossbar shuffle
#ifdef VB_EMU
VB_DEVICE float vb_pair_sum32(float x) { return x + __shfl_xor(x, 32); }
#else
VB_DEVICE float vb_pair_sum32(float x) {
    // v_permlane32_swap_b32 a, b swaps a[32..63] with b[0..31]: with a = b = x it leaves a = {lo|lo}, b = {hi|hi}.
    // Inline asm because hipcc 7.2 mis-folds __builtin_amdgcn_permlane32_swap's second result into the first (the ISA
    // showed r[0] + r[0]; caught by the LayerNorm parity tests on the device).  hipcc does not resolve hazards around an
    // asm statement: the s_nop pair covers VALU-write -> permlane read and permlane write -> VALU read.
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
#endif

// v_permlane16_swap_b32 a, b (gfx950): the ODD 16-lane rows of a are exchanged with the EVEN rows of b (a.row1 <-> b.row0,
// a.row3 <-> b.row2); the other rows stay.  Used to turn two 8-byte-per-lane results of an MFMA-layout epilogue (lane (li, lg) holds
// columns 4 lg .. 4 lg + 3 of two adjacent 16-column blocks) into ONE 16-byte-per-lane store: a store instruction costs the CU's
// vector-memory path ~64 cycles whatever its width (guide T21), so halving the instruction count halves the store tail.
// Inline asm for the same reason as vb_pair_sum32 (the builtin's second result is mis-folded by hipcc 7.2); s_nop pads cover the
// VALU-write -> permlane-read and permlane-write -> VALU-read hazards the compiler cannot see inside an asm statement.
#ifdef VB_EMU
VB_DEVICE void vb_permlane16_swap(uint32_t& a, uint32_t& b) {
    const bool odd = ((::hipemu::cur()->lane >> 4) & 1) != 0;
    const uint32_t pa = (uint32_t)__shfl_xor((int)a, 16), pb = (uint32_t)__shfl_xor((int)b, 16);
    if (odd) a = pb; else b = pa;
}
#else
VB_DEVICE void vb_permlane16_swap(uint32_t& a, uint32_t& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
#endif
// element offset of the 8 contiguous columns lane group lg owns after vb_permlane16_swap of blocks (2j, 2j + 1), inside the 32 columns
VB_DEVICE int vb_wide_col(int lg) { return (lg & 1) * 16 + (lg >> 1) * 8; }

// Inter-workgroup hand-off through a device-scope counter (guide, Guideline 16 / "in-launch split-K reduction"), write-through form:
// the producer's payload leaves by sc1 stores (vb_buf_store16_sc1), every wave drains vmcnt, workgroup barrier, ONE lane draws a relaxed
// agent-scope ticket; the consumer reads the payload with sc1 loads.  Placement-independent; no fences.
#ifdef VB_EMU
VB_DEVICE int vb_ticket_add(int* p) { return __atomic_fetch_add(p, 1, __ATOMIC_SEQ_CST); }
VB_DEVICE void vb_ticket_reset(int* p) { __atomic_store_n(p, 0, __ATOMIC_SEQ_CST); }
#else
VB_DEVICE int vb_ticket_add(int* p) { return __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
VB_DEVICE void vb_ticket_reset(int* p) { __hip_atomic_store(p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

// value known to be identical in every lane of the wave: keep it in a scalar register (addresses built from it
// become scalar arithmetic instead of per-lane VALU + v_readfirstlane at every use)
#ifdef VB_EMU
VB_DEVICE int vb_uniform(int v) { return v; }
#else
VB_DEVICE int vb_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif

// a pointer known to be wave-uniform, made OPAQUE to the optimiser as a scalar-register pair: `p + zext(per-lane 32-bit
// offset)` then selects the scalar-base + vector-offset form of global_load / global_load_lds instead of 64-bit per-lane
// address arithmetic (LLVM otherwise re-associates base + k * stride + lane_offset into per-lane 64-bit induction variables)
#ifdef VB_EMU
VB_DEVICE const unsigned char* vb_uniform_ptr(const unsigned char* p) { return p; }
#else
VB_DEVICE const unsigned char* vb_uniform_ptr(const unsigned char* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (const unsigned char*)(((uint64_t)hi << 32) | lo);
}
#endif

// wave-level ordering point for data exchanged through LDS by the lanes of ONE wave (no workgroup barrier):
// the LDS unit executes a wave's instructions in order, so only the compiler has to be held back.
#ifdef VB_EMU
VB_DEVICE void vb_wave_sync() { (void)::hipemu::shfl(0, 0); }
VB_DEVICE int vb_num_cus() { return 4; }
#else
VB_DEVICE void vb_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
static inline int vb_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}
#endif

// shader clock (cycles) for in-kernel timelines (measurement builds only)
#ifdef VB_EMU
VB_DEVICE unsigned long long vb_clock() { return 0ull; }
#else
VB_DEVICE unsigned long long vb_clock() { return __builtin_readcyclecounter(); }
#endif

// instruction-order hints for the LLVM scheduler (guide T19): emit `n` instructions of class `mask` next.
// masks: 0x8 MFMA, 0x20 VMEM read, 0x100 DS read.  No-ops in the simulator build.
#ifdef VB_EMU
#define VB_SCHED_GROUP(mask, n) do {} while (0)
#else
#define VB_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
#endif

// ------------------------------------------------------------------------------------------
// wave (64-lane) reductions
// ------------------------------------------------------------------------------------------
VB_DEVICE float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
VB_DEVICE float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
// sum over the 16 lanes of a DPP row (lanes 16 i .. 16 i + 15), result in every lane of the row: four row-rotate adds on
// the VALU (v_add_f32 ... row_ror) instead of four LDS-crossbar shuffles (ds_bpermute + lgkmcnt wait each)
#ifdef VB_EMU
VB_DEVICE float row16_sum(float v) {
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}
#else
template <int N> VB_DEVICE float vb_row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xF, 0xF, false));
}
VB_DEVICE float row16_sum(float v) {
    v += vb_row_ror<8>(v); v += vb_row_ror<4>(v); v += vb_row_ror<2>(v); v += vb_row_ror<1>(v);
    return v;
}
#endif

// "the value becomes available HERE": an empty volatile asm that redefines its operand.  Consumers of a register that a
// prefetch load fills cannot be scheduled above it, so the load's s_waitcnt lands at the pin (after the work the load was
// meant to run under) instead of right behind the load.
#ifdef VB_EMU
template <typename V> VB_DEVICE void vb_pin(V&) {}
VB_DEVICE void vb_pin_s(int&) {}
#else
template <typename V> VB_DEVICE void vb_pin(V& v) { asm volatile("" : "+v"(v)); }
VB_DEVICE void vb_pin_s(int& v) { asm volatile("" : "+s"(v)); }        // the same for a wave-uniform (scalar register) value
#endif

// sum over aligned groups of 8 lanes (quad swaps + half-row mirror on the VALU, no LDS-crossbar shuffles)
#ifdef VB_EMU
VB_DEVICE float oct_sum(float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); return v; }
#else
template <int CTRL> VB_DEVICE float vb_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
VB_DEVICE float oct_sum(float v) {
    v += vb_dpp<0xB1>(v);                                  // quad_perm [1 0 3 2]
    v += vb_dpp<0x4E>(v);                                  // quad_perm [2 3 0 1]
    v += vb_dpp<0x141>(v);                                 // row_half_mirror: lane i <-> 7 - i of its group of 8
    return v;
}
#endif

// reduction across the 32 lanes of a half-wave (lanes [0,32) and [32,64) independently)
VB_DEVICE float half_sum(float v) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// ------------------------------------------------------------------------------------------
// Counter-based random bits for dropout.  One call yields 128 bits = eight 16-bit uniforms for the 8 consecutive
// elements of "group"; element e is dropped iff u16[e] < thresh16, thresh16 = round(p * 65536).  Any kernel (forward
// or backward, any thread mapping) regenerates the same mask from (seed, stream, element index) -- no mask tensor
// in HBM.  Two words come from a KEYED 32-bit mixer (two multiplies per word, key material folded in before the first
// and between the two multiplies; the key schedule is wave-uniform, i.e. scalar); the other two are each ONE full-rate
// 24-bit multiply of one mixed word XORed with the other (w2 = f(w0) ^ w1, w3 = g(w1) ^ w0: every word uniform and
// pairwise independent of every other, which is all a Bernoulli mask needs).  32-bit multiplies are quarter-rate on CDNA:
// Philox4x32-10 needs 40 per 8 elements (27 % of the attention forward kernel in round 1), four mixer words 8 (dropout was
// still ~half of that kernel's VALU work), this form 4.  Keep rates, pair / lag / stream / seed correlations, bucket
// chi-squares and bit balance of old and new form are indistinguishable on 2^21 groups (tools/dropout_rng_check.py);
// the keep fraction and the forward/backward agreement are tested on the device (tests/test_kernels.py).
// ------------------------------------------------------------------------------------------
struct Rand8 { uint32_t w[4]; };               // eight 16-bit uniforms
static inline uint32_t vb_drop_thresh16(float p) {                         // host side: round(p * 65536), at most 65535
    const uint32_t t = (uint32_t)(p * 65536.0f + 0.5f);
    return t > 65535u ? 65535u : t;
}
VB_DEVICE uint32_t vb_mix32(uint32_t x, uint32_t k) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= k;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
#ifdef VB_EMU
VB_DEVICE uint32_t vb_mul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
#else
VB_DEVICE uint32_t vb_mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }       // v_mul_u32_u24: full rate
#endif
VB_DEVICE Rand8 vb_dropout_bits8(uint64_t seed, uint64_t group, uint32_t stream) {
    const uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
    const uint32_t k1 = vb_mix32(s0 ^ (stream * 0x9E3779B9u), s1 ^ 0x5ca1ab1eu);           // uniform: scalar unit
    const uint32_t k2 = vb_mix32(s1 + stream * 0x85EBCA6Bu, k1) ^ ((uint32_t)(group >> 31) * 0xC2B2AE35u);
    const uint32_t c = (uint32_t)group * 2u + k1;
    Rand8 o;
    o.w[0] = vb_mix32(c, k2);
    o.w[1] = vb_mix32(c + 1u, k2);
    o.w[2] = vb_mul24(o.w[0] >> 8, 0x9E3779u) ^ o.w[1];
    o.w[3] = vb_mul24(o.w[1] >> 8, 0x85EBCBu) ^ o.w[0];
    return o;
}
// keep-mask bit e (0..7) of a group.  thresh16 <= 65535 (vb_drop_thresh16), so the upper half of a word is tested without
// extracting it: (w >> 16) >= t  <=>  w >= (t << 16)
VB_DEVICE bool rand8_keep(const Rand8& r, int e, uint32_t thresh16) {
    const uint32_t w = r.w[e >> 1];
    return (e & 1) ? (w >= (thresh16 << 16)) : ((w & 0xFFFFu) >= thresh16);
}

// fast exp: one v_exp_f32 (2^x) after a multiply; relative error ~1e-7..1e-6, flushes like expf for the
// ranges used here (softmax arguments <= 0, Gaussian tails)
#ifdef VB_EMU
VB_DEVICE float fast_exp(float x) { return expf(x); }
VB_DEVICE float fast_exp2(float x) { return exp2f(x); }
VB_DEVICE float fast_rcp(float x) { return 1.0f / x; }
#else
VB_DEVICE float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
VB_DEVICE float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
VB_DEVICE float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#endif

// erf via Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7): 1 rcp + 1 exp + ~12 FMAs instead of the
// ~40-instruction branchy libm erff -- the GELU / GELU' epilogues run this 64x per lane per tile.
VB_DEVICE float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = fast_rcp(1.0f + 0.3275911f * ax);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const float r = 1.0f - p * t * fast_exp(-ax * ax);
    return x < 0.f ? -r : r;
}
// Phi(x) and phi(x) of the standard normal with ONE exponential: erf(x/sqrt2) = 1 - poly(t) exp(-x^2/2) and
// phi(x) = exp(-x^2/2)/sqrt(2 pi) share it.
VB_DEVICE void gelu_parts(float x, float& cdf, float& pdf) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float t = fast_rcp(1.0f + 0.3275911f * ax);
    const float e = fast_exp(-0.5f * x * x);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const float r = 1.0f - p * t * e;                 // erf(|x| / sqrt 2)
    cdf = 0.5f * (1.0f + (x < 0.f ? -r : r));
    pdf = 0.39894228040143267794f * e;
}
// Two elements at a time on the packed-fp32 VALU (v_pk_fma_f32 / v_pk_mul_f32: two lanes of work per issue slot):
// the same A&S 7.1.26 evaluation as gelu_parts written with explicit fused multiply-adds -- 17 packed operations + 2 rcp
// + 2 exp per PAIR where the scalar form issues ~25 full-rate operations + rcp + exp per ELEMENT (the build uses
// -ffp-contract=off).  The FFN-in epilogue runs this 128x per lane per tile: it was ~40 % of that GEMM's tile time.
VB_DEVICE f32x2 vb_fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
VB_DEVICE f32x2 vb_splat2(float c) { return f32x2{c, c}; }
VB_DEVICE void gelu_and_grad2(f32x2 x, f32x2& y, f32x2& dy) {
    const f32x2 ax = __builtin_elementwise_abs(x);
    const f32x2 den = vb_fma2(ax, vb_splat2(0.3275911f * 0.70710678118654752440f), vb_splat2(1.0f));
    const f32x2 t = f32x2{fast_rcp(den[0]), fast_rcp(den[1])};
    const f32x2 h = x * vb_splat2(-0.5f * 1.44269504088896340736f);
    const f32x2 hx = h * x;                                   // -x^2/2 in units of log 2
    const f32x2 e = f32x2{fast_exp2(hx[0]), fast_exp2(hx[1])};
    f32x2 p = vb_fma2(t, vb_splat2(1.061405429f), vb_splat2(-1.453152027f));
    p = vb_fma2(p, t, vb_splat2(1.421413741f));
    p = vb_fma2(p, t, vb_splat2(-0.284496736f));
    p = vb_fma2(p, t, vb_splat2(0.254829592f));
    const f32x2 pt = p * t;
    const f32x2 r = vb_fma2(-pt, e, vb_splat2(1.0f));         // erf(|x| / sqrt 2)
    const f32x2 sr = f32x2{__builtin_copysignf(r[0], x[0]), __builtin_copysignf(r[1], x[1])};
    const f32x2 cdf = vb_fma2(sr, vb_splat2(0.5f), vb_splat2(0.5f));
    const f32x2 xpdf = x * (e * vb_splat2(0.39894228040143267794f));
    y = x * cdf;
    dy = cdf + xpdf;
}
VB_DEVICE float gelu_f(float x) { float c, d; gelu_parts(x, c, d); return x * c; }
// d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
VB_DEVICE float gelu_grad_f(float x) { float c, d; gelu_parts(x, c, d); return c + x * d; }
VB_DEVICE void gelu_and_grad_f(float x, float& y, float& dy) { float c, d; gelu_parts(x, c, d); y = x * c; dy = c + x * d; }

#define VB_OK 0
#define VB_ERR_ARG (-1)
#define VB_ERR_LAUNCH (-2)
#define VB_ERR_UNSUPPORTED (-3)

// a plain K-contiguous x K-contiguous bf16 GEMM handed to the vendor library (vendor_gemm.hip; vb_stream_opts.nt_kernel = 200)
struct VbVendorGemm {
    const void* A; long lda; const void* B; long ldb; void* C; long ldc;
    int M, N, K; float alpha; const float* bias; const void* addend; long ld_addend; int out_f32;
};
#ifdef VB_EMU
static inline int vb_vendor_nt(const VbVendorGemm&, void*) { return VB_ERR_UNSUPPORTED; }   // no vendor library in the simulator
#else
__attribute__((visibility("hidden"))) int vb_vendor_nt(const VbVendorGemm& g, void* stream);   // internal: not part of the C ABI
#endif

// internal forms of vb_ln_fwd / vb_ln_bwd / vb_attn_fwd / vb_attn_bwd used by layer.hip in the split-operand mode (fp32 tensors
// only): the same call + the bf16 hi | lo image of the result that the next GEMM reads -- y_split [M, ld] (ld >= 2 H), dx_split
// likewise, ctx_split [B S, 2 H], dqkv_split [B S, 6 H] -- written by the producing kernel.  NULL image = the exported call.
// split_only (attention) / dx == NULL (LayerNorm backward): the fp32 form of that result is NOT written at all -- in the layer only
// GEMMs consume it, and they read the image; the q | k | v bias gradient is then summed from the image's two planes.
// rebuild (LayerNorm, device int): the forward skips z_out when the backward can take x-hat from y (layernorm.hip: ln_rebuildable) and
// records its decision there; the backward (every element type at H <= 768; fp32 uses a Newton-refined 1 / gamma) is handed y, beta and the
// same int.  gamma / beta must still hold the forward's values at backward time: a graph kept alive ACROSS an optimizer step (retain_graph) is
// not supported -- like every weight of the layer, whose bf16 shadows the backward GEMMs read ("Contracts").
__attribute__((visibility("hidden"))) int vb_ln_fwd_sp(int dtype, const void* x, const void* resid, void* z_out, void* y, float* mean,
    float* rstd, const float* gamma, const float* beta, int M, int H, float eps, float p_in, uint32_t stream_in, float p_out,
    uint32_t stream_out, uint64_t seed, void* y_split, int64_t ld_split, int* rebuild, void* stream);
__attribute__((visibility("hidden"))) int vb_ln_bwd_sp(int dtype, const void* dy, const void* z, const float* mean, const float* rstd,
    const float* gamma, void* dz, void* dx, float* dgamma, float* dbeta, float* dbias, int M, int H, float p_in, uint32_t stream_in,
    float p_out, uint32_t stream_out, uint64_t seed, float* ws, void* dx_split, int64_t ld_split, const void* y, const float* beta,
    const int* rebuild, void* stream);
__attribute__((visibility("hidden"))) int vb_attn_fwd_sp(int dtype, const void* qkv, const float* mask_add, void* ctx, float* lse,
    uint64_t* keepbits, int B, int S, int nh, int head_dim, float p_drop, uint64_t seed, uint32_t stream_id, void* ctx_split,
    int split_only, void* stream);
__attribute__((visibility("hidden"))) int vb_attn_bwd_sp(int dtype, const void* qkv, const float* mask_add, const void* dctx,
    const float* lse, const uint64_t* keepbits, float* dsum_ws, void* dqkv, const void* ctx_fwd, float* dqkv_bias, int B, int S, int nh,
    int head_dim, float p_drop, uint64_t seed, uint32_t stream_id, void* dqkv_split, int split_only, void* stream);

// Second-stage column reductions -- dst[c] += sum over rows r of src[r * stride + c]: the LayerNorm backward's dgamma | dbeta | dbias
// partials, the one-pass attention backward's per-sample bias-gradient sums -- deferred to ONE launch at the end of an encoder layer's
// backward (vb_bert_layer_bwd) instead of one launch each behind their producers: at small per-GPU batches the step is a chain of
// dependent 5-30 us kernels and these three 5 us launches per layer (+ their boundaries) are ~3 % of it.  While vb_reduce_defer_slot() points
// at a list with room, vb_ln_bwd_sp / vb_attn_bwd_sp append their reductions there (their workspaces must then stay untouched until
// vb_reduce_jobs_launch); otherwise they launch them themselves, as every other caller sees them do.
struct VbReduceJob { const float* src; float* dst; int rows, cols; long stride; int slices; };
struct VbReduceJobs { VbReduceJob j[8]; int n; };
__attribute__((visibility("hidden"))) VbReduceJobs*& vb_reduce_defer_slot();      // this thread's list (NULL: nobody defers)
__attribute__((visibility("hidden"))) bool vb_reduce_defer(const float* src, float* dst, int rows, int cols, long stride, int slices);
__attribute__((visibility("hidden"))) int vb_reduce_jobs_launch(const VbReduceJobs& jobs, void* stream);

__attribute__((visibility("hidden"))) int vb_gemm_fuse_residual_armed();
__attribute__((visibility("hidden"))) int vb_gemm_dropres(int dtype, int out_dtype, int a_layout, int b_layout, const void* A, int64_t lda,
    const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K, float alpha, const float* alpha_dev, const float* bias,
    const void* addend, int64_t ld_addend, int act, const void* aux_in, void* aux_out, int64_t ld_aux, int accumulate, float* colsum_out,
    float p_drop, uint64_t drop_seed, uint32_t drop_stream, void* stream);

__attribute__((visibility("hidden"))) int vb_colsum_image(const void* image, int64_t ld_image, float* out, int M, int C, void* stream);

static inline int vb_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VB_OK : VB_ERR_LAUNCH;
}
