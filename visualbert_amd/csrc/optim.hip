// optim.hip -- fused multi-tensor BertAdam over the flat parameter arena.
//
// Replaces BertAdam.step (pytorch_pretrained_bert/optimization.py:239-304), a Python loop over ~200
// tensors with ~10 small kernels each, by three launches over one contiguous fp32 arena:
//   1. per-TENSOR gradient sum of squares (the reference clips every tensor separately to
//      max_grad_norm with clip_grad_norm_, optimization.py:272-273)
//   2. update: m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; u = m / (sqrt(v) + e) + wd * p ;
//      p -= lr * schedule(step) * u      -- no bias correction, eps OUTSIDE the sqrt, decoupled weight
//      decay, LR multiplier from the per-tensor step counter read BEFORE its increment (:292,297);
//      the bf16 shadow copy that the MFMA GEMMs read is refreshed in the same pass
//   3. step counters += 1
// HBM traffic per parameter: read p, g, m, v (16 B) + write p, m, v (12 B) [+ 2 B shadow] = 28-30 B.
// Everything the step needs lives on the device (chunk table, step counters, hyper-parameters): no
// host<->device traffic per step, so the step can sit in a captured graph.
#include "vb_rt.h"
#include "../../include/visualbert_hip.h"

namespace {

constexpr int NT = 256;

struct AdamHyper {
    float lr, b1, b2, eps, max_grad_norm, warmup, t_total, weight_decay;
    int schedule;      // 0 = none (multiplier 1), 1 = warmup_linear
};

// chunk table entry layout (int64 x 4): tensor id, arena offset of the chunk, length, number of chunks of this tensor
// tensor table entry layout (int64 x 4): arena offset, numel, shadow offset (-1: none), flags (bit0: optimise, bit1: decay)

// Squared gradient norms per tensor, DETERMINISTICALLY: a chunk's partial goes to its own slot and one thread per tensor
// adds its chunks' partials in table order.  (A float atomic per chunk made the clip coefficient differ in the last bit
// from run to run -- and between data-parallel replicas, whose weights then drift apart ulp by ulp.)
VB_KERNEL VB_LAUNCH_BOUNDS(NT) adam_norm_kernel(const float* grads, const int64_t* chunks, float* partial) {
    VB_DYN_SMEM(smem);
    float* red = (float*)smem;
    const int64_t* c = chunks + (long)blockIdx.x * 4;
    const long off = c[1], len = c[2];
    float s = 0.f;
    if (((off | len) & 3) == 0) {
        // 16-byte loads, four independent ones in flight per trip (the scalar rolled loop paid a memory round trip per
        // element: 145 us for 448 MB)
        const f32x4* g4 = (const f32x4*)(grads + off);
        const long n4 = len >> 2;
        long i = threadIdx.x;
        for (; i + 3 * NT < n4; i += 4 * NT) {
            const f32x4 a = g4[i], b = g4[i + NT], c2 = g4[i + 2 * NT], d = g4[i + 3 * NT];
            s += (a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3]) + (b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3]) +
                 (c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2] + c2[3] * c2[3]) + (d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);
        }
        for (; i < n4; i += NT) { const f32x4 a = g4[i]; s += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3]; }
    } else {
        for (long i = threadIdx.x; i < len; i += NT) { const float g = grads[off + i]; s += g * g; }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// one wave per tensor (the workgroup of the tensor's FIRST chunk; the others leave): lane l adds the partials of chunks
// l, l + 64, ... in that order, then a fixed butterfly -- the same bits every time
VB_KERNEL VB_LAUNCH_BOUNDS(64) adam_norm_finish_kernel(const int64_t* chunks, int n_chunks, const float* partial, float* norm2) {
    const int j = blockIdx.x;
    const int64_t id = chunks[(long)j * 4];
    if (j > 0 && chunks[(long)(j - 1) * 4] == id) return;
    const int count = (int)chunks[(long)j * 4 + 3];                  // chunks of this tensor (adjacent in the table)
    float s = 0.f;
    for (int k = threadIdx.x; k < count; k += 64) s += partial[j + k];
    s = wave_sum(s);
    if (threadIdx.x == 0) norm2[id] = s;
}

VB_DEVICE float schedule_mult(const AdamHyper& h, int step) {
    if (h.schedule == 0 || h.t_total < 0.f) return 1.0f;
    // optimization.py:55-73 and :164-173, evaluated in double like the reference's Python floats
    const double progress = (double)step / (double)h.t_total;
    const double w = h.warmup > 0.f ? (double)h.warmup : 0.0;
    double r;
    if (progress < w) r = progress / w;
    else { r = (progress - 1.0) / (w - 1.0); if (r < 0.0) r = 0.0; }
    return (float)r;
}

// a tensor no backward pass has EVER written to (and whose gradient is zero): the reference's `if p.grad is None: continue`
// (optimization.py:254-255).  .grad stops being None at a parameter's first backward and zero_grad() leaves a zero tensor
// behind from then on, so the skip is sticky the other way round: once a tensor has taken a step (its counter is > 0) it
// takes every later step too (weight decay, moment decay, counter) even when this step wrote nothing to it.  Device-side,
// per step: no host-cached decision can go stale between data-parallel ranks.  The gradient norms are always computed when
// flags are given (host side below), so an unrecorded non-zero gradient is never dropped.
// ASSUMED zero_grad SEMANTICS: the reference's era (torch < 2: zero_grad() zeroes .grad in place, so .grad is None only before
// a parameter's first backward).  Under torch >= 2's default zero_grad(set_to_none=True) the REFERENCE would skip a tensor again
// in any later step that does not reach it; this library keeps the torch < 2 behaviour the reference was written and trained
// with (include/visualbert_hip.h: vb_bert_adam_step; visualbert_amd.optimization.BertAdam docstring).
VB_DEVICE bool adam_skips(const float* touched, const float* norm2, const int* steps, long tid) {
    if (!touched || touched[tid] != 0.f || steps[tid] > 0) return false;
    return !(norm2[tid] > 0.f);
}

VB_KERNEL VB_LAUNCH_BOUNDS(NT) adam_update_kernel(float* params, const float* grads, float* m, float* v, bf16* shadow,
                                                 const int64_t* chunks, const int64_t* tensors, const float* norm2,
                                                 const int* steps, const float* touched, AdamHyper h) {
    const int64_t* c = chunks + (long)blockIdx.x * 4;
    const long tid = c[0], off = c[1], len = c[2];
    const int64_t* te = tensors + tid * 4;
    const long t_off = te[0], sh_off = te[2], flags = te[3];
    if (!(flags & 1)) return;
    if (adam_skips(touched, norm2, steps, tid)) return;
    const float wd = (flags & 2) ? h.weight_decay : 0.0f;
    float clip = 1.0f;
    if (h.max_grad_norm > 0.f) {
        const float cc = h.max_grad_norm / (sqrtf(norm2[tid]) + 1e-6f);
        clip = cc < 1.0f ? cc : 1.0f;
    }
    const float lr = h.lr * schedule_mult(h, steps[tid]);
    // (4-byte accesses on purpose: a wave's load is 256 contiguous bytes per array and many of them are in flight.  The 16-byte form --
    // a quarter of the memory instructions -- measured SLOWER in the step, 632 vs 554 us at B = 8 on the 3.3 GB pass; round 6, not kept)
    for (long i = threadIdx.x; i < len; i += NT) {
        const long e = off + i;
        const float g = grads[e] * clip;
        const float p = params[e];
        const float mm = m[e] * h.b1 + (1.0f - h.b1) * g;
        const float vv = v[e] * h.b2 + (1.0f - h.b2) * g * g;
        float u = mm / (sqrtf(vv) + h.eps);
        if (wd > 0.f) u += wd * p;
        const float pn = p - lr * u;
        m[e] = mm; v[e] = vv; params[e] = pn;
        if (shadow && sh_off >= 0) shadow[sh_off + (e - t_off)] = (bf16)pn;
    }
}

VB_KERNEL adam_step_inc_kernel(int* steps, const int64_t* tensors, int n, const float* touched, const float* norm2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (tensors[(long)i * 4 + 3] & 1) && !adam_skips(touched, norm2, steps, i)) steps[i] += 1;
}

VB_KERNEL VB_LAUNCH_BOUNDS(NT) shadow_refresh_kernel(const float* params, bf16* shadow, const int64_t* chunks,
                                                    const int64_t* tensors) {
    const int64_t* c = chunks + (long)blockIdx.x * 4;
    const long tid = c[0], off = c[1], len = c[2];
    const long t_off = tensors[tid * 4], sh_off = tensors[tid * 4 + 2];
    if (sh_off < 0) return;
    for (long i = threadIdx.x; i < len; i += NT) shadow[sh_off + (off + i - t_off)] = (bf16)params[off + i];
}


// multi-tensor tiled transpose of bf16 weight shadows: for every GEMM weight W [R,C] keep W^T [C, R_pad]
// so that dgrad (dx = dy W) runs as a K-contiguous x K-contiguous GEMM with LDS-direct loads.
// tensors: int64[n][5] = {src offset, R, C, dst offset, dst ld}; tiles: int64[n_tiles][3] = {tensor, tile row, tile col}
VB_KERNEL VB_LAUNCH_BOUNDS(NT) shadow_transpose_kernel(const bf16* src, bf16* dst, const int64_t* tensors,
                                                      const int64_t* tiles) {
    VB_DYN_SMEM(smem);
    bf16* tile = (bf16*)smem;                       // [64][66]
    const int64_t* tl = tiles + (long)blockIdx.x * 3;
    const int64_t* te = tensors + tl[0] * 5;
    const long soff = te[0], R = te[1], C = te[2], doff = te[3], dld = te[4];
    const long r0 = tl[1] * 64, c0 = tl[2] * 64;
    // interior tiles of 16-byte aligned tensors move whole 16-byte vectors on both sides (the element-wise form below
    // took 232 us per step for 85 M weights: 1.4 TB/s)
    const bool fast = r0 + 64 <= R && c0 + 64 <= C && (C & 7) == 0 && (dld & 7) == 0 && (soff & 7) == 0 && (doff & 7) == 0;
    if (fast) {
        for (int i = threadIdx.x; i < 64 * 8; i += NT) {
            const int r = i >> 3, cc = (i & 7) * 8;
            const u32x4 v = *(const u32x4*)(src + soff + (r0 + r) * C + c0 + cc);
            uint32_t* t32 = (uint32_t*)(tile + r * 66 + cc);
            t32[0] = v[0]; t32[1] = v[1]; t32[2] = v[2]; t32[3] = v[3];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * 8; i += NT) {
            const int c = i >> 3, rr = (i & 7) * 8;      // output row = source column
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = tile[(rr + j) * 66 + c];
            *(bf16x8*)(dst + doff + (c0 + c) * dld + r0 + rr) = o;
        }
        return;
    }
    for (int i = threadIdx.x; i < 64 * 8; i += NT) {
        const int r = i >> 3, cc = (i & 7) * 8;
        const long gr = r0 + r, gc = c0 + cc;
        bf16 v[8];
        for (int j = 0; j < 8; ++j) v[j] = (gr < R && gc + j < C) ? src[soff + gr * C + gc + j] : (bf16)0.0f;
        for (int j = 0; j < 8; ++j) tile[r * 66 + cc + j] = v[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 8; i += NT) {
        const int c = i >> 3, rr = (i & 7) * 8;      // output row = source column
        const long gc = c0 + c, gr = r0 + rr;
        if (gc >= C) continue;
        for (int j = 0; j < 8; ++j) if (gr + j < R) dst[doff + gc * dld + gr + j] = tile[(rr + j) * 66 + c];
    }
}

}  // namespace

extern "C" int vb_bert_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                 void* bf16_shadow, const int64_t* chunk_table, int n_chunks,
                                 const int64_t* tensor_table, int n_tensors, const float* touched, float* norm2_ws,
                                 int* step_counters,
                                 float lr, float b1, float b2, float eps, float weight_decay,
                                 float max_grad_norm, float warmup, float t_total, int schedule, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !chunk_table || !tensor_table || !norm2_ws || !step_counters)
        return VB_ERR_ARG;
    if (n_chunks <= 0 || n_tensors <= 0 || (schedule != 0 && schedule != 1)) return VB_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    AdamHyper h{lr, b1, b2, eps, max_grad_norm, warmup, t_total, weight_decay, schedule};
    if (max_grad_norm > 0.f || touched) {                   // the skip decision reads the norms too
        float* partial = norm2_ws + n_tensors;              // [n_chunks]
        VB_LAUNCH(adam_norm_kernel, dim3((unsigned)n_chunks), dim3(NT), 64, s, grads, chunk_table, partial);
        VB_LAUNCH(adam_norm_finish_kernel, dim3((unsigned)n_chunks), dim3(64), 0, s, chunk_table, n_chunks,
                  (const float*)partial, norm2_ws);
    }
    VB_LAUNCH(adam_update_kernel, dim3((unsigned)n_chunks), dim3(NT), 0, s, params, grads, exp_avg, exp_avg_sq,
              (bf16*)bf16_shadow, chunk_table, tensor_table, (const float*)norm2_ws, (const int*)step_counters, touched, h);
    VB_LAUNCH(adam_step_inc_kernel, dim3((unsigned)((n_tensors + 63) / 64)), dim3(64), 0, s, step_counters,
              tensor_table, n_tensors, touched, (const float*)norm2_ws);
    return vb_check_launch();
}

extern "C" int vb_refresh_bf16_shadow(const float* params, void* bf16_shadow, const int64_t* chunk_table,
                                      int n_chunks, const int64_t* tensor_table, void* stream) {
    if (!params || !bf16_shadow || !chunk_table || !tensor_table || n_chunks <= 0) return VB_ERR_ARG;
    VB_LAUNCH(shadow_refresh_kernel, dim3((unsigned)n_chunks), dim3(NT), 0, (hipStream_t)stream, params,
              (bf16*)bf16_shadow, chunk_table, tensor_table);
    return vb_check_launch();
}

extern "C" int vb_refresh_transposed_shadow(const void* bf16_shadow, void* bf16_shadow_t, const int64_t* tensor_table5,
                                            const int64_t* tile_table3, int n_tiles, void* stream) {
    if (!bf16_shadow || !bf16_shadow_t || !tensor_table5 || !tile_table3 || n_tiles <= 0) return VB_ERR_ARG;
    VB_LAUNCH(shadow_transpose_kernel, dim3((unsigned)n_tiles), dim3(NT), 64 * 66 * 2, (hipStream_t)stream,
              (const bf16*)bf16_shadow, (bf16*)bf16_shadow_t, tensor_table5, tile_table3);
    return vb_check_launch();
}
