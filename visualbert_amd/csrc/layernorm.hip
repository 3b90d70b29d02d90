// layernorm.hip -- HBM-bound row kernels of the VisualBERT training path:
//   * BertLayerNorm forward / backward with the residual add and both dropouts fused
//       reference: BertLayerNorm.forward  pytorch_pretrained_bert/modeling.py:171-175
//                  BertSelfOutput.forward :270-274, BertOutput.forward :315-319
//                  (dense -> dropout -> LayerNorm(x + input)), embeddings LN+dropout :1255-1257,
//                  BertPredictionHeadTransform LN :400
//   * BertEmbeddingsWithVisualEmbedding gather-add forward / scatter-add backward
//       reference: modeling.py:1198-1253 (image_text_alignment=None branch)
//
// Thread mapping: a HALF-wave (32 lanes) owns one row; a lane owns 8-element chunks
// {l, l+32, l+64, ...} (16-byte bf16 / 32-byte fp32 accesses, a half-wave instruction covers 512 B of
// one row).  Row statistics are fp32, two-pass in registers (mean, then centred second moment: the
// reference's formula, biased variance, eps inside the sqrt).  Column reductions (dgamma, dbeta,
// bias gradient, position/type embedding gradients) are accumulated per lane in registers across
// the rows a half-wave visits, reduced across the workgroup in LDS and added to HBM once per block.
#include "vb_rt.h"
#include "../../include/visualbert_hip.h"

namespace {

constexpr int NT = 256;
constexpr int HW_PER_BLOCK = NT / 32;
constexpr int MAX_NC = 4;            // H <= 1024

struct DropSpec {
    float p; float scale; uint32_t thresh; uint32_t stream; uint64_t seed;
};
static inline DropSpec make_drop(float p, uint64_t seed, uint32_t stream) {
    DropSpec d; d.p = p; d.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    d.thresh = vb_drop_thresh16(p); d.stream = stream; d.seed = seed;
    return d;
}
// multiply v[0..8) by the keep mask / (1-p) of group (element index >> 3)
VB_DEVICE void apply_dropout8(float (&v)[8], const DropSpec& d, uint64_t group) {
    Rand8 r = vb_dropout_bits8(d.seed, group, d.stream);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = rand8_keep(r, j, d.thresh) ? v[j] * d.scale : 0.0f;
}

struct LnFwdArgs {
    const void* x; const void* resid; void* z_out; void* y; float* mean; float* rstd;
    const float* gamma; const float* beta; int M, H; float eps; DropSpec din, dout;
    bf16* y_split; long ld_split;      // fp32 only: also write y as a bf16 hi | lo image [M, ld_split] (split-operand mode), or NULL
    int* rebuild;                      // non-NULL: z_out is written only if the backward cannot rebuild x-hat from y (see
                                       // ln_rebuildable); the verdict (one device int) goes here
};

// May the backward take x-hat = (y - beta) / gamma from the output it already has instead of from a saved pre-LN sum z?
// Rounding y to T costs x-hat 2^-9 (|x-hat| + |beta| / |gamma|) per element where the z route costs 2^-9 |x-hat + mean * rstd|: the
// same order while |beta| <= 2 |gamma| on every channel (true at initialisation: gamma = 1, beta = 0), unbounded when a trained
// LayerNorm has a channel with |gamma| << |beta|.  So the kernels decide per launch, from the parameters themselves: each lane
// tests the channels it owns, the half-wave (which owns all H channels between its lanes) agrees, every workgroup reaches the same
// verdict, workgroup 0 records it for the backward (which must not re-derive it: an optimizer step may lie in between).
// 1 / gamma of the rebuild route.  bf16: the hardware reciprocal (1 ulp of fp32, far below the 2^-9 of the y it multiplies).  fp32 / split
// mode: one Newton step on top (r + r (1 - g r): two fused multiply-adds, correct to ~0.5 ulp) -- there y itself carries only 2^-24, so
// a 1-ulp reciprocal would be the largest error term of x-hat, and the fp32 gradients are held to the reference at the 1e-5 level.
template <typename T> VB_DEVICE float rebuild_rcp(float g) {
    const float r = fast_rcp(g);
    if constexpr (sizeof(T) == 2) return r;
    else return fmaf(r, fmaf(-g, r, 1.0f), r);
}

template <int NC>
VB_DEVICE bool ln_rebuildable(const float* gamma, const float* beta, int H, int l32) {
    float bad = 0.f;
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
        const int col = (l32 + 32 * ci) * 8;
        const int colc = col < H ? col : 0;
        float gm[8], bt[8];
        load8(gm, gamma + colc); load8(bt, beta + colc);
#pragma unroll
        for (int j = 0; j < 8; ++j) bad += (fabsf(gm[j]) > 1e-30f && fabsf(bt[j]) <= 2.f * fabsf(gm[j])) ? 0.f : 1.f;
    }
    return half_sum(bad) == 0.f;
}

template <typename T, int NC>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) ln_fwd_kernel(LnFwdArgs a) {
    const int l32 = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const int H = a.H;
    const float invH = 1.0f / (float)H;
    bool write_z = a.z_out != nullptr;
    if (a.rebuild) {
        const bool rb = ln_rebuildable<NC>(a.gamma, a.beta, H, l32);
        if (rb) write_z = false;
        if (blockIdx.x == 0 && threadIdx.x == 0) *a.rebuild = rb ? 1 : 0;
    }
    const bool has_res = a.resid != nullptr;
    const T* rp = has_res ? (const T*)a.resid : (const T*)a.x;
    // the trip count is uniform per workgroup (wave shuffles below need all 64 lanes); a half-wave
    // whose row is past the end just keeps its lanes predicated off
    for (int base = blockIdx.x * HW_PER_BLOCK; base < a.M; base += gridDim.x * HW_PER_BLOCK) {
        const int row = base + hw;
        const bool act = row < a.M;
        const long rowc = act ? row : a.M - 1;
        // loads are UNCONDITIONAL from a clamped (row, column) and everything that must not happen for a lane outside the
        // matrix is a select or a predicated store: `if (ok) load` compiled to one exec-masked branch per load with
        // s_waitcnt vmcnt(0) between them -- six serialized HBM round trips per row
        // ALL of a row's loads go out together, ahead of the first use: x, the residual (from x itself when there is none: a valid
        // address, the values dropped by a select), gamma and beta (through a scalar base the compiler cannot fold into loop-invariant
        // 64-bit lane addresses).  As `if (a.resid) { load ... }` and "gamma / beta where they are used" the row had two serialized HBM
        // round trips and three exposed L2 ones (round 4: 5.8 -> 6.3 TB/s with three tensors per launch)
        float v[NC][8], r[NC][8], gmv[NC][8], btv[NC][8];
        float s = 0.f;
        int zs = 0;
        vb_pin_s(zs);
        const float* gbase = a.gamma + zs;
        const float* bbase = a.beta + zs;
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
            const int col = (l32 + 32 * ci) * 8;
            const long e = rowc * H + (col < H ? col : 0);
            load8(v[ci], (const T*)a.x + e);
        }
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
            const int col = (l32 + 32 * ci) * 8;
            const long e = rowc * H + (col < H ? col : 0);
            load8(r[ci], rp + (has_res ? e : (long)(col < H ? col : 0)));     // (no residual: row 0 of x again and again, cache hits)
        }
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
            const int col = (l32 + 32 * ci) * 8;
            const int colc = col < H ? col : 0;
            load8(gmv[ci], gbase + colc);
        }
        if (a.din.p > 0.f) {
#pragma unroll
            for (int ci = 0; ci < NC; ++ci) {
                const int col = (l32 + 32 * ci) * 8;
                apply_dropout8(v[ci], a.din, (uint64_t)(rowc * H + (col < H ? col : 0)) >> 3);
            }
        }
#pragma unroll
        for (int ci = 0; ci < NC; ++ci)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[ci][j] += has_res ? r[ci][j] : 0.f;
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
            const int col = (l32 + 32 * ci) * 8;
            const bool ok = act && col < H;
            if (write_z && ok) store8((T*)a.z_out + (long)row * H + col, v[ci]);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[ci][j] = ok ? v[ci][j] : 0.f; s += v[ci][j]; }
        }
        const float mean = half_sum(s) * invH;
        // (beta is needed last: its L2 round trip runs under the variance pass, and issuing it here instead of with the loads above keeps
        //  the kernel inside the 128 registers of four waves per SIMD)
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
            const int col = (l32 + 32 * ci) * 8;
            load8(btv[ci], bbase + (col < H ? col : 0));
        }
        float q = 0.f;
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
            const int col = (l32 + 32 * ci) * 8;
            const bool ok = act && col < H;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = ok ? v[ci][j] - mean : 0.f; q += d * d; }
        }
        const float var = half_sum(q) * invH;
        const float rstd = 1.0f / sqrtf(var + a.eps);
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
            const int col = (l32 + 32 * ci) * 8;
            const int colc = col < H ? col : 0;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = gmv[ci][j] * ((v[ci][j] - mean) * rstd) + btv[ci][j];
            const long e = (long)row * H + col;
            if (a.dout.p > 0.f) apply_dropout8(o, a.dout, (uint64_t)(rowc * H + colc) >> 3);
            if (act && col < H) {
                store8((T*)a.y + e, o);
                if constexpr (sizeof(T) == 4) { if (a.y_split) store_split8(a.y_split + (long)row * a.ld_split + col, a.ld_split / 2, o); }
            }
        }
        if (act && l32 == 0) {
            if (a.mean) a.mean[row] = mean;
            if (a.rstd) a.rstd[row] = rstd;
        }
    }
}

struct LnBwdArgs {
    const void* dy; const void* z; const float* mean; const float* rstd; const float* gamma;
    void* dz; void* dx; float* dgamma; float* dbeta; float* dbias; int M, H; DropSpec din, dout;
    float* partials;     // [gridDim.x][3][H] when the two-stage column reduction is used, else NULL
    bf16* dx_split; long ld_split;     // fp32 only: also write dx (= dz when no dropout) as a bf16 hi | lo image, or NULL
    const void* y; const float* beta; const int* rebuild;   // *rebuild != 0: x-hat = (y - beta) / gamma (the forward wrote no z)
};

// reduce this block's per-lane column partials across its 8 half-waves through LDS ([8][H] floats, plain
// stores -- no LDS atomics), then write them to this block's row of the partials workspace (two-stage
// path, no global atomics) or add them to HBM with fp32 atomics
template <int NC>
VB_DEVICE void block_colsum_flush(float (&acc)[NC][8], float* lds, float* out, int H, int l32, int hw,
                                  float* part = nullptr) {
    __syncthreads();
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
        const int col = (l32 + 32 * ci) * 8;
        if (col < H) {
            *(f32x4*)(lds + hw * H + col) = f32x4{acc[ci][0], acc[ci][1], acc[ci][2], acc[ci][3]};
            *(f32x4*)(lds + hw * H + col + 4) = f32x4{acc[ci][4], acc[ci][5], acc[ci][6], acc[ci][7]};
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < H; i += NT) {
        float s = 0.f;
#pragma unroll
        for (int h = 0; h < HW_PER_BLOCK; ++h) s += lds[h * H + i];
        if (part) part[i] = s;
        else atomicAdd(&out[i], s);
    }
}

// second stage: out[c] += sum over blocks of partials[b][which][c].  1024 threads = 32 columns x 32 row
// groups: coalesced 128-byte reads, 32 independent chains per column, LDS tree at the end.  gridDim.z slices the
// partial rows (a 24 x 3 grid alone left 2/3 of the chip idle over 9 MB of partials); slices meet in fp32 atomics.
VB_KERNEL VB_LAUNCH_BOUNDS(1024) ln_bwd_reduce_kernel(const float* partials, int nblocks, int H, float* dgamma,
                                                     float* dbeta, float* dbias) {
    VB_DYN_SMEM(smem);
    float* red = (float*)smem;                         // [32][33]
    const int cx = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    const int which = blockIdx.y;
    const int per = (nblocks + (int)gridDim.z - 1) / (int)gridDim.z;
    const int b0 = (int)blockIdx.z * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    float* out = which == 0 ? dgamma : (which == 1 ? dbeta : dbias);
    float s = 0.f;
    if (c < H && out) {
        for (int b = b0 + rg; b < b1; b += 32) s += partials[((long)b * 3 + which) * H + c];
    }
    red[rg * 33 + cx] = s;
    __syncthreads();
    if (rg == 0 && c < H && out) {
        float t = 0.f;
        for (int r = 0; r < 32; ++r) t += red[r * 33 + cx];
        if (gridDim.z == 1) out[c] += t;
        else atomicAdd(&out[c], t);
    }
}

// LDS accumulate of 8 column partials owned exclusively by this lane (no atomics, no conflicts with other lanes)
VB_DEVICE void lds_acc8(float* p, const float (&v)[8]) {
    f32x4 lo = *(f32x4*)p, hi = *(f32x4*)(p + 4);
    lo = f32x4{lo[0] + v[0], lo[1] + v[1], lo[2] + v[2], lo[3] + v[3]};
    hi = f32x4{hi[0] + v[4], hi[1] + v[5], hi[2] + v[6], hi[3] + v[7]};
    *(f32x4*)p = lo; *(f32x4*)(p + 4) = hi;
}

// One WAVE per row: lane l owns the 8-column chunks l and l + 64 (H = 768: the second chunk on lanes 0..31 only).  The
// column accumulators (dgamma, dbeta, bias gradient) are REGISTERS of the lane that owns the columns -- 48 for H <= 1024 --
// and reach LDS once, at the end.  (The previous form gave a row to each 32-lane half: three chunks per lane, accumulators
// in LDS updated by read-modify-write every row, the halves' contributions combined with 72 v_permlane32_swap per pair of
// rows -- ~500 instructions per row, which is what bounded the kernel at 3.9 TB/s, not HBM.)  All loads of a row are
// unconditional from clamped columns and issued together.
constexpr int WAVES_PER_BLOCK = NT / 64;
VB_DEVICE float wave_sum64(float v) {
    v = row16_sum(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
template <typename T, int NC2, int MINW, int RB = 0>
VB_KERNEL VB_LAUNCH_BOUNDS2(NT, MINW) ln_bwd_kernel(LnBwdArgs a) {
    VB_DYN_SMEM(smem);
    float* lds = (float*)smem;
    const int lane = threadIdx.x & 63, wave = vb_uniform((int)threadIdx.x >> 6);
    const int H = a.H;
    const float invH = 1.0f / (float)H;
    float accg[NC2][8], accb[NC2][8], accx[NC2][8];
#pragma unroll
    for (int ci = 0; ci < NC2; ++ci)
#pragma unroll
        for (int j = 0; j < 8; ++j) { accg[ci][j] = 0.f; accb[ci][j] = 0.f; accx[ci][j] = 0.f; }

    const bool rebuild = RB && *a.rebuild != 0;
    const T* xsrc = (const T*)(rebuild ? a.y : a.z);
    for (int row = blockIdx.x * WAVES_PER_BLOCK + wave; row < a.M; row += gridDim.x * WAVES_PER_BLOCK) {
        const float mean = a.mean[row], rstd = a.rstd[row];
        float dy[NC2][8], xh[NC2][8];
#pragma unroll
        for (int ci = 0; ci < NC2; ++ci) {
            const int col = (lane + 64 * ci) * 8;
            load8(dy[ci], (const T*)a.dy + (long)row * H + (col < H ? col : 0));
        }
#pragma unroll
        for (int ci = 0; ci < NC2; ++ci) {
            const int col = (lane + 64 * ci) * 8;
            load8(xh[ci], xsrc + (long)row * H + (col < H ? col : 0));
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ci = 0; ci < NC2; ++ci) {
            const int col = (lane + 64 * ci) * 8;
            const int colc = col < H ? col : 0;
            const bool ok = col < H;
            float gm[8];
            load8(gm, a.gamma + colc);
            if (a.dout.p > 0.f) apply_dropout8(dy[ci], a.dout, (uint64_t)((long)row * H + colc) >> 3);
            if constexpr (RB == 1) {
                float bt[8];
                load8(bt, a.beta + colc);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float sub = rebuild ? bt[j] : mean, q = rebuild ? rebuild_rcp<T>(gm[j]) : rstd;
                    xh[ci][j] = (xh[ci][j] - sub) * q;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) xh[ci][j] = (xh[ci][j] - mean) * rstd;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = ok ? dy[ci][j] : 0.f;         // columns past H contribute exact zeros
                accg[ci][j] += d * xh[ci][j];                 // dgamma
                accb[ci][j] += d;                             // dbeta
                dy[ci][j] = d * gm[j];                        // g = dy * gamma
                s1 += dy[ci][j];
                s2 += dy[ci][j] * xh[ci][j];
            }
        }
        s1 = wave_sum64(s1) * invH;
        s2 = wave_sum64(s2) * invH;
#pragma unroll
        for (int ci = 0; ci < NC2; ++ci) {
            const int col = (lane + 64 * ci) * 8;
            const bool ok = col < H;
            const long e = (long)row * H + col;
            float dz[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) dz[j] = rstd * (dy[ci][j] - s1 - xh[ci][j] * s2);
            if (ok) store8((T*)a.dz + e, dz);
            if (a.dx) {
                if (a.din.p > 0.f) apply_dropout8(dz, a.din, (uint64_t)((long)row * H + (ok ? col : 0)) >> 3);
                if (a.dx != a.dz && ok) store8((T*)a.dx + e, dz);
            }
            if constexpr (sizeof(T) == 4) { if (a.dx_split && ok) store_split8(a.dx_split + (long)row * a.ld_split + col, a.ld_split / 2, dz); }
            if (a.dbias) {                                   // bias gradient of the Linear in front: column sums of dx
#pragma unroll
                for (int j = 0; j < 8; ++j) accx[ci][j] += ok ? dz[j] : 0.f;
            }
        }
    }
    // the waves' accumulators -> LDS [wave][3][H]
    float* my = lds + (long)wave * 3 * H;
#pragma unroll
    for (int ci = 0; ci < NC2; ++ci) {
        const int col = (lane + 64 * ci) * 8;
        if (col < H) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { my[col + j] = accg[ci][j]; my[H + col + j] = accb[ci][j]; my[2 * H + col + j] = accx[ci][j]; }
        }
    }
    __syncthreads();
    // reduce the waves' accumulators, then this block's row of the partials workspace (two-stage, no global
    // atomics) or fp32 atomics straight into HBM
    float* part = a.partials ? a.partials + (long)blockIdx.x * 3 * H : nullptr;
    for (int i = threadIdx.x; i < 3 * H; i += NT) {
        const int which = i / H, c = i - which * H;
        float* out = which == 0 ? a.dgamma : (which == 1 ? a.dbeta : a.dbias);
        if (!out) continue;
        float sum = 0.f;
#pragma unroll
        for (int h = 0; h < WAVES_PER_BLOCK; ++h) sum += lds[(long)h * 3 * H + i];
        if (part) part[i] = sum;
        else atomicAdd(&out[c], sum);
    }
}

// 512 < H <= 768 (BERT-base): one wave per row with TWELVE columns per lane -- the 8-column chunk `lane` and the 4-column chunk at
// 512 + 4 lane -- so that H = 768 keeps all 64 lanes busy in both chunks.  The generic kernel above gives a lane two 8-column chunks, the
// second one only on lanes 0..31: registers for 16 columns (48 accumulators + 32 row values -> the 128-register budget of four waves
// per SIMD is met with spills) and a quarter of the vector work masked off.  Here: 36 accumulators + 24 row values.
// RB: the launch may find x-hat in the LayerNorm's own output (*a.rebuild, the forward's verdict): x-hat = (v - sub) q with
// (v, sub, q) = (y, beta_c, 1 / gamma_c) or (z, mean_r, rstd_r).  ONE instruction stream -- beta is loaded unconditionally next to gamma,
// the source pointer and the operands are selected by a launch-uniform condition: a branch around the extra loads put two more
// exposed memory round trips into every row of this latency-bound kernel (188 -> 276 us per launch, profiles/r04_ln_rebuild.txt).
template <int W, typename T> VB_DEVICE void loadw(float (&v)[W], const T* p) {
    if constexpr (W == 8) load8(v, p);
    else if constexpr (sizeof(T) == 2) { const bf16x4 x = *(const bf16x4*)p; v[0] = (float)x[0]; v[1] = (float)x[1]; v[2] = (float)x[2]; v[3] = (float)x[3]; }
    else { const f32x4 x = *(const f32x4*)p; v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3]; }
}
template <int W, typename T> VB_DEVICE void storew(T* p, const float (&v)[W]) {
    if constexpr (W == 8) store8(p, v);
    else if constexpr (sizeof(T) == 2) { bf16x4 x; x[0] = (bf16)v[0]; x[1] = (bf16)v[1]; x[2] = (bf16)v[2]; x[3] = (bf16)v[3]; *(bf16x4*)p = x; }
    else *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
}
// keep mask / (1 - p) of the W elements that start at element index e (a multiple of W): half a generator group when W = 4
template <int W> VB_DEVICE void dropw(float (&v)[W], const DropSpec& d, uint64_t e) {
    Rand8 r = vb_dropout_bits8(d.seed, e >> 3, d.stream);
    if constexpr (W == 4) {                        // the group's upper half: words 2, 3 (selected, not indexed: r must stay in registers)
        const bool hi = (e & 4) != 0;
        r.w[0] = hi ? r.w[2] : r.w[0]; r.w[1] = hi ? r.w[3] : r.w[1];
    }
#pragma unroll
    for (int j = 0; j < W; ++j) v[j] = rand8_keep(r, j, d.thresh) ? v[j] * d.scale : 0.0f;
}
template <typename T, bool RB>
VB_KERNEL VB_LAUNCH_BOUNDS2(NT, 4) ln_bwd12_kernel(LnBwdArgs a) {
    VB_DYN_SMEM(smem);
    float* lds = (float*)smem;
    const int lane = threadIdx.x & 63, wave = vb_uniform((int)threadIdx.x >> 6);
    const int H = a.H;
    const float invH = 1.0f / (float)H;
    float accg[12], accb[12], accx[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) { accg[j] = 0.f; accb[j] = 0.f; accx[j] = 0.f; }
    const bool rebuild = RB && *a.rebuild != 0;
    const T* xsrc = (const T*)(rebuild ? a.y : a.z);
    const int colA = lane * 8, colB = 512 + lane * 4;
    const bool okB = colB < H;                                     // a 4-column chunk is inside or outside as a whole (H % 8 == 0)
    const int colBc = okB ? colB : 0;
    typedef std::integral_constant<int, 8> W8;
    typedef std::integral_constant<int, 4> W4;

    for (int row = blockIdx.x * WAVES_PER_BLOCK + wave; row < a.M; row += gridDim.x * WAVES_PER_BLOCK) {
        const float mean = a.mean[row], rstd = a.rstd[row];
        const long rb = (long)row * H;
        float dyA[8], dyB[4], xhA[8], xhB[4];
        loadw<8>(dyA, (const T*)a.dy + rb + colA); loadw<4>(dyB, (const T*)a.dy + rb + colBc);
        loadw<8>(xhA, xsrc + rb + colA); loadw<4>(xhB, xsrc + rb + colBc);
        float s1 = 0.f, s2 = 0.f;
        // gamma / beta through a base the compiler cannot fold into a loop-invariant 64-bit lane address: kept that way they cost
        // four VGPR pairs, which were spilled and reloaded (behind s_waitcnt vmcnt(0)) in every row; scalar base + 32-bit lane offset
        // is the form the row tensors' loads have anyway
        int zs = 0;
        vb_pin_s(zs);
        const float* gbase = a.gamma + zs;
        const float* bbase = a.beta + zs;
        // phase 1 of a chunk: x-hat, column partials, g = dy gamma (in place of dy), row sums
        // (all of a row's loads go out together, ahead of the first use)
        float gmA[8], gmB[4], btA[8], btB[4];
        loadw<8>(gmA, gbase + colA); loadw<4>(gmB, gbase + colBc);
        if constexpr (RB) { loadw<8>(btA, bbase + colA); loadw<4>(btB, bbase + colBc); }
        auto phase1 = [&](auto wtag, float (&dy)[decltype(wtag)::value], float (&xh)[decltype(wtag)::value], const float (&gm)[decltype(wtag)::value],
                          const float (&bt)[decltype(wtag)::value], int col, bool ok, float* ag, float* ab) {
            constexpr int W = decltype(wtag)::value;
            if (a.dout.p > 0.f) dropw<W>(dy, a.dout, (uint64_t)(rb + col));
            if constexpr (RB) {
#pragma unroll
                for (int j = 0; j < W; ++j) xh[j] = (xh[j] - (rebuild ? bt[j] : mean)) * (rebuild ? rebuild_rcp<T>(gm[j]) : rstd);
            } else {
#pragma unroll
                for (int j = 0; j < W; ++j) xh[j] = (xh[j] - mean) * rstd;
            }
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const float d = ok ? dy[j] : 0.f;             // columns past H contribute exact zeros
                ag[j] += d * xh[j];                           // dgamma
                ab[j] += d;                                   // dbeta
                dy[j] = d * gm[j];                            // g = dy * gamma
                s1 += dy[j];
                s2 += dy[j] * xh[j];
            }
        };
        phase1(W8(), dyA, xhA, gmA, btA, colA, true, accg, accb);
        phase1(W4(), dyB, xhB, gmB, btB, colBc, okB, accg + 8, accb + 8);
        s1 = wave_sum64(s1) * invH;
        s2 = wave_sum64(s2) * invH;
        auto phase2 = [&](auto wtag, float (&g)[decltype(wtag)::value], float (&xh)[decltype(wtag)::value], int col, bool ok, float* ax) {
            constexpr int W = decltype(wtag)::value;
            const long e = rb + col;
            float dz[W];
#pragma unroll
            for (int j = 0; j < W; ++j) dz[j] = rstd * (g[j] - s1 - xh[j] * s2);
            if (ok) storew<W>((T*)a.dz + e, dz);
            if (a.dx) {
                if (a.din.p > 0.f) dropw<W>(dz, a.din, (uint64_t)e);
                if (a.dx != a.dz && ok) storew<W>((T*)a.dx + e, dz);
            }
            if constexpr (sizeof(T) == 4) {
                if (a.dx_split && ok) {
                    if constexpr (W == 8) store_split8(a.dx_split + (long)row * a.ld_split + col, a.ld_split / 2, dz);
                    else store_split4(a.dx_split + (long)row * a.ld_split + col, a.ld_split / 2, f32x4{dz[0], dz[1], dz[2], dz[3]});
                }
            }
            if (a.dbias) {                                   // bias gradient of the Linear in front: column sums of dx
#pragma unroll
                for (int j = 0; j < W; ++j) ax[j] += ok ? dz[j] : 0.f;
            }
        };
        phase2(W8(), dyA, xhA, colA, true, accx);
        phase2(W4(), dyB, xhB, colBc, okB, accx + 8);
    }
    // the waves' accumulators -> LDS [wave][3][HP]: column c sits at c + (c >> 5), i.e. one float of padding per 32 columns, so that the
    // 32 lanes of a store (lane stride 8 or 4 columns) fall on 32 different banks -- at the plain pitch they shared four (an 8-way
    // conflict on every store: SQ_LDS_BANK_CONFLICT 34 % of this kernel's LDS cycles in round 5's PMC pass)
    const int HP = H + (H >> 5) + 1;
    float* my = lds + (long)wave * 3 * HP;
    const int pA = colA + (colA >> 5), pB = colB + (colB >> 5);
#pragma unroll
    for (int j = 0; j < 8; ++j) { my[pA + j] = accg[j]; my[HP + pA + j] = accb[j]; my[2 * HP + pA + j] = accx[j]; }
    if (okB) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { my[pB + j] = accg[8 + j]; my[HP + pB + j] = accb[8 + j]; my[2 * HP + pB + j] = accx[8 + j]; }
    }
    __syncthreads();
    float* part = a.partials ? a.partials + (long)blockIdx.x * 3 * H : nullptr;
    for (int i = threadIdx.x; i < 3 * H; i += NT) {
        const int which = i / H, c = i - which * H;
        float* out = which == 0 ? a.dgamma : (which == 1 ? a.dbeta : a.dbias);
        if (!out) continue;
        float sum = 0.f;
        const int pi = which * HP + c + (c >> 5);
#pragma unroll
        for (int h = 0; h < WAVES_PER_BLOCK; ++h) sum += lds[(long)h * 3 * HP + pi];
        if (part) part[i] = sum;
        else atomicAdd(&out[c], sum);
    }
}

// ---------------------------------------------------------------------------------------------
struct EmbArgs {
    const int64_t* ids; const int64_t* type_ids; const int64_t* vis_type;
    const void* vis_proj; const float* word; const float* pos; const float* type;
    const float* pos_vis; const float* type_vis; const float* pos_align; void* z;
    int B, T, R, H, V, TV, P;
};

template <typename TT_, int NC>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) embed_fwd_kernel(EmbArgs a) {
    const int l32 = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const int S = a.T + a.R, H = a.H;
    const long rows = (long)a.B * S;
    for (long row = (long)blockIdx.x * HW_PER_BLOCK + hw; row < rows; row += (long)gridDim.x * HW_PER_BLOCK) {
        const int b = (int)(row / S), s = (int)(row % S);
        const float *t0, *t1, *t2, *t3 = nullptr;
        const TT_* vp = nullptr;
        if (s < a.T) {
            long id = a.ids[(long)b * a.T + s]; id = id < 0 ? 0 : (id >= a.V ? a.V - 1 : id);
            long tt = a.type_ids ? a.type_ids[(long)b * a.T + s] : 0; tt = tt < 0 ? 0 : (tt >= a.TV ? a.TV - 1 : tt);
            t0 = a.word + id * H; t1 = a.pos + (long)(s < a.P ? s : a.P - 1) * H; t2 = a.type + tt * H;
        } else {
            const int r = s - a.T;
            long vt = a.vis_type ? a.vis_type[(long)b * a.R + r] : 0; vt = vt < 0 ? 0 : (vt >= a.TV ? a.TV - 1 : vt);
            vp = (const TT_*)a.vis_proj + ((long)b * a.R + r) * H;
            t0 = nullptr; t1 = a.pos_vis; t2 = a.type_vis + vt * H;      // visual position id is always 0
            if (a.pos_align) t3 = a.pos_align + ((long)b * a.R + r) * H;   // + mean position of the aligned words
        }
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
            const int col = (l32 + 32 * ci) * 8;
            if (col < H) {
                float v[8], x[8];
                if (t0) load8(v, t0 + col); else load8(v, vp + col);
                load8(x, t1 + col);
                if (t3) {                                    // reference order: vis + ((align + pos_vis) + type_vis)
                    float y[8];
                    load8(y, t3 + col);
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] += y[j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += x[j];
                load8(x, t2 + col);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += x[j];
                store8((TT_*)a.z + row * H + col, v);
            }
        }
    }
}

// word-embedding gradient: d_word[id[b, s]] += dz[b, s] for the text positions.  One workgroup per text token; thread t
// owns columns t, t + 256, ...: every atomic instruction of a wave covers 64 CONSECUTIVE floats of one table row (two
// cache lines).  (Inside embed_bwd_kernel each lane owned 8 consecutive columns, i.e. every atomic instruction touched
// 32 lanes x 4 B spread over 1 KB: 12.6 M atomics at B=128 took most of that kernel's 377 us.)
template <typename TT_>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) embed_word_scatter_kernel(const TT_* dz, const int64_t* ids, float* d_word,
                                                        int B, int T, int S, int H, int V) {
    const int tok = blockIdx.x;                        // b * T + s
    const int b = tok / T, s = tok - b * T;
    long id = ids[tok];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const TT_* src = dz + ((long)b * S + s) * H;
    float* dst = d_word + id * H;
    for (int c = threadIdx.x; c < H; c += NT) vb_atomic_add_noret(dst + c, to_f32(src[c]));
}

struct EmbBwdArgs {
    const void* dz; const int64_t* ids; const int64_t* type_ids; const int64_t* vis_type;
    float* d_word; float* d_pos; float* d_type; float* d_pos_vis; float* d_type_vis; void* d_vis_proj;
    int B, T, R, H, V, TV, P;
};

// workgroup (s, c) handles sequence position s for the batch slice c of gridDim.y slices: the position-embedding
// gradient of s is accumulated in registers over the slice (then added atomically -- one owner when there is one
// slice); word rows are scattered with fp32 atomics; the (tiny) type tables are reduced in LDS first.  164 workgroups
// (one per position) left a third of the chip idle with 16 serial trips each: 377 us at B=128.
template <typename TT_, int NC>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) embed_bwd_kernel(EmbBwdArgs a) {
    VB_DYN_SMEM(smem);
    float* lds_pos = (float*)smem;                 // [H]
    float* lds_type = lds_pos + a.H;               // [TV][H]
    const int l32 = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const int S = a.T + a.R, H = a.H, s = blockIdx.x;
    const bool text = s < a.T;
    for (int i = threadIdx.x; i < H * (1 + a.TV); i += NT) lds_pos[i] = 0.f;
    __syncthreads();
    // two token types (every BERT configuration of the reference): the type-1 rows are summed in registers next to the
    // all-rows sum and type 0 is the difference -- the general path below pays one LDS float atomic per ELEMENT (64 M of
    // them per step at B = 512: most of this kernel's 435 us)
    const bool two_types = a.TV == 2;
    float acc[NC][8], acc1[NC][8];
#pragma unroll
    for (int ci = 0; ci < NC; ++ci)
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[ci][j] = 0.f; acc1[ci][j] = 0.f; }
    const int bper = (a.B + (int)gridDim.y - 1) / (int)gridDim.y;
    const int b_lo = (int)blockIdx.y * bper, b_hi = b_lo + bper < a.B ? b_lo + bper : a.B;
    for (int b = b_lo + hw; b < b_hi; b += HW_PER_BLOCK) {
        const long row = (long)b * S + s;
        long id = 0, tt = 0;
        if (text) {
            id = a.ids[(long)b * a.T + s]; id = id < 0 ? 0 : (id >= a.V ? a.V - 1 : id);
            tt = a.type_ids ? a.type_ids[(long)b * a.T + s] : 0;
        } else {
            tt = a.vis_type ? a.vis_type[(long)b * a.R + (s - a.T)] : 0;
        }
        tt = tt < 0 ? 0 : (tt >= a.TV ? a.TV - 1 : tt);
        float vv[NC][8];                                   // the row's chunks are fetched together, unconditionally (clamped column)
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
            const int col = (l32 + 32 * ci) * 8;
            load8(vv[ci], (const TT_*)a.dz + row * H + (col < H ? col : 0));
        }
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
            const int col = (l32 + 32 * ci) * 8;
            const bool ok = col < H;
            float (&v)[8] = vv[ci];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = ok ? v[j] : 0.f;
                acc[ci][j] += v[j];
                if (two_types) acc1[ci][j] += tt ? v[j] : 0.f;
                else if (ok) atomicAdd(&lds_type[tt * H + col + j], v[j]);
            }
            if (ok) {
                if (text) {
                    if (a.d_word) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) atomicAdd(&a.d_word[id * H + col + j], v[j]);
                    }
                } else if (a.d_vis_proj) {
                    store8((TT_*)a.d_vis_proj + ((long)b * a.R + (s - a.T)) * H + col, v);
                }
            }
        }
    }
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
        const int col = (l32 + 32 * ci) * 8;
        if (col < H) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                atomicAdd(&lds_pos[col + j], acc[ci][j]);
                if (two_types) {                            // 8 half-waves per workgroup: a handful of LDS adds per column
                    atomicAdd(&lds_type[H + col + j], acc1[ci][j]);
                    atomicAdd(&lds_type[col + j], acc[ci][j] - acc1[ci][j]);
                }
            }
        }
    }
    __syncthreads();
    float* dpos = text ? (a.d_pos ? a.d_pos + (long)(s < a.P ? s : a.P - 1) * H : nullptr) : a.d_pos_vis;
    float* dtype = text ? a.d_type : a.d_type_vis;
    for (int i = threadIdx.x; i < H; i += NT) {
        if (dpos) {
            if (text && gridDim.y == 1) dpos[i] += lds_pos[i];  // sole owner of this row
            else atomicAdd(&dpos[i], lds_pos[i]);            // all visual slots share position row 0
        }
    }
    if (dtype) {
        for (int i = threadIdx.x; i < H * a.TV; i += NT) atomicAdd(&dtype[i], lds_type[i]);
    }
}

#define VB_DISPATCH_NC(KERNEL, T, H, grid, smem, stream, args)                                  \
    do {                                                                                        \
        const int nc_ = ((H) / 8 + 31) / 32;                                                    \
        if (nc_ <= 1) VB_LAUNCH((KERNEL<T, 1>), grid, dim3(NT), smem, stream, args);            \
        else if (nc_ == 2) VB_LAUNCH((KERNEL<T, 2>), grid, dim3(NT), smem, stream, args);       \
        else if (nc_ == 3) VB_LAUNCH((KERNEL<T, 3>), grid, dim3(NT), smem, stream, args);       \
        else VB_LAUNCH((KERNEL<T, 4>), grid, dim3(NT), smem, stream, args);                     \
    } while (0)

static inline bool bad_h(int H) { return H <= 0 || (H % 8) != 0 || H > 8 * 32 * MAX_NC; }
static inline unsigned row_grid(long rows, int cap) {
    long g = (rows + HW_PER_BLOCK - 1) / HW_PER_BLOCK;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" int vb_ln_fwd(int dtype, const void* x, const void* resid, void* z_out, void* y, float* mean, float* rstd,
                         const float* gamma, const float* beta, int M, int H, float eps,
                         float p_in, uint32_t stream_in, float p_out, uint32_t stream_out, uint64_t seed,
                         void* stream) {
    return vb_ln_fwd_sp(dtype, x, resid, z_out, y, mean, rstd, gamma, beta, M, H, eps, p_in, stream_in, p_out, stream_out, seed,
                        nullptr, 0, nullptr, stream);
}

int vb_ln_fwd_sp(int dtype, const void* x, const void* resid, void* z_out, void* y, float* mean, float* rstd,
                 const float* gamma, const float* beta, int M, int H, float eps,
                 float p_in, uint32_t stream_in, float p_out, uint32_t stream_out, uint64_t seed,
                 void* y_split, int64_t ld_split, int* rebuild, void* stream) {
    if (!x || !y || !gamma || !beta || M <= 0 || bad_h(H)) return VB_ERR_ARG;
    if (rebuild && (p_out > 0.f || !z_out || H > 768)) return VB_ERR_ARG;   // a dropped output cannot give x-hat back; z_out is the fallback
    if (p_in < 0.f || p_in >= 1.f || p_out < 0.f || p_out >= 1.f) return VB_ERR_ARG;
    if (y_split && (dtype != VB_F32 || (ld_split % 16) || ld_split < 2 * H || (((uintptr_t)y_split) & 15))) return VB_ERR_ARG;
    LnFwdArgs a{x, resid, z_out, y, mean, rstd, gamma, beta, M, H, eps, make_drop(p_in, seed, stream_in),
                make_drop(p_out, seed, stream_out), (bf16*)y_split, (long)ld_split, rebuild};
    dim3 grid(row_grid(M, 4096));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VB_BF16) VB_DISPATCH_NC(ln_fwd_kernel, bf16, H, grid, 0, s, a);
    else if (dtype == VB_F32) VB_DISPATCH_NC(ln_fwd_kernel, float, H, grid, 0, s, a);
    else return VB_ERR_ARG;
    return vb_check_launch();
}

extern "C" int vb_ln_fwd_rb(int dtype, const void* x, const void* resid, void* z_out, void* y, float* mean, float* rstd,
                            const float* gamma, const float* beta, int M, int H, float eps,
                            float p_in, uint32_t stream_in, uint64_t seed, int* rebuild, void* stream) {
    if (!rebuild) return VB_ERR_ARG;
    return vb_ln_fwd_sp(dtype, x, resid, z_out, y, mean, rstd, gamma, beta, M, H, eps, p_in, stream_in, 0.f, 0, seed, nullptr, 0, rebuild, stream);
}

extern "C" int64_t vb_ln_bwd_ws_bytes(int M, int H) {
    return (int64_t)row_grid(M, 1024) * 3 * H * (int64_t)sizeof(float);
}

extern "C" int vb_ln_bwd(int dtype, const void* dy, const void* z, const float* mean, const float* rstd,
                         const float* gamma, void* dz, void* dx, float* dgamma, float* dbeta, float* dbias,
                         int M, int H, float p_in, uint32_t stream_in, float p_out, uint32_t stream_out,
                         uint64_t seed, float* ws, void* stream) {
    return vb_ln_bwd_sp(dtype, dy, z, mean, rstd, gamma, dz, dx, dgamma, dbeta, dbias, M, H, p_in, stream_in, p_out, stream_out, seed,
                        ws, nullptr, 0, nullptr, nullptr, nullptr, stream);
}

int vb_ln_bwd_sp(int dtype, const void* dy, const void* z, const float* mean, const float* rstd,
                 const float* gamma, void* dz, void* dx, float* dgamma, float* dbeta, float* dbias,
                 int M, int H, float p_in, uint32_t stream_in, float p_out, uint32_t stream_out,
                 uint64_t seed, float* ws, void* dx_split, int64_t ld_split, const void* y, const float* beta, const int* rebuild,
                 void* stream) {
    if (!dy || !z || !mean || !rstd || !gamma || !dz || M <= 0 || bad_h(H)) return VB_ERR_ARG;
    if (rebuild && (!y || !beta || p_out > 0.f || H > 768)) return VB_ERR_ARG;
    if (dx_split && (dtype != VB_F32 || (ld_split % 16) || ld_split < 2 * H || (((uintptr_t)dx_split) & 15))) return VB_ERR_ARG;
    // the image is of dx (the gradient after the input dropout).  dx == NULL with an image: dx leaves ONLY as the image -- the
    // kernel still needs a.dx non-NULL to run the dropout on its registers, and a.dx == a.dz suppresses the fp32 store
    if (dx_split && !dx) dx = dz;
    else if (p_in > 0.f && (!dx || dx == dz)) return VB_ERR_ARG;   // dropped and un-dropped grads differ
    LnBwdArgs a{dy, z, mean, rstd, gamma, dz, dx, dgamma, dbeta, dbias, M, H, make_drop(p_in, seed, stream_in),
                make_drop(p_out, seed, stream_out), ws, (bf16*)dx_split, (long)ld_split, y, beta, rebuild};
    dim3 grid(row_grid(M, ws ? 1024 : 256));
    hipStream_t s = (hipStream_t)stream;
    const size_t smem = (size_t)(H + (H >> 5) + 1) * WAVES_PER_BLOCK * 3 * sizeof(float);      // padded pitch: ln_bwd12_kernel
    // generic kernel <.., 4>: four waves per SIMD (128 VGPRs, a 12-byte spill) beat three without the spill: 109 vs 130 us at M = 83,968
    // 512 < H <= 768: twelve columns per lane (113 VGPRs, no spills, every lane busy): 188.7 -> 182.4 us per launch at M = 167,936 in the
    // step, 191.8 us in the rebuild-capable form (profiles/r04_ln_rebuild.txt).  The generic kernel's rebuild-capable form exists for
    // H <= 512 only (one chunk per lane, 87 VGPRs); with two full chunks it spills (340 us) -- layer.hip does not ask for it there.
    const bool twelve = H > 512 && H <= 768;
    if (dtype == VB_BF16 && rebuild) {
        if (twelve) VB_LAUNCH((ln_bwd12_kernel<bf16, true>), grid, dim3(NT), smem, s, a);
        else if (H <= 512) VB_LAUNCH((ln_bwd_kernel<bf16, 1, 4, 1>), grid, dim3(NT), smem, s, a);
        else return VB_ERR_UNSUPPORTED;
    } else if (dtype == VB_BF16) {
        if (twelve) VB_LAUNCH((ln_bwd12_kernel<bf16, false>), grid, dim3(NT), smem, s, a);
        else if (H <= 512) VB_LAUNCH((ln_bwd_kernel<bf16, 1, 4>), grid, dim3(NT), smem, s, a);
        else VB_LAUNCH((ln_bwd_kernel<bf16, 2, 4>), grid, dim3(NT), smem, s, a);
    } else if (dtype == VB_F32 && rebuild) {
        if (twelve) VB_LAUNCH((ln_bwd12_kernel<float, true>), grid, dim3(NT), smem, s, a);
        else if (H <= 512) VB_LAUNCH((ln_bwd_kernel<float, 1, 4, 1>), grid, dim3(NT), smem, s, a);
        else return VB_ERR_UNSUPPORTED;
    } else if (dtype == VB_F32) {
        if (twelve) VB_LAUNCH((ln_bwd12_kernel<float, false>), grid, dim3(NT), smem, s, a);
        else if (H <= 512) VB_LAUNCH((ln_bwd_kernel<float, 1, 4>), grid, dim3(NT), smem, s, a);
        else VB_LAUNCH((ln_bwd_kernel<float, 2, 4>), grid, dim3(NT), smem, s, a);
    } else return VB_ERR_ARG;
    if (ws && (dgamma || dbeta || dbias)) {
        const int slices = grid.x >= 256 ? 4 : 1;
        VbReduceJobs* defer = vb_reduce_defer_slot();
        if (defer && defer->n + 3 <= 8) {       // the caller launches the second stage (vb_rt.h: VbReduceJobs)
            float* outs[3] = {dgamma, dbeta, dbias};
            for (int w = 0; w < 3; ++w)
                if (outs[w]) vb_reduce_defer(ws + (long)w * H, outs[w], (int)grid.x, H, 3L * H, slices);
            return vb_check_launch();
        }
        dim3 g2((unsigned)((H + 31) / 32), 3, slices);
        VB_LAUNCH(ln_bwd_reduce_kernel, g2, dim3(1024), 32 * 33 * sizeof(float), s, (const float*)ws, (int)grid.x, H,
                  dgamma, dbeta, dbias);
    }
    return vb_check_launch();
}

extern "C" int vb_ln_bwd_rb(int dtype, const void* dy, const void* z, const float* mean, const float* rstd,
                            const float* gamma, void* dz, void* dx, float* dgamma, float* dbeta, float* dbias,
                            int M, int H, float p_in, uint32_t stream_in, uint64_t seed, float* ws,
                            const void* y, const float* beta, const int* rebuild, void* stream) {
    if (!rebuild) return VB_ERR_ARG;
    return vb_ln_bwd_sp(dtype, dy, z, mean, rstd, gamma, dz, dx, dgamma, dbeta, dbias, M, H, p_in, stream_in, 0.f, 0, seed, ws,
                        nullptr, 0, y, beta, rebuild, stream);
}

extern "C" int vb_embed_fwd(int dtype, const int64_t* input_ids, const int64_t* token_type_ids,
                            const int64_t* visual_type, const void* vis_proj, const float* word, const float* pos,
                            const float* type, const float* pos_vis, const float* type_vis, const float* pos_align,
                            void* z, int B, int T, int R, int H, int V, int type_vocab, int max_pos, void* stream) {
    if (!input_ids || !word || !pos || !type || !z || B <= 0 || T <= 0 || R < 0 || bad_h(H)) return VB_ERR_ARG;
    if (R > 0 && (!vis_proj || !pos_vis || !type_vis)) return VB_ERR_ARG;
    if (type_vocab <= 0 || type_vocab > 8 || max_pos <= 0 || V <= 0) return VB_ERR_ARG;
    EmbArgs a{input_ids, token_type_ids, visual_type, vis_proj, word, pos, type, pos_vis, type_vis,
              R > 0 ? pos_align : nullptr, z, B, T, R, H, V, type_vocab, max_pos};
    dim3 grid(row_grid((long)B * (T + R), 4096));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VB_BF16) VB_DISPATCH_NC(embed_fwd_kernel, bf16, H, grid, 0, s, a);
    else if (dtype == VB_F32) VB_DISPATCH_NC(embed_fwd_kernel, float, H, grid, 0, s, a);
    else return VB_ERR_ARG;
    return vb_check_launch();
}

extern "C" int vb_embed_bwd(int dtype, const void* dz, const int64_t* input_ids, const int64_t* token_type_ids,
                            const int64_t* visual_type, float* d_word, float* d_pos, float* d_type,
                            float* d_pos_vis, float* d_type_vis, void* d_vis_proj,
                            int B, int T, int R, int H, int V, int type_vocab, int max_pos, void* stream) {
    if (!dz || !input_ids || B <= 0 || T <= 0 || R < 0 || bad_h(H)) return VB_ERR_ARG;
    if (type_vocab <= 0 || type_vocab > 8 || max_pos <= 0 || V <= 0) return VB_ERR_ARG;
    // the word-table scatter runs as its own coalesced-atomic kernel; the main kernel gets no d_word
    EmbBwdArgs a{dz, input_ids, token_type_ids, visual_type, nullptr, d_pos, d_type, d_pos_vis, d_type_vis,
                 d_vis_proj, B, T, R, H, V, type_vocab, max_pos};
    dim3 grid((unsigned)(T + R), (unsigned)(B >= 64 ? 8 : (B >= 16 ? 2 : 1)));
    hipStream_t s = (hipStream_t)stream;
    const size_t smem = (size_t)H * (1 + type_vocab) * sizeof(float);
    if (dtype == VB_BF16) VB_DISPATCH_NC(embed_bwd_kernel, bf16, H, grid, smem, s, a);
    else if (dtype == VB_F32) VB_DISPATCH_NC(embed_bwd_kernel, float, H, grid, smem, s, a);
    else return VB_ERR_ARG;
    if (d_word) {
        dim3 g2((unsigned)(B * T));
        if (dtype == VB_BF16) VB_LAUNCH(embed_word_scatter_kernel<bf16>, g2, dim3(NT), 0, s, (const bf16*)dz, input_ids, d_word, B, T, T + R, H, V);
        else VB_LAUNCH(embed_word_scatter_kernel<float>, g2, dim3(NT), 0, s, (const float*)dz, input_ids, d_word, B, T, T + R, H, V);
    }
    return vb_check_launch();
}
