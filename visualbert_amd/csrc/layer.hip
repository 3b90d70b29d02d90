// layer.hip -- host-side sequencing of one whole BertLayer (forward: 7 launches, backward: 15) behind a
// single C-ABI call each, so the Python side pays two FFI calls per layer instead of ~22 op dispatches.
// Pure orchestration: every launch goes through the same entry points the per-op ABI exposes.
//
// Replaces BertLayer.forward and its autograd backward
//   (pytorch_pretrained_bert/modeling.py:331-341 = BertAttention :276-293 [BertSelfAttention :231-261,
//    BertSelfOutput :270-274] -> BertIntermediate :302-305 -> BertOutput :315-319).
#include "vb_rt.h"
#include "../../include/visualbert_hip.h"

namespace {

// dtype: the caller's (VB_F32 / VB_BF16 / VB_BF16X3); edt: what the non-GEMM kernels see (bf16x3 keeps every activation
// in fp32 and splits a GEMM's operands into bf16 hi | lo planes right in front of it); es: bytes per activation element
struct Dims { int B, S, H, I, nh; long M; size_t es; int dtype, edt; bool x3; };

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

// saved-for-backward workspace of one layer
struct Saved {
    unsigned char *qkv, *ctx, *z1, *a_out, *pre, *inter, *z2;
    float *lse, *mean1, *rstd1, *mean2, *rstd2;
    uint64_t* keepbits;
    int* ln_flags;      // [2] device ints written by the two LayerNorm forwards: 1 = no z was written, the backward rebuilds x-hat from y
    // bf16x3 only: the split (hi | lo) images of the four GEMM inputs the forward made anyway -- they are the x operands of the
    // backward's weight-gradient launch, so keeping them saves four split passes per layer (19 % of the mode's split traffic)
    unsigned char *sp_hin, *sp_ctx, *sp_aout, *sp_inter;
    size_t total;
};

Saved carve_saved(unsigned char* base, const Dims& d, bool attn_dropout) {
    Saved s;
    size_t o = 0;
    auto take = [&](size_t bytes) { unsigned char* p = base ? base + o : nullptr; o += al(bytes); return p; };
    s.qkv = take((size_t)d.M * 3 * d.H * d.es);
    s.ctx = take(d.x3 ? 0 : (size_t)d.M * d.H * d.es);      // split-operand mode: the context and the FFN activation exist only as
                                                              // their images (sp_ctx, sp_inter below): only GEMMs read them
    s.z1 = take((size_t)d.M * d.H * d.es);
    s.a_out = take((size_t)d.M * d.H * d.es);
    s.pre = take((size_t)d.M * d.I * d.es);
    s.inter = take(d.x3 ? 0 : (size_t)d.M * d.I * d.es);
    s.z2 = take((size_t)d.M * d.H * d.es);
    s.lse = (float*)take((size_t)d.B * d.nh * d.S * 4);
    s.mean1 = (float*)take((size_t)d.M * 4);
    s.rstd1 = (float*)take((size_t)d.M * 4);
    s.mean2 = (float*)take((size_t)d.M * 4);
    s.rstd2 = (float*)take((size_t)d.M * 4);
    s.keepbits = (uint64_t*)take(attn_dropout ? (size_t)d.B * d.nh * vb_attn_keepbits_words(d.S) * 8 : 0);
    s.ln_flags = (int*)take(8);
    const size_t sh = d.x3 ? (size_t)d.M * 2 * d.H * 2 : 0, si = d.x3 ? (size_t)d.M * 2 * d.I * 2 : 0;
    s.sp_hin = take(sh); s.sp_ctx = take(sh); s.sp_aout = take(sh); s.sp_inter = take(si);
    s.total = o;
    return s;
}

// scratch (temporaries, reusable by every layer on the same stream)
struct Scratch {
    unsigned char *t_h0, *t_h1, *t_h2, *t_h3, *t_h4, *t_h5, *t_i, *t_3h;
    float* dsum;
    float *ln_ws, *ln_ws1;      // column partials of the output / the attention-output LayerNorm's backward (both alive to the layer's
                                // one second-stage reduction launch)
    // bf16x3 only: split (hi | lo) images of the backward's output gradients, [M, 2 x features] bf16 each, alive until the
    // grouped weight-gradient launch (the x operands' images come from the forward: Saved)
    unsigned char *sp_dfo, *sp_dpre, *sp_dao, *sp_dqkv;
    size_t total;
};
Scratch carve_scratch(unsigned char* base, const Dims& d) {
    Scratch s;
    size_t o = 0;
    auto take = [&](size_t bytes) { unsigned char* p = base ? base + o : nullptr; o += al(bytes); return p; };
    s.t_h0 = take((size_t)d.M * d.H * d.es);
    s.t_h1 = take((size_t)d.M * d.H * d.es);
    s.t_h2 = take((size_t)d.M * d.H * d.es);
    s.t_h3 = take((size_t)d.M * d.H * d.es);
    s.t_h4 = take((size_t)d.M * d.H * d.es);
    s.t_h5 = take((size_t)d.M * d.H * d.es);
    s.t_i = take((size_t)d.M * d.I * d.es);
    s.t_3h = take((size_t)d.M * 3 * d.H * d.es);
    s.dsum = (float*)take((size_t)vb_attn_bwd_ws_floats((int)d.B, (int)d.S, (int)d.nh) * 4);
    s.ln_ws = (float*)take((size_t)vb_ln_bwd_ws_bytes((int)d.M, d.H));
    s.ln_ws1 = (float*)take((size_t)vb_ln_bwd_ws_bytes((int)d.M, d.H));
    const size_t sh = d.x3 ? (size_t)d.M * 2 * d.H * 2 : 0, si = d.x3 ? (size_t)d.M * 2 * d.I * 2 : 0;
    s.sp_dfo = take(sh); s.sp_dpre = take(si); s.sp_dao = take(sh); s.sp_dqkv = take(3 * sh);
    s.total = o;
    return s;
}

bool fill_dims(Dims& d, int dtype, int B, int S, int H, int I, int nh) {
    if (B <= 0 || S <= 0 || H <= 0 || I <= 0 || nh <= 0 || nh * 64 != H || (H % 8) || (I % 8)) return false;
    if (dtype != VB_F32 && dtype != VB_BF16 && dtype != VB_BF16X3) return false;
    if (dtype == VB_BF16X3 && ((H % 64) || (I % 64))) return false;       // split-operand GEMMs reduce over whole K tiles
    d.B = B; d.S = S; d.H = H; d.I = I; d.nh = nh; d.M = (long)B * S; d.es = dtype == VB_BF16 ? 2 : 4;
    d.dtype = dtype; d.x3 = dtype == VB_BF16X3; d.edt = d.x3 ? VB_F32 : dtype;
    return true;
}

// y[M, n] = epilogue(x[M, k] W[n, k]^T): in the split-operand mode x (fp32) is split into `stage` first and W arrives split
// split_y (split-operand mode only): y is written as a split image [M, 2 n] itself (the result feeds only GEMMs)
int linear(const Dims& d, const void* x, int k, unsigned char* stage, const void* w, int64_t ldw, void* y, int n,
           const float* bias, const void* addend, int act, const void* aux_in, void* aux_out, float* colsum, void* stream,
           bool split_y = false) {
    const int M = (int)d.M;
    if (!d.x3)
        return vb_gemm(d.dtype, d.dtype, VB_KCONTIG, VB_KCONTIG, x, k, w, ldw, y, n, M, n, k, 1.f, nullptr, bias, addend, n, act,
                       aux_in, aux_out, n, 0, colsum, stream);
    if (stage) {                                               // nullptr: `x` already is the split image
        const int rc = vb_split_bf16((const float*)x, k, stage, 2 * k, M, k, stream);
        if (rc != VB_OK) return rc;
        x = stage;
    }
    return vb_gemm(VB_BF16X3, split_y ? VB_BF16X3 : VB_F32, VB_KCONTIG, VB_KCONTIG, x, 2 * k, w, ldw, y, split_y ? 2 * n : n, M, n, k,
                   1.f, nullptr, bias, addend, n, act, aux_in, aux_out, n, 0, colsum, stream);
}

#ifndef VB_DEFER_REDUCE
#define VB_DEFER_REDUCE 1       // 0: every second-stage reduction right behind its producer (the A/B arm of profiles/r06_small_batch_ab.txt)
#endif
#define VB_TRY(expr) do { int rc_ = (expr); if (rc_ != VB_OK) return rc_; } while (0)


}  // namespace

extern "C" int64_t vb_bert_layer_saved_bytes(int dtype, int B, int S, int H, int I, int nh, float p_attn) {
    Dims d;
    if (!fill_dims(d, dtype, B, S, H, I, nh)) return -1;
    return (int64_t)carve_saved(nullptr, d, p_attn > 0.f).total;
}
extern "C" int64_t vb_bert_layer_scratch_bytes(int dtype, int B, int S, int H, int I, int nh) {
    Dims d;
    if (!fill_dims(d, dtype, B, S, H, I, nh)) return -1;
    return (int64_t)carve_scratch(nullptr, d).total;
}

// weights[]: VB_LW_* order (T for matrices read by the GEMMs, fp32 for biases / LayerNorm)
extern "C" int vb_bert_layer_fwd(int dtype, const void* h_in, const float* mask_add, void* h_out,
                                 void* saved, void* scratch, const void* const* weights,
                                 int B, int S, int H, int I, int nh, float p_hidden, float p_attn, float eps,
                                 uint64_t seed, uint32_t sid, void* stream) {
    Dims d;
    if (!fill_dims(d, dtype, B, S, H, I, nh) || !h_in || !mask_add || !h_out || !saved || !scratch || !weights)
        return VB_ERR_ARG;
    Saved sv = carve_saved((unsigned char*)saved, d, p_attn > 0.f);
    Scratch sc = carve_scratch((unsigned char*)scratch, d);
    const int M = (int)d.M;
    const void* wqkv = weights[VB_LW_QKV_W]; const float* bqkv = (const float*)weights[VB_LW_QKV_B];
    const void* wo = weights[VB_LW_AO_W]; const float* bo = (const float*)weights[VB_LW_AO_B];
    const float* g1 = (const float*)weights[VB_LW_LN1_G]; const float* b1 = (const float*)weights[VB_LW_LN1_B];
    const void* wi = weights[VB_LW_FI_W]; const float* bi = (const float*)weights[VB_LW_FI_B];
    const void* wo2 = weights[VB_LW_FO_W]; const float* bo2 = (const float*)weights[VB_LW_FO_B];
    const float* g2 = (const float*)weights[VB_LW_LN2_G]; const float* b2 = (const float*)weights[VB_LW_LN2_B];

    const int edt = d.edt;
    const int64_t wk = d.x3 ? 2 : 1;                   // leading dimension of a weight matrix per K element (split: hi | lo)
    // 1. packed Q|K|V projection
    VB_TRY(linear(d, h_in, H, sv.sp_hin, wqkv, wk * H, sv.qkv, 3 * H, bqkv, nullptr, VB_ACT_NONE, nullptr, nullptr, nullptr, stream));
    // 2. fused attention
    //    (split-operand mode: the kernels that PRODUCE a GEMM input write its hi | lo image themselves -- the context here, the
    //     LayerNorm output below, the three output gradients in the backward -- so the only stand-alone split pass left per layer
    //     is the one over h_in, which the previous layer / the embeddings produced)
    //    and where ONLY GEMMs read a result -- the context, and in the backward dfo / dao / dqkv -- its fp32 form is not written
    //    at all (sv.ctx, sc.t_h1 / t_h4 / t_3h stay untouched in this mode)
    VB_TRY(vb_attn_fwd_sp(d.dtype, sv.qkv, mask_add, sv.ctx, sv.lse, sv.keepbits, B, S, nh, 64, p_attn, seed, sid,
                          d.x3 ? sv.sp_ctx : nullptr, 1, stream));
    // 3 + 4. attention output projection, dropout + residual + LayerNorm
    //    (the pre-LN sum z is NOT written by the LayerNorm when the backward can rebuild x-hat from the output it reads anyway --
    //     decided in the kernel from gamma / beta and recorded in sv.ln_flags)
    //    Round 6 built the other split of this block -- the GEMM's epilogue applies dropout + residual and writes z straight into the saved
    //    slot, the LayerNorm launch reads ONE tensor -- measured it, and declined it: with dropout the mask
    //    generator in the persistent kernel's epilogue costs more than the LayerNorm saves (step 118.0 -> 118.45 ms), and even the
    //    residual-only form normalises a bf16-rounded residual stream (max |dlogit| against the reference 3.2e-3 -> 5.8e-3).  The arm
    //    stays reachable in the developer library (debug bit 29: gemm.hip vb_gemm_dropres; profiles/r06_dropres_epilogue_ab.txt).
    const bool rb = H <= 768;                           // (wider rows: the backward's rebuild-capable form does not pay, layernorm.hip)
    int* rb1 = rb ? sv.ln_flags : nullptr;
    int* rb2 = rb ? sv.ln_flags + 1 : nullptr;
    auto out_block = [&](const void* x, int k, const void* w, const float* bias, const void* resid, unsigned char* tmp, unsigned char* zslot,
                         void* y, float* mean, float* rstd, const float* gam, const float* bet, uint32_t site, int* rbf,
                         void* y_split) -> int {
        if (d.dtype == VB_BF16 && vb_gemm_fuse_residual_armed()) {
            const int rc = vb_gemm_dropres(VB_BF16, VB_BF16, VB_KCONTIG, VB_KCONTIG, x, k, w, k, zslot, H, M, H, k, 1.f, nullptr, bias, resid, H,
                                           VB_ACT_NONE, nullptr, nullptr, H, 0, nullptr, p_hidden, seed, site, stream);
            if (rc == VB_OK)                            // z is in its slot already: x = z_out = the slot (the kernel reads a row before it writes it)
                return vb_ln_fwd_sp(edt, zslot, nullptr, zslot, y, mean, rstd, gam, bet, M, H, eps, 0.f, site, 0.f, 0, seed, nullptr, 0, rbf, stream);
            if (rc != VB_ERR_UNSUPPORTED) return rc;    // (small problems: declined, nothing launched)
        }
        VB_TRY(linear(d, x, k, nullptr, w, wk * k, tmp, H, bias, nullptr, VB_ACT_NONE, nullptr, nullptr, nullptr, stream));
        return vb_ln_fwd_sp(edt, tmp, resid, zslot, y, mean, rstd, gam, bet, M, H, eps, p_hidden, site, 0.f, 0, seed, y_split, 2 * H, rbf, stream);
    };
    VB_TRY(out_block(d.x3 ? (const void*)sv.sp_ctx : (const void*)sv.ctx, H, wo, bo, h_in, sc.t_h0, sv.z1, sv.a_out, sv.mean1, sv.rstd1,
                     g1, b1, sid + 1, rb1, d.x3 ? sv.sp_aout : nullptr));
    // 5. FFN in + erf-GELU (GELU' kept for backward)
    //    (split-operand mode: the activation leaves the GEMM as a split image -- only GEMMs read it: FFN-out and its wgrad)
    VB_TRY(linear(d, d.x3 ? (const void*)sv.sp_aout : (const void*)sv.a_out, H, nullptr, wi, wk * H,
                  d.x3 ? (void*)sv.sp_inter : (void*)sv.inter, I, bi, nullptr, VB_ACT_GELU_SAVE_GRAD, nullptr, sv.pre, nullptr, stream, d.x3));
    // 6 + 7. FFN out, dropout + residual + LayerNorm (the same block: modeling.py:316-318)
    VB_TRY(out_block(d.x3 ? (const void*)sv.sp_inter : (const void*)sv.inter, I, wo2, bo2, sv.a_out, sc.t_h1, sv.z2, h_out, sv.mean2, sv.rstd2,
                     g2, b2, sid + 4, rb2, nullptr));
    return VB_OK;
}

// grads[]: fp32 accumulation targets in VB_LW_* order (all required)
extern "C" int vb_bert_layer_bwd(int dtype, const void* h_in, const void* h_out, const float* mask_add, const void* d_out, void* d_in,
                                 const void* saved, void* scratch, const void* const* weights, void* const* grads,
                                 const void* const* weights_t, const int64_t* ld_t,
                                 int B, int S, int H, int I, int nh, float p_hidden, float p_attn,
                                 uint64_t seed, uint32_t sid, void* stream) {
    Dims d;
    if (!fill_dims(d, dtype, B, S, H, I, nh) || !h_in || !mask_add || !d_out || !d_in || !saved || !scratch ||
        !weights || !grads)
        return VB_ERR_ARG;
    if (H <= 768 && !h_out) return VB_ERR_ARG;               // the output LayerNorm's backward may rebuild x-hat from it
    Saved sv = carve_saved((unsigned char*)saved, d, p_attn > 0.f);
    Scratch sc = carve_scratch((unsigned char*)scratch, d);
    const int M = (int)d.M;
    const void* wqkv = weights[VB_LW_QKV_W];
    const void* wo = weights[VB_LW_AO_W];
    const float* g1 = (const float*)weights[VB_LW_LN1_G]; const float* b1 = (const float*)weights[VB_LW_LN1_B];
    const void* wi = weights[VB_LW_FI_W];
    const void* wo2 = weights[VB_LW_FO_W];
    const float* g2 = (const float*)weights[VB_LW_LN2_G]; const float* b2 = (const float*)weights[VB_LW_LN2_B];
    const bool rb = H <= 768;
    const int* rb1 = rb ? sv.ln_flags : nullptr;
    const int* rb2 = rb ? sv.ln_flags + 1 : nullptr;
    float* G[VB_LW_COUNT];
    for (int i = 0; i < VB_LW_COUNT; ++i) { G[i] = (float*)grads[i]; if (!G[i]) return VB_ERR_ARG; }
    // dgrad dx[M,in] = dy[M,out] W[out,in]: with W^T [in, ld>=out] both operands are K-contiguous (LDS-direct
    // loads); otherwise W is read K-strided and transposed on the fly
    const int edt = d.edt;
    if (d.x3 && (!weights_t || !weights_t[0] || !weights_t[1] || !weights_t[2] || !weights_t[3] || !ld_t)) return VB_ERR_UNSUPPORTED;
    // split-operand mode: dy (fp32) is split into `stage` -- kept for the grouped weight-gradient launch -- and W^T arrives split
    auto dgrad = [&](const void* dy, int n_out, const void* w, int which_t, int n_in, void* dx, const void* addend,
                     int act, const void* aux, float* colsum = nullptr, unsigned char* stage = nullptr,
                     bool split_dx = false) -> int {
        const void* wt = weights_t ? weights_t[which_t] : nullptr;
        if (d.x3)
            return linear(d, dy, n_out, stage, wt, ld_t[which_t], dx, n_in, nullptr, addend, act, aux, nullptr, colsum, stream,
                          split_dx);
        if (wt)
            return vb_gemm(dtype, dtype, VB_KCONTIG, VB_KCONTIG, dy, n_out, wt, ld_t[which_t], dx, n_in, M, n_in, n_out,
                           1.f, nullptr, nullptr, addend, n_in, act, aux, nullptr, n_in, 0, colsum, stream);
        return vb_gemm(dtype, dtype, VB_KCONTIG, VB_KSTRIDED, dy, n_out, w, n_in, dx, n_in, M, n_in, n_out, 1.f, nullptr,
                       nullptr, addend, n_in, act, aux, nullptr, n_in, 0, colsum, stream);
    };

    // Every output gradient (dfo, dpre, dao, dqkv) stays alive to the end of the layer so that the four weight
    // gradients run as ONE grouped launch (vb_wgrad_grouped): 108 output tiles x 2 token slices fill the chip with
    // 164-K-tile items, where four separate launches had 9..36 tiles each and needed 7..28 slices (atomic traffic x4).
    // the second stages of the two LayerNorm backwards' column reductions and of the attention backward's bias gradient run as ONE
    // launch at the end of the layer (vb_rt.h: VbReduceJobs)
    VbReduceJobs tail{};
    struct DeferScope {
        explicit DeferScope(VbReduceJobs* j) { vb_reduce_defer_slot() = VB_DEFER_REDUCE ? j : nullptr; }
        ~DeferScope() { vb_reduce_defer_slot() = nullptr; }
    } defer_scope(&tail);
    unsigned char* dz2 = sc.t_h0;                        // d(a_out) through the residual of the output LN
    unsigned char* dfo = p_hidden > 0.f ? sc.t_h1 : dz2; // d(FFN-out dense output)
    // 1. output LayerNorm backward (+ bias gradient of the FFN-out dense as a by-product)
    //    (split-operand mode: + the hi | lo image of dfo, the next dgrad's and the weight-gradient launch's operand)
    VB_TRY(vb_ln_bwd_sp(edt, d_out, sv.z2, sv.mean2, sv.rstd2, g2, dz2, d.x3 ? nullptr : dfo, G[VB_LW_LN2_G], G[VB_LW_LN2_B],
                        G[VB_LW_FO_B], M, H, p_hidden, sid + 4, 0.f, 0, seed, sc.ln_ws, d.x3 ? sc.sp_dfo : nullptr, 2 * H, h_out, b2, rb2, stream));
    // 2. dgrad FFN-out with the saved GELU' folded into the epilogue: dpre = (dfo Wo2) * gelu'(pre)
    //    (+ bias gradient of FFN-in = column sums of dpre, accumulated by the same epilogue)
    //    (split-operand mode: dpre leaves the GEMM as a split image -- only the next dgrad and the wgrad launch read it)
    unsigned char* dpre = d.x3 ? sc.sp_dpre : sc.t_i;
    VB_TRY(dgrad(d.x3 ? (const void*)sc.sp_dfo : (const void*)dfo, H, wo2, VB_LWT_FO, I, dpre, nullptr, VB_ACT_MUL_AUX, sv.pre,
                 G[VB_LW_FI_B], nullptr, d.x3));
    // 3. dgrad FFN-in + residual gradient: da = dpre Wi + dz2
    VB_TRY(dgrad(dpre, I, wi, VB_LWT_FI, H, sc.t_h2, dz2, VB_ACT_NONE, nullptr, nullptr, nullptr));
    // 4. attention-output LayerNorm backward
    unsigned char* dz1 = sc.t_h5;
    unsigned char* dao = p_hidden > 0.f ? sc.t_h4 : dz1;
    VB_TRY(vb_ln_bwd_sp(edt, sc.t_h2, sv.z1, sv.mean1, sv.rstd1, g1, dz1, d.x3 ? nullptr : dao, G[VB_LW_LN1_G], G[VB_LW_LN1_B],
                        G[VB_LW_AO_B], M, H, p_hidden, sid + 1, 0.f, 0, seed, sc.ln_ws1, d.x3 ? sc.sp_dao : nullptr, 2 * H, sv.a_out, b1, rb1, stream));
    // 5. dgrad attention-out: dctx = dao Wo
    VB_TRY(dgrad(d.x3 ? (const void*)sc.sp_dao : (const void*)dao, H, wo, VB_LWT_AO, H, sc.t_h3, nullptr, VB_ACT_NONE, nullptr, nullptr,
                 nullptr));
    // 6-8. attention backward (one pass for bf16 and S <= 192, else dQ pass + dK/dV pass) + the q | k | v bias gradient
    //      (per-sample sums out of the one-pass kernel's accumulators; a column-sum pass over dqkv otherwise)
    VB_TRY(vb_attn_bwd_sp(d.dtype, sv.qkv, mask_add, sc.t_h3, sv.lse, sv.keepbits, sc.dsum, sc.t_3h, d.x3 ? nullptr : sv.ctx, G[VB_LW_QKV_B], B, S, nh,
                          64, p_attn, seed, sid, d.x3 ? sc.sp_dqkv : nullptr, 1, stream));
    // 9. dgrad QKV + residual gradient: dh = dqkv Wqkv + dz1
    VB_TRY(dgrad(d.x3 ? (const void*)sc.sp_dqkv : (const void*)sc.t_3h, 3 * H, wqkv, VB_LWT_QKV, H, d_in, dz1, VB_ACT_NONE, nullptr,
                 nullptr, nullptr));
    // 10. the four weight gradients: dW_fo[H,I] += dfo^T inter, dW_fi[I,H] += dpre^T a_out, dW_ao[H,H] += dao^T ctx,
    //     dW_qkv[3H,H] += dqkv^T h_in
    if (d.x3) {
        // the dy images were split for the dgrads above; the four layer inputs' images were kept by the forward
        const void* dys[4] = {sc.sp_dfo, sc.sp_dpre, sc.sp_dao, sc.sp_dqkv};
        const int64_t ld_dy[4] = {2 * H, 2 * I, 2 * H, 6 * H};
        const void* xs[4] = {sv.sp_inter, sv.sp_aout, sv.sp_ctx, sv.sp_hin};
        const int64_t ld_x[4] = {2 * I, 2 * H, 2 * H, 2 * H};
        void* dws[4] = {G[VB_LW_FO_W], G[VB_LW_FI_W], G[VB_LW_AO_W], G[VB_LW_QKV_W]};
        const int64_t ld_dw[4] = {I, H, H, H};
        const int n_out[4] = {H, I, H, 3 * H};
        const int n_in[4] = {I, H, H, H};
        VB_TRY(vb_wgrad_grouped(VB_BF16X3, 4, dys, ld_dy, xs, ld_x, dws, ld_dw, n_out, n_in, M, 1.f, nullptr, stream));
    } else {
        const void* dys[4] = {dfo, sc.t_i, dao, sc.t_3h};
        const int64_t ld_dy[4] = {H, I, H, 3 * H};
        const void* xs[4] = {sv.inter, sv.a_out, sv.ctx, h_in};
        const int64_t ld_x[4] = {I, H, H, H};
        void* dws[4] = {G[VB_LW_FO_W], G[VB_LW_FI_W], G[VB_LW_AO_W], G[VB_LW_QKV_W]};
        const int64_t ld_dw[4] = {I, H, H, H};
        const int n_out[4] = {H, I, H, 3 * H};
        const int n_in[4] = {I, H, H, H};
        VB_TRY(vb_wgrad_grouped(dtype, 4, dys, ld_dy, xs, ld_x, dws, ld_dw, n_out, n_in, M, 1.f, nullptr, stream));
    }
    return vb_reduce_jobs_launch(tail, stream);
}
