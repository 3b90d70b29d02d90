// heads.hip -- the small kernels of the branches outside BASELINE.json's configs:
// image_text_alignment position means and the Flickr30k grounding head.  All of it is HBM/latency-bound
// index work over a few thousand rows -- coalesced rows, no MFMA (the projections around it are vb_gemm calls).
#include "vb_rt.h"
#include "../../include/visualbert_hip.h"

namespace {

constexpr int NT = 256;

VB_DEVICE long clamp_index(long i, long n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// ---- image_text_alignment (pytorch_pretrained_bert/modeling.py:1223-1245) -------------------------------------------
// out[b*R + r, :] = mean over the aligned words a (alignment[b, r, a] != -1) of pos[alignment[b, r, a], :];
// a region with no aligned word gets zeros (the reference divides a zero sum by the guarded count 1).
VB_KERNEL VB_LAUNCH_BOUNDS(NT) align_pos_fwd_kernel(const int64_t* alignment, const float* pos, float* out,
                                                   int R, int Ra, int A, int H, int P) {
    const int row = blockIdx.x, b = row / R, r = row - b * R;
    const int64_t* al = alignment + ((long)b * Ra + r) * A;
    int cnt = 0;
    for (int a = 0; a < A; ++a) cnt += al[a] != -1;
    const float inv = 1.0f / (float)(cnt > 0 ? cnt : 1);
    for (int c = threadIdx.x; c < H; c += NT) {
        float acc = 0.f;
        for (int a = 0; a < A; ++a) {
            const long idx = al[a];
            if (idx != -1) acc += pos[clamp_index(idx, P) * H + c];
        }
        out[(long)row * H + c] = acc * inv;
    }
}

// d_pos[alignment[b, r, a], :] += dz[b, T + r, :] / count   (fp32 atomics, one coalesced row per aligned word)
template <typename T>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) align_pos_bwd_kernel(const T* dz, const int64_t* alignment, float* d_pos,
                                                   int Tlen, int R, int Ra, int A, int H, int P) {
    const int row = blockIdx.x, b = row / R, r = row - b * R;
    const int64_t* al = alignment + ((long)b * Ra + r) * A;
    int cnt = 0;
    for (int a = 0; a < A; ++a) cnt += al[a] != -1;
    if (cnt == 0) return;
    const float inv = 1.0f / (float)cnt;
    const T* src = dz + ((long)b * (Tlen + R) + Tlen + r) * H;
    for (int c = threadIdx.x; c < H; c += NT) {
        const float g = to_f32(src[c]) * inv;
        for (int a = 0; a < A; ++a) {
            const long idx = al[a];
            if (idx != -1) vb_atomic_add_noret(d_pos + clamp_index(idx, P) * H + c, g);
        }
    }
}

// ---- batched_index_select (modeling.py:1713-1716) and its adjoint ----------------------------------------------------
// out[b*E + e, :] = x[b, index[b, e], :]; the reference turns the -1 padding into position 0 first (:1574).
template <typename T>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) gather_index_kernel(const T* x, const int64_t* index, T* out, int S, int E, int H) {
    const int row = blockIdx.x, b = row / E;
    const long s = clamp_index(index[row], S);
    const T* src = x + ((long)b * S + s) * H;
    for (int c = threadIdx.x; c < H; c += NT) out[(long)row * H + c] = src[c];
}

// dx[b, s, :] = addend[b, s, :] + sum over the entities e with index[b, e] == s of dsel[b*E + e, :]
// (one workgroup per destination row: deterministic, and two entities may point at the same word)
template <typename T>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) scatter_index_kernel(const T* dsel, const int64_t* index, const T* addend, T* dx,
                                                   int S, int E, int H) {
    const int row = blockIdx.x, b = row / S, s = row - b * S;
    for (int c = threadIdx.x; c < H; c += NT) {
        float acc = addend ? to_f32(addend[(long)row * H + c]) : 0.f;
        for (int e = 0; e < E; ++e)
            if (clamp_index(index[(long)b * E + e], S) == s) acc += to_f32(dsel[((long)b * E + e) * H + c]);
        dx[(long)row * H + c] = from_f32<T>(acc);
    }
}

// ---- FlickrAttention scores (modeling.py:1624-1648) + compute_score_with_logits_flickr (:1650-1673) -----------------
// One workgroup per (sample, entity): scores[b, e, r] = q[b, e, :] . k[b, T + r, :] / sqrt(d) + (1 - image_mask[b, r]) * -1e4
// stats[0] += label[b, e, argmax_r scores] != 0;  stats[1] += sum_r label[b, e, r];  stats[2] += position[b, e] != -1
template <typename T>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) flickr_scores_kernel(const T* q, long ldq, const T* k, long ldk, const int64_t* image_mask,
                                                   const float* label, const int64_t* position, float* scores,
                                                   float* stats, int E, int R, int S, int Tlen, int d) {
    VB_DYN_SMEM(smem);
    float* qs = (float*)smem;                       // [d]
    float* bestv = qs + d;                          // [NT]
    int* besti = (int*)(bestv + NT);                // [NT]
    const int row = blockIdx.x, b = row / E;
    for (int c = threadIdx.x; c < d; c += NT) qs[c] = to_f32(q[(long)row * ldq + c]);
    __syncthreads();
    const float rs = sqrtf((float)d);
    float bv = -INFINITY; int bi = R;
    for (int r = threadIdx.x; r < R; r += NT) {
        const T* kr = k + ((long)b * S + Tlen + r) * ldk;
        float acc = 0.f;
        for (int c = 0; c < d; ++c) acc += qs[c] * to_f32(kr[c]);
        const float sc = acc / rs + (1.0f - (float)image_mask[(long)b * R + r]) * -10000.0f;
        scores[(long)row * R + r] = sc;
        if (sc > bv) { bv = sc; bi = r; }
    }
    bestv[threadIdx.x] = bv; besti[threadIdx.x] = bi;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int t = 1; t < NT; ++t)
            if (bestv[t] > bv || (bestv[t] == bv && besti[t] < bi)) { bv = bestv[t]; bi = besti[t]; }
        float up = 0.f;
        for (int r = 0; r < R; ++r) up += label[(long)row * R + r];
        if (bi < R && label[(long)row * R + bi] != 0.0f) atomicAdd(stats + 0, 1.0f);
        atomicAdd(stats + 1, up);
        if (position[row] != -1) atomicAdd(stats + 2, 1.0f);
    }
}

// One workgroup per sample.  dq[b, e, :] = alpha/sqrt(d) * sum_r ds[b, e, r] k[b, T + r, :];
// dk[b, T + r, :] = alpha/sqrt(d) * sum_e ds[b, e, r] q[b, e, :];  dk over the text rows = 0.
template <typename T>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) flickr_scores_bwd_kernel(const float* ds, const T* q, long ldq, const T* k, long ldk,
                                                       T* dq, T* dk, const float* scale_dev, float alpha,
                                                       int E, int R, int S, int Tlen, int d) {
    const int b = blockIdx.x;
    const float sc = alpha * (scale_dev ? scale_dev[0] : 1.f) / sqrtf((float)d);
    for (int i = threadIdx.x; i < E * d; i += NT) {
        const int e = i / d, c = i - e * d;
        float acc = 0.f;
        for (int r = 0; r < R; ++r)
            acc += ds[((long)b * E + e) * R + r] * to_f32(k[((long)b * S + Tlen + r) * ldk + c]);
        dq[((long)b * E + e) * ldq + c] = from_f32<T>(acc * sc);
    }
    for (int i = threadIdx.x; i < S * d; i += NT) {
        const int s = i / d, c = i - s * d;
        float acc = 0.f;
        if (s >= Tlen)
            for (int e = 0; e < E; ++e)
                acc += ds[((long)b * E + e) * R + (s - Tlen)] * to_f32(q[((long)b * E + e) * ldq + c]);
        dk[((long)b * S + s) * ldk + c] = from_f32<T>(acc * sc);
    }
}

// ---- output_attention_weights (modeling.py:241-261 returning attention_probs) ----------------------------------------
// The training kernels never materialise the [B, nh, S, S] probabilities; this forward-only kernel writes them for the
// visualisation path.  One workgroup per (sample, head, query row); any S.
template <typename T>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) attn_probs_kernel(const T* qkv, const float* mask_add, float* probs, int S, int nh, int d) {
    VB_DYN_SMEM(smem);
    float* qs = (float*)smem;                       // [d]
    float* red = qs + d;                            // [NT / 64]
    const long row = blockIdx.x;
    const int i = (int)(row % S);
    const int h = (int)((row / S) % nh), b = (int)(row / S / nh);
    const long ld = 3L * nh * d;
    const T* q = qkv + ((long)b * S + i) * ld + (long)h * d;
    const T* kbase = qkv + (long)b * S * ld + (long)nh * d + (long)h * d;
    for (int c = threadIdx.x; c < d; c += NT) qs[c] = to_f32(q[c]);
    __syncthreads();
    float* out = probs + row * S;
    const float rs = sqrtf((float)d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < S; j += NT) {
        const T* k = kbase + (long)j * ld;
        float acc = 0.f;
        for (int c = 0; c < d; ++c) acc += qs[c] * to_f32(k[c]);
        const float sc = acc / rs + mask_add[(long)b * S + j];
        out[j] = sc;
        m = fmaxf(m, sc);
    }
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int j = threadIdx.x; j < S; j += NT) { const float e = expf(out[j] - m); out[j] = e; sum += e; }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    for (int j = threadIdx.x; j < S; j += NT) out[j] = out[j] / sum;
}

}  // namespace

extern "C" int vb_attn_probs(int dtype, const void* qkv, const float* mask_add, float* probs, int B, int S, int nh,
                             int head_dim, void* stream) {
    if (!qkv || !mask_add || !probs || B <= 0 || S <= 0 || nh <= 0 || head_dim <= 0 || head_dim > 1024) return VB_ERR_ARG;
    if ((long)B * nh * S > 0x7fffffffL) return VB_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((long)B * nh * S));
    const size_t smem = (size_t)(head_dim + NT / 64) * sizeof(float);
    if (dtype == VB_BF16) VB_LAUNCH(attn_probs_kernel<bf16>, grid, dim3(NT), smem, s, (const bf16*)qkv, mask_add, probs, S, nh, head_dim);
    else if (dtype == VB_F32) VB_LAUNCH(attn_probs_kernel<float>, grid, dim3(NT), smem, s, (const float*)qkv, mask_add, probs, S, nh, head_dim);
    else return VB_ERR_ARG;
    return vb_check_launch();
}

extern "C" int vb_align_pos_fwd(const int64_t* alignment, const float* pos, float* out, int B, int R, int Ra, int A,
                                int H, int max_pos, void* stream) {
    if (!alignment || !pos || !out || B <= 0 || R <= 0 || Ra < R || A <= 0 || H <= 0 || max_pos <= 0) return VB_ERR_ARG;
    VB_LAUNCH(align_pos_fwd_kernel, dim3((unsigned)(B * R)), dim3(NT), 0, (hipStream_t)stream, alignment, pos, out,
              R, Ra, A, H, max_pos);
    return vb_check_launch();
}

extern "C" int vb_align_pos_bwd(int dtype, const void* dz, const int64_t* alignment, float* d_pos, int B, int T, int R,
                                int Ra, int A, int H, int max_pos, void* stream) {
    if (!dz || !alignment || !d_pos || B <= 0 || T <= 0 || R <= 0 || Ra < R || A <= 0 || H <= 0 || max_pos <= 0)
        return VB_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)(B * R));
    if (dtype == VB_BF16) VB_LAUNCH(align_pos_bwd_kernel<bf16>, grid, dim3(NT), 0, s, (const bf16*)dz, alignment, d_pos, T, R, Ra, A, H, max_pos);
    else if (dtype == VB_F32) VB_LAUNCH(align_pos_bwd_kernel<float>, grid, dim3(NT), 0, s, (const float*)dz, alignment, d_pos, T, R, Ra, A, H, max_pos);
    else return VB_ERR_ARG;
    return vb_check_launch();
}

extern "C" int vb_gather_index_rows(int dtype, const void* x, const int64_t* index, void* out, int B, int S, int E,
                                    int H, void* stream) {
    if (!x || !index || !out || B <= 0 || S <= 0 || E <= 0 || H <= 0) return VB_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)(B * E));
    if (dtype == VB_BF16) VB_LAUNCH(gather_index_kernel<bf16>, grid, dim3(NT), 0, s, (const bf16*)x, index, (bf16*)out, S, E, H);
    else if (dtype == VB_F32) VB_LAUNCH(gather_index_kernel<float>, grid, dim3(NT), 0, s, (const float*)x, index, (float*)out, S, E, H);
    else return VB_ERR_ARG;
    return vb_check_launch();
}

extern "C" int vb_scatter_index_rows(int dtype, const void* dsel, const int64_t* index, const void* addend, void* dx,
                                     int B, int S, int E, int H, void* stream) {
    if (!dsel || !index || !dx || B <= 0 || S <= 0 || E <= 0 || H <= 0) return VB_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)(B * S));
    if (dtype == VB_BF16) VB_LAUNCH(scatter_index_kernel<bf16>, grid, dim3(NT), 0, s, (const bf16*)dsel, index, (const bf16*)addend, (bf16*)dx, S, E, H);
    else if (dtype == VB_F32) VB_LAUNCH(scatter_index_kernel<float>, grid, dim3(NT), 0, s, (const float*)dsel, index, (const float*)addend, (float*)dx, S, E, H);
    else return VB_ERR_ARG;
    return vb_check_launch();
}

extern "C" int vb_flickr_scores_fwd(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk,
                                    const int64_t* image_mask, const float* label, const int64_t* position,
                                    float* scores, float* stats, int B, int E, int R, int S, int T, int d,
                                    void* stream) {
    if (!q || !k || !image_mask || !label || !position || !scores || !stats) return VB_ERR_ARG;
    if (B <= 0 || E <= 0 || R <= 0 || T < 0 || S != T + R || d <= 0 || d > 1024) return VB_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(stats, 0, 3 * sizeof(float), s) != hipSuccess) return VB_ERR_LAUNCH;
    dim3 grid((unsigned)(B * E));
    const size_t smem = (size_t)(d + 2 * NT) * sizeof(float);
    if (dtype == VB_BF16) VB_LAUNCH(flickr_scores_kernel<bf16>, grid, dim3(NT), smem, s, (const bf16*)q, (long)ldq, (const bf16*)k, (long)ldk, image_mask, label, position, scores, stats, E, R, S, T, d);
    else if (dtype == VB_F32) VB_LAUNCH(flickr_scores_kernel<float>, grid, dim3(NT), smem, s, (const float*)q, (long)ldq, (const float*)k, (long)ldk, image_mask, label, position, scores, stats, E, R, S, T, d);
    else return VB_ERR_ARG;
    return vb_check_launch();
}

extern "C" int vb_flickr_scores_bwd(int dtype, const float* dscores, const void* q, int64_t ldq, const void* k,
                                    int64_t ldk, void* dq, void* dk, const float* scale_dev, float alpha,
                                    int B, int E, int R, int S, int T, int d, void* stream) {
    if (!dscores || !q || !k || !dq || !dk) return VB_ERR_ARG;
    if (B <= 0 || E <= 0 || R <= 0 || T < 0 || S != T + R || d <= 0) return VB_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)B);
    if (dtype == VB_BF16) VB_LAUNCH(flickr_scores_bwd_kernel<bf16>, grid, dim3(NT), 0, s, dscores, (const bf16*)q, (long)ldq, (const bf16*)k, (long)ldk, (bf16*)dq, (bf16*)dk, scale_dev, alpha, E, R, S, T, d);
    else if (dtype == VB_F32) VB_LAUNCH(flickr_scores_bwd_kernel<float>, grid, dim3(NT), 0, s, dscores, (const float*)q, (long)ldq, (const float*)k, (long)ldk, (float*)dq, (float*)dk, scale_dev, alpha, E, R, S, T, d);
    else return VB_ERR_ARG;
    return vb_check_launch();
}
