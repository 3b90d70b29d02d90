// misc.hip -- small data-movement kernels around the hot path: input preparation (integer work,
// bit-exact with the reference), dtype casts, row gather / scatter for the VQA head.
#include "vb_rt.h"
#include "../../include/visualbert_hip.h"

namespace {

constexpr int NT = 256;

// models/model.py:262-268: image_mask[b,r] = (r < image_dim[b]);
// modeling.py:1417: attention_mask = cat(input_mask, image_mask);  :1293-1294: (1 - mask) * -10000;
// modeling.py:1419-1426: LM labels extended with -1 over the visual slots.
VB_KERNEL VB_LAUNCH_BOUNDS(NT) prepare_inputs_kernel(const int64_t* input_mask, const int64_t* image_dim,
                                                    const int64_t* image_mask_in, const int64_t* lm_labels,
                                                    int64_t* attention_mask, float* mask_add, int64_t* labels_ext,
                                                    int B, int T, int R) {
    const int S = T + R;
    const long n = (long)B * S;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const int b = (int)(i / S), s = (int)(i % S);
        int64_t mval, lab;
        if (s < T) {
            mval = input_mask[(long)b * T + s];
            lab = lm_labels ? lm_labels[(long)b * T + s] : -1;
        } else {
            const int r = s - T;
            mval = image_mask_in ? image_mask_in[(long)b * R + r] : ((int64_t)r < image_dim[b] ? 1 : 0);
            lab = -1;
        }
        attention_mask[i] = mval;
        mask_add[i] = (1.0f - (float)mval) * -10000.0f;
        if (labels_ext) labels_ext[i] = lab;
    }
}

template <typename TI, typename TO>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) cast_kernel(const TI* in, TO* out, long n) {
    for (long i = ((long)blockIdx.x * NT + threadIdx.x) * 8; i < n; i += (long)gridDim.x * NT * 8) {
        if (i + 8 <= n && ((((uintptr_t)(in + i)) | ((uintptr_t)(out + i))) & 15) == 0) {
            float v[8]; load8(v, in + i); store8(out + i, v);
        } else {
            for (long j = i; j < n && j < i + 8; ++j) out[j] = from_f32<TO>(to_f32(in[j]));
        }
    }
}

// out[b, :] = x[b, index[b], :]   (VQA head: hidden state at position input_mask.sum(1) - 2, modeling.py:1503-1505)
template <typename T>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) gather_rows_kernel(const T* x, const int64_t* input_mask, T* out, int64_t* index_out,
                                                 int B, int S, int Tlen, int H) {
    const int b = blockIdx.x;
    long cnt = 0;
    for (int s = 0; s < Tlen; ++s) cnt += input_mask[(long)b * Tlen + s];
    long idx = cnt - 2;
    if (idx < 0) idx += S;                   // torch.gather would raise; keep in range
    if (idx >= S) idx = S - 1;
    if (threadIdx.x == 0 && index_out) index_out[b] = cnt - 2;
    for (int j = threadIdx.x; j < H; j += NT) out[(long)b * H + j] = x[((long)b * S + idx) * H + j];
}
template <typename T>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) scatter_rows_kernel(const T* dout, const int64_t* index, T* dx, int B, int S, int H) {
    const int b = blockIdx.x;
    long idx = index[b];
    if (idx < 0) idx += S;
    if (idx >= S) idx = S - 1;
    for (int j = threadIdx.x; j < H; j += NT) dx[((long)b * S + idx) * H + j] = dout[(long)b * H + j];
}


// column sums of a T [M,N] matrix, ACCUMULATED into fp32 out[N] (bias gradients):
// a lane owns 8 consecutive columns, a wave 512; the 4 waves of a block take interleaved rows.
template <typename T>
// fold > 0: the matrix is a split image ([M, 2 fold]: hi | lo planes): column c and column fold + c add to the same out[c]
VB_KERNEL VB_LAUNCH_BOUNDS(NT) colsum_kernel(const T* x, long ld, float* out, const float* scale_dev, int M, int N,
                                            int rows_per_block, int fold) {
    VB_DYN_SMEM(smem);
    constexpr int RED_PITCH = 64 * 9;
    float* red = (float*)smem;                       // [4 waves][64 lanes][9]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 512 + lane * 8;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (col < N) {
        const bool vec = (col + 8 <= N) && ((ld & 7) == 0);
        int r = r0 + wave;
        if (vec) {
            for (; r + 12 < r1; r += 16) {                         // 4 independent 16-byte loads in flight
                float v0[8], v1[8], v2[8], v3[8];
                load8(v0, x + (long)r * ld + col); load8(v1, x + (long)(r + 4) * ld + col);
                load8(v2, x + (long)(r + 8) * ld + col); load8(v3, x + (long)(r + 12) * ld + col);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (v0[j] + v1[j]) + (v2[j] + v3[j]);
            }
        }
        for (; r < r1; r += 4) {
            const T* p = x + (long)r * ld + col;
            if (vec) {
                float v[8]; load8(v, p);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            } else {
                for (int j = 0; j < 8 && col + j < N; ++j) acc[j] += to_f32(p[j]);
            }
        }
    }
    // a lane's 8 values at a pitch of 9 floats: lane stride 9 is coprime with the 32 banks a 4-byte LDS access is spread over (the
    // pitch-8 form was an 8-way conflict on every store: SQ_LDS_BANK_CONFLICT 35 % of this kernel's LDS cycles, round 5's PMC pass)
#pragma unroll
    for (int j = 0; j < 8; ++j) red[wave * RED_PITCH + lane * 9 + j] = acc[j];
    __syncthreads();
    const float sc = scale_dev ? scale_dev[0] : 1.f;
    for (int c = threadIdx.x; c < 512; c += NT) {
        const int cc = blockIdx.x * 512 + c, ci = c + (c >> 3);
        if (cc < N) atomicAdd(&out[fold > 0 && cc >= fold ? cc - fold : cc], (red[ci] + red[RED_PITCH + ci] + red[2 * RED_PITCH + ci] + red[3 * RED_PITCH + ci]) * sc);
    }
}

// dx = dy * act'(.)  for act in {GELU (aux = pre-activation), TANH (aux = tanh output)}
template <typename T>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) act_bwd_kernel(const T* dy, const T* aux, T* dx, long n, int act) {
    for (long i = ((long)blockIdx.x * NT + threadIdx.x) * 8; i < n; i += (long)gridDim.x * NT * 8) {
        if (i + 8 <= n) {
            float g[8], a[8];
            load8(g, dy + i); load8(a, aux + i);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] *= (act == VB_ACT_GELU) ? gelu_grad_f(a[j]) : (1.0f - a[j] * a[j]);
            store8(dx + i, g);
        } else {
            for (long j = i; j < n; ++j) {
                const float a = to_f32(aux[j]);
                dx[j] = from_f32<T>(to_f32(dy[j]) * ((act == VB_ACT_GELU) ? gelu_grad_f(a) : (1.0f - a * a)));
            }
        }
    }
}

// zero a 16-byte aligned region: the gradient arena (448 MB at BERT-base) before every backward.  Grid-stride 16-byte
// stores, 4 per thread per trip.
VB_KERNEL VB_LAUNCH_BOUNDS(NT) zero_kernel(u32x4* p, long n16) {
    const long stride = (long)gridDim.x * NT;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n16; i += stride) p[i] = u32x4{0u, 0u, 0u, 0u};
}

}  // namespace

extern "C" const char* vb_version(void) {
#ifdef VB_EMU
    return "visualbert_hip EMULATOR (developer tool, not the product) r1";
#else
    return "visualbert_hip gfx950 r1";
#endif
}

extern "C" int vb_prepare_inputs(const int64_t* input_mask, const int64_t* image_dim, const int64_t* image_mask,
                                 const int64_t* masked_lm_labels, int64_t* attention_mask, float* mask_add,
                                 int64_t* labels_ext, int B, int T, int R, void* stream) {
    if (!input_mask || !attention_mask || !mask_add || B <= 0 || T <= 0 || R < 0) return VB_ERR_ARG;
    if (R > 0 && !image_dim && !image_mask) return VB_ERR_ARG;
    const long n = (long)B * (T + R);
    dim3 grid((unsigned)((n + NT - 1) / NT > 1024 ? 1024 : (n + NT - 1) / NT));
    VB_LAUNCH(prepare_inputs_kernel, grid, dim3(NT), 0, (hipStream_t)stream, input_mask, image_dim, image_mask,
              masked_lm_labels, attention_mask, mask_add, labels_ext, B, T, R);
    return vb_check_launch();
}

extern "C" int vb_zero(void* dst, int64_t bytes, void* stream) {
    if (!dst || bytes <= 0 || (bytes & 15) || (((uintptr_t)dst) & 15)) return VB_ERR_ARG;
    const long n16 = bytes / 16;
    long blocks = (n16 + NT - 1) / NT;
    if (blocks > 8192) blocks = 8192;
    VB_LAUNCH(zero_kernel, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, (u32x4*)dst, n16);
    return vb_check_launch();
}

extern "C" int vb_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n, void* stream) {
    if (!src || !dst || n <= 0) return VB_ERR_ARG;
    long blocks = (n / 8 + NT - 1) / NT + 1;
    if (blocks > 4096) blocks = 4096;
    dim3 grid((unsigned)blocks);
    hipStream_t s = (hipStream_t)stream;
    if (src_dtype == VB_F32 && dst_dtype == VB_BF16) VB_LAUNCH((cast_kernel<float, bf16>), grid, dim3(NT), 0, s, (const float*)src, (bf16*)dst, (long)n);
    else if (src_dtype == VB_BF16 && dst_dtype == VB_F32) VB_LAUNCH((cast_kernel<bf16, float>), grid, dim3(NT), 0, s, (const bf16*)src, (float*)dst, (long)n);
    else if (src_dtype == VB_F32 && dst_dtype == VB_F32) VB_LAUNCH((cast_kernel<float, float>), grid, dim3(NT), 0, s, (const float*)src, (float*)dst, (long)n);
    else if (src_dtype == VB_BF16 && dst_dtype == VB_BF16) VB_LAUNCH((cast_kernel<bf16, bf16>), grid, dim3(NT), 0, s, (const bf16*)src, (bf16*)dst, (long)n);
    else return VB_ERR_ARG;
    return vb_check_launch();
}

// ---- split operands of the VB_BF16X3 GEMM mode (include/visualbert_hip.h) ------------------------------------------
// x = hi + lo with hi = bf16(x) (round to nearest even) and lo = bf16(x - hi): the residual x - hi is exact in fp32 (it has
// at most 16 significant bits), so |x - hi - lo| <= 2^-9 |x - hi| <= 2^-18 |x|.
// A thread handles 8 consecutive columns of a row: one 32-byte read, two 16-byte writes (hi plane, lo plane).
VB_KERNEL VB_LAUNCH_BOUNDS(NT) split_rows_kernel(const float* src, long ld_src, bf16* dst, long ld_dst, long rows, int cols) {
    const int half = (int)(ld_dst / 2), chunks = half / 8;               // ld_dst % 16 == 0
    const bool vec_src = ((ld_src % 4) == 0) && ((((uintptr_t)src) & 15) == 0);
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < rows * chunks; i += (long)gridDim.x * NT) {
        const long r = i / chunks;
        const int c0 = (int)(i - r * chunks) * 8;
        float v[8];
        if (c0 + 8 <= cols && vec_src) load8(v, src + r * ld_src + c0);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (c0 + j < cols) ? src[r * ld_src + c0 + j] : 0.0f;
        }
        store_split8(dst + r * ld_dst + c0, half, v);
    }
}
// transposed: a 64 x 64 tile of src through LDS; dst[c, r] = hi, dst[c, half + r] = lo
VB_KERNEL VB_LAUNCH_BOUNDS(NT) split_rows_t_kernel(const float* src, long ld_src, bf16* dst, long ld_dst, int rows, int cols, int tiles_c) {
    VB_DYN_SMEM(smem_raw);
    float (*tile)[65] = (float (*)[65])smem_raw;
    const int tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
    const int r0 = tr * 64, c0 = tc * 64, half = (int)(ld_dst / 2);
    for (int i = threadIdx.x; i < 64 * 64; i += NT) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(long)(r0 + r) * ld_src + c0 + c] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += NT) {
        const int c = i >> 6, r = i & 63;                                 // output row = source column
        if (c0 + c >= cols || r0 + r >= half) continue;
        const float v = tile[r][c];
        const bf16 h = (bf16)v;
        dst[(long)(c0 + c) * ld_dst + r0 + r] = h;
        dst[(long)(c0 + c) * ld_dst + half + r0 + r] = (bf16)(v - (float)h);
    }
}

extern "C" int vb_split_bf16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int cols, void* stream) {
    if (!src || !dst || rows <= 0 || cols <= 0 || (ld_dst % 16) || cols > ld_dst / 2 || ld_src < cols || (((uintptr_t)dst) & 15))
        return VB_ERR_ARG;
    const long work = rows * (ld_dst / 16);
    long blocks = (work + NT - 1) / NT;
    if (blocks > 8192) blocks = 8192;
    VB_LAUNCH(split_rows_kernel, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, src, (long)ld_src, (bf16*)dst, (long)ld_dst,
              (long)rows, cols);
    return vb_check_launch();
}
extern "C" int vb_split_bf16_t(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int rows, int cols, void* stream) {
    if (!src || !dst || rows <= 0 || cols <= 0 || (ld_dst % 16) || rows > ld_dst / 2 || ld_src < cols) return VB_ERR_ARG;
    const int tiles_r = (int)((ld_dst / 2 + 63) / 64), tiles_c = (cols + 63) / 64;       // row tiles cover the zero padding too
    VB_LAUNCH(split_rows_t_kernel, dim3((unsigned)(tiles_r * tiles_c)), dim3(NT), 64 * 65 * 4, (hipStream_t)stream, src, (long)ld_src, (bf16*)dst,
              (long)ld_dst, rows, cols, tiles_c);
    return vb_check_launch();
}

// nn.Dropout on a small tensor (the pooled / gathered [B, H] state in front of the fine-tuning heads, modeling.py:1495, 1509,
// 1557): y = keep ? x / (1 - p) : 0 with the counter-based bits every other dropout site uses (vb_dropout_bits8: keyed by
// seed, stream id and the 8-element group index) -- the backward pass calls it again on dy with the same key and gets the
// same mask, so no mask tensor exists.
template <typename T>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) dropout_kernel(const T* x, T* y, long n, float inv_keep, uint32_t thresh, uint64_t seed, uint32_t sid) {
    for (long grp = (long)blockIdx.x * NT + threadIdx.x; grp * 8 < n; grp += (long)gridDim.x * NT) {
        const Rand8 rnd = vb_dropout_bits8(seed, (uint64_t)grp, sid);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const long i = grp * 8 + e;
            if (i < n) y[i] = rand8_keep(rnd, e, thresh) ? from_f32<T>(to_f32(x[i]) * inv_keep) : from_f32<T>(0.0f);
        }
    }
}

extern "C" int vb_dropout(int dtype, const void* x, void* y, int64_t n, float p, uint64_t seed, uint32_t stream_id, void* stream) {
    if (!x || !y || n <= 0 || p < 0.f || p >= 1.f) return VB_ERR_ARG;
    long blocks = ((n + 7) / 8 + NT - 1) / NT;
    if (blocks > 2048) blocks = 2048;
    const float inv_keep = 1.0f / (1.0f - p);
    const uint32_t thresh = vb_drop_thresh16(p);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VB_BF16) VB_LAUNCH(dropout_kernel<bf16>, dim3((unsigned)blocks), dim3(NT), 0, s, (const bf16*)x, (bf16*)y, (long)n, inv_keep, thresh, seed, stream_id);
    else if (dtype == VB_F32) VB_LAUNCH(dropout_kernel<float>, dim3((unsigned)blocks), dim3(NT), 0, s, (const float*)x, (float*)y, (long)n, inv_keep, thresh, seed, stream_id);
    else return VB_ERR_ARG;
    return vb_check_launch();
}

extern "C" int vb_gather_rows(int dtype, const void* x, const int64_t* input_mask, void* out, int64_t* index_out,
                              int B, int S, int T, int H, void* stream) {
    if (!x || !input_mask || !out || B <= 0 || S <= 0 || T <= 0 || H <= 0) return VB_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VB_BF16) VB_LAUNCH(gather_rows_kernel<bf16>, dim3((unsigned)B), dim3(NT), 0, s, (const bf16*)x, input_mask, (bf16*)out, index_out, B, S, T, H);
    else if (dtype == VB_F32) VB_LAUNCH(gather_rows_kernel<float>, dim3((unsigned)B), dim3(NT), 0, s, (const float*)x, input_mask, (float*)out, index_out, B, S, T, H);
    else return VB_ERR_ARG;
    return vb_check_launch();
}

extern "C" int vb_scatter_rows(int dtype, const void* dout, const int64_t* index, void* dx, int B, int S, int H,
                               void* stream) {
    if (!dout || !index || !dx || B <= 0 || S <= 0 || H <= 0) return VB_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VB_BF16) VB_LAUNCH(scatter_rows_kernel<bf16>, dim3((unsigned)B), dim3(NT), 0, s, (const bf16*)dout, index, (bf16*)dx, B, S, H);
    else if (dtype == VB_F32) VB_LAUNCH(scatter_rows_kernel<float>, dim3((unsigned)B), dim3(NT), 0, s, (const float*)dout, index, (float*)dx, B, S, H);
    else return VB_ERR_ARG;
    return vb_check_launch();
}

static int colsum_launch(int dtype, const void* x, int64_t ld, float* out, const float* scale_dev, int M, int N, int fold, void* stream) {
    if (!x || !out || M <= 0 || N <= 0) return VB_ERR_ARG;
    int rb = 64;                                // rows per block (4 waves x 16 rows)
    while ((long)((M + rb - 1) / rb) * ((N + 511) / 512) > 4096) rb *= 2;
    dim3 grid((unsigned)((N + 511) / 512), (unsigned)((M + rb - 1) / rb));
    hipStream_t s = (hipStream_t)stream;
    const size_t smem = 4 * 64 * 9 * sizeof(float);
    if (dtype == VB_BF16) VB_LAUNCH(colsum_kernel<bf16>, grid, dim3(NT), smem, s, (const bf16*)x, (long)ld, out, scale_dev, M, N, rb, fold);
    else if (dtype == VB_F32) VB_LAUNCH(colsum_kernel<float>, grid, dim3(NT), smem, s, (const float*)x, (long)ld, out, scale_dev, M, N, rb, fold);
    else return VB_ERR_ARG;
    return vb_check_launch();
}

extern "C" int vb_colsum(int dtype, const void* x, int64_t ld, float* out, const float* scale_dev, int M, int N,
                         void* stream) {
    return colsum_launch(dtype, x, ld, out, scale_dev, M, N, 0, stream);
}

// internal (attention.hip): out[c] += sum over rows of (hi + lo)[r, c] of a split image [M, 2 C] (ld_image elements per row)
int vb_colsum_image(const void* image, int64_t ld_image, float* out, int M, int C, void* stream) {
    if ((C % 8) || ld_image < 2 * C) return VB_ERR_ARG;
    if (ld_image != 2 * C) {                     // padded planes: one pass per plane
        const int rc = colsum_launch(VB_BF16, image, ld_image, out, nullptr, M, C, 0, stream);
        return rc != VB_OK ? rc : colsum_launch(VB_BF16, (const bf16*)image + ld_image / 2, ld_image, out, nullptr, M, C, 0, stream);
    }
    return colsum_launch(VB_BF16, image, ld_image, out, nullptr, M, 2 * C, C, stream);
}

// ---- deferred second-stage column reductions (vb_rt.h: VbReduceJobs) ---------------------------------------------------------
namespace {
struct ReduceLaunch { VbReduceJob j[8]; int first[9]; int n; };
// 1024 threads = 32 columns x 32 row groups (coalesced 128-byte reads, 32 independent chains per column, LDS tree at the end -- the
// shape of layernorm.hip's ln_bwd_reduce_kernel, whose results this reproduces bit for bit); workgroup -> (job, column block, row slice)
VB_KERNEL VB_LAUNCH_BOUNDS(1024) reduce_jobs_kernel(ReduceLaunch L) {
    VB_DYN_SMEM(smem);
    float* red = (float*)smem;                         // [32][33]
    int b = blockIdx.x, k = 0;
    while (k + 1 < L.n && b >= L.first[k + 1]) ++k;
    b -= L.first[k];
    const float* src = L.j[k].src;
    float* dst = L.j[k].dst;
    const int rows = L.j[k].rows, cols = L.j[k].cols, slices = L.j[k].slices;
    const long stride = L.j[k].stride;
    const int colblocks = (cols + 31) / 32;
    const int cb = b % colblocks, sl = b / colblocks;
    const int cx = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = cb * 32 + cx;
    const int per = (rows + slices - 1) / slices;
    const int r0 = sl * per, r1 = r0 + per < rows ? r0 + per : rows;
    float s = 0.f;
    if (c < cols)
        for (int r = r0 + rg; r < r1; r += 32) s += src[(long)r * stride + c];
    red[rg * 33 + cx] = s;
    __syncthreads();
    if (rg == 0 && c < cols) {
        float t = 0.f;
        for (int r = 0; r < 32; ++r) t += red[r * 33 + cx];
        if (slices == 1) dst[c] += t;
        else atomicAdd(&dst[c], t);
    }
}
}  // namespace

VbReduceJobs*& vb_reduce_defer_slot() {
    static thread_local VbReduceJobs* slot = nullptr;
    return slot;
}

bool vb_reduce_defer(const float* src, float* dst, int rows, int cols, long stride, int slices) {
    VbReduceJobs* J = vb_reduce_defer_slot();
    if (!J || J->n >= 8 || !src || !dst || rows <= 0 || cols <= 0 || slices <= 0) return false;
    J->j[J->n++] = VbReduceJob{src, dst, rows, cols, stride, slices};
    return true;
}

int vb_reduce_jobs_launch(const VbReduceJobs& jobs, void* stream) {
    if (jobs.n <= 0) return VB_OK;
    ReduceLaunch L{};
    int blocks = 0;
    for (int k = 0; k < jobs.n; ++k) {
        L.j[k] = jobs.j[k];
        L.first[k] = blocks;
        blocks += ((jobs.j[k].cols + 31) / 32) * jobs.j[k].slices;
    }
    L.first[jobs.n] = blocks;
    L.n = jobs.n;
    VB_LAUNCH(reduce_jobs_kernel, dim3((unsigned)blocks), dim3(1024), 32 * 33 * sizeof(float), (hipStream_t)stream, L);
    return vb_check_launch();
}

extern "C" int vb_act_bwd(int dtype, const void* dy, const void* aux, void* dx, int64_t n, int act, void* stream) {
    if (!dy || !aux || !dx || n <= 0 || (act != VB_ACT_GELU && act != VB_ACT_TANH)) return VB_ERR_ARG;
    if (((uintptr_t)dy | (uintptr_t)aux | (uintptr_t)dx) & 15) return VB_ERR_ARG;
    long blocks = (n / 8 + NT - 1) / NT + 1;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VB_BF16) VB_LAUNCH(act_bwd_kernel<bf16>, dim3((unsigned)blocks), dim3(NT), 0, s, (const bf16*)dy, (const bf16*)aux, (bf16*)dx, (long)n, act);
    else if (dtype == VB_F32) VB_LAUNCH(act_bwd_kernel<float>, dim3((unsigned)blocks), dim3(NT), 0, s, (const float*)dy, (const float*)aux, (float*)dx, (long)n, act);
    else return VB_ERR_ARG;
    return vb_check_launch();
}


// ---- per-stream launch options (include/visualbert_hip.h) ---------------------------------------------------------------
// A handful of entries at most (one per stream that ever set options): a vector under a mutex; a lookup is a few
// nanoseconds next to a kernel launch.
#include "vb_opts.h"
#include <mutex>
#include <vector>
#include <utility>
namespace {
std::mutex g_opts_mutex;
std::vector<std::pair<void*, vb_stream_opts>>& opts_table() {
    static std::vector<std::pair<void*, vb_stream_opts>> t;
    return t;
}
}  // namespace

vb_stream_opts vb_opts_for(void* stream) {
    std::lock_guard<std::mutex> lock(g_opts_mutex);
    for (auto& e : opts_table()) if (e.first == stream) return e.second;
    return vb_stream_opts{0, 0, 0, 0};
}

extern "C" int vb_stream_set_opts(void* stream, const vb_stream_opts* opts) {
    if (opts) {
        const int k = opts->nt_kernel;
        if (opts->reserved != 0) return VB_ERR_ARG;
        // the product library pins only kernels its own dispatcher can choose; the experiment arms (80, 91, 100, 101) and the
        // vendor yardstick (200) exist in libvisualbert_hip_dev.so only (include/visualbert_hip_dev.h)
        bool ok = (k == 0 || k == 1 || k == 14 || k == 22 || k == 24 || k == 42 || k == 81 || k == 90);
#ifdef VB_DEV_KNOBS
        ok = ok || k == 80 || k == 82 || k == 91 || k == 92 || k == 100 || k == 101 || k == 200;
#endif
        if (opts->persistent_workgroups < 0 || !ok) return VB_ERR_ARG;
    }
    std::lock_guard<std::mutex> lock(g_opts_mutex);
    auto& t = opts_table();
    for (size_t i = 0; i < t.size(); ++i)
        if (t[i].first == stream) {
            if (opts) t[i].second = *opts; else t.erase(t.begin() + i);
            return VB_OK;
        }
    if (opts) t.emplace_back(stream, *opts);
    return VB_OK;
}

extern "C" int vb_stream_get_opts(void* stream, vb_stream_opts* out) {
    if (!out) return VB_ERR_ARG;
    *out = vb_opts_for(stream);
    return VB_OK;
}

// ---- per-stream scratch for in-launch reductions (include/visualbert_hip.h: vb_stream_set_scratch) --------------------------
namespace {
std::vector<std::pair<void*, vb_scratch>>& scratch_table() {
    static std::vector<std::pair<void*, vb_scratch>> t;
    return t;
}
}  // namespace

vb_scratch vb_scratch_for(void* stream) {
    std::lock_guard<std::mutex> lock(g_opts_mutex);
    for (auto& e : scratch_table()) if (e.first == stream) return e.second;
    return vb_scratch{nullptr, 0};
}

extern "C" int vb_stream_set_scratch(void* stream, void* scratch, int64_t bytes) {
    if (scratch && (bytes < VB_SCRATCH_MIN_BYTES || (((uintptr_t)scratch) & 255))) return VB_ERR_ARG;
    if (scratch) {
        // the arrival counters at the head of the buffer start at zero; every kernel that uses one leaves it at zero again
#ifndef VB_EMU
        if (hipMemsetAsync(scratch, 0, VB_SCRATCH_COUNTER_BYTES, (hipStream_t)stream) != hipSuccess) return VB_ERR_LAUNCH;
#else
        memset(scratch, 0, VB_SCRATCH_COUNTER_BYTES);
#endif
    }
    std::lock_guard<std::mutex> lock(g_opts_mutex);
    auto& t = scratch_table();
    for (size_t i = 0; i < t.size(); ++i)
        if (t[i].first == stream) {
            if (scratch) t[i].second = vb_scratch{scratch, bytes}; else t.erase(t.begin() + i);
            return VB_OK;
        }
    if (scratch) t.emplace_back(stream, vb_scratch{scratch, bytes});
    return VB_OK;
}

// ---- developer library and simulator only: one wave through vb_mma_f8 / vb_cvt4_fp8 (the documented lane layout is what the test pins) ----
#if defined(VB_DEV_KNOBS) || defined(VB_EMU)
namespace {
// A [16][128], B [16][128] e4m3 bytes (row-major, K contiguous); sa, sb [16][4] E8M0 bytes per (row, 32-element K block); D [16][16] fp32
VB_KERNEL VB_LAUNCH_BOUNDS(64) mma_f8_probe_kernel(const unsigned char* A, const unsigned char* B, const unsigned char* sa,
                                                  const unsigned char* sb, float* D) {
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    i32x8 av, bv;
    unsigned char* ab = (unsigned char*)&av;
    unsigned char* bb = (unsigned char*)&bv;
    for (int j = 0; j < 32; ++j) {
        const int k = 64 * (j >> 4) + 16 * q + (j & 15);
        ab[j] = A[r * 128 + k]; bb[j] = B[r * 128 + k];
    }
    const int s_a = sa[r * 4 + q], s_b = sb[r * 4 + q];      // lane 16 q + r: block q of row r
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    acc = vb_mma_f8(av, bv, acc, s_a, s_b);
    for (int i = 0; i < 4; ++i) D[(4 * q + i) * 16 + r] = acc[i];
}
VB_KERNEL VB_LAUNCH_BOUNDS(64) cvt_fp8_probe_kernel(const float* x, uint32_t* y, int n4) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n4) y[i] = vb_cvt4_fp8(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]);
}
}  // namespace
namespace {
// vb_split_f8: one wave per row.  image row = [ hi: cols bf16 (zero padded to K) | hi8: K e4m3 | lo8: K e4m3 ], K = ld_img / 2;
// hi = bf16(x), lo = x - hi; each fp8 plane holds its values times 2^e with e = floor(log2(448 / max|row|)) (one per row and plane),
// scale byte = 127 - e (E8M0: the value is byte-encoded x 2^(scale - 127)), stored at (r & ~63) | ((r & 15) << 2) | ((r >> 4) & 3) of
// scale_hi / scale_lo (round_up(rows, 64) bytes each)
VB_DEVICE int f8_row_exp(float amax) {
    if (!(amax > 0.f)) return 0;
    int ex;
    const float m = frexpf(amax, &ex);                      // amax = m 2^ex, m in [0.5, 1)
    int e = (m <= 0.875f ? 9 : 8) - ex;                     // 448 / amax = (448 / m) 2^-ex, 448 / m in (448, 896]
    return e > 126 ? 126 : (e < -127 ? -127 : e);
}
VB_KERNEL VB_LAUNCH_BOUNDS(256) split_f8_kernel(const float* x, long ldx, unsigned char* img, long ld_img, int rows, int cols,
                                               unsigned char* s_hi, unsigned char* s_lo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = (int)(ld_img / 2);
    for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
        const float* xr = x + (long)r * ldx;
        float mh = 0.f, ml = 0.f;
        for (int c = lane; c < cols; c += 64) {
            const float v = xr[c];
            const float h = (float)(bf16)v;
            mh = fmaxf(mh, fabsf(h)); ml = fmaxf(ml, fabsf(v - h));
        }
        for (int o = 32; o > 0; o >>= 1) { mh = fmaxf(mh, __shfl_xor(mh, o)); ml = fmaxf(ml, __shfl_xor(ml, o)); }
        const int eh = f8_row_exp(mh), el = f8_row_exp(ml);
        const int pr = (r & ~63) | ((r & 15) << 2) | ((r >> 4) & 3);      // the GEMM reads four fragments' scales as one dword
        if (lane == 0) { s_hi[pr] = (unsigned char)(127 - eh); s_lo[pr] = (unsigned char)(127 - el); }
        unsigned char* row = img + (long)r * ld_img * 2;
        for (int c = lane * 4; c < K; c += 256) {
            float v[4], h[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = c + j < cols ? xr[c + j] : 0.f;
                h[j] = (float)(bf16)v[j]; l[j] = v[j] - h[j];
                ((bf16*)row)[c + j] = (bf16)v[j];
            }
            *(uint32_t*)(row + 2 * K + c) = vb_cvt4_fp8(ldexpf(h[0], eh), ldexpf(h[1], eh), ldexpf(h[2], eh), ldexpf(h[3], eh));
            *(uint32_t*)(row + 3 * K + c) = vb_cvt4_fp8(ldexpf(l[0], el), ldexpf(l[1], el), ldexpf(l[2], el), ldexpf(l[3], el));
        }
    }
}
}  // namespace
extern "C" int vb_split_f8(const float* x, int64_t ldx, void* image, int64_t ld_img, int rows, int cols, void* scale_hi, void* scale_lo,
                           void* stream) {
    if (!x || !image || !scale_hi || !scale_lo || rows <= 0 || cols <= 0 || (ld_img % 8) || cols > ld_img / 2 || ldx < cols) return VB_ERR_ARG;
    long blocks = (rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    VB_LAUNCH(split_f8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (unsigned char*)image, (long)ld_img,
              rows, cols, (unsigned char*)scale_hi, (unsigned char*)scale_lo);
    return vb_check_launch();
}
extern "C" int vb_mma_f8_probe(const void* A, const void* B, const void* scale_a, const void* scale_b, float* D, void* stream) {
    if (!A || !B || !scale_a || !scale_b || !D) return VB_ERR_ARG;
    VB_LAUNCH(mma_f8_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned char*)A, (const unsigned char*)B,
              (const unsigned char*)scale_a, (const unsigned char*)scale_b, D);
    return vb_check_launch();
}
extern "C" int vb_cvt_fp8_probe(const float* x, void* y, int n, void* stream) {
    if (!x || !y || n <= 0 || (n & 3)) return VB_ERR_ARG;
    VB_LAUNCH(cvt_fp8_probe_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(64), 0, (hipStream_t)stream, x, (uint32_t*)y, n / 4);
    return vb_check_launch();
}
#endif
