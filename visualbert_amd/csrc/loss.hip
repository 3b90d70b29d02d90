// loss.hip -- losses and the tiny task heads of the VisualBERT training path.
//   * CrossEntropyLoss(ignore_index) forward + backward in one sweep over fp32 logits
//       reference: TrainVisualBERTObjective.forward, pytorch_pretrained_bert/modeling.py:1471-1477
//                  (masked-LM over [B*S, V] with ignore_index=-1, image-text-match over [B, 2]) and
//                  :1563-1565 (NLVR2, CrossEntropyLoss())
//   * KLDivLoss(batchmean) on log_softmax + VQA score            modeling.py:1517-1523, :1697-1711
//   * "small linear": heads whose output width is tiny (seq_relationship 768->2, NLVR2 768->2),
//     where an MFMA tile would be >98 % padding                  modeling.py:451, :1558
//
// Memory plan for the [B*S, V] masked-LM logits (V = 30522, fp32, leading dimension padded to a
// multiple of 64): one workgroup per row.  ~88 % of rows carry label -1: they are never read -- only
// their dlogits row is zero-filled.  A labelled row is read twice (online max/sum, then gradient);
// its 122 KB stay L2-resident between the two sweeps.
#include "vb_rt.h"
#include "../../include/visualbert_hip.h"

namespace {

constexpr int NT = 256;

VB_DEVICE float block_reduce_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < NT / 64; ++w) r += red[w];
    return r;
}
VB_DEVICE float block_reduce_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < NT / 64; ++w) r = fmaxf(r, red[w]);
    return r;
}

// acc[0] = sum of row losses, acc[1] = number of counted rows (as float), acc[2 .. 2 + CE_SLOTS) = partial sums:
// a row adds its loss to slot (row & 63) -- thousands of workgroups adding to ONE address serialise in L2 at ~100 ns
// per atomic (2492 labelled rows: 247 us, the whole kernel); 64 addresses take that off the critical path.
constexpr int CE_SLOTS = 64;
VB_KERNEL VB_LAUNCH_BOUNDS(NT) ce_count_kernel(const int64_t* labels, int M, int V, int ignore_index, float* acc) {
    VB_DYN_SMEM(smem);
    float* red = (float*)smem;
    // acc[] was zeroed by the launcher; every workgroup adds the count of its slice (a single workgroup walking 20992
    // labels with one load in flight took 40 us)
    float c = 0.f;
    for (int i = blockIdx.x * NT + threadIdx.x; i < M; i += gridDim.x * NT) {
        const int64_t l = labels[i];
        c += (l != ignore_index && l >= 0 && l < V) ? 1.f : 0.f;
    }
    c = block_reduce_sum(c, red);
    if (threadIdx.x == 0 && c != 0.f) atomicAdd(&acc[1], c);
}

template <typename T>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) ce_row_kernel(const float* logits, long ld, const int64_t* labels, int ignore_index,
                                            float* acc, T* dlogits, long ldd, int M, int V,
                                            const int64_t* rows, int n_rows) {
    VB_DYN_SMEM(smem);
    float* red = (float*)smem;
    // compact form: workgroup r handles source row rows[r] and writes row r of dlogits; r >= n_rows are pad rows
    const int row = rows ? ((int)blockIdx.x < n_rows ? (int)rows[blockIdx.x] : -1) : (int)blockIdx.x;
    const int64_t label = row >= 0 ? labels[row] : (int64_t)ignore_index;
    const bool counted = (row >= 0 && label != ignore_index && label >= 0 && label < V);
    T* drow = dlogits ? dlogits + (long)blockIdx.x * ldd : nullptr;
    // 8-element vectors whenever the row pitches keep 16-byte alignment (they do for the padded MLM buffers)
    const bool vec = ((ld & 7) == 0) && ((ldd & 7) == 0) && ((((uintptr_t)logits) | ((uintptr_t)dlogits)) & 31) == 0;
    if (!counted) {                                      // workgroup-uniform branch: ~88 % of MLM rows
        if (drow) {
            if (vec) {
                float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (long j = (long)threadIdx.x * 8; j < ldd; j += NT * 8) store8(drow + j, z);
            } else {
                for (long j = threadIdx.x; j < ldd; j += NT) drow[j] = from_f32<T>(0.f);
            }
        }
        return;
    }
    const float* x = logits + (long)row * ld;
    // Rows of up to 32768 columns live in registers (16 x 8 floats per thread): every load of the row is issued before
    // the first use (the rolled online-softmax loop below pays one HBM round trip per trip, twice over), the logits are
    // read once, and the gradient is written from the registers.
    constexpr int CACHE_IT = 16;
    if (vec && V <= CACHE_IT * NT * 8) {
        const float x_label = x[label];                  // issued with the row, not after the reductions
        const float count = acc[1];
        float v[CACHE_IT][8];
#pragma unroll
        for (int k = 0; k < CACHE_IT; ++k) {
            const int j = (k * NT + (int)threadIdx.x) * 8;
            if (j < V) load8(v[k], x + j);               // pad columns (j + e >= V) are readable; masked below
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[k][e] = -INFINITY;
            }
        }
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < CACHE_IT; ++k) {
            const int j = (k * NT + (int)threadIdx.x) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (j + e >= V) v[k][e] = -INFINITY;
                m = fmaxf(m, v[k][e]);
            }
        }
        const float gm = block_reduce_max(m, red);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < CACHE_IT; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += fast_exp(v[k][e] - gm);      // exp(-inf) = 0 for the masked tail
        const float gs = block_reduce_sum(s, red);
        const float lse = gm + logf(gs);
        if (threadIdx.x == 0) atomicAdd(&acc[2 + (blockIdx.x & (CE_SLOTS - 1))], lse - x_label);
        if (drow) {
            const float invc = 1.0f / count;
#pragma unroll
            for (int k = 0; k < CACHE_IT; ++k) {
                const long j = ((long)k * NT + threadIdx.x) * 8;
                if (j < ldd) {
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        o[e] = (j + e < V) ? (fast_exp(v[k][e] - lse) - ((j + e) == label ? 1.f : 0.f)) * invc : 0.f;
                    store8(drow + j, o);
                }
            }
            for (long j = ((long)CACHE_IT * NT + threadIdx.x) * 8; j < ldd; j += NT * 8) {   // pad beyond the cached span
                float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                store8(drow + j, z);
            }
        }
        return;
    }
    float m = -INFINITY, s = 0.f;                        // online max / sum-exp
    if (vec) {
        for (int j = threadIdx.x * 8; j < V; j += NT * 8) {
            float v[8]; load8(v, x + j);                 // pad columns (j+e >= V) are readable; they are masked here
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float ve = (j + e < V) ? v[e] : -INFINITY;
                if (ve > m) { s = s * fast_exp(m - ve) + 1.f; m = ve; }
                else if (ve != -INFINITY) s += fast_exp(ve - m);
            }
        }
    } else {
        for (int j = threadIdx.x; j < V; j += NT) {
            const float v = x[j];
            if (v > m) { s = s * expf(m - v) + 1.f; m = v; }
            else s += expf(v - m);
        }
    }
    const float gm = block_reduce_max(m, red);
    s = (m == -INFINITY) ? 0.f : s * expf(m - gm);
    const float gs = block_reduce_sum(s, red);
    const float lse = gm + logf(gs);
    if (threadIdx.x == 0) atomicAdd(&acc[2 + (blockIdx.x & (CE_SLOTS - 1))], lse - x[label]);
    if (drow) {
        const float invc = 1.0f / acc[1];
        if (vec) {
            for (long j = (long)threadIdx.x * 8; j < ldd; j += NT * 8) {
                float v[8], o[8];
                if (j < V) load8(v, x + j);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    o[e] = (j + e < V) ? (expf(v[e] - lse) - ((j + e) == label ? 1.f : 0.f)) * invc : 0.f;
                store8(drow + j, o);
            }
        } else {
            for (long j = threadIdx.x; j < ldd; j += NT) {
                float gval = 0.f;
                if (j < V) gval = (expf(x[j] - lse) - (j == label ? 1.f : 0.f)) * invc;
                drow[j] = from_f32<T>(gval);
            }
        }
    }
}

// loss[0] = sum(partials) / acc[1]   (NaN when nothing is counted, like the reference); acc[0] = the sum
VB_KERNEL ce_finish_kernel(float* acc, float* loss) {
    float v = threadIdx.x < CE_SLOTS ? acc[2 + threadIdx.x] : 0.f;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if (threadIdx.x == 0) { acc[0] = v; loss[0] = v / acc[1]; }
}

// KLDivLoss(reduction=batchmean)(log_softmax(logits), target) + gradient + VQA score, one block per row
VB_KERNEL VB_LAUNCH_BOUNDS(NT) kldiv_row_kernel(const float* logits, long ld, const float* target, long ldt,
                                               float* loss, float* score, float* dlogits, long ldd, int M, int V) {
    VB_DYN_SMEM(smem);
    float* red = (float*)smem;
    const int row = blockIdx.x;
    const float* x = logits + (long)row * ld;
    const float* tg = target + (long)row * ldt;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < V; j += NT) m = fmaxf(m, x[j]);
    m = block_reduce_max(m, red);
    float s = 0.f, tsum = 0.f;
    for (int j = threadIdx.x; j < V; j += NT) { s += expf(x[j] - m); tsum += tg[j]; }
    s = block_reduce_sum(s, red);
    tsum = block_reduce_sum(tsum, red);
    const float lse = m + logf(s);
    float l = 0.f;
    // argmax over classes 1..V-1 (masked_unk_softmax zeroes class 0), first index on ties
    float bv = -INFINITY; int bi = V;
    for (int j = threadIdx.x; j < V; j += NT) {
        const float tj = tg[j], lsm = x[j] - lse;
        if (tj > 0.f) l += tj * (logf(tj) - lsm);
        if (j >= 1 && (x[j] > bv || (x[j] == bv && j < bi))) { bv = x[j]; bi = j; }
        if (dlogits) dlogits[(long)row * ldd + j] = (expf(lsm) * tsum - tj) / (float)M;
    }
    l = block_reduce_sum(l, red);
    const float gbv = block_reduce_max(bv, red);
    // smallest index among the maxima
    float cand = (bv == gbv) ? (float)bi : 3.0e38f;
    cand = -block_reduce_max(-cand, red);
    if (threadIdx.x == 0) {
        atomicAdd(loss, l / (float)M);
        if (score) {
            const int am = (int)cand;
            atomicAdd(score, (am >= 0 && am < V) ? tg[am] / (float)M : 0.f);
        }
    }
}

// ---- small linear -------------------------------------------------------------------------------
template <typename T>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) small_linear_fwd_kernel(const T* x, long ldx, const float* W, const float* bias,
                                                      float* y, int M, int N, int K) {
    // one wave per (m, n)
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const bool act = wid < M * N;                        // wave-uniform
    const int m = act ? wid / N : 0, n = act ? wid % N : 0;
    float s = 0.f;
    if (act) for (int k = lane; k < K; k += 64) s += to_f32(x[(long)m * ldx + k]) * W[(long)n * K + k];
    s = wave_sum(s);
    if (act && lane == 0) y[(long)m * N + n] = s + (bias ? bias[n] : 0.f);
}

constexpr int SMALL_LINEAR_ROWS = 32;
template <typename T>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) small_linear_bwd_kernel(const float* dy, const T* x, long ldx, const float* W,
                                                      T* dx, long lddx, float* dW, float* db,
                                                      const float* scale_dev, int M, int N, int K) {
    if (blockIdx.y > 0 && (int)blockIdx.x * NT >= N * K) return;       // row slices > 0 only carry the dW / db sums
    const float sc = scale_dev ? scale_dev[0] : 1.f;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (blockIdx.y == 0 && i < M * K) {                  // dx[m,k] = sum_n dy[m,n] W[n,k]
        if (dx) {
            const int m = i / K, k = i % K;
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += dy[(long)m * N + n] * W[(long)n * K + k];
            dx[(long)m * lddx + k] = from_f32<T>(s * sc);
        }
    }
    // the reductions over m run in slices of SMALL_LINEAR_ROWS rows (gridDim.y), one fp32 atomic per slice and element:
    // a single thread walking all M rows was 160 us of pure load latency at M = 512
    const int m0 = blockIdx.y * SMALL_LINEAR_ROWS, m1 = m0 + SMALL_LINEAR_ROWS < M ? m0 + SMALL_LINEAR_ROWS : M;
    if (i < N * K) {                                     // dW[n,k] += sum_m dy[m,n] x[m,k]
        if (dW) {
            const int n = i / K, k = i % K;
            float s = 0.f;
            for (int m = m0; m < m1; ++m) s += dy[(long)m * N + n] * to_f32(x[(long)m * ldx + k]);
            vb_atomic_add_noret(dW + i, s * sc);
        }
    }
    if (i < N && db) {
        float s = 0.f;
        for (int m = m0; m < m1; ++m) s += dy[(long)m * N + i];
        vb_atomic_add_noret(db + i, s * sc);
    }
}

}  // namespace

extern "C" int vb_ce_fwd_bwd(int dtype, const float* logits, int64_t ld_logits, const int64_t* labels,
                             int ignore_index, float* acc2, float* loss, void* dlogits, int64_t ld_dlogits,
                             int M, int V, void* stream) {
    if (!logits || !labels || !acc2 || !loss || M <= 0 || V <= 0) return VB_ERR_ARG;
    if (dlogits && ld_dlogits < V) return VB_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(acc2, 0, sizeof(float) * (2 + CE_SLOTS), s) != hipSuccess) return VB_ERR_LAUNCH;
    VB_LAUNCH(ce_count_kernel, dim3((unsigned)((M + NT - 1) / NT > 128 ? 128 : (M + NT - 1) / NT)), dim3(NT), 64, s, labels, M, V, ignore_index, acc2);
    if (dtype == VB_BF16)
        VB_LAUNCH(ce_row_kernel<bf16>, dim3((unsigned)M), dim3(NT), 64, s, logits, (long)ld_logits, labels,
                  ignore_index, acc2, (bf16*)dlogits, (long)ld_dlogits, M, V, (const int64_t*)nullptr, 0);
    else if (dtype == VB_F32)
        VB_LAUNCH(ce_row_kernel<float>, dim3((unsigned)M), dim3(NT), 64, s, logits, (long)ld_logits, labels,
                  ignore_index, acc2, (float*)dlogits, (long)ld_dlogits, M, V, (const int64_t*)nullptr, 0);
    else return VB_ERR_ARG;
    VB_LAUNCH(ce_finish_kernel, dim3(1), dim3(64), 0, s, acc2, loss);
    return vb_check_launch();
}

extern "C" int vb_ce_fwd_bwd_rows(int dtype, const float* logits, int64_t ld_logits, const int64_t* labels,
                                  int ignore_index, const int64_t* rows, int n_rows, int n_rows_padded, float* acc2,
                                  float* loss, void* dlogits_compact, int64_t ld_dlogits, int M, int V, void* stream) {
    if (!logits || !labels || !acc2 || !loss || !dlogits_compact || M <= 0 || V <= 0) return VB_ERR_ARG;
    if (n_rows < 0 || n_rows > M || n_rows_padded < n_rows || n_rows_padded <= 0 || (n_rows > 0 && !rows)) return VB_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(acc2, 0, sizeof(float) * (2 + CE_SLOTS), s) != hipSuccess) return VB_ERR_LAUNCH;
    VB_LAUNCH(ce_count_kernel, dim3((unsigned)((M + NT - 1) / NT > 128 ? 128 : (M + NT - 1) / NT)), dim3(NT), 64, s, labels, M, V, ignore_index, acc2);
    const int64_t* r = rows ? rows : labels;               // never dereferenced when n_rows == 0
    if (dtype == VB_BF16)
        VB_LAUNCH(ce_row_kernel<bf16>, dim3((unsigned)n_rows_padded), dim3(NT), 64, s, logits, (long)ld_logits, labels,
                  ignore_index, acc2, (bf16*)dlogits_compact, (long)ld_dlogits, M, V, r, n_rows);
    else if (dtype == VB_F32)
        VB_LAUNCH(ce_row_kernel<float>, dim3((unsigned)n_rows_padded), dim3(NT), 64, s, logits, (long)ld_logits, labels,
                  ignore_index, acc2, (float*)dlogits_compact, (long)ld_dlogits, M, V, r, n_rows);
    else return VB_ERR_ARG;
    VB_LAUNCH(ce_finish_kernel, dim3(1), dim3(64), 0, s, acc2, loss);
    return vb_check_launch();
}

extern "C" int vb_kldiv_fwd_bwd(const float* logits, int64_t ld_logits, const float* target, int64_t ld_target,
                                float* loss, float* score, float* dlogits, int64_t ld_dlogits, int M, int V,
                                void* stream) {
    if (!logits || !target || !loss || M <= 0 || V <= 0) return VB_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(loss, 0, sizeof(float), s) != hipSuccess) return VB_ERR_LAUNCH;
    if (score && hipMemsetAsync(score, 0, sizeof(float), s) != hipSuccess) return VB_ERR_LAUNCH;
    VB_LAUNCH(kldiv_row_kernel, dim3((unsigned)M), dim3(NT), 64, s, logits, (long)ld_logits, target, (long)ld_target,
              loss, score, dlogits, (long)ld_dlogits, M, V);
    return vb_check_launch();
}

extern "C" int vb_small_linear_fwd(int dtype, const void* x, int64_t ldx, const float* W, const float* bias, float* y,
                                   int M, int N, int K, void* stream) {
    if (!x || !W || !y || M <= 0 || N <= 0 || K <= 0) return VB_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((M * N + NT / 64 - 1) / (NT / 64)));
    if (dtype == VB_BF16) VB_LAUNCH(small_linear_fwd_kernel<bf16>, grid, dim3(NT), 0, s, (const bf16*)x, (long)ldx, W, bias, y, M, N, K);
    else if (dtype == VB_F32) VB_LAUNCH(small_linear_fwd_kernel<float>, grid, dim3(NT), 0, s, (const float*)x, (long)ldx, W, bias, y, M, N, K);
    else return VB_ERR_ARG;
    return vb_check_launch();
}

extern "C" int vb_small_linear_bwd(int dtype, const float* dy, const void* x, int64_t ldx, const float* W,
                                   void* dx, int64_t lddx, float* dW, float* db, const float* scale_dev,
                                   int M, int N, int K, void* stream) {
    if (!dy || !x || !W || M <= 0 || N <= 0 || K <= 0) return VB_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int work = (M > N ? M : N) * K;
    dim3 grid((unsigned)((work + NT - 1) / NT), (unsigned)((M + SMALL_LINEAR_ROWS - 1) / SMALL_LINEAR_ROWS));
    if (dtype == VB_BF16) VB_LAUNCH(small_linear_bwd_kernel<bf16>, grid, dim3(NT), 0, s, dy, (const bf16*)x, (long)ldx, W, (bf16*)dx, (long)lddx, dW, db, scale_dev, M, N, K);
    else if (dtype == VB_F32) VB_LAUNCH(small_linear_bwd_kernel<float>, grid, dim3(NT), 0, s, dy, (const float*)x, (long)ldx, W, (float*)dx, (long)lddx, dW, db, scale_dev, M, N, K);
    else return VB_ERR_ARG;
    return vb_check_launch();
}
