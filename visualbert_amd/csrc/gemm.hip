// gemm.hip -- MFMA GEMM for every Linear on the VisualBERT training path (forward, dgrad, wgrad).
//
//   C[M,N] = epilogue( sum_k Aop[m,k] * Bop[n,k] )
//
// Replaces the cuBLAS calls behind nn.Linear / F.linear in the reference:
//   forward  y = x W^T          pytorch_pretrained_bert/modeling.py:232-234 (Q,K,V, packed), :271 (attn out),
//                               :303 (FFN in), :316 (FFN out), :1220 (region projection), :383-385 (pooler),
//                               :398 (MLM transform), :419 (tied decoder), :451 (seq_relationship)
//   dgrad    dx = dy W           autograd of the same lines
//   wgrad    dW += dy^T x        autograd of the same lines (fp32 accumulate straight into the grad arena)
//
// Operand layouts (per operand, independent):
//   VB_KCONTIG : stored [rows][K], K contiguous   (x, dy as A; W as B in forward; W^T shadows as B in dgrad)
//   VB_KSTRIDED: stored [K][rows], rows contiguous (dy and x in wgrad; W as B in dgrad when no W^T shadow exists)
//
// Kernels in this file (details next to each):
//   gemm_nt_8ph_kernel   persistent 256x256-tile kernel for K-contiguous x K-contiguous bf16 GEMMs (every forward and
//                        dgrad GEMM of the training step): LDS-direct copy stream of half-tiles that runs across output
//                        tiles, four-slot schedule with the two wave groups one barrier apart, counted vmcnt waits,
//                        wave-private epilogue slabs, epilogue specialised by template (ACT, OPT).
//   gemm_tn_8ph_kernel   grouped weight gradients: both operands token-major, copied as stored, MFMA fragments gathered with ds_read_b64_tr_b16; (problem, tile, token slice) work items, fp32 atomics (+ gemm_tn_tail_kernel: the ragged tokens % 64 rows of all problems in one launch)
//   gemm_nt_dual_kernel  256x128 tiles, two workgroups per CU (one's epilogue under the other's K loop): the GEMMs with heavy epilogues (GELU + GELU', x GELU' + column sums, fp32 logits)
//   gemm_nt_pipe_kernel  two-barrier 64x128 / 128x128 / 256x128 kernels, two or four LDS stages: small or odd problems (fewer than ~160 big tiles), fp32.
//   gemm_kernel          generic 128x128 kernel for any layout / dtype / ragged K, register-staged transposes, split-K:
//                        the strict-fp32 parity mode and the fallbacks.
// Common: 128-byte LDS rows (64 bf16 / 32 fp32 of K) with a 16-byte-chunk XOR swizzle so fragment reads (ds_read_b128) are
// conflict-free; K-contiguous operands with whole K tiles go global -> LDS directly (global_load_lds_dwordx4), the swizzle
// applied to the per-lane SOURCE address because the LDS image of such a copy is lane-linear; accumulators leave through LDS
// so every global access of the epilogue is a full 128-byte row segment.
#include "vb_rt.h"
#include "../../include/visualbert_hip.h"
#include "vb_opts.h"
#include "vb_prof.h"
#ifdef VB_DEV_KNOBS
#include "../../include/visualbert_hip_dev.h"
#endif
#include <mutex>
#include <vector>
#include <type_traits>
#include <utility>

namespace {

constexpr int BM = 128, BN = 128, NT = 256;
constexpr int EPI_PITCH = 64 * 4 + 16;          // bytes per staged fp32 row of a wave's 32x64 slab
constexpr int EPI_BYTES_PER_WAVE = 32 * EPI_PITCH;
constexpr int TILE_BYTES = 128 * 128;            // one operand tile in LDS
constexpr int SMEM_BYTES = 4 * TILE_BYTES;       // [A0 | B0 | A1 | B1]  (>= the epilogue's 4 slabs)

template <typename T> struct TT {
    static constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
    static constexpr int BK = 8 * EPC;                // K extent of one LDS tile (128 B per row)
    static constexpr int KSTEPS = BK / 32;            // MFMA K steps (of 32) per tile
};

// 16-byte-chunk XOR swizzle of a [rows][128 B] LDS tile.  ds_read_b128 is served in four 16-lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS): a fragment read puts rows
// {0-3,12-15} with chunk c and rows {4-11} with chunk c^1 in one group, and a 256-byte bank row holds
// two tile rows, so the 16-byte slot is (row&1)*8 + chunk'.  chunk' = chunk ^ ((row>>1) & 7) makes the 16
// slots of every group distinct (conflict-free); the extra (row>>4) term keeps the 8-lane groups of the
// register-staged K-strided stores (rows 8 apart) on distinct slots as well.
VB_DEVICE int swz(int row) { return ((row >> 1) ^ (row >> 4)) & 7; }
VB_DEVICE int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ swz(row)) << 4); }

struct GemmArgs {
    const void* A; const void* B; void* C;
    long lda, ldb, ldc;
    int M, N, K;
    const float* bias;
    const void* addend; long ld_addend;
    const void* aux_in; void* aux_out; long ld_aux;
    float alpha;
    const float* alpha_dev;
    int act, accumulate;
    int tiles_m, tiles_n;
    int splits, kt_per_split;     // split-K: slice s covers K tiles [s*kt_per_split, (s+1)*kt_per_split)
    int fast_a, fast_b;           // operand may be copied with global_load_lds (K-contiguous, no K tail)
    unsigned long long* trace;    // measurement builds: per-iteration timestamps of waves 0 and 4 of workgroup 0
    float* colsum;                // optional fp32 [N]: += column sums of the stored values (bias gradient)
    int debug;                    // ablation bits (measurement only): 1 skip tile loads, 2 skip fragment reads, 4 skip MFMAs
    int stripe;                   // dual kernel: column tiles per stripe of the tile walk (tiles_n = one stripe = row-major)
    // VB_BF16X3 (split-operand mode): each operand row holds a hi plane [0, ld/2) and a lo plane [ld/2, ld) of bf16 with
    // x = hi + lo to ~2^-17; the kernel walks 3 K segments of K / 64 tiles each -- hi.hi, lo.hi, hi.lo -- into the same
    // fp32 accumulators.  a_lo / b_lo: ELEMENT offset of the lo plane inside a row; kseg: K tiles per segment.
    int x3, a_lo, b_lo, kseg;
    // split_out (VB_BF16X3 output): C is a bf16 [M, ldc] SPLIT operand -- the fp32 result leaves as hi = bf16(v) in column n and
    // lo = bf16(v - hi) in column ldc / 2 + n (what vb_split_bf16 would make of it), ready to be the next GEMM's operand
    int split_out;
    // streaming (nt) stores for the results (vb_rt.h: store8_nt).  Decided per call by vb_gemm: short reductions (K <= 1024) -- many
    // output tiles per second, whose dirty lines would push the operand panels out of the XCD's L2 -- gain 5-13 % alone at M = 167,936; long
    // ones (K >= 2048) lose up to 12 % with them (profiles/r04_gemm_nt_stores.txt), and so does nothing in the split-operand mode
    int nt_store;
    // in-launch split-K of the small-problem kernels (gemm_nt_pipe_kernel<..., SK = true>): workgroup (tile, slice) reduces K tiles
    // [slice sk_kps, (slice + 1) sk_kps), parks its fp32 partial tile in sk_slabs and draws a ticket from sk_cnt[tile]; the last
    // arrival sums the slices in slice order and runs the epilogue (vb_stream_set_scratch owns the memory)
    int sk_splits, sk_kps;
    float* sk_slabs;
    int* sk_cnt;
    // EPI_DROP (vb_gemm_dropres): inverted dropout of (alpha acc + bias) BEFORE the addend -- BertSelfOutput / BertOutput's
    // dropout(dense(x)) + residual (modeling.py:271-273, 316-318) in the producing GEMM's epilogue; the generator, key and group
    // indexing are the LayerNorm kernels' (layernorm.hip: apply_dropout8 on element index m N + n), whose backward regenerates the mask
    uint32_t drop_thresh, drop_stream;
    float drop_scale;
    uint64_t drop_seed;
};
// K tile `v` of the (virtual) K loop -> element offset of its first column inside a row of A / of B
VB_DEVICE int x3_col_a(const GemmArgs& g, int v, int bk) {
    const int seg = v >= 2 * g.kseg ? 2 : (v >= g.kseg ? 1 : 0);
    return (v - seg * g.kseg) * bk + (seg == 1 ? g.a_lo : 0);
}
VB_DEVICE int x3_col_b(const GemmArgs& g, int v, int bk) {
    const int seg = v >= 2 * g.kseg ? 2 : (v >= g.kseg ? 1 : 0);
    return (v - seg * g.kseg) * bk + (seg == 2 ? g.b_lo : 0);
}

// ---- global -> register staging -------------------------------------------------------------
template <typename T>
VB_DEVICE void zero_tail(u32x4& v, int kvalid) {      // keep the first kvalid (< EPC) elements
    T* e = (T*)&v;
#pragma unroll
    for (int j = 0; j < TT<T>::EPC; ++j) if (j >= kvalid) e[j] = from_f32<T>(0.0f);
}

// K-contiguous operand: 128 rows x 8 chunks; thread t -> chunk t&7, rows (t>>3) + 32 i
template <typename T>
VB_DEVICE void gload_kcontig(u32x4 (&r)[4], const T* P, long ld, int R, int K, int r0, int k0, int t) {
    const int c = t & 7, rr = t >> 3;
    const int k = k0 + c * TT<T>::EPC;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = r0 + rr + 32 * i;
        row = row < R ? row : R - 1;
        if (k < K) {
            r[i] = *(const u32x4*)(P + (long)row * ld + k);
            if (k + TT<T>::EPC > K) zero_tail<T>(r[i], K - k);
        } else {
            r[i] = u32x4{0u, 0u, 0u, 0u};
        }
    }
}
template <typename T>
VB_DEVICE void sstore_kcontig(const u32x4 (&r)[4], unsigned char* lds, int t) {
    const int c = t & 7, rr = t >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) *(u32x4*)(lds + lds_off(rr + 32 * i, c)) = r[i];
}

// K-contiguous operand, direct global -> LDS: wave w issues 4 instructions; instruction i covers tile
// rows (4w+i)*8 .. +7 (1 KiB of LDS, lane-linear): lane -> row = +lane/8, LDS chunk slot c' = lane%8,
// which must hold global chunk c = c' ^ swz(row)  (the same involution the fragment reads apply)
template <typename T>
VB_DEVICE void glds_kcontig(unsigned char* lds, const T* P, long ld, int R, int r0, int k0, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rbase = (wave * 4 + i) * 8;
        const int row = rbase + (lane >> 3);
        const int c = (lane & 7) ^ swz(row);
        int grow = r0 + row;
        grow = grow < R ? grow : R - 1;
        vb_glds16(P + (long)grow * ld + k0 + c * TT<T>::EPC, lds + rbase * 128);
    }
}

// K-strided operand: a thread owns EPC k-rows x 8 tile rows; threads [tbase, tbase+128)
//   rc = u & 15 -> tile rows rc*8 .. rc*8+7 ; kc = u >> 4 -> k = k0 + kc*EPC .. +EPC-1
template <typename T>
VB_DEVICE void gload_kstrided(u32x4 (&r)[8], const T* P, long ld, int R, int K, int r0, int k0, int u) {
    constexpr int EPC = TT<T>::EPC;
    constexpr int VPR = 8 / EPC;                     // 16-byte vectors per 8 rows (1 bf16, 2 fp32)
    const int rc = u & 15, kc = u >> 4;
    const int row = r0 + rc * 8;
#pragma unroll
    for (int kk = 0; kk < EPC; ++kk) {
        const int k = k0 + kc * EPC + kk;
        const bool ok = (k < K) && (row < R);
#pragma unroll
        for (int v = 0; v < VPR; ++v) {
            if (ok) r[kk * VPR + v] = *(const u32x4*)(P + (long)k * ld + row + v * EPC);
            else r[kk * VPR + v] = u32x4{0u, 0u, 0u, 0u};
        }
    }
}
VB_DEVICE void sstore_kstrided_bf16(const u32x4 (&r)[8], unsigned char* lds, int u) {
    const int rc = u & 15, kc = u >> 4;
    // r[kk][jp]: k = kk, rows (2jp, 2jp+1) packed lo/hi.  out row j chunk word kp = k (2kp, 2kp+1)
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
        u32x4 lo, hi;
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) {
            uint32_t w0 = r[2 * kp][jp], w1 = r[2 * kp + 1][jp];
            lo[kp] = (w0 & 0xFFFFu) | (w1 << 16);
            hi[kp] = (w0 >> 16) | (w1 & 0xFFFF0000u);
        }
        *(u32x4*)(lds + lds_off(rc * 8 + 2 * jp, kc)) = lo;
        *(u32x4*)(lds + lds_off(rc * 8 + 2 * jp + 1, kc)) = hi;
    }
}
VB_DEVICE void sstore_kstrided_f32(const u32x4 (&r)[8], unsigned char* lds, int u) {
    const int rc = u & 15, kc = u >> 4;
    // r[kk*2 + v][e]: k = kk (0..3), row = v*4 + e
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        u32x4 o;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) o[kk] = r[kk * 2 + (j >> 2)][j & 3];
        *(u32x4*)(lds + lds_off(rc * 8 + j, kc)) = o;
    }
}
VB_DEVICE void sstore_kstrided(const u32x4 (&r)[8], unsigned char* lds, int u, bf16) { sstore_kstrided_bf16(r, lds, u); }
VB_DEVICE void sstore_kstrided(const u32x4 (&r)[8], unsigned char* lds, int u, float) { sstore_kstrided_f32(r, lds, u); }

// ---- LDS -> MFMA fragment ----------------------------------------------------------------------
VB_DEVICE bf16x8 load_frag(const unsigned char* lds, int row, int ks, int g, bf16) {
    return *(const bf16x8*)(lds + lds_off(row, ks * 4 + g));
}
VB_DEVICE f32x8 load_frag(const unsigned char* lds, int row, int ks, int g, float) {
    f32x4 lo = *(const f32x4*)(lds + lds_off(row, 2 * g));
    f32x4 hi = *(const f32x4*)(lds + lds_off(row, 2 * g + 1));
    (void)ks;
    return f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// launch options of the call being served: copied from the stream's entry (vb_stream_set_opts) at every extern "C" entry of
// this file; thread-local, so concurrent callers on different streams never see each other's settings
static thread_local vb_stream_opts t_opts = {0, 0, 0, 0};
// developer builds only (include/visualbert_hip_dev.h): ablation bits and the timeline buffer; constants in the product
#ifdef VB_DEV_KNOBS
static int g_debug = 0;
static unsigned long long* g_trace = nullptr;
#else
static constexpr int g_debug = 0;
static constexpr unsigned long long* g_trace = nullptr;
#endif

// XCD-aware, bijective remap of the linear workgroup id: hardware places workgroup b on XCD b % 8
// (observed, speed only); give each XCD a contiguous run of logical tiles so that tiles sharing an
// A row-panel hit the same L2 (guide T1, bijective form for nwg % 8 != 0).
VB_DEVICE int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// ---------------- epilogue ------------------------------------------------------------------------------
// The accumulators leave through LDS so that a lane owns 8 consecutive columns of a row and every global
// access is a full 128-byte row segment.  CODE SIZE IS A FIRST-ORDER COST HERE: the epilogue runs once per
// output tile, straight-line, and an unrolled body that carries every activation and every ragged-edge
// variant (~120 KB of instructions) cost 7-20 us per tile in instruction fetch alone on MI355X
// (profiles/r01_gemm_epilogue_codesize.txt: same kernel, bias-only body: 68 -> 33 us at K=64).  Hence:
//   * ACT is a template parameter (-1 = decided at run time) so a kernel carries one activation;
//   * the ragged / unaligned path is a ROLLED per-element loop that reads the staged value back from LDS.
// A register-direct epilogue with swapped MFMA operand roles (4 consecutive columns per lane, no LDS) was
// measured 8-15 % slower (profiles/r01_gemm_register_epilogue_experiment.txt).
// OPT (template) says which optional code an epilogue instantiation carries; a launcher picks the smallest
// instantiation that covers the call (epi_needs)
enum { EPI_ADD = 1,          // addend and/or accumulate operands
       EPI_RAGGED = 2,       // N % 8 != 0 or a pointer / leading dimension that is not 16-byte aligned
       EPI_COLSUM = 4,       // fused column sums
       EPI_ALL = 7,          // everything above, decided at run time (incl. a split result when g.split_out is set)
       EPI_SPLIT = 8,        // specialised split-operand epilogues: the result ALWAYS leaves as a bf16 hi | lo image (g.split_out)
       EPI_DROP = 16 };      // inverted dropout of (alpha acc + bias) in front of the addend (specialised bf16 instantiations only; ldc == N)
struct EpiLane {             // per-lane constants of an epilogue call
    float bb[8], cs[8];
    float alpha;
    int ncol, nv;
    bool vec;                // this lane's 8 columns are all valid and every row pointer is 16-byte aligned
};
template <typename T, typename TO, int OPT>
VB_DEVICE void epi_lane_init(EpiLane& e, const GemmArgs& g, int nw0, int lane) {
    e.alpha = g.alpha_dev ? g.alpha * g.alpha_dev[0] : g.alpha;
    e.ncol = nw0 + (lane & 7) * 8;
    const bool full = (e.ncol + 8 <= g.N);
    e.nv = full ? 8 : (e.ncol < g.N ? g.N - e.ncol : 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) { e.bb[j] = 0.f; e.cs[j] = 0.f; }
    if constexpr (OPT & EPI_RAGGED) {
        e.vec = full && (((g.ldc * (g.split_out ? 2 : sizeof(TO))) | (uintptr_t)g.C) & 15) == 0 &&
                (!g.addend || (((g.ld_addend * sizeof(T)) | (uintptr_t)g.addend) & 15) == 0) &&
                (!(g.aux_in || g.aux_out) || (((g.ld_aux * sizeof(T)) | (uintptr_t)g.aux_in | (uintptr_t)g.aux_out) & 15) == 0);
        if (g.bias && full) {
            if ((((uintptr_t)(g.bias + e.ncol)) & 15) == 0) load8(e.bb, g.bias + e.ncol);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) e.bb[j] = g.bias[e.ncol + j];
            }
        }
    } else {
        e.vec = true;                                   // the launcher checked N % 8 == 0 and every alignment
        if (g.bias && e.ncol < g.N) load8(e.bb, g.bias + e.ncol);
    }
}
// what a call needs from its epilogue (host side)
static int epi_needs(const GemmArgs& g, size_t t_size, size_t to_size) {
    int n = 0;
    if (g.addend || g.accumulate) n |= EPI_ADD;
    if (g.accumulate) n |= EPI_RAGGED;                 // "+= C" keeps the run-time epilogue: the specialised EPI_ADD instantiation means "addend, no accumulate"
    if (g.colsum) n |= EPI_COLSUM;
    if (g.drop_thresh) n |= EPI_DROP;
    bool aligned = (g.N % 8) == 0 && (((g.ldc * to_size) | (uintptr_t)g.C) & 15) == 0 &&
                   (!g.bias || ((uintptr_t)g.bias & 15) == 0) &&
                   (!g.addend || (((g.ld_addend * t_size) | (uintptr_t)g.addend) & 15) == 0) &&
                   (!(g.aux_in || g.aux_out) || (((g.ld_aux * t_size) | (uintptr_t)g.aux_in | (uintptr_t)g.aux_out) & 15) == 0);
    if (!aligned) n |= EPI_RAGGED;
    return n;
}
// one element, every option decided at run time (ragged edges and unaligned leading dimensions only)
template <typename T, typename TO>
VB_DEVICE void epi_scalar(float x, const GemmArgs& g, float alpha, long m, int n) {
    x = x * alpha + (g.bias ? g.bias[n] : 0.f);
    if (g.act == VB_ACT_GELU) {
        if (g.aux_out) ((T*)g.aux_out)[m * g.ld_aux + n] = from_f32<T>(x);
        x = gelu_f(x);
    } else if (g.act == VB_ACT_TANH) {
        x = tanhf(x);
    } else if (g.act == VB_ACT_GELU_GRAD) {
        x *= gelu_grad_f(to_f32(((const T*)g.aux_in)[m * g.ld_aux + n]));
    } else if (g.act == VB_ACT_GELU_SAVE_GRAD) {
        float y, dy;
        gelu_and_grad_f(x, y, dy);
        ((T*)g.aux_out)[m * g.ld_aux + n] = from_f32<T>(dy);
        x = y;
    } else if (g.act == VB_ACT_MUL_AUX) {
        x *= to_f32(((const T*)g.aux_in)[m * g.ld_aux + n]);
    }
    if (g.addend) x += to_f32(((const T*)g.addend)[m * g.ld_addend + n]);
    TO* cp = (TO*)g.C + m * g.ldc + n;
    if (g.accumulate) x += to_f32(*cp);
    if (!(g.debug & 128)) *cp = from_f32<TO>(x);
    if (g.colsum && g.splits == 1) vb_atomic_add_noret(g.colsum + n, x);
}
// per-row operands of the vector path, loaded ahead of the stores of the same pass: interleaved with the
// stores each load would pay a full memory round trip (the compiler must assume C may alias them)
template <typename T, typename TO, int ACT, int OPT>
VB_DEVICE void epi_load8(float (&xa)[8], float (&xd)[8], float (&xc)[8], const GemmArgs& g, long m, int ncol) {
    const int act = ACT >= 0 ? ACT : g.act;
    if (act == VB_ACT_GELU_GRAD || act == VB_ACT_MUL_AUX) load8(xa, (const T*)g.aux_in + m * g.ld_aux + ncol);
    if constexpr (OPT & EPI_ADD) {
        if (g.addend) load8(xd, (const T*)g.addend + m * g.ld_addend + ncol);
        if (g.accumulate) load8(xc, (const TO*)g.C + m * g.ldc + ncol);
    }
}
// offc / offa: element offsets of (m, e.ncol) in C and in the aux matrix.  The FFN-in (GELU) epilogue of the persistent
// kernel steps them from row to row with 64-bit adds instead of forming m * ld per row (two matrices written per row:
// ~100 quarter-rate integer multiplies per lane per tile); measured on one box, A/B/A/B at M = 83,968
// (profiles/r01_gemm_8phase_notes.txt): FFN-in 623 -> 595 us with it, while the plain epilogues got SLOWER with the same
// change (QKV 324 -> 334 us, FFN-out 340 -> 364 us) -- so it is compiled in for that activation only.
template <typename T, typename TO, int ACT, int OPT>
VB_DEVICE void epi_vec8(float (&v)[8], const GemmArgs& g, EpiLane& e, long offc, long offa,
                        const float (&xa)[8], const float (&xd)[8], const float (&xc)[8]) {
    const int act = ACT >= 0 ? ACT : g.act;
    if constexpr (ACT == VB_ACT_GELU_SAVE_GRAD) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {                  // packed fp32 FMA: two columns per issue slot
            const f32x2 r = vb_fma2(f32x2{v[j], v[j + 1]}, vb_splat2(e.alpha), f32x2{e.bb[j], e.bb[j + 1]});
            v[j] = r[0]; v[j + 1] = r[1];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * e.alpha + e.bb[j];
    }
    if constexpr ((OPT & EPI_DROP) != 0 && OPT != EPI_ALL) {
        // the 8 columns of this lane are one generator group: element index m N + n = offc (the launcher checked ldc == N)
        const Rand8 r = vb_dropout_bits8(g.drop_seed, (uint64_t)offc >> 3, g.drop_stream);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = rand8_keep(r, j, g.drop_thresh) ? v[j] * g.drop_scale : 0.0f;
    }
    if (act == VB_ACT_GELU) {
        if (g.aux_out) { if (g.nt_store) store8_nt((T*)g.aux_out + offa, v); else store8((T*)g.aux_out + offa, v); }   // pre-activation, kept for backward
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
    } else if (act == VB_ACT_TANH) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tanhf(v[j]);
    } else if (act == VB_ACT_GELU_GRAD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_f(xa[j]);
    } else if (act == VB_ACT_GELU_SAVE_GRAD) {
        float d[8];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            f32x2 y2, d2;
            gelu_and_grad2(f32x2{v[j], v[j + 1]}, y2, d2);
            v[j] = y2[0]; v[j + 1] = y2[1]; d[j] = d2[0]; d[j + 1] = d2[1];
        }
        if (g.nt_store) store8_nt((T*)g.aux_out + offa, d);               // gelu'(pre), what backward multiplies by
        else store8((T*)g.aux_out + offa, d);
    } else if (act == VB_ACT_MUL_AUX) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= xa[j];
    }
    if constexpr (OPT & EPI_ADD) {
        if (g.addend) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += xd[j];
        }
        if (g.accumulate) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += xc[j];
        }
    }
    TO* cp = (TO*)g.C + offc;
    if constexpr (sizeof(TO) == 4 && (OPT & (EPI_RAGGED | EPI_SPLIT)) != 0) {   // run-time epilogue of the fp32-output kernels, or
        if ((OPT & EPI_SPLIT) != 0 || g.split_out) {                            // a split-operand instantiation that always splits
            bf16* hp = (bf16*)g.C + offc;
            bf16x8 h;
            float lo[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { h[j] = (bf16)v[j]; lo[j] = v[j] - (float)h[j]; }
            *(bf16x8*)hp = h;
            store8(hp + g.ldc / 2, lo);
            if constexpr (OPT & EPI_COLSUM) {
#pragma unroll
                for (int j = 0; j < 8; ++j) e.cs[j] += v[j];
            }
            return;
        }
    }
#ifdef VB_DEV_KNOBS
    // store ablations (developer library; results WRONG or merely differently cached): bit 22 = every tile's rows wrap into the first 256
    // rows of C (the stores hit the same L2-resident lines: no HBM write traffic); bits 23-24 = cache-policy bits on the store
    // instruction (1: sc1, 2: sc0 sc1 = system scope, 3: nt)
    if constexpr (sizeof(TO) == 2) {
        if (g.debug & (0xF << 22)) {
            if (g.debug & (1 << 22)) cp = (TO*)g.C + (offc % (256 * g.ldc));
            bf16x8 xv;
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = (bf16)v[j];
            const u32x4 w = *(const u32x4*)&xv;
            const int pol = (g.debug >> 23) & 3;
            if (pol == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(cp), "v"(w) : "memory");
            else if (pol == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(cp), "v"(w) : "memory");
            else if (pol == 3) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(cp), "v"(w) : "memory");
            else *(u32x4*)cp = w;
            return;
        }
    }
#endif
#ifdef VB_DEV_KNOBS
    if (g.debug & 128) { if (v[0] == 123.456f) store8(cp, v); }           // ablation (developer library): no global stores
    else
#endif
    { if (g.nt_store) store8_nt(cp, v); else store8(cp, v); }             // streaming or plain store: GemmArgs::nt_store
    if constexpr (OPT & EPI_COLSUM) {
#pragma unroll
        for (int j = 0; j < 8; ++j) e.cs[j] += v[j];
    }
}
// column sums (bias gradient): lanes with equal (lane & 7) own the same 8 columns on different rows: reduce
// over lane bits 3..5, then one fp32 atomic per column per wave
VB_DEVICE void epi_colsum_flush(EpiLane& e, const GemmArgs& g, int nw0, int lane) {
    if (!(g.colsum && g.splits == 1)) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        e.cs[j] += __shfl_xor(e.cs[j], 8);
        e.cs[j] += __shfl_xor(e.cs[j], 16);
        e.cs[j] += __shfl_xor(e.cs[j], 32);
    }
    if (lane < 8) {
        const int n = nw0 + lane * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (n + j < g.N) vb_atomic_add_noret(g.colsum + n + j, e.cs[j]);
    }
}

// Shared-buffer form (kernels whose tile buffers are dead by now): a wave stages 32 rows x 64 columns per pass in
// its slab of the tile buffers.  mw0 / nw0: first row / column of the wave's 64x64 sub-tile.  Uses __syncthreads():
// every wave of the workgroup must call it.
template <typename T, typename TO, int ACT, int OPT>
VB_DEVICE void gemm_epilogue(f32x4 (&acc)[4][4], unsigned char* smem, const GemmArgs& g, int mw0, int nw0,
                             int wave, int lane) {
    const int li = lane & 15, lg = lane >> 4;
    unsigned char* slab = smem + wave * EPI_BYTES_PER_WAVE;
    EpiLane e;
    epi_lane_init<T, TO, OPT>(e, g, nw0, lane);
    const int cc = lane & 7;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();                 // LDS free (pass 0: main loop done; pass 1: previous reads done)
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = mh * 16 + lg * 4 + r, col = ni * 16 + li;
                    *(float*)(slab + row * EPI_PITCH + col * 4) = acc[pass * 2 + mh][ni][r];
                }
        __syncthreads();
        if (g.splits > 1) {
            // partial tile of an fp32 accumulator: one fire-and-forget atomic per element, a wave covering
            // 64 CONSECUTIVE columns of one row per instruction (4 cache lines, not 64 scattered words)
            if constexpr (sizeof(TO) == 4) {
                const int n = nw0 + lane;
                for (int row = 0; row < 32; ++row) {
                    const int m = mw0 + pass * 32 + row;
                    if (m < g.M && n < g.N)
                        vb_atomic_add_noret((float*)g.C + (long)m * g.ldc + n, e.alpha * *(const float*)(slab + row * EPI_PITCH + lane * 4));
                }
            }
            continue;
        }
        float xa[4][8], xd[4][8], xc[4][8];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int m = mw0 + pass * 32 + it * 8 + (lane >> 3);
            if (e.vec && m < g.M && e.ncol < g.N) epi_load8<T, TO, ACT, OPT>(xa[it], xd[it], xc[it], g, m, e.ncol);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3);
            const int m = mw0 + pass * 32 + row;
            if (m >= g.M || e.ncol >= g.N) continue;
            const unsigned char* src = slab + row * EPI_PITCH + cc * 32;
            if (e.vec) {
                float v[8];
                f32x4 lo = *(const f32x4*)src;
                f32x4 hi = *(const f32x4*)(src + 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
                epi_vec8<T, TO, ACT, OPT>(v, g, e, (long)m * g.ldc + e.ncol, (long)m * g.ld_aux + e.ncol,
                                          xa[it], xd[it], xc[it]);
            }
        }
        if constexpr (OPT & EPI_RAGGED) {
            // ragged / unaligned lanes: one rolled loop over (row, column) of the pass
            if (!e.vec) {
                for (int q = 0; q < 4 * e.nv; ++q) {
                    const int it = q / e.nv, j = q - it * e.nv;
                    const int row = it * 8 + (lane >> 3);
                    const int m = mw0 + pass * 32 + row;
                    if (m < g.M) epi_scalar<T, TO>(*(const float*)(slab + row * EPI_PITCH + cc * 32 + j * 4), g, e.alpha, m, e.ncol + j);
                }
            }
        }
    }
    if constexpr (OPT & EPI_COLSUM) epi_colsum_flush(e, g, nw0, lane);
}

// TE: element type of the epilogue's T operands (addend, aux) -- T, except in the split-operand mode (bf16 operands, fp32 rest)
template <typename T, typename TO, int AL, int BL, typename TE = T>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) gemm_kernel(GemmArgs g) {
    constexpr int EPC = TT<T>::EPC, BK = TT<T>::BK, KSTEPS = TT<T>::KSTEPS;
    (void)EPC;
    VB_DYN_SMEM(smem);

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;

    const int nwg = g.tiles_m * g.tiles_n;
    const int split = (int)blockIdx.x / nwg;
    const int tile = xcd_remap((int)blockIdx.x % nwg, nwg);
    const int m0 = (tile / g.tiles_n) * BM, n0 = (tile % g.tiles_n) * BN;

    const T* A = (const T*)g.A;
    const T* B = (const T*)g.B;

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // register staging (K-strided operands, or K-contiguous with a ragged tail): K-contiguous uses 4
    // vectors, K-strided 8 (threads [0,128) for A, or for B when A is K-contiguous; [128,256) for B when
    // both are K-strided)
    u32x4 ra[AL == VB_KCONTIG ? 4 : 8];
    u32x4 rb[BL == VB_KCONTIG ? 4 : 8];
    constexpr int B_TBASE = (AL == VB_KSTRIDED && BL == VB_KSTRIDED) ? 128 : 0;
    const bool a_active = (AL == VB_KCONTIG) || (t < 128);
    const bool b_active = (BL == VB_KCONTIG) || (t >= B_TBASE && t < B_TBASE + 128);
    const bool glds_a = (AL == VB_KCONTIG) && g.fast_a;
    const bool glds_b = (BL == VB_KCONTIG) && g.fast_b;

    const int nk_all = g.x3 ? 3 * g.kseg : (g.K + BK - 1) / BK;
    const int kbound = g.x3 ? (1 << 30) : g.K;       // split operands: whole K tiles only, columns past K belong to the lo plane
    const int kt0 = split * g.kt_per_split;
    const int kt1 = (kt0 + g.kt_per_split < nk_all) ? kt0 + g.kt_per_split : nk_all;

    // issue(): start moving K tile `kt` towards LDS buffer `buf`; commit(): finish it for the
    // register-staged operands (LDS-direct copies need no commit, only the barrier)
    auto issue = [&](int kt, int buf) {
        const int k0 = g.x3 ? x3_col_a(g, kt, BK) : kt * BK;
        const int k0b = g.x3 ? x3_col_b(g, kt, BK) : k0;
        unsigned char* la = smem + buf * 2 * TILE_BYTES;
        unsigned char* lb = la + TILE_BYTES;
        if constexpr (AL == VB_KCONTIG) {
            if (glds_a) glds_kcontig<T>(la, A, g.lda, g.M, m0, k0, wave, lane);
            else gload_kcontig<T>(ra, A, g.lda, g.M, kbound, m0, k0, t);
        } else {
            if (a_active) gload_kstrided<T>(ra, A, g.lda, g.M, g.K, m0, k0, t);
        }
        if constexpr (BL == VB_KCONTIG) {
            if (glds_b) glds_kcontig<T>(lb, B, g.ldb, g.N, n0, k0b, wave, lane);
            else gload_kcontig<T>(rb, B, g.ldb, g.N, kbound, n0, k0b, t);
        } else {
            if (b_active) gload_kstrided<T>(rb, B, g.ldb, g.N, g.K, n0, k0b, t - B_TBASE);
        }
    };
    auto commit = [&](int buf) {
        unsigned char* la = smem + buf * 2 * TILE_BYTES;
        unsigned char* lb = la + TILE_BYTES;
        if constexpr (AL == VB_KCONTIG) { if (!glds_a) sstore_kcontig<T>(ra, la, t); }
        else { if (a_active) sstore_kstrided(ra, la, t, T()); }
        if constexpr (BL == VB_KCONTIG) { if (!glds_b) sstore_kcontig<T>(rb, lb, t); }
        else { if (b_active) sstore_kstrided(rb, lb, t - B_TBASE, T()); }
    };

    if (kt0 < kt1) issue(kt0, 0);
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        commit(buf);
        // one barrier per K tile: (a) this tile is in LDS (register-staged stores done; LDS-direct copies
        // drained by the vmcnt(0) the barrier carries), (b) every wave has finished reading the OTHER
        // buffer (tile kt-1), which issue() below starts to overwrite
        __syncthreads();
        if (kt + 1 < kt1) issue(kt + 1, buf ^ 1);
        const unsigned char* ldsA = smem + buf * 2 * TILE_BYTES;
        const unsigned char* ldsB = ldsA + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            typename VecOf<T>::v8 fa[4], fb[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) fa[mi] = load_frag(ldsA, wm * 64 + mi * 16 + li, ks, lg, T());
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) fb[ni] = load_frag(ldsB, wn * 64 + ni * 16 + li, ks, lg, T());
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = vb_mma(fa[mi], fb[ni], acc[mi][ni]);
        }
    }

    gemm_epilogue<TE, TO, -1, EPI_ALL>(acc, smem, g, m0 + wm * 64, n0 + wn * 64, wave, lane);
}

// ---- optional per-launch HIP-event timing (bench.py's roofline leg) -----------------------------------
// Events are recorded on the stream the kernel is launched on, immediately around the launch, so the
// elapsed time is the kernel's own duration even when the host is the bottleneck.

template <typename T, typename TO, int AL, int BL, typename TE = T>
int launch_gemm(const GemmArgs& g, hipStream_t stream) {
    dim3 grid((unsigned)(g.tiles_m * g.tiles_n * g.splits)), block(NT);
    return vb_prof_launch(2.0 * g.M * g.N * g.K, (sizeof(T) == 2 ? 0 : 8) | (sizeof(TO) == 4 ? 4 : 0) | (AL << 1) | BL | (g.x3 ? 256 : 0), stream, [&]() { VB_LAUNCH((gemm_kernel<T, TO, AL, BL, TE>), grid, block, SMEM_BYTES, stream, g); });
}


// =================================================================================================
// Pipelined kernel for the hot case: both operands K-contiguous with whole K tiles (every forward
// GEMM and, through the W^T shadows, every dgrad).  WM x 2 waves, each 64x64 -> tile (64 WM) x 128.
// STAGES LDS stages filled by global_load_lds; tile kt+STAGES-1 is issued while tile kt is computed and
// the wait before each barrier is COUNTED (vmcnt(P) leaves the newest tile in flight), so HBM/L2
// latency is covered by STAGES-1 tiles of MFMA work instead of one.
// =================================================================================================
// per-lane source pointers of the LDS-direct copies are loop invariants (row clamp + swizzle done once);
// a K tile only advances them by BK elements
template <typename T, int WM>
struct FastPtrs {
    static constexpr int NW = WM * 2, BMX = WM * 64;
    static constexpr int A_INSTR = (BMX / 8) / NW, B_INSTR = (128 / 8) / NW;
    // per-lane BYTE offsets from the (wave-uniform) operand bases: 32 bits, checked by the launcher.  A K tile advances
    // the scalar base, so issuing a copy costs no vector arithmetic (same as the persistent kernel).
    unsigned a[A_INSTR];
    unsigned b[B_INSTR];
    const unsigned char* A;
    const unsigned char* B;
};
template <typename T, int WM>
VB_DEVICE void fast_setup(FastPtrs<T, WM>& p, const T* A, const T* B, const GemmArgs& g, int m0, int n0, int wave, int lane) {
    p.A = (const unsigned char*)A;
    p.B = (const unsigned char*)B;
#pragma unroll
    for (int i = 0; i < FastPtrs<T, WM>::A_INSTR; ++i) {
        const int row = (wave * FastPtrs<T, WM>::A_INSTR + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz(row);
        int grow = m0 + row;
        grow = grow < g.M ? grow : g.M - 1;
        p.a[i] = (unsigned)(((long)grow * g.lda + c * TT<T>::EPC) * (long)sizeof(T));
    }
#pragma unroll
    for (int i = 0; i < FastPtrs<T, WM>::B_INSTR; ++i) {
        const int row = (wave * FastPtrs<T, WM>::B_INSTR + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz(row);
        int grow = n0 + row;
        grow = grow < g.N ? grow : g.N - 1;
        p.b[i] = (unsigned)(((long)grow * g.ldb + c * TT<T>::EPC) * (long)sizeof(T));
    }
}
template <typename T, int WM>
VB_DEVICE void fast_issue(unsigned char* stage, const FastPtrs<T, WM>& p, int k0, int k0b, int wave) {
    unsigned char* la = stage;
    unsigned char* lb = stage + FastPtrs<T, WM>::BMX * 128;
    const unsigned char* sa = p.A + (long)k0 * (long)sizeof(T);
    const unsigned char* sb = p.B + (long)k0b * (long)sizeof(T);
#pragma unroll
    for (int i = 0; i < FastPtrs<T, WM>::A_INSTR; ++i)
        vb_glds16(sa + p.a[i], la + (wave * FastPtrs<T, WM>::A_INSTR + i) * 8 * 128);
#pragma unroll
    for (int i = 0; i < FastPtrs<T, WM>::B_INSTR; ++i)
        vb_glds16(sb + p.b[i], lb + (wave * FastPtrs<T, WM>::B_INSTR + i) * 8 * 128);
}

template <typename T, typename TO, int WM, int STAGES, int DBG = 0, int ACT = -1, int OPT = EPI_ALL, typename TE = T, bool X3 = false, bool SK = false>
VB_KERNEL VB_LAUNCH_BOUNDS(WM * 128) gemm_nt_pipe_kernel(GemmArgs g) {
    constexpr int NW = WM * 2, BMX = WM * 64;
    constexpr int BK = TT<T>::BK, KSTEPS = TT<T>::KSTEPS;
    constexpr int STAGE_BYTES = (BMX + 128) * 128;
    constexpr int PER_TILE = (BMX / 8) / NW + (128 / 8) / NW;       // LDS-direct instructions per wave per tile
    VB_DYN_SMEM(smem);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = vb_uniform(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int nwg = g.tiles_m * g.tiles_n;
    // SK: workgroup id -> (tile, K slice); consecutive ids (one XCD's run of the remap) are the slices of one tile, so the last
    // arrival reads its partners' partial tiles from its own XCD's L2 (speed only: the hand-off below is placement-independent)
    const int wid = xcd_remap((int)blockIdx.x, SK ? nwg * g.sk_splits : nwg);
    const int tile = SK ? wid / g.sk_splits : wid;
    const int kt_lo = SK ? (wid - tile * g.sk_splits) * g.sk_kps : 0;
    const int m0 = (tile / g.tiles_n) * BMX, n0 = (tile % g.tiles_n) * BN;
    const T* A = (const T*)g.A;
    const T* B = (const T*)g.B;

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk_all = X3 ? 3 * g.kseg : g.K / BK;
    const int nk = SK ? (nk_all - kt_lo < g.sk_kps ? nk_all - kt_lo : g.sk_kps) : nk_all;
    FastPtrs<T, WM> ptrs;
    fast_setup<T, WM>(ptrs, A, B, g, m0, n0, wave, lane);
    typename VecOf<T>::v8 fa[4], fb[4];
    auto issue_tile = [&](int kt) {                       // K tile kt of this workgroup's (virtual) K loop into its ring stage
        unsigned char* stage = smem + (kt % STAGES) * STAGE_BYTES;
        if constexpr (X3) fast_issue<T, WM>(stage, ptrs, x3_col_a(g, kt, BK), x3_col_b(g, kt, BK), wave);
        else fast_issue<T, WM>(stage, ptrs, (kt_lo + kt) * BK, (kt_lo + kt) * BK, wave);
    };
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) issue_tile(s);

    for (int kt = 0; kt < nk; ++kt) {
        // tile kt must have landed; tiles kt+1 .. kt+STAGES-2 (if they exist) may stay in flight
        if (STAGES >= 3 && kt + STAGES - 2 < nk) vb_wait_vmcnt<(STAGES - 2) * PER_TILE>();
        else if (STAGES >= 4 && kt + STAGES - 3 < nk) vb_wait_vmcnt<(STAGES >= 4 ? STAGES - 3 : 0) * PER_TILE>();
        else vb_wait_vmcnt<0>();
        const bool tr = (DBG & 64) && g.trace && blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0 && kt < 64;
        unsigned long long* trp = (DBG & 64) && g.trace ? g.trace + ((wave >> 2) * 64 + (kt < 64 ? kt : 63)) * 8 : nullptr;
        if (tr) trp[0] = vb_clock();          // tile landed (after the vmcnt wait)
        if (!(DBG & 32) || (kt & 3) == 0)
            vb_raw_barrier(); // (a) everyone's part of tile kt is in LDS, (b) everyone finished reading tile kt-1
        if (tr) trp[1] = vb_clock();          // barrier released
        // DBG is a compile-time ablation / experiment mask (0 in production): 1 skip tile loads, 2 skip fragment
        // reads, 4 skip MFMAs, 8 raise wave priority around the MFMA block, 16 issue the next tile's copies
        // after the first K step instead of before it
        if (!(DBG & 16) && kt + STAGES - 1 < nk && !(DBG & 1)) issue_tile(kt + STAGES - 1);
        const unsigned char* ldsA = smem + (kt % STAGES) * STAGE_BYTES;
        const unsigned char* ldsB = ldsA + BMX * 128;
        if (tr) trp[2] = vb_clock();          // tile kt+1 copies issued
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            if (!(DBG & 2) || kt == 0) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) fa[mi] = load_frag(ldsA, wm * 64 + mi * 16 + li, ks, lg, T());
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) fb[ni] = load_frag(ldsB, wn * 64 + ni * 16 + li, ks, lg, T());
            }
            if (DBG & 64) {                   // measurement build: fragments in registers before the clock is read
                vb_wait_lgkmcnt0();
                vb_sched_fence();
                if (tr) trp[3 + 2 * ks] = vb_clock();
            }
            if (!(DBG & 4)) {
                if (DBG & 8) vb_setprio<1>();
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = vb_mma(fa[mi], fb[ni], acc[mi][ni]);
                if (DBG & 8) vb_setprio<0>();
            } else {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[mi][0][0] += to_f32(fa[mi][0]) + to_f32(fb[mi][0]);   // keep the reads live
            }
            if (DBG & 64) {
                vb_sched_fence();
                if (tr) trp[4 + 2 * ks] = vb_clock();    // this K step's MFMAs issued
            }
            if ((DBG & 16) && ks == 0 && kt + STAGES - 1 < nk) issue_tile(kt + STAGES - 1);
        }
    }
    if constexpr (SK) {
        // partial tile -> this slice's slab, register-major ([16 accumulator vectors][threads]: every store / load instruction of a wave
        // covers 1 KB of consecutive bytes), WRITE-THROUGH (sc1) stores: once a wave's vmcnt has drained they are in L2 / on the fabric,
        // so the hand-off needs no release fence (which would write back the whole XCD L2: first version, 28.8 vs 30.0 us unsplit --
        // profiles/r06_small_batch_ab.txt); workgroup barrier, ONE relaxed device-scope ticket; the workgroup that draws the last ticket
        // reads every slice with sc1 loads (never served by this CU's L1: no acquire needed) and sums them IN SLICE ORDER -- its own from
        // memory too, so the result does not depend on who arrived last.  Placement-independent (guide, Guideline 16).
        constexpr int NTH = WM * 128;
        const int nsl = g.sk_splits;
        const vb_buf sbuf = vb_make_buf(g.sk_slabs);
        const unsigned tile_off = (unsigned)tile * (unsigned)nsl * (16u * NTH * 16u);
        const unsigned my_off = tile_off + (unsigned)(wid - tile * nsl) * (16u * NTH * 16u);
#pragma unroll
        for (int i = 0; i < 16; ++i) vb_buf_store16_sc1(sbuf, (unsigned)(i * NTH + t) * 16u, my_off, __builtin_bit_cast(u32x4, acc[i >> 2][i & 3]));
        vb_wait_vmcnt<0>();
        __syncthreads();                                   // every wave's stores are out; the ring is dead
        int* flag = (int*)smem;
        if (t == 0) *flag = vb_ticket_add(g.sk_cnt + tile);
        __syncthreads();
        if (*flag != nsl - 1) return;                      // uniform over the workgroup
        if (t == 0) vb_ticket_reset(g.sk_cnt + tile);      // zero again for the next launch on this stream
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] = __builtin_bit_cast(f32x4, vb_buf_load16_sc1(sbuf, (unsigned)(i * NTH + t) * 16u, tile_off));
        for (int sl = 1; sl < nsl; ++sl) {
            const unsigned off = tile_off + (unsigned)sl * (16u * NTH * 16u);
            f32x4 part[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) part[i] = __builtin_bit_cast(f32x4, vb_buf_load16_sc1(sbuf, (unsigned)(i * NTH + t) * 16u, off));
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] += part[i];
        }
    }
    gemm_epilogue<TE, TO, ACT, OPT>(acc, smem, g, m0 + wm * 64, n0 + wn * 64, wave, lane);
}

// activation baked into the kernel where it matters (bf16 in / bf16 out: FFN-in forward GELU, FFN-out dgrad GELU');
// everything else takes the run-time epilogue (ACT = -1), except the plain bias epilogue (ACT = 0)
template <typename T, typename TO>
constexpr bool kActSpecialised = (sizeof(T) == 2 && sizeof(TO) == 2);

template <typename T, typename TO, int WM, int STAGES, int ACT, int OPT, typename TE = T, bool X3 = false, bool SK = false>
int launch_pipe_act(const GemmArgs& g, dim3 grid, dim3 block, int smem_bytes, hipStream_t stream) {
    return vb_prof_launch(2.0 * g.M * g.N * g.K, (sizeof(T) == 2 ? 0 : 8) | (sizeof(TO) == 4 ? 4 : 0) | (X3 ? 256 : 0) | (SK ? 1024 : 0), stream, [&]() { VB_LAUNCH((gemm_nt_pipe_kernel<T, TO, WM, STAGES, 0, ACT, OPT, TE, X3, SK>), grid, block, smem_bytes, stream, g); });
}

template <typename T, typename TO, int WM, int STAGES>
int launch_pipe(GemmArgs g, hipStream_t stream) {
    constexpr int BMX = WM * 64;
    constexpr int SM = STAGES * (BMX + 128) * 128;
    static_assert(WM * 2 * EPI_BYTES_PER_WAVE <= SM, "epilogue slabs must fit");
    // 32-bit byte offsets inside the kernel: larger operands take the generic kernel
    if ((long)g.M * g.lda * (long)sizeof(T) >= (1L << 32) || (long)g.N * g.ldb * (long)sizeof(T) >= (1L << 32)) {
        if constexpr (sizeof(T) == 2 && sizeof(TO) == 4) { if (g.x3) return launch_gemm<T, TO, VB_KCONTIG, VB_KCONTIG, float>(g, stream); }
        return launch_gemm<T, TO, VB_KCONTIG, VB_KCONTIG>(g, stream);
    }
    g.tiles_m = (g.M + BMX - 1) / BMX;
    dim3 grid((unsigned)(g.tiles_m * g.tiles_n)), block(WM * 128);
    if constexpr (sizeof(T) == 2 && sizeof(TO) == 4) {
        if (g.x3) return launch_pipe_act<T, TO, WM, STAGES, -1, EPI_ALL, float, true>(g, grid, block, SM, stream);
    }
    if constexpr (sizeof(T) == 2 && sizeof(TO) == 2 && WM == 4 && STAGES == 2) {
        switch (g.debug & 127) {                  // ablation / timeline builds of this kernel (measurement only)
            case 0: break;
#define VB_DBG_CASE(D) case D: VB_LAUNCH((gemm_nt_pipe_kernel<T, TO, WM, STAGES, D, 0, 0>), grid, block, SM, stream, g); return vb_check_launch();
            VB_DBG_CASE(1) VB_DBG_CASE(2) VB_DBG_CASE(4) VB_DBG_CASE(64)
#undef VB_DBG_CASE
            default: return VB_ERR_ARG;
        }
    }
    const int needs = epi_needs(g, sizeof(T), sizeof(TO));
    if constexpr (kActSpecialised<T, TO> && STAGES == 4) {
        if (g.sk_splits > 1) {                             // K slices on different compute units (dispatch_pipe decided; light epilogues only)
            dim3 gridk((unsigned)(g.tiles_m * g.tiles_n * g.sk_splits));
            if (g.act == VB_ACT_NONE && needs == 0) return launch_pipe_act<T, TO, WM, STAGES, VB_ACT_NONE, 0, T, false, true>(g, gridk, block, SM, stream);
            if (g.act == VB_ACT_NONE && (needs & ~EPI_ADD) == 0) return launch_pipe_act<T, TO, WM, STAGES, VB_ACT_NONE, EPI_ADD, T, false, true>(g, gridk, block, SM, stream);
            g.sk_splits = 1;
        }
    }
    g.sk_splits = 1;
#define VB_TRY_EPI(A, O) if (g.act == (A) && (needs & ~(O)) == 0) return launch_pipe_act<T, TO, WM, STAGES, A, O>(g, grid, block, SM, stream)
    if constexpr (kActSpecialised<T, TO>) {
        VB_TRY_EPI(VB_ACT_NONE, 0);                        // forward projections, dgrad attention-out
        VB_TRY_EPI(VB_ACT_GELU_SAVE_GRAD, 0);              // FFN-in forward
        VB_TRY_EPI(VB_ACT_MUL_AUX, EPI_COLSUM);            // FFN-out dgrad (+ FFN-in bias gradient)
        VB_TRY_EPI(VB_ACT_NONE, EPI_ADD);                  // dgrads that add the residual gradient
    } else if constexpr (sizeof(T) == 2) {
        VB_TRY_EPI(VB_ACT_NONE, EPI_RAGGED);               // MLM decoder logits (N = vocabulary size)
    }
#undef VB_TRY_EPI
    return launch_pipe_act<T, TO, WM, STAGES, -1, EPI_ALL>(g, grid, block, SM, stream);
}

// =================================================================================================
// 256x256 tile, 8 waves as 2 (M) x 4 (N), EIGHT-PHASE schedule (guide "256^2 8-phase template", T3+T4+T5).
// The in-kernel timeline of the two-barrier kernels above (profiles/r01_gemm_timeline.txt) shows every
// wave spending ~25 % of a K tile ISSUING its LDS-direct copies (all 8 waves queue on the CU's one
// texture-address unit at the same moment), ~22 % waiting for fragment reads and ~15 % for the next
// tile, with the matrix pipe busy ~40 %.  Here a K tile is cut into 4 phases (one 64x32 quadrant of the
// wave's 128x64 output each: 16 MFMAs), a phase is {fragment reads + ONE half-tile's copies + counted
// vmcnt -> barrier -> MFMAs -> barrier}, and the two waves that share a SIMD (w, w+4) run ONE BARRIER
// apart: while one issues memory work the other owns the matrix pipe (s_setprio 1).
//
// LDS: 2 buffers x {A0, A1, B0, B1} half-tiles of [128 rows][128 B] (128 KB).  Half-tile A_mh holds tile
// rows (r>>6)*128 + mh*64 + (r&63), B_nh holds tile rows (r>>5)*64 + nh*32 + (r&31): a wave's quadrant
// (mh, nh) reads rows wr*64.. of A_mh and wc*32.. of B_nh, while its output stays a contiguous 128x64 block.
// Copy stream (half-tile index h = 4*tile + {A0,B0,B1,A1}): h = 0..5 in the prologue, phase p issues
// h = p + 6 and waits until only the newest 4 half-tiles are in flight; phase p reads h <= p + 1 (landed
// and barrier-published one phase earlier) and overwrites a half-tile last read >= 2 phases ago.
// =================================================================================================
// Epilogue of the persistent kernel: a wave drains its 128x64 block through a PRIVATE 4 KB LDS slab (16 rows
// x 64 fp32, 16-column groups XOR-swizzled by the row quad so the MFMA-layout stores are conflict-free), one
// 16-row fragment row per pass.  No workgroup barrier: LDS executes one wave's instructions in order, so the
// wave's own stores are visible to its own loads; the slabs live ABOVE the two 64 KB tile buffers, which keep
// receiving the next tile's copies meanwhile.
constexpr int EPI8_BYTES_PER_WAVE = 16 * 256;
// one 16-row fragment row (4 fragments = 16 rows x 64 columns) of a wave's block; mrow0 = its first global row
// offc / offa: element offsets of this lane's FIRST row (mrow0 + lane/8) in C / aux; its second row is 8 rows further
template <typename T, typename TO, int ACT, int OPT>
VB_DEVICE void gemm_epilogue_fragrow(f32x4 (&a)[4], unsigned char* slab, const GemmArgs& g, int mrow0, int lane, EpiLane& e,
                                      const u32x4* pre, int pre_kind, long offc, long offa) {
    const int li = lane & 15, lg = lane >> 4, cc = lane & 7;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *(float*)(slab + (lg * 4 + r) * 256 + (((ni ^ lg) * 16 + li) << 2)) = a[ni][r];
    vb_wave_sync();
    float xa[2][8], xd[2][8], xc[2][8];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int m = mrow0 + it * 8 + (lane >> 3);
        // pre_kind is a COMPILE-TIME fact of the specialised bf16 instantiations (1: aux_in, 2: addend): with a run-time fallback to
        // epi_load8 next to it the compiler had to assume a load might be pending at every later use and waited vmcnt(0) -- for the
        // previous fragment row's stores -- eight times per tile
        if constexpr (sizeof(T) == 2 && !(OPT & EPI_RAGGED) && ACT >= 0 && ((OPT & EPI_ADD) || ACT == VB_ACT_MUL_AUX || ACT == VB_ACT_GELU_GRAD)) {
            const bf16x8 x = *(const bf16x8*)&pre[it];
#pragma unroll
            for (int j = 0; j < 8; ++j) { if (pre_kind == 1) xa[it][j] = (float)x[j]; else xd[it][j] = (float)x[j]; }
            continue;
        }
        if (e.vec && m < g.M && e.ncol < g.N) epi_load8<T, TO, ACT, OPT>(xa[it], xd[it], xc[it], g, m, e.ncol);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = it * 8 + (lane >> 3);
        const int m = mrow0 + row;
        const unsigned char* src = slab + row * 256 + (((((cc >> 1) ^ (row >> 2)) & 3) * 16 + (cc & 1) * 8) << 2);
        if (m < g.M && e.ncol < g.N && e.vec) {
            float v[8];
            f32x4 lo = *(const f32x4*)src;
            f32x4 hi = *(const f32x4*)(src + 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
            if constexpr (ACT == VB_ACT_GELU_SAVE_GRAD)
                epi_vec8<T, TO, ACT, OPT>(v, g, e, offc + it * 8 * g.ldc, offa + it * 8 * g.ld_aux, xa[it], xd[it], xc[it]);
            else
                epi_vec8<T, TO, ACT, OPT>(v, g, e, (long)m * g.ldc + e.ncol, (long)m * g.ld_aux + e.ncol, xa[it], xd[it], xc[it]);
        }
    }
    if constexpr (OPT & EPI_RAGGED) {
        if (!e.vec) {                                   // ragged / unaligned lanes: one rolled loop per fragment row
            for (int q = 0; q < 2 * e.nv; ++q) {
                const int it = q / e.nv, j = q - it * e.nv;
                const int row = it * 8 + (lane >> 3);
                const int m = mrow0 + row;
                const unsigned char* src = slab + row * 256 + (((((cc >> 1) ^ (row >> 2)) & 3) * 16 + (cc & 1) * 8) << 2);
                if (m < g.M) epi_scalar<T, TO>(((const float*)src)[j], g, e.alpha, m, e.ncol + j);
            }
        }
    }
    vb_wave_sync();
}
// developer library: pause after every fragment row of the epilogue (debug bits 20-21: 256 / 1024 / 3072 cycles) -- spreads a wave's 16
// stores in time so that the co-resident workgroup's copies can interleave with them (profiles/r04_gemm_store_ablation_b1024.txt)
VB_DEVICE void epi_pause(const GemmArgs& g) {
#ifdef VB_DEV_KNOBS
    const int p = (g.debug >> 20) & 3;
    if (p == 1) __builtin_amdgcn_s_sleep(4);
    else if (p == 2) __builtin_amdgcn_s_sleep(16);
    else if (p == 3) __builtin_amdgcn_s_sleep(48);
#else
    (void)g;
#endif
}
// FLUSH: one compiler-visible vmcnt(0) after the epilogue's loads (below).  (Loading the bias BEFORE the K loop of the one-tile-per-workgroup
// kernel, so that this wait costs nothing, does not fit: 8 more registers through the K loop spill, and an EpiLane handed over by
// POINTER lives in scratch memory altogether -- 128 bytes per lane, 200 scratch instructions.)
template <typename T, typename TO, int ACT, int OPT, bool FLUSH = false>
VB_DEVICE void gemm_epilogue_private(f32x4 (&acc)[8][4], unsigned char* slab, const GemmArgs& g, int mw0, int nw0, int lane) {
    EpiLane e;
    epi_lane_init<T, TO, OPT>(e, g, nw0, lane);
    // The per-row operand of the specialised epilogues (saved GELU' / residual gradient) is fetched for the WHOLE 128x64
    // block as 16 independent 16-byte loads per lane, ALL ahead of the first store (the fragment registers are dead by now: 64 registers),
    // instead of two per fragment row, each waiting out an HBM round trip before its multiply (8 round trips per tile).  Until round 5
    // they went out in two batches of 8 with four fragment rows of stores in between: the wait for the second batch was then a wait
    // for those stores' acknowledgements as well (one in-order vmcnt), and in the persistent kernel -- LDS-direct copies of the next
    // tile in flight, exec-masked blocks per fragment row -- the compiler re-waited vmcnt(0) at later uses of the bias and batch
    // registers too, i.e. for every store issued since (74 such waits in the "+ addend" instantiation: 667 us where the same source in
    // the developer build, laid out differently, ran 603).  Now: every load of the epilogue first, ONE compiler-visible vmcnt(0), then
    // eight fragment rows of LDS traffic and stores with nothing left to wait for.
    constexpr bool PRE_AUX = sizeof(T) == 2 && !(OPT & EPI_RAGGED) && (ACT == VB_ACT_MUL_AUX || ACT == VB_ACT_GELU_GRAD);
    constexpr bool PRE_ADD = sizeof(T) == 2 && !(OPT & EPI_RAGGED) && ACT >= 0 && !PRE_AUX && (OPT & EPI_ADD);
    u32x4 pre[8][2];                                // all 8 fragment rows (16 loads in flight, 64 VGPRs)
    constexpr int pre_kind = PRE_AUX ? 1 : (PRE_ADD ? 2 : 0);   // EPI_ADD here = "addend, no accumulate" (epi_needs sends "+= C" to the run-time epilogue)
    const unsigned char* pbase = nullptr;
    long pld = 0;
    if constexpr (PRE_AUX || PRE_ADD) {
        if (PRE_AUX) { pbase = (const unsigned char*)g.aux_in; pld = g.ld_aux; }
        else { pbase = (const unsigned char*)g.addend; pld = g.ld_addend; }
    }
    // row addressing without per-row 64-bit multiplies: one product per matrix here, 64-bit adds from then on
    const int mlane = mw0 + (lane >> 3);
    const long stepc = 16 * g.ldc, stepa = 16 * g.ld_aux;
    long offc = (long)mlane * g.ldc + e.ncol, offa = (long)mlane * g.ld_aux + e.ncol;
    if constexpr (PRE_AUX || PRE_ADD) {
        if (pre_kind) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    int m = mw0 + mi * 16 + it * 8 + (lane >> 3);
                    m = m < g.M ? m : g.M - 1;                  // clamped rows are loaded but never stored
                    const int n = e.ncol < g.N ? e.ncol : 0;
                    pre[mi][it] = *(const u32x4*)(pbase + ((long)m * pld + n) * 2);
                }
        }
    }
    // ONE compiler-visible vmcnt(0) for everything the epilogue loaded -- the bias, the row operands (+ whatever copies of the next tile
    // were in flight).  Without it the compiler, which finds the first use of those registers inside the exec-masked block of fragment
    // row 0, RE-WAITS at every later use -- `s_waitcnt vmcnt(1)`, `vmcnt(0)` in front of each store's arithmetic -- and since a wave has
    // one in-order vmcnt for loads and stores, each of those waits is a wait for the PREVIOUS STORE's acknowledgement: sixteen
    // serialised store round trips per wave and tile.  That, not the store path's rate, was the per-tile "store cost" of rounds 1-4
    // ("Round 5").  Where the waits PACED the stores to the benefit of a co-resident workgroup, the caller leaves
    // FLUSH off (the two-workgroup kernel's fp32 and x GELU' epilogues).
    if constexpr (FLUSH) vb_wait_vmcnt0_visible();
    // constant indices spelled out: the accumulators must never be addressed by a loop variable
    gemm_epilogue_fragrow<T, TO, ACT, OPT>(acc[0], slab, g, mw0 + 0, lane, e, pre[0], pre_kind, offc, offa);
    epi_pause(g);
    gemm_epilogue_fragrow<T, TO, ACT, OPT>(acc[1], slab, g, mw0 + 16, lane, e, pre[1], pre_kind, offc + stepc, offa + stepa);
    epi_pause(g);
    gemm_epilogue_fragrow<T, TO, ACT, OPT>(acc[2], slab, g, mw0 + 32, lane, e, pre[2], pre_kind, offc + 2 * stepc, offa + 2 * stepa);
    epi_pause(g);
    gemm_epilogue_fragrow<T, TO, ACT, OPT>(acc[3], slab, g, mw0 + 48, lane, e, pre[3], pre_kind, offc + 3 * stepc, offa + 3 * stepa);
    epi_pause(g);
    offc += 4 * stepc; offa += 4 * stepa;
    gemm_epilogue_fragrow<T, TO, ACT, OPT>(acc[4], slab, g, mw0 + 64, lane, e, pre[4], pre_kind, offc, offa);
    epi_pause(g);
    gemm_epilogue_fragrow<T, TO, ACT, OPT>(acc[5], slab, g, mw0 + 80, lane, e, pre[5], pre_kind, offc + stepc, offa + stepa);
    epi_pause(g);
    gemm_epilogue_fragrow<T, TO, ACT, OPT>(acc[6], slab, g, mw0 + 96, lane, e, pre[6], pre_kind, offc + 2 * stepc, offa + 2 * stepa);
    epi_pause(g);
    gemm_epilogue_fragrow<T, TO, ACT, OPT>(acc[7], slab, g, mw0 + 112, lane, e, pre[7], pre_kind, offc + 3 * stepc, offa + 3 * stepa);
    epi_pause(g);
    if constexpr (OPT & EPI_COLSUM) epi_colsum_flush(e, g, nw0, lane);
}

// =================================================================================================
// REGISTER-DIRECT epilogue (round 5; developer-library experiment arms nt_kernel 92 / 82 -- it LOST, see kDirectEpilogue below) of the
// specialised instantiations (one activation, no ragged edge) of the two fast kernels.
// The K loop runs with the MFMA operand roles SWAPPED -- acc = W-fragment x activation-fragment -- so lane (li, lg) of a wave holds,
// per 16x16 fragment (mi, ni), FOUR CONSECUTIVE COLUMNS of ONE row: C[mw0 + 16 mi + li][nw0 + 16 ni + 4 lg + 0..3].  Two
// v_permlane16_swap_b32 per register pair (vb_rt.h: the ODD 16-lane rows of one register exchanged with the EVEN rows of the other)
// turn the fragments (mi, 2 j) and (mi, 2 j + 1) into EIGHT contiguous columns per lane, nw0 + 32 j + vb_wide_col(lg) + 0..7 -- the
// unit every element-wise epilogue above (epi_vec8) is written for -- and the result leaves in ONE 16-byte store per lane, 16 store
// instructions per wave and tile exactly like the LDS route, but with no LDS round trip at all: 128 ds_write_b32, 32 ds_read_b128,
// 16 waits per wave and tile are gone, and with them the 2-way bank conflicts of the slab reads (7.7 % of the LDS-active cycles of
// these kernels, profiles/r04_pmc_step_bf16.txt).  Round 1 tried the swapped roles with 8-byte stores (twice the store instructions:
// 8-15 % slower, profiles/r01_gemm_register_epilogue_experiment.txt); what prices an epilogue is its store-instruction count
// ("Round 5"), and the lane exchange keeps it.
// A store instruction covers 16 rows x 64 bytes here (8 rows x 128 bytes on the LDS route); the two 64-byte halves of a 128-byte
// line are written by the two stores of the same fragment row, back to back.
VB_DEVICE void vb_permlane16_swap_f(float& a, float& b) {
    uint32_t x = __builtin_bit_cast(uint32_t, a), y = __builtin_bit_cast(uint32_t, b);
    vb_permlane16_swap(x, y);
    a = __builtin_bit_cast(float, x); b = __builtin_bit_cast(float, y);
}
template <typename T, typename TO>
VB_DEVICE void epi_lane_init_direct(EpiLane& e, const GemmArgs& g, int ncol) {
    e.alpha = g.alpha_dev ? g.alpha * g.alpha_dev[0] : g.alpha;
    e.ncol = ncol;
    e.nv = 8; e.vec = true;                             // the launcher checked N % 8 == 0 and every alignment
#pragma unroll
    for (int j = 0; j < 8; ++j) { e.bb[j] = 0.f; e.cs[j] = 0.f; }
    if (g.bias && e.ncol < g.N) load8(e.bb, g.bias + e.ncol);
}
// column sums: the 16 lanes of a lane row (equal lg) own the same 8 columns on 16 different rows
VB_DEVICE void epi_colsum_flush_direct(EpiLane& e, const GemmArgs& g, int lane) {
    if (!(g.colsum && g.splits == 1)) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        e.cs[j] += __shfl_xor(e.cs[j], 1);
        e.cs[j] += __shfl_xor(e.cs[j], 2);
        e.cs[j] += __shfl_xor(e.cs[j], 4);
        e.cs[j] += __shfl_xor(e.cs[j], 8);
    }
    if ((lane & 15) == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (e.ncol + j < g.N) vb_atomic_add_noret(g.colsum + e.ncol + j, e.cs[j]);
    }
}
template <typename T, typename TO, int ACT, int OPT>
VB_DEVICE void gemm_epilogue_direct(f32x4 (&acc)[8][4], const GemmArgs& g, int mw0, int nw0, int lane) {
    static_assert(ACT >= 0 && !(OPT & EPI_RAGGED), "specialised, aligned instantiations only");
    const int li = lane & 15, lg = lane >> 4;
    EpiLane e0, e1;
    epi_lane_init_direct<T, TO>(e0, g, nw0 + vb_wide_col(lg));
    epi_lane_init_direct<T, TO>(e1, g, nw0 + 32 + vb_wide_col(lg));
    // the per-row operand of the specialised epilogues (saved GELU' / residual gradient): two batches of 8 independent 16-byte loads
    // per lane, issued ahead of the batch's first multiply (one round trip per batch instead of one per row)
    constexpr bool PRE_AUX = sizeof(T) == 2 && (ACT == VB_ACT_MUL_AUX || ACT == VB_ACT_GELU_GRAD);
    constexpr bool PRE_ADD = sizeof(T) == 2 && !PRE_AUX && (OPT & EPI_ADD);
    u32x4 pre[4][2];
    int pre_kind = 0;
    const unsigned char* pbase = nullptr;
    long pld = 0;
    if constexpr (PRE_AUX || PRE_ADD) {
        if (PRE_AUX) { pbase = (const unsigned char*)g.aux_in; pld = g.ld_aux; pre_kind = 1; }
        else if (g.addend && !g.accumulate) { pbase = (const unsigned char*)g.addend; pld = g.ld_addend; pre_kind = 2; }
    }
    auto preload = [&](int mi0) {
        if constexpr (PRE_AUX || PRE_ADD) {
            if (pre_kind) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        int m = mw0 + (mi0 + mi) * 16 + li;
                        m = m < g.M ? m : g.M - 1;                  // clamped rows are loaded but never stored
                        const int nc = j ? e1.ncol : e0.ncol;
                        pre[mi][j] = *(const u32x4*)(pbase + ((long)m * pld + (nc < g.N ? nc : 0)) * 2);
                    }
            }
        }
    };
    const long rowc = (long)(mw0 + li) * g.ldc, rowa = (long)(mw0 + li) * g.ld_aux;
    const long stepc = 16 * g.ldc, stepa = 16 * g.ld_aux;
    // one fragment row (16 rows x 64 columns of the wave's block); mi is a compile-time constant at every call
    auto fragrow = [&](f32x4 (&a)[4], int mi, const u32x4 (&pr)[2]) {
        const int m = mw0 + mi * 16 + li;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            EpiLane& e = j ? e1 : e0;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                       // every lane takes part; only the store is predicated
                float x = a[2 * j][r], y = a[2 * j + 1][r];
                vb_permlane16_swap_f(x, y);
                v[r] = x; v[4 + r] = y;
            }
            float xa[8], xd[8], xc[8];
            const bool ok = m < g.M && e.ncol < g.N;
            if constexpr (PRE_AUX || PRE_ADD) {
                if (pre_kind) {
                    const bf16x8 x = *(const bf16x8*)&pr[j];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { if (pre_kind == 1) xa[q] = (float)x[q]; else xd[q] = (float)x[q]; }
                } else if (ok) epi_load8<T, TO, ACT, OPT>(xa, xd, xc, g, m, e.ncol);
            } else {
                if (ok) epi_load8<T, TO, ACT, OPT>(xa, xd, xc, g, m, e.ncol);
            }
            if (ok) epi_vec8<T, TO, ACT, OPT>(v, g, e, rowc + mi * stepc + e.ncol, rowa + mi * stepa + e.ncol, xa, xd, xc);
        }
    };
    // constant indices spelled out: the accumulators must never be addressed by a loop variable
    preload(0);
    fragrow(acc[0], 0, pre[0]); fragrow(acc[1], 1, pre[1]); fragrow(acc[2], 2, pre[2]); fragrow(acc[3], 3, pre[3]);
    preload(4);
    fragrow(acc[4], 4, pre[0]); fragrow(acc[5], 5, pre[1]); fragrow(acc[6], 6, pre[2]); fragrow(acc[7], 7, pre[3]);
    if constexpr (OPT & EPI_COLSUM) { epi_colsum_flush_direct(e0, g, lane); epi_colsum_flush_direct(e1, g, lane); }
}
// which instantiations take it (and therefore run their K loop with swapped operand roles): the developer library's arms 92 / 82 only --
// MEASURED SLOWER than the LDS route on every shape of the step (profiles/r05_gemm_direct_epilogue_ab.txt: 69.96 vs 66.68 ms per step on
// the two-workgroup kernel, 69.72 vs 65.44 on the persistent one; the GELU + GELU' epilogue 1134 vs 1007 us): a store instruction that
// covers 16 rows x 64 bytes costs more than one that covers 8 rows x 128 bytes -- the price follows the LINES an instruction touches,
// not only the instruction count -- which is what the LDS transposition buys.  Kept as an experiment arm, not in the product.
template <int ACT, int OPT, bool ON> constexpr bool kDirectEpilogue = ACT >= 0 && !(OPT & EPI_RAGGED) && ON;

// X3: split-operand mode (see GemmArgs): the copy stream walks 3 K / 64 virtual K tiles whose column offsets come from
// x3_col_a / x3_col_b; the epilogue's row operands are fp32 (TE).  With a virtual K of >= 2304 the per-tile epilogue is a small
// share and this kernel's lower LDS traffic per FLOP is what counts (profiles/r04_power_by_kernel.txt: +9 % at K = 3072 in bf16).
template <typename T, typename TO, int ACT, int OPT, int SCHED, bool X3 = false>
VB_KERNEL VB_LAUNCH_BOUNDS(512) gemm_nt_8ph_kernel(GemmArgs g) {
    typedef typename std::conditional<X3, float, T>::type TE;
    constexpr int BK = TT<T>::BK, KSTEPS = TT<T>::KSTEPS, EPC = TT<T>::EPC;
    constexpr int HALF = 128 * 128, BUF = 4 * HALF;
    constexpr int SLOT_A0 = 0, SLOT_A1 = 1, SLOT_B0 = 2, SLOT_B1 = 3;
    // SCHED: bit 0 = four-slot schedule (what ships), bit 1 = the register-direct epilogue where it applies (developer library:
    // nt_kernel 82, an experiment arm that lost)
    constexpr bool DIRECT = kDirectEpilogue<ACT, OPT, (SCHED & 2) != 0>;
    VB_DYN_SMEM(smem);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = vb_uniform(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int li = lane & 15, lg = lane >> 4;
    const int ntiles = g.tiles_m * g.tiles_n;
    const int G = (int)gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;      // persistent: tiles blockIdx.x + G j
    const unsigned char* A = (const unsigned char*)g.A;
    const unsigned char* B = (const unsigned char*)g.B;
    unsigned char* slab = smem + 2 * BUF + wave * EPI8_BYTES_PER_WAVE;

    f32x4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // LDS-direct copies: a half-tile is 16 one-KiB instructions, 2 per wave: half-tile rows 16 wave + 8 i + lane/8.
    // Their tile rows are rowA + 64 mh + 8 i (A) / rowB + 32 nh + 8 i (B); the source chunk is the LDS chunk slot
    // XOR the row swizzle (swz(r + 8) = swz(r) ^ 4).  Per-lane BYTE offsets from the operand base (32 bits, checked by
    // the launcher) are recomputed only when the load stream moves to another output tile; a K tile advances the
    // SCALAR base, so issuing a copy costs no vector arithmetic.
    const int l3 = lane >> 3;
    const int rowA = (wave >> 2) * 128 + (wave & 3) * 16 + l3;
    const int rowB = (wave >> 1) * 64 + (wave & 1) * 16 + l3;
    const int csrc0 = ((lane & 7) ^ swz(wave * 16 + l3)) * 16;
    const int csrc1 = csrc0 ^ 64;
    unsigned offA[2][2], offB[2][2];
    int ld_j = 0, ld_t = 0;                        // load stream position: K tile ld_t of my tile number ld_j
    auto origin = [&](int j, int& m0, int& n0) {
        const int tile = xcd_remap((int)blockIdx.x + G * j, ntiles);
        m0 = (tile / g.tiles_n) * 256; n0 = (tile % g.tiles_n) * 256;
    };
    auto set_load_tile = [&](int j) {
        int m0, n0;
        origin(j, m0, n0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int a0 = m0 + rowA + h * 64, a1 = a0 + 8, b0 = n0 + rowB + h * 32, b1 = b0 + 8;
            a0 = a0 < g.M ? a0 : g.M - 1; a1 = a1 < g.M ? a1 : g.M - 1;
            b0 = b0 < g.N ? b0 : g.N - 1; b1 = b1 < g.N ? b1 : g.N - 1;
            offA[h][0] = (unsigned)(a0 * (int)g.lda) * (unsigned)sizeof(T) + csrc0;
            offA[h][1] = (unsigned)(a1 * (int)g.lda) * (unsigned)sizeof(T) + csrc1;
            offB[h][0] = (unsigned)(b0 * (int)g.ldb) * (unsigned)sizeof(T) + csrc0;
            offB[h][1] = (unsigned)(b1 * (int)g.ldb) * (unsigned)sizeof(T) + csrc1;
        }
    };
    set_load_tile(0);
    auto issueA = [&](int mh, int par) {
        unsigned char* dst = smem + par * BUF + (mh ? SLOT_A1 : SLOT_A0) * HALF + wave * 2048;
        const unsigned char* src = A + (X3 ? (long)x3_col_a(g, ld_t, BK) * 2 : (long)ld_t * (BK * (int)sizeof(T)));
        vb_glds16(src + offA[mh][0], dst);
        vb_glds16(src + offA[mh][1], dst + 1024);
    };
    auto issueB = [&](int nh, int par) {
        unsigned char* dst = smem + par * BUF + (nh ? SLOT_B1 : SLOT_B0) * HALF + wave * 2048;
        const unsigned char* src = B + (X3 ? (long)x3_col_b(g, ld_t, BK) * 2 : (long)ld_t * (BK * (int)sizeof(T)));
        vb_glds16(src + offB[nh][0], dst);
        vb_glds16(src + offB[nh][1], dst + 1024);
    };
    const int nk = X3 ? 3 * g.kseg : g.K / BK;     // K tiles per output tile (split-operand mode: hi.hi, lo.hi, hi.lo segments)
    auto ld_advance = [&]() {
        if (++ld_t == nk) { ld_t = 0; ++ld_j; set_load_tile(ld_j); }
    };

    typename VecOf<T>::v8 fa[4][KSTEPS], fb0[2][KSTEPS], fb1[2][KSTEPS];
    auto readA = [&](const unsigned char* half) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) fa[f][ks] = load_frag(half, wr * 64 + f * 16 + li, ks, lg, T());
    };
    auto readB = [&](typename VecOf<T>::v8 (&fb)[2][KSTEPS], const unsigned char* half) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) fb[f][ks] = load_frag(half, wc * 32 + f * 16 + li, ks, lg, T());
    };
    auto quad = [&](int mh, int nh, typename VecOf<T>::v8 (&fb)[2][KSTEPS]) {
        vb_setprio<1>();
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if constexpr (DIRECT) acc[mh * 4 + f][nh * 2 + q] = vb_mma(fb[q][ks], fa[f][ks], acc[mh * 4 + f][nh * 2 + q]);    // swapped roles: 4 columns per lane
                    else acc[mh * 4 + f][nh * 2 + q] = vb_mma(fa[f][ks], fb[q][ks], acc[mh * 4 + f][nh * 2 + q]);
                }
        vb_setprio<0>();
    };

    const int GK = my_tiles * nk;                  // K tiles in this workgroup's stream
    if constexpr ((SCHED & 1) != 0) {
        // ---- FOUR-SLOT schedule: two quadrants (32 MFMAs) per slot, half the barriers.  Per K tile g (buffer g & 1):
        //   E(g): reads A0 B0 B1 | issues A1 of tile g+1 | MFMAs (0,0) (0,1)
        //   O(g): reads A1       | issues A0 B0 B1 of tile g+2 (into the buffer E(g) just drained) | MFMAs (1,1) (1,0)
        // A wave waits for its own fragment reads BEFORE the barrier that ends its memory slot, so a half-tile may be
        // refilled one phase after its last read; every counted wait is "all but the newest 8".
        issueA(0, 0); issueB(0, 0); issueB(1, 0); issueA(1, 0);
        if (GK > 1) { ld_advance(); issueA(0, 1); issueB(0, 1); issueB(1, 1); vb_wait_vmcnt<6>(); }
        else vb_wait_vmcnt<0>();
        vb_phase_barrier();
        if (wr == 1) vb_phase_barrier();           // waves 4-7 run one barrier behind waves 0-3
        int ct = 0, cj = 0;
        for (int gk = 0; gk < GK; ++gk) {
            const int par = gk & 1;
            const unsigned char* buf = smem + par * BUF;
            const bool n1 = gk + 1 < GK, n2 = gk + 2 < GK;
            // ---- E
            readB(fb0, buf + SLOT_B0 * HALF);
            readA(buf + SLOT_A0 * HALF);
            readB(fb1, buf + SLOT_B1 * HALF);
            if (n1) { issueA(1, par ^ 1); vb_wait_vmcnt<8>(); } else vb_wait_vmcnt<0>();
            vb_raw_barrier();                      // lgkmcnt(0) first: my reads of A0 B0 B1 are done
            vb_sched_fence();
            quad(0, 0, fb0);
            quad(0, 1, fb1);
            vb_phase_barrier();
            // ---- O
            readA(buf + SLOT_A1 * HALF);
            if (n2) { ld_advance(); issueA(0, par); issueB(0, par); issueB(1, par); vb_wait_vmcnt<8>(); }
            else if (n1) vb_wait_vmcnt<2>();
            else vb_wait_vmcnt<0>();
            vb_raw_barrier();
            vb_sched_fence();
            quad(1, 1, fb1);
            quad(1, 0, fb0);
            vb_phase_barrier();
            if (++ct == nk) {
                // output tile finished: drain it while the next tile's first K tiles are already landing
                int m0, n0;
                origin(cj, m0, n0);
                if constexpr (DIRECT) gemm_epilogue_direct<TE, TO, ACT, OPT>(acc, g, m0 + wr * 128, n0 + wc * 64, lane);
                else gemm_epilogue_private<TE, TO, ACT, OPT, true>(acc, slab, g, m0 + wr * 128, n0 + wc * 64, lane);
    #pragma unroll
                for (int mi = 0; mi < 8; ++mi)
    #pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
                ct = 0; ++cj;
            }
        }
        if (wr == 0) vb_phase_barrier();           // balance the stagger barrier
        return;
    }
    // prologue: K tile 0 (A0 B0 B1 A1) and the first two half-tiles of K tile 1
    issueA(0, 0); issueB(0, 0); issueB(1, 0); issueA(1, 0);
    if (GK > 1) { ld_advance(); issueA(0, 1); issueB(0, 1); vb_wait_vmcnt<8>(); }
    else vb_wait_vmcnt<4>();
    vb_phase_barrier();
    if (wr == 1) vb_phase_barrier();               // waves 4-7 run one barrier behind waves 0-3

    // vmcnt counts the epilogue's loads and stores too.  The vector-memory operations of a wave retire in issue order, so
    // "all but the newest 8" stays correct after an epilogue: it only waits for MORE (the epilogue's stores).  Admitting
    // those stores explicitly (vmcnt(8 + 16) for one K tile) measured no gain: the write burst is bandwidth, not
    // acknowledgement latency (profiles/r01_gemm_8phase_notes.txt).
    auto wait_steady = [&]() { vb_wait_vmcnt<8>(); };
    int ct = 0, cj = 0;
    for (int gk = 0; gk < GK; ++gk) {
        const int par = gk & 1;
        const unsigned char* buf = smem + par * BUF;
        const bool n1 = gk + 1 < GK, n2 = gk + 2 < GK;
        // ---- phase 0: quadrant (0, 0)
        readB(fb0, buf + SLOT_B0 * HALF);
        readA(buf + SLOT_A0 * HALF);
        if (n1) { issueB(1, par ^ 1); wait_steady(); } else vb_wait_vmcnt<2>();
        vb_phase_barrier();
        quad(0, 0, fb0);
        vb_phase_barrier();
        // ---- phase 1: quadrant (0, 1)
        readB(fb1, buf + SLOT_B1 * HALF);
        if (n1) { issueA(1, par ^ 1); wait_steady(); } else vb_wait_vmcnt<0>();
        vb_phase_barrier();
        quad(0, 1, fb1);
        vb_phase_barrier();
        // ---- phase 2: quadrant (1, 1)
        readA(buf + SLOT_A1 * HALF);
        if (n2) { ld_advance(); issueA(0, par); wait_steady(); } else if (n1) vb_wait_vmcnt<6>(); else vb_wait_vmcnt<0>();
        vb_phase_barrier();
        quad(1, 1, fb1);
        vb_phase_barrier();
        // ---- phase 3: quadrant (1, 0)
        if (n2) { issueB(0, par); wait_steady(); } else if (n1) vb_wait_vmcnt<4>(); else vb_wait_vmcnt<0>();
        vb_phase_barrier();
        quad(1, 0, fb0);
        vb_phase_barrier();
        if (++ct == nk) {
            // output tile finished: drain it while the next tile's first K tiles are already landing
            int m0, n0;
            origin(cj, m0, n0);
            if constexpr (DIRECT) gemm_epilogue_direct<TE, TO, ACT, OPT>(acc, g, m0 + wr * 128, n0 + wc * 64, lane);
            else gemm_epilogue_private<TE, TO, ACT, OPT, true>(acc, slab, g, m0 + wr * 128, n0 + wc * 64, lane);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            ct = 0; ++cj;
        }
    }
    if (wr == 0) vb_phase_barrier();               // balance the stagger barrier
}

template <typename T, typename TO, int ACT, int OPT, int SCHED, bool X3 = false>
int launch_8ph_sched(const GemmArgs& g, dim3 grid, dim3 block, int smem_bytes, hipStream_t stream) {
    return vb_prof_launch(2.0 * g.M * g.N * g.K, (sizeof(T) == 2 ? 0 : 8) | (sizeof(TO) == 4 ? 4 : 0) | 16 | (X3 ? 256 : 0), stream, [&]() { VB_LAUNCH((gemm_nt_8ph_kernel<T, TO, ACT, OPT, SCHED, X3>), grid, block, smem_bytes, stream, g); });
}
template <typename T, typename TO, int ACT, int OPT>
int launch_8ph_act(const GemmArgs& g, dim3 grid, dim3 block, int smem_bytes, hipStream_t stream) {
#ifdef VB_DEV_KNOBS
    if (t_opts.nt_kernel == 80) return launch_8ph_sched<T, TO, ACT, OPT, 0>(g, grid, block, smem_bytes, stream);   // eight slots
    if (t_opts.nt_kernel == 82) return launch_8ph_sched<T, TO, ACT, OPT, 3>(g, grid, block, smem_bytes, stream);   // four slots, register-direct epilogue (experiment arm)
#endif
    return launch_8ph_sched<T, TO, ACT, OPT, 1>(g, grid, block, smem_bytes, stream);                                // four slots
}
template <typename T, typename TO>
int launch_8ph(GemmArgs g, hipStream_t stream) {
    constexpr int SM = 2 * 4 * 128 * 128 + 8 * EPI8_BYTES_PER_WAVE;
    // 32-bit element offsets inside the kernel; bf16 operands only (fp32 is the parity path, not the fast path)
    if (sizeof(T) != 2 || (long)g.M * g.lda >= (1L << 30) || (long)g.N * g.ldb >= (1L << 30))
        return launch_pipe<T, TO, 4, 2>(g, stream);
    if constexpr (sizeof(T) == 2) {
        g.tiles_m = (g.M + 255) / 256;
        g.tiles_n = (g.N + 255) / 256;
        const int ntiles = g.tiles_m * g.tiles_n;
        // persistent: one workgroup per CU walks tiles b, b + wgs, ...; the XCD-aware tile remap assumes workgroups b and
        // b + wgs share an XCD (wgs % 8 == 0), which only matters when a workgroup has more than one tile
        int wgs = t_opts.persistent_workgroups > 0 ? t_opts.persistent_workgroups : vb_num_cus();
        if (wgs >= ntiles) wgs = ntiles;
        else if (wgs >= 8) wgs &= ~7;
        dim3 grid((unsigned)wgs), block(512);
        if constexpr (sizeof(TO) == 4) {
            if (g.x3) {                                     // split operands: the same four specialised epilogues as launch_dual
                const int needs = epi_needs(g, 4, 4);
                if (!(needs & EPI_RAGGED)) {
                    if (!g.split_out) {
                        if (g.act == VB_ACT_NONE && needs == 0) return launch_8ph_sched<T, TO, VB_ACT_NONE, 0, 1, true>(g, grid, block, SM, stream);
                        if (g.act == VB_ACT_NONE && needs == EPI_ADD) return launch_8ph_sched<T, TO, VB_ACT_NONE, EPI_ADD, 1, true>(g, grid, block, SM, stream);
                    } else {
                        if (g.act == VB_ACT_GELU_SAVE_GRAD && needs == 0)
                            return launch_8ph_sched<T, TO, VB_ACT_GELU_SAVE_GRAD, EPI_SPLIT, 1, true>(g, grid, block, SM, stream);
                        if (g.act == VB_ACT_MUL_AUX && (needs & ~EPI_COLSUM) == 0)
                            return launch_8ph_sched<T, TO, VB_ACT_MUL_AUX, EPI_COLSUM | EPI_SPLIT, 1, true>(g, grid, block, SM, stream);
                    }
                }
                return launch_8ph_sched<T, TO, -1, EPI_ALL, 1, true>(g, grid, block, SM, stream);
            }
        }
        const int needs = epi_needs(g, sizeof(T), sizeof(TO));
        if (needs & EPI_DROP) {                             // dropout + residual: the one specialised instantiation (developer library), or nothing
#ifdef VB_DEV_KNOBS
            if constexpr (kActSpecialised<T, TO>) {
                if (g.act == VB_ACT_NONE && g.addend && needs == (EPI_ADD | EPI_DROP) && g.ldc == g.N)
                    return launch_8ph_act<T, TO, VB_ACT_NONE, EPI_ADD | EPI_DROP>(g, grid, block, SM, stream);
            }
#endif
            return VB_ERR_UNSUPPORTED;
        }
#define VB_TRY_EPI(A, O) if (g.act == (A) && (needs & ~(O)) == 0) return launch_8ph_act<T, TO, A, O>(g, grid, block, SM, stream)
        if constexpr (kActSpecialised<T, TO>) {
            VB_TRY_EPI(VB_ACT_NONE, 0);
            VB_TRY_EPI(VB_ACT_GELU_SAVE_GRAD, 0);
            VB_TRY_EPI(VB_ACT_MUL_AUX, EPI_COLSUM);
            VB_TRY_EPI(VB_ACT_NONE, EPI_ADD);
        } else {
            VB_TRY_EPI(VB_ACT_NONE, EPI_RAGGED);
        }
#undef VB_TRY_EPI
        return launch_8ph_act<T, TO, -1, EPI_ALL>(g, grid, block, SM, stream);
    }
    return VB_ERR_UNSUPPORTED;
}

// =================================================================================================
// 256x128 tile, 4 waves as 2 (M) x 2 (N), TWO WORKGROUPS PER COMPUTE UNIT (nt_kernel 90).
//
// Why: the persistent 256x256 kernel above holds a CU alone (136 KB of LDS), so nothing overlaps its per-tile epilogue
// -- LDS transpose, bias / GELU arithmetic, a 128 KB store burst: ~9 us of a ~26 us tile at K = 768, ~20 us with the GELU
// + GELU' epilogue of the FFN-in GEMM (more than the tile's 17 us of MFMA work).  The matrix pipe and the VALU / store
// path are different resources: here every SIMD hosts one wave of EACH of two independent workgroups, so one workgroup's
// epilogue (VALU, LDS, stores) and its prologue's HBM latency run under the other workgroup's MFMA phases, and the two
// drift apart by themselves because nothing synchronises them.  Per wave the work is unchanged: a 128x64 output block,
// 128 accumulator registers, the same fragment reads per MFMA as the 256x256 kernel.
//
// LDS: exactly half a CU, 80 KB = a ring of FIVE 16-KB half-tiles; a K tile is three of them (A0, B, A1; A_mh holds tile
// rows (r>>6)*128 + mh*64 + (r&63) like the kernel above, B holds the tile's 128 rows of W).  Half-tile index
// h = 3k + {0: A0, 1: B, 2: A1} lives in slot h % 5, so the K loop is unrolled five times and every LDS address is a
// constant.  Per K tile k, two phases, one barrier each:
//   E(k): wait until only the newest 2 half-tiles are in flight | barrier | issue B(k+1) | read A0 B | 32 MFMAs (0,0) (0,1)
//   O(k): wait (same count)                                      | barrier | issue A1(k+1) A0(k+2) | read A1 | 32 MFMAs (1,1) (1,0)
// A half-tile is needed >= one K tile after it was issued (A0(k): O(k-2), B(k): E(k-1), A1(k): O(k-1)); the barrier
// that opens a phase publishes everyone's landed copies AND proves every wave has finished the previous phase's reads,
// whose slots the phase then refills (E(k): slot of A1(k-1); O(k): slots of A0(k) and B(k)).  No wave-group stagger: the
// co-resident workgroup is the other half of the pipeline.  The epilogue is the wave-private one of the kernel above; its
// slabs alias the (by then idle) ring.
// =================================================================================================
// VAR: bit 0 = fragment reads before the phase's copies are issued, bit 1 = s_setprio 1 around the MFMA block.  Both on
// (VAR 3, nt_kernel 90) is what ships; VAR 0 (nt_kernel 91) is kept for A/B.  Measured and dropped
// (profiles/r02_gemm_dual_notes.txt): copies issued one by one between the MFMAs; a half-tile start offset for one of
// the two workgroups that open a CU; v_mfma_f32_32x32x16_bf16 instead of 16x16x32 (same fragment reads, 7 % slower).
// X3: split-operand mode (see GemmArgs): the K loop runs over 3 K / 64 virtual tiles whose column offsets come from
// x3_col_a / x3_col_b (scalar arithmetic per copy batch); the epilogue's T operands are fp32.
template <typename TO, int ACT, int OPT, int VAR, bool X3 = false>
VB_KERNEL VB_LAUNCH_BOUNDS2(256, 2) gemm_nt_dual_kernel(GemmArgs g) {
    typedef bf16 T;
    typedef typename std::conditional<X3, float, bf16>::type TE;
    constexpr int BK = 64, KSTEPS = 2, HALF = 128 * 128;
    constexpr bool DIRECT = kDirectEpilogue<ACT, OPT, (VAR & 4) != 0>;    // VAR bit 2 (developer library, nt_kernel 92): the register-direct epilogue (experiment arm)
    VB_DYN_SMEM(smem);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = vb_uniform(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int ntiles = g.tiles_m * g.tiles_n;
    // tile walk: column stripes of g.stripe tiles, row panels inside a stripe, columns inside a row panel.  An XCD (a
    // contiguous run of the walk) keeps one stripe of B in its L2 while it streams row panels of A past it.
    const int tile = xcd_remap((int)blockIdx.x, ntiles);
    const int per = g.tiles_m * g.stripe;
    const int sidx = tile / per, srem = tile - sidx * per;
    const int sleft = g.tiles_n - sidx * g.stripe;
    const int swid = g.stripe < sleft ? g.stripe : sleft;
    const int m0 = (srem / swid) * 256, n0 = (sidx * g.stripe + srem % swid) * 128;
    const vb_buf A = vb_make_buf(g.A);
    const vb_buf B = vb_make_buf(g.B);

    f32x4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // LDS-direct copies: a half-tile is 16 one-KiB instructions, 4 per wave; instruction i of wave w fills half-tile rows
    // (4 w + i) 8 + lane/8, chunk slot lane % 8 <- global chunk (lane % 8) ^ swz(row).  Per-lane BYTE offsets from the
    // operand bases (32 bits, checked by the launcher) go in a VGPR, the K tile's offset in an SGPR, the base in a buffer
    // descriptor: issuing a copy is one s_mov m0 + one buffer_load ... lds, no vector arithmetic.
    unsigned offA[2][4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const unsigned csrc = (unsigned)(((lane & 7) ^ swz(r)) * 16);
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            int a = m0 + (r >> 6) * 128 + mh * 64 + (r & 63);
            a = a < g.M ? a : g.M - 1;                          // clamped rows are computed but never stored
            offA[mh][i] = (unsigned)(a * (int)g.lda) * 2u + csrc;
        }
        int b = n0 + r;
        b = b < g.N ? b : g.N - 1;
        offB[i] = (unsigned)(b * (int)g.ldb) * 2u + csrc;
    }
    auto issue_at = [&](const vb_buf& buf, const unsigned (&off)[4], unsigned koff, int slot) {
        unsigned char* dst = smem + slot * HALF + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) vb_glds16_buf(buf, off[i], koff, dst + i * 1024);
    };
    // is_b: the copy belongs to operand B (the K tile -> column map of the split-operand mode differs per operand)
    auto issue_a = [&](const unsigned (&off)[4], int kt, int slot) {
#ifdef VB_DEV_KNOBS
        if (g.debug & (1 << 25)) {                             // developer experiment: the A (activation) copies marked streaming
            unsigned char* dst = smem + slot * HALF + wave * 4096;
            const unsigned koff = X3 ? (unsigned)x3_col_a(g, kt, BK) * 2u : (unsigned)kt * (BK * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) vb_glds16_buf_nt(A, off[i], koff, dst + i * 1024);
            return;
        }
#endif
        issue_at(A, off, X3 ? (unsigned)x3_col_a(g, kt, BK) * 2u : (unsigned)kt * (BK * 2), slot);
    };
    auto issue_b = [&](int kt, int slot) {
        issue_at(B, offB, X3 ? (unsigned)x3_col_b(g, kt, BK) * 2u : (unsigned)kt * (BK * 2), slot);
    };

    bf16x8 fa[4][KSTEPS], fb0[2][KSTEPS], fb1[2][KSTEPS];
    auto readA = [&](const unsigned char* half) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) fa[f][ks] = load_frag(half, wr * 64 + f * 16 + li, ks, lg, T());
    };
    auto readB = [&](bf16x8 (&fb)[2][KSTEPS], const unsigned char* half, int nh) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) fb[f][ks] = load_frag(half, wc * 64 + nh * 32 + f * 16 + li, ks, lg, T());
    };
    auto quad = [&](int mh, int nh, bf16x8 (&fb)[2][KSTEPS]) {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if constexpr (DIRECT) acc[mh * 4 + f][nh * 2 + q] = vb_mma(fb[q][ks], fa[f][ks], acc[mh * 4 + f][nh * 2 + q]);    // swapped roles
                    else acc[mh * 4 + f][nh * 2 + q] = vb_mma(fa[f][ks], fb[q][ks], acc[mh * 4 + f][nh * 2 + q]);
                }
    };
    auto prio = [&](int p) {
        if constexpr ((VAR & 2) != 0) { if (p) vb_setprio<1>(); else vb_setprio<0>(); }
        (void)p;
    };

    const int nk = X3 ? 3 * g.kseg : g.K / BK;
    // prologue: h = 0 .. 3 (A0(0) B(0) A1(0) A0(1)); E(0) adds h = 4
    issue_a(offA[0], 0, 0); issue_b(0, 1); issue_a(offA[1], 0, 2);
    if (nk > 1) issue_a(offA[0], 1, 3);

    // one K tile; J = k % 5 fixes the ring slots: A0 -> (3J) % 5, B -> (3J+1) % 5, A1 -> (3J+2) % 5
    auto step = [&](auto jtag, int k) {
        constexpr int J = decltype(jtag)::value;
        constexpr int SA0 = (3 * J) % 5, SB = (3 * J + 1) % 5, SA1 = (3 * J + 2) % 5;
        constexpr int SE = (3 * J + 4) % 5;                    // E refills the slot of A1(k-1)
        const bool last = (k + 1 == nk);
        // ---- E(k): A0(k), B(k) must have landed; A1(k) and A0(k+1) may still be in flight
        if (last) vb_wait_vmcnt<4>(); else vb_wait_vmcnt<8>();
        vb_phase_barrier();
        if constexpr ((VAR & 1) == 0) { if (!last) issue_b(k + 1, SE); }
        readB(fb0, smem + SB * HALF, 0);
        readA(smem + SA0 * HALF);
        readB(fb1, smem + SB * HALF, 1);
        if constexpr ((VAR & 1) != 0) { vb_sched_fence(); if (!last) issue_b(k + 1, SE); vb_sched_fence(); }
        prio(1);
        quad(0, 0, fb0);
        quad(0, 1, fb1);
        prio(0);
        // ---- O(k): A1(k) must have landed; A0(k+1) and B(k+1) may still be in flight
        if (last) vb_wait_vmcnt<0>(); else vb_wait_vmcnt<8>();
        vb_phase_barrier();
        if constexpr ((VAR & 1) != 0) { readA(smem + SA1 * HALF); vb_sched_fence(); }
        if (!last) issue_a(offA[1], k + 1, SA0);               // A1(k+1) into the slot A0(k) just left
        if (k + 2 < nk) issue_a(offA[0], k + 2, SB);           // A0(k+2) into the slot B(k) just left
        if constexpr ((VAR & 1) == 0) readA(smem + SA1 * HALF);
        else vb_sched_fence();
        prio(1);
        quad(1, 1, fb1);
        quad(1, 0, fb0);
        prio(0);
    };
    for (int k = 0;;) {
        step(std::integral_constant<int, 0>(), k); if (++k == nk) break;
        step(std::integral_constant<int, 1>(), k); if (++k == nk) break;
        step(std::integral_constant<int, 2>(), k); if (++k == nk) break;
        step(std::integral_constant<int, 3>(), k); if (++k == nk) break;
        step(std::integral_constant<int, 4>(), k); if (++k == nk) break;
    }
    if constexpr (DIRECT) {
        gemm_epilogue_direct<TE, TO, ACT, OPT>(acc, g, m0 + wr * 128, n0 + wc * 64, lane);     // no LDS: no barrier, a wave leaves when it is done
    } else {
        vb_phase_barrier();                                    // every wave is done with the ring: the slabs may alias it
        // FLUSH (see gemm_epilogue_private): measured per epilogue in the step (profiles/r05_gemm_epilogue_waits.txt) -- on for the plain
        // and the GELU + GELU' epilogues (-1 ... -2 %), off where un-paced store bursts cost the co-resident workgroup more than the waits
        // cost this one: the fp32 logits (+12 % with it), x GELU' + column sums (+2 %); no difference in the split-operand mode
        constexpr bool FLUSH = !X3 && sizeof(TO) == 2 && (ACT == VB_ACT_NONE || ACT == VB_ACT_GELU_SAVE_GRAD);
        gemm_epilogue_private<TE, TO, ACT, OPT, FLUSH>(acc, smem + wave * EPI8_BYTES_PER_WAVE, g, m0 + wr * 128, n0 + wc * 64, lane);
    }
}

template <typename TO, int ACT, int OPT, int VAR, bool X3 = false>
int launch_dual_var(const GemmArgs& g, dim3 grid, hipStream_t stream) {
    constexpr int SM = 5 * 128 * 128;                           // 80 KB: two workgroups per compute unit
    return vb_prof_launch(2.0 * g.M * g.N * g.K, (sizeof(TO) == 4 ? 4 : 0) | 64 | (X3 ? 256 : 0), stream, [&]() { VB_LAUNCH((gemm_nt_dual_kernel<TO, ACT, OPT, VAR, X3>), grid, dim3(256), SM, stream, g); });
}
template <typename TO, int ACT, int OPT>
int launch_dual_act(const GemmArgs& g, dim3 grid, hipStream_t stream) {
#ifdef VB_DEV_KNOBS
    if (t_opts.nt_kernel == 91) return launch_dual_var<TO, ACT, OPT, 0>(g, grid, stream);
    if (t_opts.nt_kernel == 92) return launch_dual_var<TO, ACT, OPT, 7>(g, grid, stream);      // register-direct epilogue (experiment arm)
#endif
    return launch_dual_var<TO, ACT, OPT, 3>(g, grid, stream);
}
template <typename T, typename TO>
int launch_dual(GemmArgs g, hipStream_t stream) {
    // bf16 operands, 32-bit byte offsets inside the kernel (rows x pitch + K below 2^31 elements = 2^32 bytes)
    if (sizeof(T) != 2 || (long)g.M * g.lda >= (1L << 30) || (long)g.N * g.ldb >= (1L << 30))
        return launch_pipe<T, TO, 4, 2>(g, stream);
    if constexpr (sizeof(T) == 2) {
        g.tiles_m = (g.M + 255) / 256;
        g.tiles_n = (g.N + 127) / 128;
        // tile walk (see the kernel): one stripe = row-major.  Measured at M = 83,968 (profiles/r02_gemm_raster.txt): thirds of
        // the columns help the 18-column QKV projection (312 -> 299 us: a 1.2 MB stripe of B stays in the XCD's L2); every
        // other shape of the step is within noise or slower with stripes (the decoder +5 %: A re-read per stripe costs more
        // than B re-read per row panel, which the memory-side cache serves)
        g.stripe = (g.tiles_n > 12 && g.tiles_n <= 20) ? (g.tiles_n + 2) / 3 : g.tiles_n;
        if (((g.debug >> 8) & 0xFFF) > 0) g.stripe = ((g.debug >> 8) & 0xFFF) < g.tiles_n ? ((g.debug >> 8) & 0xFFF) : g.tiles_n;      // developer library only: walk override
        dim3 grid((unsigned)(g.tiles_m * g.tiles_n));
        if constexpr (sizeof(TO) == 4) {
            if (g.x3) {
                // split-operand mode: the epilogue's row operands are fp32.  The four epilogues an encoder layer uses get their own
                // instantiations like the bf16 ones below (one activation, no run-time option code: the run-time body is ~4x the
                // instructions and every tile fetches it); anything else takes the run-time epilogue
                const int needs = epi_needs(g, 4, 4);
                if (!(needs & EPI_RAGGED)) {
                    if (!g.split_out) {
                        if (g.act == VB_ACT_NONE && needs == 0) return launch_dual_var<TO, VB_ACT_NONE, 0, 3, true>(g, grid, stream);
                        if (g.act == VB_ACT_NONE && needs == EPI_ADD) return launch_dual_var<TO, VB_ACT_NONE, EPI_ADD, 3, true>(g, grid, stream);
                    } else {
                        if (g.act == VB_ACT_GELU_SAVE_GRAD && needs == 0)
                            return launch_dual_var<TO, VB_ACT_GELU_SAVE_GRAD, EPI_SPLIT, 3, true>(g, grid, stream);
                        if (g.act == VB_ACT_MUL_AUX && (needs & ~EPI_COLSUM) == 0)
                            return launch_dual_var<TO, VB_ACT_MUL_AUX, EPI_COLSUM | EPI_SPLIT, 3, true>(g, grid, stream);
                    }
                }
                return launch_dual_var<TO, -1, EPI_ALL, 3, true>(g, grid, stream);
            }
        }
        const int needs = epi_needs(g, sizeof(T), sizeof(TO));
        if (needs & EPI_DROP) {
#ifdef VB_DEV_KNOBS
            if constexpr (kActSpecialised<T, TO>) {
                if (g.act == VB_ACT_NONE && g.addend && needs == (EPI_ADD | EPI_DROP) && g.ldc == g.N)
                    return launch_dual_act<TO, VB_ACT_NONE, EPI_ADD | EPI_DROP>(g, grid, stream);
            }
#endif
            return VB_ERR_UNSUPPORTED;
        }
#define VB_TRY_EPI(A, O) if (g.act == (A) && (needs & ~(O)) == 0) return launch_dual_act<TO, A, O>(g, grid, stream)
        if constexpr (kActSpecialised<T, TO>) {
            VB_TRY_EPI(VB_ACT_NONE, 0);
            VB_TRY_EPI(VB_ACT_GELU_SAVE_GRAD, 0);
            VB_TRY_EPI(VB_ACT_MUL_AUX, EPI_COLSUM);
            VB_TRY_EPI(VB_ACT_NONE, EPI_ADD);
        } else {
            VB_TRY_EPI(VB_ACT_NONE, EPI_RAGGED);
        }
#undef VB_TRY_EPI
        return launch_dual_act<TO, -1, EPI_ALL>(g, grid, stream);
    }
    return VB_ERR_UNSUPPORTED;
}

// =================================================================================================
// 256x256 tile, FOUR waves as 2 (M) x 2 (N) -- ONE WAVE PER SIMD, 128x128 outputs per wave (nt_kernel 100).
//
// Why: per MFMA the 8-wave kernels read (128 + 64) fragment rows from LDS per 128x64 block; a 128x128 block reads
// (128 + 128) for twice the MFMAs -- a third less LDS traffic per FLOP on a chip whose GEMM clock is power-limited -- and a wave
// that owns its SIMD needs no partner to hide its fragment reads: it software-pipelines them itself.  The 256 accumulator
// registers live in the AGPR half of the unified file; the fragments of one MFMA K step (8 A + 8 B = 64 VGPRs) are double
// buffered, so the 16 ds_read_b128 of K step s+1 and the 16 LDS-direct copies of K tile k+2 are issued BETWEEN the 64 MFMAs
// of K step s (LLVM sched_group_barrier hints fix the interleave).  One workgroup barrier per K tile (at its middle: it
// publishes the landed copies of K tile k+1 and proves everyone is done reading K tile k's stage, which the copies of k+2
// then overwrite), against eight in the persistent 8-wave kernel.  LDS: two 64-KB stages (A0 | A1 | B0 | B1 half-tiles of 128
// rows x 128 B, the usual chunk swizzle) + the waves' epilogue slabs.  Persistent; the copy stream runs across tile
// boundaries, so a tile's epilogue has the next tile's first two K tiles landing underneath it.
// The K loop's MFMAs and fragment reads are INLINE ASM: hipcc 7.2 does not keep 256 loop-carried accumulators in place in the
// AGPR half of a 512-register wave (it copied every accumulator tuple in front of its MFMA, parked fragments in AGPRs and
// spilled 540 bytes); "+a" constraints pin them, and the statements' source order IS the instruction interleave (4 MFMAs,
// one ds_read_b128, ...).  The compiler therefore knows nothing about these reads' lgkmcnt: the waits are explicit.

template <typename TO, int ACT, int OPT, int ABL = 0>     // ABL (developer library): 1 = no copies, 2 = no fragment reads, 4 = no MFMAs
VB_KERNEL VB_LAUNCH_BOUNDS2(256, 1) gemm_nt_big_kernel(GemmArgs g) {
    typedef bf16 T;
    constexpr int BK = 64, HALF = 128 * 128, STAGE = 4 * HALF;
    VB_DYN_SMEM(smem);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = vb_uniform(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int ntiles = g.tiles_m * g.tiles_n;
    const int G = (int)gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;      // persistent: tiles blockIdx.x + G j
    const int nk = g.K / BK;
    const vb_buf A = vb_make_buf(g.A);
    const vb_buf B = vb_make_buf(g.B);
    unsigned char* slab = smem + 2 * STAGE + wave * EPI8_BYTES_PER_WAVE;

    f32x4 accL[8][4], accR[8][4];                  // columns 0..63 | 64..127 of the wave's block (the epilogue's unit)
    auto zero_acc = [&]() {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) { accL[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f}; accR[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        big_settle();
    };

    auto origin = [&](int j, int& m0, int& n0) {
        const int tile = xcd_remap((int)blockIdx.x + G * j, ntiles);
        m0 = (tile / g.tiles_n) * 256; n0 = (tile % g.tiles_n) * 256;
    };
    // copy stream: a half-tile is 16 one-KiB pieces, 4 per wave; piece i of wave w fills half-tile rows (4 w + i) 8 + lane/8,
    // chunk slot lane % 8 <- global chunk (lane % 8) ^ swz(row); buffer form: descriptor + lane offset VGPR + K offset SGPR
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::true_type Yes;
    typedef std::false_type No;
    unsigned offA[2][4], offB[2][4];
    int ld_j = 0, ld_t = 0;
    auto set_load_tile = [&](int j) {
        int m0, n0;
        origin(j, m0, n0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (wave * 4 + i) * 8 + (lane >> 3);
            const unsigned csrc = (unsigned)(((lane & 7) ^ swz(r)) * 16);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int a = m0 + h * 128 + r, b = n0 + h * 128 + r;
                a = a < g.M ? a : g.M - 1;                  // clamped rows are computed but never stored
                b = b < g.N ? b : g.N - 1;
                offA[h][i] = (unsigned)(a * (int)g.lda) * 2u + csrc;
                offB[h][i] = (unsigned)(b * (int)g.ldb) * 2u + csrc;
            }
        }
    };
    // copy c (0..15) of the load stream's K tile into stage S: c = 8 (operand) + 4 h + i
    // (Measured and dropped: B through registers -- plain buffer loads half a K tile ahead, ds_write_b128 in the next K tile's
    //  first half -- to relieve the LDS-direct path: the copy stream alone went 284 -> 335 us on the FFN-out shape.)
    auto copy_piece = [&](auto stag, auto ctag) {
        constexpr int S = decltype(stag)::value, c = decltype(ctag)::value, h = (c >> 2) & 1, i = c & 3;
        unsigned char* dst = smem + S * STAGE + wave * 4096 + i * 1024;
        const unsigned koff = (unsigned)ld_t * (BK * 2);
        if constexpr (c < 8) vb_glds16_buf(A, offA[h][i], koff, dst + h * HALF);
        else vb_glds16_buf(B, offB[h][i], koff, dst + (2 + h) * HALF);
    };
    // past the workgroup's last K tile the stream stays on it (the unconditional copies of the last two K tiles of the loop
    // below re-fetch it into a stage nobody reads again: two K tiles of traffic per workgroup buy a branch-free K loop)
    auto ld_advance = [&]() {
        if (ld_t + 1 < nk) ++ld_t;
        else if (ld_j + 1 < my_tiles) { ld_t = 0; ++ld_j; set_load_tile(ld_j); }
    };

    // fragment addresses: row f 16 + li of a half-tile, chunk (4 ks + lg) ^ swz(row), and swz(f 16 + li) = ((li >> 1) ^ f) & 7:
    // address = half + f 2048 + li 128 + ((c0 ^ f ^ 4 ks) << 4) with c0 = lg ^ (li >> 1) -- eight lane-dependent bases V[k],
    // k = f ^ 4 ks, per operand and stage; everything else is an immediate
    unsigned VA[2][8], VB_[2][8];
    {
        const unsigned base = big_lds_base(smem);
        const int c0 = lg ^ (li >> 1);
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned v = (unsigned)(li * 128 + ((c0 ^ k) << 4));
                VA[sidx][k] = base + sidx * STAGE + wr * HALF + v;
                VB_[sidx][k] = base + sidx * STAGE + (2 + wc) * HALF + v;
            }
    }
    bf16x8 fa0[8], fb0[8], fa1[8], fb1[8];         // the fragments of MFMA K step 0 / 1 of a K tile
    // read r (0..15) of K step KS from stage S: r < 8 -> A fragment r, else B fragment r - 8
    auto frag_read = [&](auto stag, auto kstag, auto rtag, bf16x8 (&fa)[8], bf16x8 (&fb)[8]) {
        constexpr int S = decltype(stag)::value, KS = decltype(kstag)::value, r = decltype(rtag)::value, f = r & 7;
        if constexpr (r < 8) big_read<f * 2048>(fa[f], smem, VA[S][f ^ (4 * KS)]);
        else big_read<f * 2048>(fb[f], smem, VB_[S][f ^ (4 * KS)]);
    };
    // one half of a K tile: the 64 MFMAs of a K step from (fa, fb); between them the 16 reads of the NEXT K step into (na, nb)
    // from stage RS / K step RKS and, if COPY, the 16 copies of K tile +2 into stage CS.  No branch anywhere near the
    // accumulators: every alternative path through them made the register allocator shuffle all 256 between AGPRs and VGPRs.
    constexpr bool no_copy = (ABL & 1) != 0, no_read = (ABL & 2) != 0, no_mma = (ABL & 4) != 0;   // timing-only builds
    // COPY (second half): the 16 copies of the K tile after next into stage CS, one after every fourth MFMA
    auto half = [&](auto rstag, auto rkstag, auto cstag, auto copytag, bf16x8 (&fa)[8], bf16x8 (&fb)[8],
                    bf16x8 (&na)[8], bf16x8 (&nb)[8]) {
        constexpr bool COPY = decltype(copytag)::value;
        vb_static_for<0, 8>([&](auto mitag) {
            constexpr int mi = decltype(mitag)::value;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) { if constexpr (!no_mma) VB_BIG_MMA(accL[mi][ni], fa[mi], fb[ni]); }
            if constexpr (!no_read) frag_read(rstag, rkstag, std::integral_constant<int, 2 * mi>(), na, nb);
            if constexpr (COPY && !no_copy) copy_piece(cstag, std::integral_constant<int, 2 * mi>());
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) { if constexpr (!no_mma) VB_BIG_MMA(accR[mi][ni], fa[mi], fb[4 + ni]); }
            if constexpr (!no_read) frag_read(rstag, rkstag, std::integral_constant<int, 2 * mi + 1>(), na, nb);
            if constexpr (COPY && !no_copy) copy_piece(cstag, std::integral_constant<int, 2 * mi + 1>());
        });
    };
    set_load_tile(0);
    vb_static_for<0, 16>([&](auto c) { copy_piece(I0(), c); });
    ld_advance();
    vb_static_for<0, 16>([&](auto c) { copy_piece(I1(), c); });
    ld_advance();
    vb_wait_vmcnt<16>();
    vb_raw_barrier();
    vb_static_for<0, 16>([&](auto r) { frag_read(I0(), I0(), r, fa0, fb0); });
    vb_raw_barrier();                               // lgkmcnt(0): the first fragments are in

    // one K tile held in stage S (compile-time)
    auto ktile = [&](auto stag) {
        constexpr int S = decltype(stag)::value;
        typedef std::integral_constant<int, S ^ 1> SN;
        half(stag, I1(), stag, No(), fa0, fb0, fa1, fb1);                 // K step 0 | reads of (this K tile, K step 1)
        if constexpr ((ABL & 8) != 0) vb_wait_vmcnt<16>();      // timing-only: no drain (racy)
        else vb_wait_vmcnt<0>();                    // my copies of the next K tile (issued a K tile ago) have landed
        vb_raw_barrier();                           // lgkmcnt(0) | everyone is done reading this stage, everyone's copies are in
        half(SN(), I0(), stag, Yes(), fa1, fb1, fa0, fb0);                // K step 1 | reads of (next K tile, K step 0) | copies of K tile +2
        ld_advance();
        vb_wait_lgkmcnt0();                         // the next K tile's first fragments
    };
    for (int cj = 0; cj < my_tiles; ++cj) {         // nk is even (launcher): every tile starts in stage 0
        zero_acc();
        for (int kt = 0; kt < nk; kt += 2) { ktile(I0()); ktile(I1()); }
        big_settle();
        int m0, n0;
        origin(cj, m0, n0);
        gemm_epilogue_private<T, TO, ACT, OPT>(accL, slab, g, m0 + wr * 128, n0 + wc * 128, lane);
        gemm_epilogue_private<T, TO, ACT, OPT>(accR, slab, g, m0 + wr * 128, n0 + wc * 128 + 64, lane);
    }
    vb_wait_vmcnt<0>();                             // the re-fetched tail copies must not outlive the workgroup's LDS
}

// =================================================================================================
// nt_kernel 101: the four-wave 256x256 kernel above with the B operand fetched STRAIGHT from global memory into MFMA-layout
// registers (no LDS for B).
//
// Why (the byte budget): with 128x64 outputs per wave the 8-wave kernels book the LDS port at 98-109 % of
// its 128 B/clk at the full MFMA rate; the four-wave kernel (128x128 per wave) needs 94 B/clk but feeds BOTH operands through
// LDS-direct copies, whose queue holds ~4 KB per wave: 4 waves deliver 49 GB/s per CU where 58 are needed.  Here the waves
// are laid out 1 x 4 (a wave owns all 256 rows x 64 columns, so its B fragments are nobody else's) and a lane loads
// the 16 bytes B[n0 + 64 w + 16 f + (lane & 15)][k .. k + 7], k = 32 ks + 8 (lane >> 4) -- exactly its B fragment of MFMA
// K step ks -- with an ordinary buffer load one K tile ahead (two register sets: 128 VGPRs next to the 256 accumulators in
// the AGPR half), so B travels through the deep ordinary load queue, the LDS-direct queue carries A only
// (8 pieces per wave and K tile instead of 16) and the LDS port sees 160 KB per K tile instead of 192 (78 B/clk).  (With the
// 2 x 2 wave layout of the kernel above every B byte was requested by two waves: 30 % slower than that kernel, measured.)
// Same persistent tile walk, same epilogue, one barrier per K tile.  N must be a multiple of 256 (no per-fragment row clamp).
// With B out of LDS there is room for FOUR A stages (128 KB): the A copies run three K tiles ahead of the MFMAs.
// vmcnt: the loads of a wave retire in issue order; the first half of a K tile issues the 8 B loads of the next K tile, the
// second half the 8 A copies of K tile + 3, so "all but the newest 24" at the mid-tile barrier means: the A copies of the NEXT
// K tile (issued two K tiles ago) have landed.  The compiler adds its own (weaker) counted waits in
// front of the MFMAs that consume a loaded register.
template <typename TO, int ACT, int OPT>
VB_KERNEL VB_LAUNCH_BOUNDS2(256, 1) gemm_nt_bdir_kernel(GemmArgs g) {
    typedef bf16 T;
    constexpr int BK = 64, HALF = 128 * 128, STAGE = 2 * HALF;          // a stage holds A only: A0 | A1
    VB_DYN_SMEM(smem);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = vb_uniform(t >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int ntiles = g.tiles_m * g.tiles_n;
    const int G = (int)gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int nk = g.K / BK;
    const vb_buf A = vb_make_buf(g.A);
    const vb_buf B = vb_make_buf(g.B);
    constexpr int NSTG = 4;                                              // A stages: copies run three K tiles ahead of the MFMAs
    unsigned char* slab = smem + NSTG * STAGE + wave * EPI8_BYTES_PER_WAVE;

    // wave w owns ALL 256 rows x columns 64 w .. 64 w + 63 of the tile: its B fragments (4 per K step) are nobody else's -- fetched
    // straight from global memory, no duplicate requests -- and A (shared by the four waves) goes through LDS
    f32x4 acc[2][8][4];                             // [rows 0..127 | 128..255][fragment row][fragment column]
    auto zero_acc = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[h][mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        big_settle();
    };
    auto origin = [&](int j, int& m0, int& n0) {
        const int tile = xcd_remap((int)blockIdx.x + G * j, ntiles);
        m0 = (tile / g.tiles_n) * 256; n0 = (tile % g.tiles_n) * 256;
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    typedef std::true_type Yes;
    typedef std::false_type No;
    // ---- A: LDS-direct copy stream (8 one-KiB pieces per wave and K tile)
    unsigned offA[2][4];
    int ld_j = 0, ld_t = 0;
    auto set_load_tile = [&](int j) {
        int m0, n0;
        origin(j, m0, n0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (wave * 4 + i) * 8 + (lane >> 3);
            const unsigned csrc = (unsigned)(((lane & 7) ^ swz(r)) * 16);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int a = m0 + h * 128 + r;
                a = a < g.M ? a : g.M - 1;                  // clamped rows are computed but never stored
                offA[h][i] = (unsigned)(a * (int)g.lda) * 2u + csrc;
            }
        }
    };
    auto copy_piece = [&](auto stag, auto ctag) {            // piece c (0..7) of the load stream's K tile into stage S
        constexpr int S = decltype(stag)::value, c = decltype(ctag)::value, h = (c >> 2) & 1, i = c & 3;
        unsigned char* dst = smem + S * STAGE + h * HALF + wave * 4096 + i * 1024;
        vb_glds16_buf(A, offA[h][i], (unsigned)ld_t * (BK * 2), dst);
    };
    auto ld_advance = [&]() {
        if (ld_t + 1 < nk) ++ld_t;
        else if (ld_j + 1 < my_tiles) { ld_t = 0; ++ld_j; set_load_tile(ld_j); }
    };
    // ---- B: straight into fragment registers, one K tile ahead; both K steps of a fragment row back to back (the two 64-byte
    //      halves of the same 128-byte lines)
    unsigned bvoff = 0;                                      // (n0 + 64 w + li) ldb 2 + 16 lg
    const unsigned brow16 = (unsigned)(16 * (int)g.ldb) * 2u; // bytes from fragment f to fragment f + 1
    int b_j = 0, b_t = 0;
    auto set_b_tile = [&](int j) {
        int m0, n0;
        origin(j, m0, n0);
        bvoff = (unsigned)((n0 + wave * 64 + li) * (int)g.ldb) * 2u + (unsigned)lg * 16u;
    };
    auto b_load = [&](bf16x8& dst, auto ftag, auto kstag) {
        constexpr int f = decltype(ftag)::value, ks = decltype(kstag)::value;
        const u32x4 v = vb_buf_load16(B, bvoff, (unsigned)f * brow16 + (unsigned)(b_t * BK + ks * 32) * 2u);
        dst = *(const bf16x8*)&v;
    };
    auto b_advance = [&]() {                                 // past the workgroup's last K tile the stream stays on it
        if (b_t + 1 < nk) ++b_t;
        else if (b_j + 1 < my_tiles) { b_t = 0; ++b_j; set_b_tile(b_j); }
    };
    unsigned VA[2][8];                              // base of stage PAIR p (stages 2 p, 2 p + 1: the second one is an immediate offset)
    {
        const unsigned base = big_lds_base(smem);
        const int c0 = lg ^ (li >> 1);
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
            for (int k = 0; k < 8; ++k) VA[sidx][k] = base + sidx * 2 * STAGE + (unsigned)(li * 128 + ((c0 ^ k) << 4));
    }
    bf16x8 fa0[16], fa1[16];                        // A fragments (16 fragment rows = 256 rows) of MFMA K step 0 / 1 of a K tile
    bf16x8 fbt[2][2][4];                            // B fragments: [K tile parity][K step][fragment]
    auto a_read = [&](auto stag, auto kstag, auto ftag, bf16x8 (&fa)[16]) {      // fragment row f: half f >> 3, rows (f & 7) 16 ..
        constexpr int S = decltype(stag)::value, KS = decltype(kstag)::value, f = decltype(ftag)::value;
        big_read<(S & 1) * STAGE + (f >> 3) * HALF + (f & 7) * 2048>(fa[f], smem, VA[S >> 1][(f & 7) ^ (4 * KS)]);
    };
    // one half of a K tile: 64 MFMAs from (fa, fb); between them the 16 A reads of the NEXT K step into na (stage RS, K step RKS);
    // first half (BLOAD): the 8 B loads of the NEXT K tile into nb; second half (COPY): the 8 A copies of K tile +2 into stage CS
    auto half = [&](auto rstag, auto rkstag, auto cstag, auto copytag, auto bloadtag, bf16x8 (&fa)[16], bf16x8 (&fb)[4],
                    bf16x8 (&na)[16], bf16x8 (&nb)[2][4]) {
        constexpr bool COPY = decltype(copytag)::value, BLOAD = decltype(bloadtag)::value;
        vb_static_for<0, 16>([&](auto mitag) {
            constexpr int mi = decltype(mitag)::value;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) VB_BIG_MMA(acc[mi >> 3][mi & 7][ni], fa[mi], fb[ni]);
            // the 16 A reads of the next K step go out in the first 8 of the 16 MFMA groups: the last one is 32 MFMAs old when the
            // half ends, so the lgkmcnt(0) behind the half finds nothing outstanding
            if constexpr (mi < 8) { a_read(rstag, rkstag, std::integral_constant<int, 2 * mi>(), na); a_read(rstag, rkstag, std::integral_constant<int, 2 * mi + 1>(), na); }
            if constexpr (COPY && (mi & 1) == 0) copy_piece(cstag, std::integral_constant<int, mi / 2>());
            if constexpr (BLOAD && (mi & 1) == 0) {
                if constexpr ((mi & 2) == 0) b_load(nb[0][mi / 4], std::integral_constant<int, mi / 4>(), I0());
                else b_load(nb[1][mi / 4], std::integral_constant<int, mi / 4>(), I1());
            }
        });
        if constexpr (BLOAD) b_advance();
    };
    set_load_tile(0);
    vb_static_for<0, 8>([&](auto c) { copy_piece(I0(), c); });
    ld_advance();
    vb_static_for<0, 8>([&](auto c) { copy_piece(I1(), c); });
    ld_advance();
    vb_static_for<0, 8>([&](auto c) { copy_piece(I2(), c); });
    ld_advance();
    set_b_tile(0);
    vb_static_for<0, 4>([&](auto f) { b_load(fbt[0][0][decltype(f)::value], f, I0()); b_load(fbt[0][1][decltype(f)::value], f, I1()); });
    b_advance();
    vb_wait_vmcnt<0>();
    vb_raw_barrier();
    vb_static_for<0, 16>([&](auto f) { a_read(I0(), I0(), f, fa0); });
    vb_raw_barrier();                               // lgkmcnt(0): the first A fragments are in

    auto ktile = [&](auto stag) {                   // one K tile: A stage S (of 4), B register set S & 1
        constexpr int S = decltype(stag)::value, P = S & 1;
        typedef std::integral_constant<int, (S + 1) & 3> SN;                        // the next K tile's stage
        typedef std::integral_constant<int, (S + 3) & 3> SC;                        // = the previous K tile's: refilled with K tile + 3
        half(stag, I1(), SC(), No(), Yes(), fa0, fbt[P][0], fa1, fbt[P ^ 1]);       // K step 0 | A reads of K step 1 | B of the next K tile
        vb_wait_vmcnt<24>();                        // the A copies of the NEXT K tile (issued two K tiles ago) have landed
        vb_raw_barrier();                           // lgkmcnt(0) | everyone is done reading the previous stage, everyone's copies are in
        half(SN(), I0(), SC(), Yes(), No(), fa1, fbt[P][1], fa0, fbt[P ^ 1]);       // K step 1 | A reads of (next tile, K step 0) | A copies
        ld_advance();
        vb_wait_lgkmcnt0();
    };
    for (int cj = 0; cj < my_tiles; ++cj) {         // nk % 4 == 0 (launcher): every tile starts in stage 0 / register set 0
        zero_acc();
        for (int kt = 0; kt < nk; kt += 4) { ktile(I0()); ktile(I1()); ktile(I2()); ktile(I3()); }
        big_settle();
        int m0, n0;
        origin(cj, m0, n0);
        gemm_epilogue_private<T, TO, ACT, OPT>(acc[0], slab, g, m0, n0 + wave * 64, lane);
        gemm_epilogue_private<T, TO, ACT, OPT>(acc[1], slab, g, m0 + 128, n0 + wave * 64, lane);
    }
    vb_wait_vmcnt<0>();
}

template <typename TO, int ACT, int OPT>
int launch_bdir_act(const GemmArgs& g, dim3 grid, hipStream_t stream) {
    constexpr int SM = 4 * 2 * 128 * 128 + 4 * EPI8_BYTES_PER_WAVE;          // four A stages of 32 KB + the epilogue slabs
    return vb_prof_launch(2.0 * g.M * g.N * g.K, (sizeof(TO) == 4 ? 4 : 0) | 128, stream,
                          [&]() { VB_LAUNCH((gemm_nt_bdir_kernel<TO, ACT, OPT>), grid, dim3(256), SM, stream, g); });
}

template <typename TO, int ACT, int OPT>
int launch_big_act(const GemmArgs& g, dim3 grid, hipStream_t stream) {
    constexpr int SM = 2 * 4 * 128 * 128 + 4 * EPI8_BYTES_PER_WAVE;
    return vb_prof_launch(2.0 * g.M * g.N * g.K, (sizeof(TO) == 4 ? 4 : 0) | 128, stream, [&]() { VB_LAUNCH((gemm_nt_big_kernel<TO, ACT, OPT>), grid, dim3(256), SM, stream, g); });
}
template <typename T, typename TO>
int launch_big(GemmArgs g, hipStream_t stream, bool b_direct = false) {
    if (sizeof(T) != 2 || (long)g.M * g.lda >= (1L << 30) || (long)g.N * g.ldb >= (1L << 30))
        return launch_pipe<T, TO, 4, 2>(g, stream);
    if ((g.K / 64) % 2 != 0) return launch_dual<T, TO>(g, stream);      // the K loop alternates two stages per trip
    if (b_direct && ((g.N % 256) != 0 || (g.K / 64) % 4 != 0)) return launch_dual<T, TO>(g, stream);   // no B row clamp; 4 A stages per trip
    if constexpr (sizeof(T) == 2) {
        g.tiles_m = (g.M + 255) / 256;
        g.tiles_n = (g.N + 255) / 256;
        const int ntiles = g.tiles_m * g.tiles_n;
        int wgs = t_opts.persistent_workgroups > 0 ? t_opts.persistent_workgroups : vb_num_cus();
        if (wgs >= ntiles) wgs = ntiles;
        else if (wgs >= 8) wgs &= ~7;
        dim3 grid((unsigned)wgs);
        const int needs = epi_needs(g, sizeof(T), sizeof(TO));
#ifdef VB_DEV_KNOBS
        if constexpr (sizeof(TO) == 2) {
            if (g.act == VB_ACT_NONE && needs == 0 && (g.debug & 7)) {
                constexpr int SMB = 2 * 4 * 128 * 128 + 4 * EPI8_BYTES_PER_WAVE;
                switch (g.debug & 7) {
                    case 1: VB_LAUNCH((gemm_nt_big_kernel<TO, 0, 0, 1>), grid, dim3(256), SMB, stream, g); break;
                    case 2: VB_LAUNCH((gemm_nt_big_kernel<TO, 0, 0, 2>), grid, dim3(256), SMB, stream, g); break;
                    case 3: VB_LAUNCH((gemm_nt_big_kernel<TO, 0, 0, 3>), grid, dim3(256), SMB, stream, g); break;
                    case 4: VB_LAUNCH((gemm_nt_big_kernel<TO, 0, 0, 4>), grid, dim3(256), SMB, stream, g); break;
                    case 5: VB_LAUNCH((gemm_nt_big_kernel<TO, 0, 0, 5>), grid, dim3(256), SMB, stream, g); break;
                    case 6: VB_LAUNCH((gemm_nt_big_kernel<TO, 0, 0, 6>), grid, dim3(256), SMB, stream, g); break;
                    default: VB_LAUNCH((gemm_nt_big_kernel<TO, 0, 0, 14>), grid, dim3(256), SMB, stream, g); break;
                }
                return vb_check_launch();
            }
        }
#endif
#define VB_TRY_EPI(A, O) if (g.act == (A) && (needs & ~(O)) == 0) return b_direct ? launch_bdir_act<TO, A, O>(g, grid, stream) : launch_big_act<TO, A, O>(g, grid, stream)
        if constexpr (kActSpecialised<T, TO>) {
            VB_TRY_EPI(VB_ACT_NONE, 0);
            VB_TRY_EPI(VB_ACT_GELU_SAVE_GRAD, 0);
            VB_TRY_EPI(VB_ACT_MUL_AUX, EPI_COLSUM);
            VB_TRY_EPI(VB_ACT_NONE, EPI_ADD);
        } else {
            VB_TRY_EPI(VB_ACT_NONE, EPI_RAGGED);
        }
#undef VB_TRY_EPI
        return b_direct ? launch_bdir_act<TO, -1, EPI_ALL>(g, grid, stream) : launch_big_act<TO, -1, EPI_ALL>(g, grid, stream);
    }
    return VB_ERR_UNSUPPORTED;
}

// =================================================================================================
// Weight gradients: dW_p[out_p, in_p] += alpha * dY_p^T X_p for a GROUP of problems that share the token count
// (the four Linears of an encoder layer), one persistent launch.  Both operands are K-strided ([token][feature]):
// tiles are copied AS STORED (LDS-direct, a [64 token][128 feature] half-tile per copy stream step) and the MFMA
// operand fragments are gathered with ds_read_b64_tr_b16, so nothing is transposed in HBM, registers or LDS.
// Same eight-phase schedule, half-tile ring, counted waits and private epilogue slab as the kernel above.
//
// LDS image of a half-tile (16 KB, rows r = 0..127 of the operand's feature range, k = 0..63 tokens), in 16-byte
// pieces T[k][8 c .. 8 c + 7]:  chunk q = (((r>>6) 2 + (k>>5)) 2 + ((k>>2)&1)) 2 + ((k>>4)&1)   (1 KB = one copy)
//                                piece  i = ((r>>4)&3) 16 + (((k>>3)&1) 4 + (k&3)) 2 + ((r>>3)&1)
// so that (a) one copy instruction fetches 8 token rows x 128 contiguous bytes (full cache lines) and (b) the 32
// lanes of a ds_read_b64_tr_b16 half-wave read 256 CONSECUTIVE bytes: conflict-free by construction.
//
// Work: item = (problem, 256x256 output tile, token slice); workgroup b walks items b, b + G, ... (XCD-aware remap:
// an XCD gets consecutive items = the tiles of one problem and token slice, which share operand panels in its L2).
// Partial tiles are added with fp32 atomics (the output is an accumulator anyway).
// =================================================================================================
constexpr int VB_TN_MAX = 8;
struct TnProblem {
    const void* A; const void* B; float* C;      // dY [tokens][out], X [tokens][in], dW [out][in]
    long lda, ldb, ldc;
    int Mo, Ni;                                   // out, in
    int tiles_n, tiles;                           // 256x256 tiles: columns, total
    int tiles_m;                                  // rows; tiles are walked along the SHORTER dimension first (see decode)
    int item0;                                    // first item of this problem
    int tile0;                                    // first tile of this problem in the group-wide tile numbering
};
struct TnArgs {
    TnProblem p[VB_TN_MAX];
    int nprob, KT, splits, kps, nitems;           // K tiles of 64 tokens; token slices; K tiles per slice; items
    // "helper" layout (rem > 0), for item counts that do not fill the chip -- an encoder layer has 108 tiles, i.e. 216
    // two-slice items for 256 CUs: the slices cover only the first KT - rem K tiles of every tile, and the compute units the
    // items leave idle (nhelp_x per XCD next to nmain_x item-owning workgroups) take the last `rem` K tiles of the tiles
    // h, h + nhelp, ... (all of them in the same token range: shared panels in the XCD's L2).
    int rem, nmain_x, nhelp_x, ntiles;
    float alpha;
    const float* alpha_dev;
};

VB_KERNEL VB_LAUNCH_BOUNDS(512) gemm_tn_8ph_kernel(TnArgs g) {
    typedef bf16 T;
    constexpr int HALF = 128 * 128, BUF = 4 * HALF;
    constexpr int SLOT_A0 = 0, SLOT_A1 = 1, SLOT_B0 = 2, SLOT_B1 = 3;
    VB_DYN_SMEM(smem);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = vb_uniform(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int G = (int)gridDim.x;
    // helper layout: workgroup b sits on XCD b % 8 (observed placement: speed only); slots 0 .. nmain_x - 1 of an XCD own one
    // (tile, slice) item each, the remaining nhelp_x slots are helpers
    const int xslot = (int)blockIdx.x >> 3;
    const bool helper = g.rem > 0 && xslot >= g.nmain_x;
    const int nhelp = 8 * g.nhelp_x;
    const int hidx = ((int)blockIdx.x & 7) * g.nhelp_x + (xslot - g.nmain_x);
    const int my_items = g.rem == 0 ? (g.nitems - (int)blockIdx.x + G - 1) / G
                                    : (helper ? (g.ntiles - hidx + nhelp - 1) / nhelp : 1);
    unsigned char* slab = smem + 2 * BUF + wave * EPI8_BYTES_PER_WAVE;

    f32x4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    struct Item { int p, m0, n0, kt0, nk; };
    auto decode = [&](int j) {
        Item it;
        int p = 0, tl, s = 0;
        if (helper) {
            const int gt = hidx + nhelp * j;                // group-wide tile number
            for (int q = 1; q < g.nprob; ++q) if (gt >= g.p[q].tile0) p = q;
            tl = gt - g.p[p].tile0;
        } else {
            const int v = g.rem == 0 ? xcd_remap((int)blockIdx.x + G * j, g.nitems) : ((int)blockIdx.x & 7) * g.nmain_x + xslot;
            for (int q = 1; q < g.nprob; ++q) if (v >= g.p[q].item0) p = q;
            const int local = v - g.p[p].item0;
            s = local / g.p[p].tiles;
            tl = local - s * g.p[p].tiles;
        }
        it.p = p;
        // consecutive items (= one XCD's concurrent workgroups) cover whole rows / columns of the shorter tile
        // dimension: the fewest distinct operand panels per XCD L2
        const int tm = g.p[p].tiles_m, tn = g.p[p].tiles_n;
        if (tm < tn) { it.m0 = (tl % tm) * 256; it.n0 = (tl / tm) * 256; }
        else { it.m0 = (tl / tn) * 256; it.n0 = (tl % tn) * 256; }
        const int kend = g.KT - g.rem;                      // the slices end here; the helpers take the rest
        if (helper) { it.kt0 = kend; it.nk = g.rem; }
        else {
            it.kt0 = s * g.kps;
            it.nk = kend - it.kt0 < g.kps ? kend - it.kt0 : g.kps;
        }
        return it;
    };

    // ---- copy stream: wave w, instruction i fills chunk q = 2 w + i of a half-tile; lane -> piece (see header)
    const int ck = ((wave >> 1) & 1) * 32 + ((lane >> 3) & 1) * 8 + (wave & 1) * 4 + ((lane >> 1) & 3);   // + 16 i
    const int cr = (wave >> 2) * 64 + (lane >> 4) * 16 + (lane & 1) * 8;
    unsigned offA[2][2], offB[2][2];
    const unsigned char* srcA = nullptr;           // operand base of the load stream's K tile (uniform)
    const unsigned char* srcB = nullptr;
    long stepA = 0, stepB = 0;
    int ld_j = 0, ld_t = 0, ld_nk = 0;
    auto set_load_item = [&](int j) {
        const Item it = decode(j);
        const TnProblem& P = g.p[it.p];
        ld_nk = it.nk;
        stepA = 64 * P.lda * 2; stepB = 64 * P.ldb * 2;
        srcA = (const unsigned char*)P.A + (long)it.kt0 * stepA;
        srcB = (const unsigned char*)P.B + (long)it.kt0 * stepB;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int ca = it.m0 + (cr >> 6) * 128 + h * 64 + (cr & 63);
                int cb = it.n0 + (cr >> 5) * 64 + h * 32 + (cr & 31);
                const int mo8 = (P.Mo + 7) & ~7, ni8 = (P.Ni + 7) & ~7;   // readable up to round_up(rows, 8) per token (ABI)
                ca = ca <= mo8 - 8 ? ca : mo8 - 8;            // clamped pieces land in rows the epilogue masks
                cb = cb <= ni8 - 8 ? cb : ni8 - 8;
                offA[h][i] = (unsigned)(((ck + 16 * i) * (int)P.lda + ca) * 2);
                offB[h][i] = (unsigned)(((ck + 16 * i) * (int)P.ldb + cb) * 2);
            }
    };
    auto issueA = [&](int mh, int par) {
        unsigned char* dst = smem + par * BUF + (mh ? SLOT_A1 : SLOT_A0) * HALF + wave * 2048;
        vb_glds16(srcA + offA[mh][0], dst);
        vb_glds16(srcA + offA[mh][1], dst + 1024);
    };
    auto issueB = [&](int nh, int par) {
        unsigned char* dst = smem + par * BUF + (nh ? SLOT_B1 : SLOT_B0) * HALF + wave * 2048;
        vb_glds16(srcB + offB[nh][0], dst);
        vb_glds16(srcB + offB[nh][1], dst + 1024);
    };
    auto ld_advance = [&]() {
        if (++ld_t == ld_nk) { ld_t = 0; ++ld_j; set_load_item(ld_j); }
        else { srcA += stepA; srcB += stepB; }
    };

    // ---- fragment gathers: lane (s = lane & 15, kg = lane >> 4) of a transposing read
    const int s16 = lane & 15, kg = lane >> 4;
    const int lane_off = (kg >> 1) * 1024 + ((kg & 1) * 4 + (s16 >> 2)) * 32 + ((s16 >> 1) & 1) * 16 + (s16 & 1) * 8;
    const int offa_w = wr * 8192 + lane_off;                                   // F = wr
    const int offb_w = (wc >> 1) * 8192 + (wc & 1) * 512 + lane_off;           // F = wc >> 1, f16l = (wc & 1) 2 + g
    // fragments stay in their two transposed halves (k + 0..3 | k + 4..7) until the MFMA that consumes them: the reads are
    // inline asm (vb_lds_read_tr_pair: no compiler-inserted vmcnt(0) in front of them), so nothing may touch their
    // results before the phase's lgkmcnt(0) (vb_raw_barrier)
    bf16x4 fal[4][2], fah[4][2], fb0l[2][2], fb0h[2][2], fb1l[2][2], fb1h[2][2];
    auto readA = [&](auto slot, const unsigned char* buf) {
        constexpr int BASE = decltype(slot)::value * HALF;
        const unsigned char* p = buf + offa_w;
        vb_static_for<0, 8>([&](auto i) {
            constexpr int f = decltype(i)::value >> 1, ks = decltype(i)::value & 1;
            vb_lds_read_tr_pair<BASE + ks * 4096 + f * 256>(fal[f][ks], fah[f][ks], p);
        });
    };
    auto readB = [&](bf16x4 (&lo)[2][2], bf16x4 (&hi)[2][2], auto slot, const unsigned char* buf) {
        constexpr int BASE = decltype(slot)::value * HALF;
        const unsigned char* p = buf + offb_w;
        vb_static_for<0, 4>([&](auto i) {
            constexpr int f = decltype(i)::value >> 1, ks = decltype(i)::value & 1;
            vb_lds_read_tr_pair<BASE + ks * 4096 + f * 256>(lo[f][ks], hi[f][ks], p);
        });
    };
    auto quad = [&](int mh, int nh, bf16x4 (&bl)[2][2], bf16x4 (&bh)[2][2]) {
        vb_setprio<1>();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    acc[mh * 4 + f][nh * 2 + q] = vb_mma(vb_join(fal[f][ks], fah[f][ks]), vb_join(bl[q][ks], bh[q][ks]),
                                                         acc[mh * 4 + f][nh * 2 + q]);
        vb_setprio<0>();
    };
    // partial tile -> fp32 atomics, a wave covering 64 consecutive columns of one row per instruction
    auto drain = [&](const Item& it) {
        const TnProblem& P = g.p[it.p];
        const float alpha = g.alpha_dev ? g.alpha * g.alpha_dev[0] : g.alpha;
        const int li = lane & 15, lg = lane >> 4;
        const int n = it.n0 + wc * 64 + lane;
        auto fragrow = [&](f32x4 (&a)[4], int mrow0) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *(float*)(slab + (lg * 4 + r) * 256 + (((ni ^ lg) * 16 + li) << 2)) = a[ni][r];
            vb_wave_sync();
            for (int row = 0; row < 16; ++row) {
                const int m = mrow0 + row;
                const float v = *(const float*)(slab + row * 256 + (((((lane >> 4) ^ (row >> 2)) & 3) * 16 + (lane & 15)) << 2));
                if (m < P.Mo && n < P.Ni) vb_atomic_add_noret(P.C + (long)m * P.ldc + n, alpha * v);
            }
            vb_wave_sync();
        };
        const int mw0 = it.m0 + wr * 128;
        fragrow(acc[0], mw0 + 0);  fragrow(acc[1], mw0 + 16); fragrow(acc[2], mw0 + 32); fragrow(acc[3], mw0 + 48);
        fragrow(acc[4], mw0 + 64); fragrow(acc[5], mw0 + 80); fragrow(acc[6], mw0 + 96); fragrow(acc[7], mw0 + 112);
    };

    int GK = 0;
    for (int j = 0; j < my_items; ++j) GK += decode(j).nk;
    set_load_item(0);
    // four-slot schedule (see gemm_nt_8ph_kernel, SCHED = 1): E(g) reads A0 B0 B1, issues A1 of tile g+1, MFMAs (0,0) (0,1);
    // O(g) reads A1, issues A0 B0 B1 of tile g+2, MFMAs (1,1) (1,0); every counted wait is "all but the newest 8"
    issueA(0, 0); issueB(0, 0); issueB(1, 0); issueA(1, 0);
    if (GK > 1) { ld_advance(); issueA(0, 1); issueB(0, 1); issueB(1, 1); vb_wait_vmcnt<6>(); }
    else vb_wait_vmcnt<0>();
    vb_phase_barrier();
    if (wr == 1) vb_phase_barrier();               // waves 4-7 run one barrier behind waves 0-3

    int ct = 0, cj = 0;
    Item cur = decode(0);
    for (int gk = 0; gk < GK; ++gk) {
        const int par = gk & 1;
        const unsigned char* buf = smem + par * BUF;
        const bool n1 = gk + 1 < GK, n2 = gk + 2 < GK;
        // ---- E
        readB(fb0l, fb0h, std::integral_constant<int, SLOT_B0>(), buf);
        readA(std::integral_constant<int, SLOT_A0>(), buf);
        readB(fb1l, fb1h, std::integral_constant<int, SLOT_B1>(), buf);
        if (n1) { issueA(1, par ^ 1); vb_wait_vmcnt<8>(); } else vb_wait_vmcnt<0>();
        vb_raw_barrier();                          // lgkmcnt(0) first: my gathers of A0 B0 B1 are done
        vb_sched_fence();
        quad(0, 0, fb0l, fb0h);
        quad(0, 1, fb1l, fb1h);
        vb_phase_barrier();
        // ---- O
        readA(std::integral_constant<int, SLOT_A1>(), buf);
        if (n2) { ld_advance(); issueA(0, par); issueB(0, par); issueB(1, par); vb_wait_vmcnt<8>(); }
        else if (n1) vb_wait_vmcnt<2>();
        else vb_wait_vmcnt<0>();
        vb_raw_barrier();
        vb_sched_fence();
        quad(1, 1, fb1l, fb1h);
        quad(1, 0, fb0l, fb0h);
        vb_phase_barrier();
        if (++ct == cur.nk) {
            drain(cur);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            ct = 0; ++cj;
            if (cj < my_items) cur = decode(cj);
        }
    }
    if (wr == 0) vb_phase_barrier();               // balance the stagger barrier
}

// token slices: few enough to keep the atomic traffic (items x 256 KB) low, many enough to fill the chip
static int tn_pick_splits(int tiles, int KT, int wgs) {
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 32 && s <= KT; ++s) {
        const int kps = (KT + s - 1) / s;
        const int real = (KT + kps - 1) / kps;
        if (real != s) continue;
        const long items = (long)tiles * s;
        const long rounds = (items + wgs - 1) / wgs;
        const double cost = rounds * (kps + 24.0);           // draining a tile with atomics ~ 24 K tiles' worth of time (measured)
        if (cost < best_cost * 0.97) { best_cost = cost; best = s; }
    }
    return best;
}

// problems must share K (tokens, a multiple of 64); every dimension a multiple of 8; bf16 operands
static int launch_tn_group(TnArgs& g, int K, hipStream_t stream) {
    constexpr int SM = 2 * 4 * 128 * 128 + 8 * EPI8_BYTES_PER_WAVE;
    g.KT = K / 64;
    int tiles = 0;
    for (int i = 0; i < g.nprob; ++i) {
        TnProblem& P = g.p[i];
        P.tiles_n = (P.Ni + 255) / 256;
        P.tiles_m = (P.Mo + 255) / 256;
        P.tiles = P.tiles_m * P.tiles_n;
        tiles += P.tiles;
    }
    int wgs = t_opts.persistent_workgroups > 0 ? t_opts.persistent_workgroups : vb_num_cus();
    g.splits = tn_pick_splits(tiles, g.KT, wgs);
    g.kps = (g.KT + g.splits - 1) / g.splits;
    g.rem = 0; g.nmain_x = g.nhelp_x = 0; g.ntiles = tiles;
    {
        // helper layout (see TnArgs): s slices per tile leave wgs - tiles * s compute units idle for the whole launch (an
        // encoder layer: 108 tiles x 2 = 216 items on 256 CUs).  Give those CUs the tail `rem` of every tile's token range,
        // u tiles each, sized so that a slice + its drain takes as long as a helper's u pieces + u drains.
        const double drain = 24.0;                         // K tiles' worth of time to add a 256x256 tile with atomics (measured)
        const int s = tiles > 0 ? wgs / tiles : 0;
        const int nmain = tiles * s, nhelp = wgs - nmain;
        if (s >= 1 && (wgs % 8) == 0 && (nmain % 8) == 0 && nhelp >= 8) {
            const int u = (tiles + nhelp - 1) / nhelp;
            int rem = (int)(((double)g.KT / s - (u - 1) * drain) / (u + 1.0 / s));
            const int lmain = rem > 0 ? (g.KT - rem + s - 1) / s : 0;
            const double classic = ((tiles * g.splits + wgs - 1) / wgs) * (g.kps + drain);     // rounds x (slice + drain)
            if (rem >= 16 && lmain >= 16 && lmain + drain < 0.97 * classic) {
                g.rem = rem; g.splits = s; g.kps = lmain;
                g.nmain_x = nmain / 8; g.nhelp_x = nhelp / 8;
            }
        }
    }
    int item0 = 0, tile0 = 0;
    for (int i = 0; i < g.nprob; ++i) {
        g.p[i].item0 = item0; item0 += g.p[i].tiles * g.splits;
        g.p[i].tile0 = tile0; tile0 += g.p[i].tiles;
    }
    g.nitems = item0;
    if (g.rem == 0) {
        if (wgs >= g.nitems) wgs = g.nitems;
        else if (wgs >= 8) wgs &= ~7;
    }
    dim3 grid((unsigned)wgs), block(512);
    double flops = 0;
    for (int i = 0; i < g.nprob; ++i) flops += 2.0 * g.p[i].Mo * g.p[i].Ni * K;
    return vb_prof_launch(flops, 4 | 2 | 1 | 16, stream, [&]() { VB_LAUNCH(gemm_tn_8ph_kernel, grid, block, SM, stream, g); });
}
// The last tokens % 64 rows of a grouped weight-gradient call (ragged B x S: per-GPU batch 8 x 164 tokens = 20 K tiles + 32 rows), ALL
// problems in ONE launch: dW_p[o][i] += alpha sum_r dY_p[r][o] X_p[r][i], r < rows <= 63.  One 64 x 64 output tile per workgroup, both
// row panels staged in LDS as fp32, 4 x 4 outputs per thread, plain read-modify-write of dW: every output element belongs to exactly one
// thread of the launch, and the launch runs after the grouped kernel on the same stream.  (fp32 atomics instead: 25 -> 85 us per launch.)
// Problems whose dW ranges OVERLAP (nothing in this repo does that; the ABI does not forbid it) are launched one after the other.  Before round 5's last step these rows went through the generic kernel, one launch
// per problem: 4 x 11.6 us per encoder layer at B = 8 -- 0.59 ms of a 5.9 ms step (profiles/r05_kernel_stats_b8.txt).
struct TnTailArgs {
    TnProblem p[VB_TN_MAX];
    int blk0[VB_TN_MAX + 1];                      // first workgroup of each problem
    int nprob, rows;
    float alpha;
    const float* alpha_dev;
};
VB_KERNEL VB_LAUNCH_BOUNDS(256) gemm_tn_tail_kernel(TnTailArgs g) {
    VB_DYN_SMEM(smem);
    float* sa = (float*)smem;                     // [rows][64] dY columns o0 .. o0 + 63
    float* sb = sa + 64 * 64;                     // [rows][64] X  columns i0 .. i0 + 63
    int pi = 0;
    while (pi + 1 < g.nprob && (int)blockIdx.x >= g.blk0[pi + 1]) ++pi;
    const TnProblem& P = g.p[pi];
    const int tn = (P.Ni + 63) / 64;
    const int tile = (int)blockIdx.x - g.blk0[pi];
    const int o0 = (tile / tn) * 64, i0 = (tile % tn) * 64;
    const int t = threadIdx.x;
    const bf16* A = (const bf16*)P.A;
    const bf16* B = (const bf16*)P.B;
    for (int c = t; c < g.rows * 8; c += 256) {   // 16-byte chunks: the leading dimensions cover the columns rounded up to 8 (tn_eligible)
        const int r = c >> 3, ch = (c & 7) * 8;
        float va[8], vb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { va[j] = 0.f; vb[j] = 0.f; }
        if (o0 + ch < P.Mo) load8(va, A + (long)r * P.lda + o0 + ch);
        if (i0 + ch < P.Ni) load8(vb, B + (long)r * P.ldb + i0 + ch);
#pragma unroll
        for (int j = 0; j < 8; ++j) { sa[r * 64 + ch + j] = va[j]; sb[r * 64 + ch + j] = vb[j]; }
    }
    __syncthreads();
    const int to = (t >> 4) * 4, ti = (t & 15) * 4;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    for (int r = 0; r < g.rows; ++r) {
        const f32x4 x = *(const f32x4*)(sa + r * 64 + to), y = *(const f32x4*)(sb + r * 64 + ti);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] += x[a] * y[b];
    }
    const float alpha = g.alpha_dev ? g.alpha * g.alpha_dev[0] : g.alpha;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int o = o0 + to + a;
        if (o >= P.Mo) continue;
        float* crow = P.C + (long)o * P.ldc + i0 + ti;
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (i0 + ti + b < P.Ni) crow[b] += alpha * acc[a][b];
    }
}
static int launch_tn_tail_range(const TnArgs& main, int first, int count, int row0, int rows, hipStream_t stream) {
    TnTailArgs g;
    g.nprob = count; g.rows = rows; g.alpha = main.alpha; g.alpha_dev = main.alpha_dev;
    int blocks = 0;
    double flops = 0;
    for (int i = 0; i < count; ++i) {
        const TnProblem& src = main.p[first + i];
        g.p[i] = src;
        g.p[i].A = (const bf16*)src.A + (long)row0 * src.lda;
        g.p[i].B = (const bf16*)src.B + (long)row0 * src.ldb;
        g.blk0[i] = blocks;
        blocks += ((src.Mo + 63) / 64) * ((src.Ni + 63) / 64);
        flops += 2.0 * src.Mo * src.Ni * rows;
    }
    g.blk0[count] = blocks;
    return vb_prof_launch(flops, 4 | 2 | 1, stream, [&]() { VB_LAUNCH(gemm_tn_tail_kernel, dim3((unsigned)blocks), dim3(256), 2 * 64 * 64 * 4, stream, g); });
}
static int launch_tn_tail(const TnArgs& main, int row0, int rows, hipStream_t stream) {
    bool overlap = false;                         // two problems writing the same dW elements: no single launch of plain read-modify-writes
    for (int i = 0; i < main.nprob && !overlap; ++i)
        for (int j = i + 1; j < main.nprob && !overlap; ++j) {
            const float *a0 = main.p[i].C, *a1 = a0 + (long)(main.p[i].Mo - 1) * main.p[i].ldc + main.p[i].Ni;
            const float *b0 = main.p[j].C, *b1 = b0 + (long)(main.p[j].Mo - 1) * main.p[j].ldc + main.p[j].Ni;
            overlap = a0 < b1 && b0 < a1;
        }
    if (!overlap) return launch_tn_tail_range(main, 0, main.nprob, row0, rows, stream);
    for (int i = 0; i < main.nprob; ++i) {
        const int rc = launch_tn_tail_range(main, i, 1, row0, rows, stream);
        if (rc != VB_OK) return rc;
    }
    return VB_OK;
}
// =================================================================================================
// Weight gradients at SMALL token counts (per-GPU batch <= ~24 at S = 164: the regime of every reference config at DP = 8,
// configs/vqa/coco-pre-train.json:17 / models/train.py:146).  There the persistent kernel above spends most of its time adding 256x256
// partial tiles with fp32 atomics (a drain costs 24 K tiles' worth of time against 10-20 K tiles of work: 62.7 us per encoder layer at
// B = 8 for a 28 MB dW, plus a 25 us tail launch for the tokens % 64 rows -- profiles/r05_fin3_kernel_stats_b8.txt).  This kernel gives every
// 128 x 128 tile of dW to ONE workgroup that walks the WHOLE token range -- whole 64-token K tiles by LDS-direct copies (the half-tile image
// and the transposing fragment reads of the kernel above), the last tokens % 64 rows through registers with zero fill -- and adds the
// tile to dW with plain 16-byte read-modify-writes: no atomics, no token slices, no tail launch, every dW element touched by exactly
// one thread.  The dW tile is FETCHED AT THE START (64 floats per lane) so its HBM read runs under the K loop.
// 4 waves as 2 x 2 (64 x 64 each); STAGES x 32 KB ring (one A and one B half-tile per stage): 4 stages when the tiles do not outnumber the
// compute units, 2 stages (two workgroups per CU) otherwise.
// =================================================================================================
struct TnSmallArgs {
    TnProblem p[VB_TN_MAX];
    int tile0[VB_TN_MAX + 1];                     // first 128x128 tile of each problem
    int nprob, tokens;                            // the reduction length (the kernel cuts it into whole K tiles + a ragged one)
    float alpha;
    const float* alpha_dev;
};
#ifdef VB_EMU
template <int N> VB_DEVICE void vb_wait_lgkmcnt() {}
#else
template <int N> VB_DEVICE void vb_wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
#endif
// dW block (64 x 64 per wave) += alpha * acc: through a wave-private slab (in the ring, dead by now) so that a lane owns 8 consecutive
// columns; cin = the block's old values, fetched during the K loop.  Every wave of the workgroup must call it (one workgroup barrier).
VB_DEVICE void tn_small_epilogue(f32x4 (&acc)[4][4], f32x4 (&cin)[2][4][2], unsigned char* smem, const TnSmallArgs& g, const TnProblem& P,
                                 int mw0, int ncol, bool col_ok, int wave, int lane) {
    const float alpha = g.alpha_dev ? g.alpha * g.alpha_dev[0] : g.alpha;
    const int li = lane & 15, lg = lane >> 4;
    unsigned char* slab = smem + wave * EPI_BYTES_PER_WAVE;
    __syncthreads();                           // every wave has finished reading the ring
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) vb_wave_sync();
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *(float*)(slab + (mh * 16 + lg * 4 + r) * EPI_PITCH + (ni * 16 + li) * 4) = acc[pass * 2 + mh][ni][r];
        vb_wave_sync();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3);
            const int m = mw0 + pass * 32 + row;
            const unsigned char* src = slab + row * EPI_PITCH + (lane & 7) * 32;
            const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 16);
            if (m < P.Mo && col_ok) {
                float* cp = P.C + (long)m * P.ldc + ncol;
                *(f32x4*)cp = cin[pass][it][0] + alpha * lo;
                *(f32x4*)(cp + 4) = cin[pass][it][1] + alpha * hi;
            }
        }
    }
}
// KB: tokens per K tile.  64 = the persistent kernel's half-tile image as is (16 KB per operand, 32 MFMAs per wave and barrier);
// 32 = the k < 32 half of that image (8 KB per operand: chunk = (r >> 6) 4 + ((k >> 2) & 1) 2 + ((k >> 4) & 1)), so that a FOUR-stage ring is
// 64 KB and two workgroups still share a compute unit.  Built on the guess that the two-stage 64-token ring (one L2 round trip exposed per
// K tile and workgroup) was what held an encoder layer's launch at 37 us; measured (profiles/r06_small_batch_ab.txt, session 10): the same
// 36.9 us, as was fetching the dW tile behind the ring's first tiles instead of ahead of them.  What bounds the launch is the operand
// DELIVERY of a 128x128 tile: 32 KB from L2 per 2.1 MFLOP, 432 tiles x 21 K tiles = 290 MB in the ~21 us the K loops take = 13.8 TB/s, 70 %
// of the 19.8 TB/s this chip delivers L2 -> LDS (profiles/r02_glds_stream.txt), plus ~16 us of prologue / dW read-modify-write that nothing
// overlaps because all 432 workgroups run in lockstep.  KB = 32 is compiled only with -DVB_TN_SMALL_KB32=1 (tools/build_variant.sh).
template <int STAGES, int KB>
VB_KERNEL VB_LAUNCH_BOUNDS2(256, 2) gemm_tn_small_kernel(TnSmallArgs g) {      // two waves per SIMD = two workgroups per CU: at most 256 registers
                                                                              // (without the bound one build came out at 164 + 96 and ran ONE per CU)
    static_assert(KB == 64 || KB == 32, "K tile depth");
    constexpr int HALF = KB * 128 * 2, STAGE_BYTES = 2 * HALF;
    constexpr int CI = KB / 16;                  // copy instructions per wave, operand and K tile (1 KB each)
    constexpr int PER_TILE = 2 * CI;
    constexpr int KS = KB / 32;                  // MFMA K steps per tile
    constexpr int RH = KB * 128;                 // byte offset of tile rows 64 .. 127 in the image (the r >> 6 bit of the chunk index)
    VB_DYN_SMEM(smem);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = vb_uniform(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = g.tile0[g.nprob];
    const int v = xcd_remap((int)blockIdx.x, ntiles);
    int pi = 0;
    for (int q = 1; q < g.nprob; ++q) if (v >= g.tile0[q]) pi = q;
    const TnProblem& P = g.p[pi];
    const int tl = v - g.tile0[pi];
    const int tm = (P.Mo + 127) / 128, tn = (P.Ni + 127) / 128;
    int m0, n0;                                  // consecutive tiles (one XCD's workgroups) share a panel of the longer operand
    if (tm < tn) { m0 = (tl % tm) * 128; n0 = (tl / tm) * 128; }
    else { m0 = (tl / tn) * 128; n0 = (tl % tn) * 128; }

    // ---- the dW tile this wave will add to: rows mw0 + pass 32 + it 8 + lane / 8, columns nw0 + (lane & 7) 8 .. + 7 (fetched now, used last)
    const int mw0 = m0 + wm * 64, nw0 = n0 + wn * 64;
    const int ncol = nw0 + (lane & 7) * 8;
    const bool col_ok = ncol < P.Ni;             // Ni % 8 == 0 (launcher): a lane's 8 columns are inside or outside together
    f32x4 cin[2][4][2];
    constexpr int CIN_LOADS = 16;                // issued BEHIND the ring's first STAGES - 1 tiles (below): a wave has ONE in-order vmcnt

    // ---- copy stream: wave w, instruction i fills chunk CI w + i of a half-tile (image: see gemm_tn_8ph_kernel's header); token row of
    //      instruction i inside the K tile: KB = 64: (w & 1) 32 + (i & 1) 16 + (i >> 1) 4 + lane bits; KB = 32: i 16 + (w & 1) 4 + lane bits
    const int kl = (KB == 64 ? (wave & 1) * 32 : (wave & 1) * 4) + ((lane >> 3) & 1) * 8 + ((lane >> 1) & 3);
    auto krow = [&](int i) { return kl + (KB == 64 ? (i & 1) * 16 + (i >> 1) * 4 : i * 16); };
    const int cr = (wave >> 1) * 64 + (lane >> 4) * 16 + (lane & 1) * 8;
    const int mo8 = (P.Mo + 7) & ~7, ni8 = (P.Ni + 7) & ~7;                           // readable up to round_up(rows, 8) per token (ABI)
    int ca = m0 + cr, cb = n0 + cr;
    ca = ca <= mo8 - 8 ? ca : mo8 - 8;                                               // clamped pieces land in rows the epilogue masks
    cb = cb <= ni8 - 8 ? cb : ni8 - 8;
    unsigned offA[CI], offB[CI];
#pragma unroll
    for (int i = 0; i < CI; ++i) {
        offA[i] = (unsigned)((krow(i) * (int)P.lda + ca) * 2);
        offB[i] = (unsigned)((krow(i) * (int)P.ldb + cb) * 2);
    }
    const long stepA = (long)KB * P.lda * 2, stepB = (long)KB * P.ldb * 2;
    const unsigned char* baseA = (const unsigned char*)P.A;
    const unsigned char* baseB = (const unsigned char*)P.B;
    const int KT = g.tokens / KB, rows_tail = g.tokens - KT * KB;
    const int nk = KT + (rows_tail > 0 ? 1 : 0);
    auto issue = [&](int kt) {                               // K tile kt -> ring stage kt % STAGES
        unsigned char* dst = smem + (kt % STAGES) * STAGE_BYTES + wave * (CI * 1024);
        if (kt < KT) {
            const unsigned char* sa = baseA + (long)kt * stepA;
            const unsigned char* sb = baseB + (long)kt * stepB;
#pragma unroll
            for (int i = 0; i < CI; ++i) vb_glds16(sa + offA[i], dst + i * 1024);
#pragma unroll
            for (int i = 0; i < CI; ++i) vb_glds16(sb + offB[i], dst + HALF + i * 1024);
        } else {
            // the ragged tile: rows >= rows_tail are zeros; through registers (a masked LDS-direct lane would leave stale LDS behind);
            // loads from a clamped (always readable) row, the zero chosen afterwards
            const unsigned char* sa = baseA + (long)KT * stepA;
            const unsigned char* sb = baseB + (long)KT * stepB;
            u32x4 va[CI], vb[CI];
#pragma unroll
            for (int i = 0; i < CI; ++i) {
                const int k = krow(i);
                const int kc = k < rows_tail ? k : rows_tail - 1;
                va[i] = *(const u32x4*)(sa + (long)(kc * (int)P.lda + ca) * 2);
                vb[i] = *(const u32x4*)(sb + (long)(kc * (int)P.ldb + cb) * 2);
            }
#pragma unroll
            for (int i = 0; i < CI; ++i) {
                const bool ok = krow(i) < rows_tail;
                const u32x4 z = u32x4{0u, 0u, 0u, 0u};
                *(u32x4*)(dst + i * 1024 + lane * 16) = ok ? va[i] : z;
                *(u32x4*)(dst + HALF + i * 1024 + lane * 16) = ok ? vb[i] : z;
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment gathers (lane map of the transposing read: gemm_tn_8ph_kernel)
    const int s16 = lane & 15, kg = lane >> 4;
    const int lane_off = (kg >> 1) * 1024 + ((kg & 1) * 4 + (s16 >> 2)) * 32 + ((s16 >> 1) & 1) * 16 + (s16 & 1) * 8;
    const int offa_w = wm * RH + lane_off, offb_w = HALF + wn * RH + lane_off;
    bf16x4 fal[4][KS], fah[4][KS], fbl[4][KS], fbh[4][KS];

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) issue(s);
    // the dW tile, fetched NOW: behind the first tiles in the wave's in-order vmcnt queue, so that the K loop starts as soon as tile 0 has
    // landed instead of sitting through the read of the launch's dW tiles first (measured: no difference at B = 8 -- the launch is bound by
    // operand delivery, see the header); the waits for tiles 0 .. STAGES - 2 count these loads as allowed to stay in flight
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            int m = mw0 + pass * 32 + it * 8 + (lane >> 3);
            m = m < P.Mo ? m : P.Mo - 1;                                 // clamped: always readable, never stored
            const float* cp = P.C + (long)m * P.ldc + (col_ok ? ncol : 0);
            cin[pass][it][0] = *(const f32x4*)cp;
            cin[pass][it][1] = *(const f32x4*)(cp + 4);
        }
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed; the LDS-direct tiles issued after it (at most STAGES - 2, and only whole tiles: index < KT) may stay in flight
        // -- and, for the tiles of the prologue (kt <= STAGES - 2), the dW loads issued behind them
        const int last = KT - 1;
        if (kt <= STAGES - 2) {
            // in flight behind tile kt: prologue tiles kt + 1 .. STAGES - 2, the dW loads, then tiles STAGES - 1 .. kt + STAGES - 2 (issued in
            // iterations 0 .. kt - 1); every one of them that is a whole tile counts PER_TILE
            int whole = 0;
            for (int j = kt + 1; j <= kt + STAGES - 2; ++j) whole += j <= last ? 1 : 0;
            if (kt > last) vb_wait_vmcnt<0>();               // (a ragged-only launch: the tile came through registers)
            else if (whole >= 2 && STAGES >= 4) vb_wait_vmcnt<2 * PER_TILE + CIN_LOADS>();
            else if (whole == 1 && STAGES >= 3) vb_wait_vmcnt<PER_TILE + CIN_LOADS>();
            else vb_wait_vmcnt<CIN_LOADS>();
        }
        else if (STAGES >= 3 && kt + STAGES - 2 <= last) vb_wait_vmcnt<(STAGES - 2) * PER_TILE>();
        else if (STAGES >= 4 && kt + STAGES - 3 <= last) vb_wait_vmcnt<(STAGES >= 4 ? STAGES - 3 : 0) * PER_TILE>();
        else vb_wait_vmcnt<0>();
        vb_raw_barrier();                      // everyone's part of tile kt is in LDS; everyone finished reading tile kt - 1
        if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1);
        const unsigned char* buf = smem + (kt % STAGES) * STAGE_BYTES;
        const unsigned char* pa = buf + offa_w;
        const unsigned char* pb = buf + offb_w;
        vb_static_for<0, 4>([&](auto f) { vb_lds_read_tr_pair<decltype(f)::value * 256>(fal[decltype(f)::value][0], fah[decltype(f)::value][0], pa); });
        vb_static_for<0, 4>([&](auto f) { vb_lds_read_tr_pair<decltype(f)::value * 256>(fbl[decltype(f)::value][0], fbh[decltype(f)::value][0], pb); });
        if constexpr (KS == 2) {
            vb_static_for<0, 4>([&](auto f) { vb_lds_read_tr_pair<4096 + decltype(f)::value * 256>(fal[decltype(f)::value][KS - 1], fah[decltype(f)::value][KS - 1], pa); });
            vb_static_for<0, 4>([&](auto f) { vb_lds_read_tr_pair<4096 + decltype(f)::value * 256>(fbl[decltype(f)::value][KS - 1], fbh[decltype(f)::value][KS - 1], pb); });
            vb_wait_lgkmcnt<15>();             // 32 reads issued, the LDS returns them in order: the first 16 (the k = 0 .. 31 fragments) are in
        } else {
            vb_wait_lgkmcnt<0>();
        }
        vb_sched_fence();
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = vb_mma(vb_join(fal[mi][0], fah[mi][0]), vb_join(fbl[ni][0], fbh[ni][0]), acc[mi][ni]);
        if constexpr (KS == 2) {
            vb_sched_fence();
            vb_wait_lgkmcnt<0>();
            vb_sched_fence();
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = vb_mma(vb_join(fal[mi][KS - 1], fah[mi][KS - 1]), vb_join(fbl[ni][KS - 1], fbh[ni][KS - 1]), acc[mi][ni]);
        }
    }

    static_assert(4 * EPI_BYTES_PER_WAVE <= STAGES * STAGE_BYTES, "epilogue slabs live in the ring");
    tn_small_epilogue(acc, cin, smem, g, P, mw0, ncol, col_ok, wave, lane);
}
// The same job on 256 x 128 tiles, EIGHT waves as 4 (M) x 2 (N): an A panel of two half-tiles and one B half-tile per K tile (48 KB: three
// stages = 144 KB, one workgroup per compute unit).  The 128x128 form is bound by operand delivery -- 32 KB from L2 per 2.1 MFLOP; this
// tile pulls 48 KB per 4.2 MFLOP, 1.33x the FLOPs per delivered byte -- and an encoder layer's four weight gradients are 216 such tiles:
// one round on 256 compute units, eight waves per unit issuing copies instead of two workgroups of four in lockstep.  Copy stream and
// fragment gathers are gemm_tn_8ph_kernel's (wave w, instruction i fills chunk 2 w + i of every half-tile); the epilogue is the
// 128x128 kernel's (a wave owns a 64 x 64 block of dW, fetched behind the ring's first tiles, plain read-modify-write).
VB_KERNEL VB_LAUNCH_BOUNDS(512) gemm_tn_small256_kernel(TnSmallArgs g) {
    constexpr int STAGES = 3, HALF = 128 * 128, STAGE_BYTES = 3 * HALF, PER_TILE = 6, CIN_LOADS = 16;
    VB_DYN_SMEM(smem);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = vb_uniform(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = g.tile0[g.nprob];
    const int v = xcd_remap((int)blockIdx.x, ntiles);
    int pi = 0;
    for (int q = 1; q < g.nprob; ++q) if (v >= g.tile0[q]) pi = q;
    const TnProblem& P = g.p[pi];
    const int tl = v - g.tile0[pi];
    const int tm = (P.Mo + 255) / 256, tn = (P.Ni + 127) / 128;
    int m0, n0;
    if (tm < tn) { m0 = (tl % tm) * 256; n0 = (tl / tm) * 128; }
    else { m0 = (tl / tn) * 256; n0 = (tl % tn) * 128; }
    const int mw0 = m0 + wm * 64, nw0 = n0 + wn * 64;
    const int ncol = nw0 + (lane & 7) * 8;
    const bool col_ok = ncol < P.Ni;
    f32x4 cin[2][4][2];

    // ---- copy stream (gemm_tn_8ph_kernel's lane map): token row ck + 16 i, feature offset cr inside a 128-row half-tile
    const int ck = ((wave >> 1) & 1) * 32 + ((lane >> 3) & 1) * 8 + (wave & 1) * 4 + ((lane >> 1) & 3);
    const int cr = (wave >> 2) * 64 + (lane >> 4) * 16 + (lane & 1) * 8;
    const int mo8 = (P.Mo + 7) & ~7, ni8 = (P.Ni + 7) & ~7;
    int ca0 = m0 + cr, ca1 = m0 + 128 + cr, cb = n0 + cr;
    ca0 = ca0 <= mo8 - 8 ? ca0 : mo8 - 8;
    ca1 = ca1 <= mo8 - 8 ? ca1 : mo8 - 8;
    cb = cb <= ni8 - 8 ? cb : ni8 - 8;
    unsigned offA0[2], offA1[2], offB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        offA0[i] = (unsigned)(((ck + 16 * i) * (int)P.lda + ca0) * 2);
        offA1[i] = (unsigned)(((ck + 16 * i) * (int)P.lda + ca1) * 2);
        offB[i] = (unsigned)(((ck + 16 * i) * (int)P.ldb + cb) * 2);
    }
    const long stepA = 64L * P.lda * 2, stepB = 64L * P.ldb * 2;
    const unsigned char* baseA = (const unsigned char*)P.A;
    const unsigned char* baseB = (const unsigned char*)P.B;
    const int KT = g.tokens / 64, rows_tail = g.tokens - KT * 64;
    const int nk = KT + (rows_tail > 0 ? 1 : 0);
    auto issue = [&](int kt) {
        unsigned char* dst = smem + (kt % STAGES) * STAGE_BYTES + wave * 2048;
        if (kt < KT) {
            const unsigned char* sa = baseA + (long)kt * stepA;
            const unsigned char* sb = baseB + (long)kt * stepB;
#pragma unroll
            for (int i = 0; i < 2; ++i) vb_glds16(sa + offA0[i], dst + i * 1024);
#pragma unroll
            for (int i = 0; i < 2; ++i) vb_glds16(sa + offA1[i], dst + HALF + i * 1024);
#pragma unroll
            for (int i = 0; i < 2; ++i) vb_glds16(sb + offB[i], dst + 2 * HALF + i * 1024);
        } else {                                                 // the ragged tile: through registers, rows >= rows_tail are zeros
            const unsigned char* sa = baseA + (long)KT * stepA;
            const unsigned char* sb = baseB + (long)KT * stepB;
            u32x4 v0[2], v1[2], vb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int k = ck + 16 * i;
                const int kc = k < rows_tail ? k : rows_tail - 1;
                v0[i] = *(const u32x4*)(sa + (long)(kc * (int)P.lda + ca0) * 2);
                v1[i] = *(const u32x4*)(sa + (long)(kc * (int)P.lda + ca1) * 2);
                vb[i] = *(const u32x4*)(sb + (long)(kc * (int)P.ldb + cb) * 2);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool ok = ck + 16 * i < rows_tail;
                const u32x4 z = u32x4{0u, 0u, 0u, 0u};
                *(u32x4*)(dst + i * 1024 + lane * 16) = ok ? v0[i] : z;
                *(u32x4*)(dst + HALF + i * 1024 + lane * 16) = ok ? v1[i] : z;
                *(u32x4*)(dst + 2 * HALF + i * 1024 + lane * 16) = ok ? vb[i] : z;
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int s16 = lane & 15, kg = lane >> 4;
    const int lane_off = (kg >> 1) * 1024 + ((kg & 1) * 4 + (s16 >> 2)) * 32 + ((s16 >> 1) & 1) * 16 + (s16 & 1) * 8;
    const int offa_w = (wm >> 1) * HALF + (wm & 1) * 8192 + lane_off, offb_w = 2 * HALF + wn * 8192 + lane_off;
    bf16x4 fal[4][2], fah[4][2], fbl[4][2], fbh[4][2];

#pragma unroll
    for (int s2 = 0; s2 < STAGES - 1; ++s2)
        if (s2 < nk) issue(s2);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            int m = mw0 + pass * 32 + it * 8 + (lane >> 3);
            m = m < P.Mo ? m : P.Mo - 1;
            const float* cp = P.C + (long)m * P.ldc + (col_ok ? ncol : 0);
            cin[pass][it][0] = *(const f32x4*)cp;
            cin[pass][it][1] = *(const f32x4*)(cp + 4);
        }
    for (int kt = 0; kt < nk; ++kt) {
        const int last = KT - 1;
        if (kt <= STAGES - 2) {                              // behind tile kt: at most one whole tile of the ring + the dW loads
            if (kt > last) vb_wait_vmcnt<0>();
            else if (kt + 1 <= last) vb_wait_vmcnt<PER_TILE + CIN_LOADS>();
            else vb_wait_vmcnt<CIN_LOADS>();
        }
        else if (kt + 1 <= last) vb_wait_vmcnt<PER_TILE>();
        else vb_wait_vmcnt<0>();
        vb_raw_barrier();
        if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1);
        const unsigned char* buf = smem + (kt % STAGES) * STAGE_BYTES;
        const unsigned char* pa = buf + offa_w;
        const unsigned char* pb = buf + offb_w;
        vb_static_for<0, 4>([&](auto f) { vb_lds_read_tr_pair<decltype(f)::value * 256>(fal[decltype(f)::value][0], fah[decltype(f)::value][0], pa); });
        vb_static_for<0, 4>([&](auto f) { vb_lds_read_tr_pair<decltype(f)::value * 256>(fbl[decltype(f)::value][0], fbh[decltype(f)::value][0], pb); });
        vb_static_for<0, 4>([&](auto f) { vb_lds_read_tr_pair<4096 + decltype(f)::value * 256>(fal[decltype(f)::value][1], fah[decltype(f)::value][1], pa); });
        vb_static_for<0, 4>([&](auto f) { vb_lds_read_tr_pair<4096 + decltype(f)::value * 256>(fbl[decltype(f)::value][1], fbh[decltype(f)::value][1], pb); });
        vb_wait_lgkmcnt<15>();
        vb_sched_fence();
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = vb_mma(vb_join(fal[mi][0], fah[mi][0]), vb_join(fbl[ni][0], fbh[ni][0]), acc[mi][ni]);
        vb_sched_fence();
        vb_wait_lgkmcnt<0>();
        vb_sched_fence();
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = vb_mma(vb_join(fal[mi][1], fah[mi][1]), vb_join(fbl[ni][1], fbh[ni][1]), acc[mi][ni]);
    }
    tn_small_epilogue(acc, cin, smem, g, P, mw0, ncol, col_ok, wave, lane);
}
static bool tn_eligible(const void* A, long lda, const void* B, long ldb, const float* C, long ldc, int Mo, int Ni, int K) {
    return K >= 64 && (K % 64) == 0 && Mo >= 1 && Ni >= 1 && lda >= ((Mo + 7) & ~7) && ldb >= ((Ni + 7) & ~7) && (lda % 8) == 0 && (ldb % 8) == 0 &&
           ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0 && C != nullptr && ldc >= Ni &&
           64L * lda * 2 + 2L * Mo < (1L << 31) && 64L * ldb * 2 + 2L * Ni < (1L << 31);
}

// the small-token kernel takes a group when: few K tiles (the persistent kernel's atomic drain would dominate), vector-aligned dW
// rows, and no two problems adding into the same dW (plain read-modify-writes).  VB_TN_SMALL_MAX_KT: crossover measured on MI355X
// (profiles/r06_small_batch_ab.txt): the persistent kernel's 256x256 tiles move half the LDS bytes per FLOP and win once the K loop is long
#ifndef VB_TN_SMALL_256
#define VB_TN_SMALL_256 1
#endif
#ifndef VB_TN_SMALL_KB32
#define VB_TN_SMALL_KB32 0           // 1: the two-workgroups-per-CU case on four 16-KB stages of 32 tokens (measured: no gain, see the kernel)
#endif
#ifndef VB_TN_SMALL_MAX_KT
#define VB_TN_SMALL_MAX_KT 128
#endif
static bool tn_small_eligible(const TnArgs& g, int tokens) {
    if ((tokens + 63) / 64 > VB_TN_SMALL_MAX_KT) return false;
    for (int i = 0; i < g.nprob; ++i) {
        const TnProblem& P = g.p[i];
        if ((P.Ni % 8) || (P.ldc % 4) || (((uintptr_t)P.C) & 15)) return false;
        for (int j = i + 1; j < g.nprob; ++j) {
            const float *a0 = P.C, *a1 = a0 + (long)(P.Mo - 1) * P.ldc + P.Ni;
            const float *b0 = g.p[j].C, *b1 = b0 + (long)(g.p[j].Mo - 1) * g.p[j].ldc + g.p[j].Ni;
            if (a0 < b1 && b0 < a1) return false;
        }
    }
    return true;
}
static int launch_tn_small(const TnArgs& main, int tokens, hipStream_t stream) {
    TnSmallArgs g;
    g.nprob = main.nprob; g.alpha = main.alpha; g.alpha_dev = main.alpha_dev;
    g.tokens = tokens;
    int tiles = 0;
    double flops = 0;
    for (int i = 0; i < main.nprob; ++i) {
        g.p[i] = main.p[i];
        g.tile0[i] = tiles;
        tiles += ((main.p[i].Mo + 127) / 128) * ((main.p[i].Ni + 127) / 128);
        flops += 2.0 * main.p[i].Mo * main.p[i].Ni * tokens;
    }
    g.tile0[main.nprob] = tiles;
    const int cus = (t_opts.persistent_workgroups > 0 && t_opts.persistent_workgroups < vb_num_cus()) ? t_opts.persistent_workgroups : vb_num_cus();
    dim3 grid((unsigned)tiles), block(256);
    // key: weight-gradient family (4 | 2 | 1) + 32 = the small-token kernel
    if (tiles <= cus)                                        // one workgroup per CU at most: four 32-KB stages
        return vb_prof_launch(flops, 4 | 2 | 1 | 32, stream, [&]() { VB_LAUNCH((gemm_tn_small_kernel<4, 64>), grid, block, 4 * 2 * 128 * 128, stream, g); });
#if VB_TN_SMALL_256
    {   // more 128x128 tiles than compute units, but the 256x128 tiles fit in one round: the eight-wave kernel (an encoder layer: 216 tiles)
        TnSmallArgs g2 = g;
        int t256 = 0;
        for (int i = 0; i < main.nprob; ++i) {
            g2.tile0[i] = t256;
            t256 += ((main.p[i].Mo + 255) / 256) * ((main.p[i].Ni + 127) / 128);
        }
        g2.tile0[main.nprob] = t256;
        if (t256 <= cus)
            return vb_prof_launch(flops, 4 | 2 | 1 | 32, stream, [&]() { VB_LAUNCH(gemm_tn_small256_kernel, dim3((unsigned)t256), dim3(512), 3 * 3 * 128 * 128, stream, g2); });
    }
#endif
#if VB_TN_SMALL_KB32
    // more tiles than compute units: two workgroups per CU, four 16-KB stages of 32 tokens (64 KB each)
    return vb_prof_launch(flops, 4 | 2 | 1 | 32, stream, [&]() { VB_LAUNCH((gemm_tn_small_kernel<4, 32>), grid, block, 4 * 2 * 32 * 128 * 2, stream, g); });
#else
    return vb_prof_launch(flops, 4 | 2 | 1 | 32, stream, [&]() { VB_LAUNCH((gemm_tn_small_kernel<2, 64>), grid, block, 2 * 2 * 128 * 128, stream, g); });
#endif
}

// kernel for a K-contiguous x K-contiguous problem (vb_stream_opts.nt_kernel): 0 = chosen from the shape; 1 = the generic
// register-staged kernel; tens = waves in M (2 -> 128-row tile, 4 -> 256-row tile), units = LDS stages; 80 / 81 persistent
#ifndef VB_DROPRES_SHORT_K_PERSISTENT
#define VB_DROPRES_SHORT_K_PERSISTENT 1   // the K = 768 attention-out GEMM with the dropout + residual epilogue: 1 = persistent kernel, 0 = two-workgroup kernel
#endif
#ifndef VB_SPLITK_PREFER_128
#define VB_SPLITK_PREFER_128 1       // split-K problems on 128x128 tiles with more slices rather than 64x128 tiles with fewer (session 12: B = 16 5.90 -> 5.75 ms, B = 8 unchanged)
#endif
#ifndef VB_SPLITK_MIN_KT
#define VB_SPLITK_MIN_KT 24          // reductions shorter than this (K < 1536) are not cut
#endif
#ifndef VB_SPLITK_MIN_SLICE
#define VB_SPLITK_MIN_SLICE 8        // K tiles per slice at least
#endif
template <typename T, typename TO>
int dispatch_pipe(const GemmArgs& g, hipStream_t s) {
    int variant = t_opts.nt_kernel;
    // compute units this stream may fill: the device's, or fewer while DataParallelGradSync keeps some for RCCL (persistent_workgroups)
    const int all_cus = vb_num_cus();
    const int cus = (t_opts.persistent_workgroups > 0 && t_opts.persistent_workgroups < all_cus) ? t_opts.persistent_workgroups : all_cus;
    if (g.drop_thresh) {
        // dropout + residual epilogue (vb_gemm_dropres): bf16 -> bf16 on the two big-tile kernels only -- problems that fill the chip;
        // everything else is declined and the caller (layer.hip) keeps dropout + residual in its LayerNorm launch
        if (sizeof(T) != 2 || sizeof(TO) != 2 || g.x3) return VB_ERR_UNSUPPORTED;
        const long t256d = (long)((g.M + 255) / 256) * ((g.N + 255) / 256);
        if (t256d < 160 || (variant != 0 && variant != 81 && variant != 90)) return VB_ERR_UNSUPPORTED;
        if (variant == 0) variant = (g.K >= 2048 || VB_DROPRES_SHORT_K_PERSISTENT) ? 81 : 90;
        if (variant == 81 && cus < all_cus) variant = 90;
        return variant == 81 ? launch_8ph<T, TO>(g, s) : launch_dual<T, TO>(g, s);
    }
    if (g.x3) {                                            // split operands: the two-workgroup kernel or the two-barrier ones
        const long t256 = (long)((g.M + 255) / 256) * ((g.N + 255) / 256);
        const long t128 = (long)((g.M + 255) / 256) * ((g.N + 127) / 128);
        // the persistent 256x256 kernel when its tiles fill the chip (virtual K >= 3 x 64 tiles per segment: the epilogue share is
        // small and its lower LDS traffic per FLOP wins); pinning 90 keeps the two-workgroup kernel (A/B runs)
        // Measured inside the step at B = 512 (profiles/r04_x3_gemm_ab.txt): the persistent kernel wins where the epilogue is light
        // (bias only: 614 -> 576 us; + residual gradient: 917 -> 814 us), the two-workgroup kernel where it is heavy -- GELU + GELU'
        // with a split result (1195 vs 1253 us), x GELU' + column sums (1240 vs 1271 us) and the 30522-wide fp32 logits (5.96 vs 8.15 ms)
        // Round 5, after the epilogue's re-waits were removed (FLUSH): GELU + GELU' with a split result moved to the persistent kernel too --
        // 2 329 vs 2 477 us inside the step at B = 1024 (profiles/r05_x3_nt_kernel_ab.txt); x GELU' + column sums (2 436 vs 2 407) and
        // the fp32 logits (33.7 vs 22.7 ms) stay
        const bool light = g.act == VB_ACT_NONE && !g.aux_in && !g.aux_out && !g.colsum && !g.split_out && g.N <= 4096;
        const bool gelu_split = g.act == VB_ACT_GELU_SAVE_GRAD && g.split_out && !g.aux_in && !g.colsum && g.N <= 4096;
        const long t22 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);          // at most one 128x128 tile per CU: the four-stage ring (see below)
        // a pinned kernel is honoured here too (14 / 22 / 24 / 42 / 81 / 90: what vb_stream_set_opts accepts in the product library)
        if (variant != 14 && variant != 22 && variant != 24 && variant != 42 && variant != 90 && variant != 81)
            variant = t256 >= 160 ? ((light || gelu_split) ? 81 : 90) : (t128 >= cus ? 42 : (t22 <= cus ? 24 : 22));   // (never 100 / 101)
    }
#ifdef VB_DEV_KNOBS
    if (variant == 200) {
        // the vendor yardstick inside the step (vendor_gemm.hip): plain GEMMs -- bias only, or "+ addend" as beta = 1 -- go to
        // hipBLASLt; whatever it does not take (fused epilogues, no library on the box) runs on the kernels chosen below
        if constexpr (sizeof(T) == 2) {
            const bool addend_ok = !g.addend || sizeof(TO) == 2;
            if (!g.x3 && g.act == VB_ACT_NONE && !g.aux_in && !g.aux_out && !g.colsum && !g.alpha_dev && !g.accumulate && addend_ok) {
                VbVendorGemm v{g.A, g.lda, g.B, g.ldb, g.C, g.ldc, g.M, g.N, g.K, g.alpha, g.bias, g.addend, g.ld_addend, sizeof(TO) == 4 ? 1 : 0};
                int rc = VB_ERR_UNSUPPORTED;
                const int prc = vb_prof_launch(2.0 * g.M * g.N * g.K, (sizeof(TO) == 4 ? 4 : 0) | 512, s, [&]() { rc = vb_vendor_nt(v, s); });
                if (rc != VB_ERR_UNSUPPORTED) return rc != VB_OK ? rc : prc;
                vb_prof_drop_last();
            }
        }
        variant = 0;
    }
#endif
    if (variant == 0) {
        // measured on MI355X (profiles/r01_gemm_variant_sweep_b128.txt)
        const long t256 = (long)((g.M + 255) / 256) * ((g.N + 255) / 256);
        const long t128 = (long)((g.M + 255) / 256) * ((g.N + 127) / 128);
        // big persistent tiles when they fill most of the chip; otherwise the two-barrier kernels, with the 128x128 tile
        // when even 256x128 tiles would leave CUs idle (small batches)
        // Problems that fill the chip (B >= ~128 at S = 164): the two-workgroups-per-CU kernel, whose epilogues run under the
        // partner workgroup's MFMAs -- except long-K GEMMs with a plain epilogue (FFN-out forward: K = 3072), where the
        // persistent 256x256 kernel's K loop is ~10 % faster and the epilogue is 1/48 of the tile
        // (profiles/r02_gemm_ab_*.txt: per step 36.6 -> 34.6 ms of NT GEMMs at B = 512).
        // Round 5: "+ residual gradient" (an addend, no activation) counts as light too -- with the epilogue's re-waits gone
        // (gemm_epilogue_private, FLUSH) the two long-K dgrads of a layer (FFN-in: K = 3072, QKV: K = 2304) run 565 us on the persistent
        // kernel against 624 us on the two-workgroup one inside the step (profiles/r05_gemm_epilogue_waits.txt); before that fix the
        // product build's persistent "+ addend" instantiation was the slower one (667 us) although the developer build's, laid out
        // differently, was not (603) -- A/B product builds, never the developer library, when the question is what ships.
        // (developer library: debug bit 27 restores the round-4 rule)
        // Last step of round 5: bias-only GEMMs go to the persistent kernel at ANY K (QKV / attention-out forward, attention-out dgrad:
        // K = 768) -- 321 vs 343 us per launch inside the step, product build against product build on one box 119.3 -> 119.0 ms per step
        // (profiles/r05_gemm_plain_short_k_ab.txt).  The two-workgroup kernel keeps what has a heavy epilogue: GELU + GELU' (1 038 vs
        // 1 110 us), x GELU' + column sums (tie), the fp32 logits (9.0 vs 9.6 ms).
        const bool plain = !g.addend && !g.aux_in && !g.aux_out && !g.colsum && g.act == VB_ACT_NONE && sizeof(TO) == 2;
        const bool light = !g.aux_in && !g.aux_out && !g.colsum && g.act == VB_ACT_NONE && !g.accumulate && sizeof(TO) == 2 && !(g.debug & (1 << 27));
        // small problems (per-GPU batch <= ~32 at S = 164: at most one 128x128 tile per compute unit): the FOUR-stage ring (128 KB, one
        // workgroup per CU -- there is no second one to host anyway) keeps three K tiles in flight where the two-stage form exposes an
        // L2 / HBM round trip per K tile: the whole step 6.43 -> 6.18 ms at B = 8, 6.73 -> 6.38 at 16, 7.63 -> 7.21 at 32; with more
        // tiles than CUs it loses the second resident workgroup (B = 64: 10.68 -> 11.46 ms) -- profiles/r05_small_batch_four_stage_ab.txt
        // ... and 64x128 tiles on the same ring (nt_kernel 14) while even those leave half the chip idle: 6.01 -> 5.90 ms at B = 8, 6.34 -> 6.19 at 16
        const long t22 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
        variant = (sizeof(T) == 2 && t256 >= 160) ? ((plain || (g.K >= 2048 && light)) ? 81 : 90) : (t128 >= cus ? 42 : (t22 <= cus / 2 ? 14 : (t22 <= cus ? 24 : 22)));
        // compute units reserved for a collective (cus < all_cus): the persistent kernel's grid IS the CU count and its N = 768 shapes pay
        // a whole second round for any grid below 246 workgroups -- those launches take the one-workgroup-per-tile kernel instead; every
        // other choice of the rule stands (pinning 90 from Python bypassed the small-problem kernels)
        if (variant == 81 && cus < all_cus) variant = 90;
#ifdef VB_DEV_KNOBS
        if (variant == 90 && (g.debug & (1 << 28)) && (plain || light)) variant = 81;      // A/B: the short-K plain / "+ addend" shapes on the persistent kernel too
#endif
    }
    if constexpr (sizeof(T) == 2 && sizeof(TO) == 2) {
        // split-K across compute units for long reductions whose tiles leave at least half of the chip idle (per-GPU batch 8 at S = 164: the
        // K = 3072 / 2304 GEMMs with N = 768 are 126 tiles of 64x128 walking 36-48 K tiles each -- 30 us for 6 GFLOP,
        // profiles/r05_fin3_kernel_stats_b8.txt): slices of at least VB_SPLITK_MIN_SLICE K tiles, as many as fill the chip, partial tiles
        // through the stream's scratch (vb_stream_set_scratch; none registered = not chosen)
        if ((variant == 14 || variant == 24) && !g.x3 && g.act == VB_ACT_NONE && !g.aux_in && !g.aux_out && !g.colsum && !g.accumulate &&
            g.K / 64 >= VB_SPLITK_MIN_KT) {
            const vb_scratch sc = vb_scratch_for((void*)s);
#if VB_SPLITK_PREFER_128
            // 128x128 tiles with more slices instead of 64x128 tiles with fewer: a third less operand traffic per FLOP, four waves per workgroup
            if (variant == 14 && (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * 2 <= cus) variant = 24;
#endif
            const long tiles = (long)((g.M + (variant == 14 ? 63 : 127)) / (variant == 14 ? 64 : 128)) * ((g.N + 127) / 128);
            const int nk = g.K / 64;
            int want = (int)(cus / tiles);
            if (want > nk / VB_SPLITK_MIN_SLICE) want = nk / VB_SPLITK_MIN_SLICE;
            if (want > 8) want = 8;
            if (sc.ptr && want >= 2 && tiles * 4 <= VB_SCRATCH_COUNTER_BYTES) {
                GemmArgs gs = g;
                gs.sk_kps = (nk + want - 1) / want;
                gs.sk_splits = (nk + gs.sk_kps - 1) / gs.sk_kps;
                const long slab_bytes = tiles * gs.sk_splits * (variant == 14 ? 128 : 256) * 256L;
                if (gs.sk_splits >= 2 && VB_SCRATCH_COUNTER_BYTES + slab_bytes <= sc.bytes) {
                    gs.sk_cnt = (int*)sc.ptr;
                    gs.sk_slabs = (float*)((unsigned char*)sc.ptr + VB_SCRATCH_COUNTER_BYTES);
                    return variant == 14 ? launch_pipe<T, TO, 1, 4>(gs, s) : launch_pipe<T, TO, 2, 4>(gs, s);
                }
            }
        }
    }
    switch (variant) {
        case 22: return launch_pipe<T, TO, 2, 2>(g, s);
        case 24: return launch_pipe<T, TO, 2, 4>(g, s);
        case 14: return launch_pipe<T, TO, 1, 4>(g, s);
        case 42: return launch_pipe<T, TO, 4, 2>(g, s);
        case 81: return launch_8ph<T, TO>(g, s);
        case 90: return launch_dual<T, TO>(g, s);
#ifdef VB_DEV_KNOBS
        case 80: case 82: return launch_8ph<T, TO>(g, s);
        case 91: case 92: return launch_dual<T, TO>(g, s);
        case 100: return launch_big<T, TO>(g, s);
        case 101: return launch_big<T, TO>(g, s, true);
#endif
        default: return VB_ERR_UNSUPPORTED;
    }
}

template <typename T>
int dispatch(int out_dtype_is_f32, int al, int bl, const GemmArgs& g, hipStream_t s) {
    if (al == VB_KCONTIG && bl == VB_KCONTIG) {
        if (g.fast_a && g.fast_b && g.splits == 1 && t_opts.nt_kernel != 1)
            return out_dtype_is_f32 ? dispatch_pipe<T, float>(g, s) : dispatch_pipe<T, T>(g, s);
    }
    if (g.drop_thresh) return VB_ERR_UNSUPPORTED;           // the dropout epilogue exists on the fast K-contiguous kernels only
    if (al == VB_KCONTIG && bl == VB_KCONTIG) {
        if constexpr (sizeof(T) == 2) { if (g.x3) return launch_gemm<T, float, VB_KCONTIG, VB_KCONTIG, float>(g, s); }
        return out_dtype_is_f32 ? launch_gemm<T, float, VB_KCONTIG, VB_KCONTIG>(g, s)
                                : launch_gemm<T, T, VB_KCONTIG, VB_KCONTIG>(g, s);
    }
    if (al == VB_KCONTIG && bl == VB_KSTRIDED)
        return out_dtype_is_f32 ? launch_gemm<T, float, VB_KCONTIG, VB_KSTRIDED>(g, s)
                                : launch_gemm<T, T, VB_KCONTIG, VB_KSTRIDED>(g, s);
    if (al == VB_KSTRIDED && bl == VB_KSTRIDED)
        return out_dtype_is_f32 ? launch_gemm<T, float, VB_KSTRIDED, VB_KSTRIDED>(g, s)
                                : launch_gemm<T, T, VB_KSTRIDED, VB_KSTRIDED>(g, s);
    return VB_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int vb_gemm(int dtype, int out_dtype, int a_layout, int b_layout,
                       const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                       int M, int N, int K, float alpha, const float* alpha_dev, const float* bias,
                       const void* addend, int64_t ld_addend, int act,
                       const void* aux_in, void* aux_out, int64_t ld_aux, int accumulate,
                       float* colsum_out, void* stream) {
    return vb_gemm_dropres(dtype, out_dtype, a_layout, b_layout, A, lda, B, ldb, C, ldc, M, N, K, alpha, alpha_dev, bias, addend, ld_addend,
                           act, aux_in, aux_out, ld_aux, accumulate, colsum_out, 0.f, 0, 0, stream);
}

// internal (layer.hip): is the "dropout + residual in the producing GEMM" form of a BertLayer's output blocks armed?  Never in the product
// library (it lost its A/B AND costs bf16 parity: below); developer library: debug bit 29.
int vb_gemm_fuse_residual_armed() {
#ifdef VB_DEV_KNOBS
    return (g_debug >> 29) & 1;
#else
    return 0;
#endif
}

// internal (layer.hip): vb_gemm + inverted dropout (p_drop, seed, drop_stream: the LayerNorm kernels' generator and indexing) of
// alpha acc + bias in front of the addend -- BertSelfOutput / BertOutput's dropout(dense(x)) + residual in the producing GEMM, so that
// the LayerNorm launch reads one tensor instead of two.
// MEASURED, round 6 (profiles/r06_dropres_epilogue_ab.txt; three product builds on one box, three interleaved rounds at B = 1024):
//   p_drop == 0 ("+ residual" only): free on every kernel (the "+ addend" epilogue) -- but the LayerNorm then normalises a bf16-ROUNDED
//   residual stream z where the unfused pair adds bf16(y) + bf16(x) exactly in fp32: max |dlogit| against the real reference's golden
//   3.2e-3 -> 5.8e-3 on micro_pretraining (tests/test_model_parity.py caught it; the residual stream is the largest term of the bf16 error
//   budget, profiles/r02_bf16_error_budget.txt).  Not shipped either;
//   p_drop  > 0: the LayerNorm forward drops 132.3 -> 112.2 us per launch (-0.48 ms per step) but the generator's ~50 VALU instructions
//   per 8 elements sit in the persistent kernel's epilogue, where the matrix pipe idles: 397 -> 454 us per launch on the 24 fused GEMMs
//   of a step (+1.09 ms), step 118.0 -> 118.45 ms (two-workgroup kernel for the K = 768 shape: 118.5).  So the dropout form LOSES and is
//   compiled into the DEVELOPER library only (debug bit 29 arms it; tests/test_bench_shape.py replays its mask); the product declines
//   p_drop > 0 with VB_ERR_UNSUPPORTED, NOTHING launched, and the caller keeps dropout + residual in its LayerNorm launch.
int vb_gemm_dropres(int dtype, int out_dtype, int a_layout, int b_layout,
                    const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                    int M, int N, int K, float alpha, const float* alpha_dev, const float* bias,
                    const void* addend, int64_t ld_addend, int act,
                    const void* aux_in, void* aux_out, int64_t ld_aux, int accumulate,
                    float* colsum_out, float p_drop, uint64_t drop_seed, uint32_t drop_stream, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return VB_ERR_ARG;
    if (p_drop < 0.f || p_drop >= 1.f) return VB_ERR_ARG;
    if (p_drop > 0.f) {
#ifdef VB_DEV_KNOBS
        if (!(g_debug & (1 << 29))) return VB_ERR_UNSUPPORTED;
#else
        return VB_ERR_UNSUPPORTED;
#endif
        if (dtype != VB_BF16 || out_dtype != VB_BF16 || !addend || act != VB_ACT_NONE || accumulate || colsum_out || alpha_dev ||
            a_layout != VB_KCONTIG || b_layout != VB_KCONTIG || (K % 64) || ldc != N)
            return VB_ERR_UNSUPPORTED;
    }
    if (dtype != VB_F32 && dtype != VB_BF16 && dtype != VB_BF16X3) return VB_ERR_ARG;
    if (out_dtype != VB_F32 && out_dtype != dtype) return VB_ERR_ARG;
    const bool x3 = dtype == VB_BF16X3;
    const bool split_out = x3 && out_dtype == VB_BF16X3;
    if (x3) {
        // split operands: fp32 everywhere but the MFMA inputs; whole K tiles; hi | lo halves of 16-byte-aligned rows
        if (a_layout != VB_KCONTIG || b_layout != VB_KCONTIG) return VB_ERR_UNSUPPORTED;
        // a split RESULT: whole 8-column groups, room for both planes, nothing to accumulate into
        if (split_out && ((N % 8) || (ldc % 16) || N > ldc / 2 || accumulate || (((uintptr_t)C) & 15))) return VB_ERR_UNSUPPORTED;
        if ((K % 64) || (lda % 16) || (ldb % 16) || K > lda / 2 || K > ldb / 2) return VB_ERR_UNSUPPORTED;
    }
    t_opts = vb_opts_for(stream);
    vb_prof_select(stream);
    const int epc = dtype == VB_BF16 ? 8 : 4;
    // vector loads are 16 bytes: leading dimensions and base pointers must keep them aligned
    if ((lda % 8) || (ldb % 8) || (((uintptr_t)A | (uintptr_t)B) & 15)) return VB_ERR_ARG;
    (void)epc;
    if ((act == VB_ACT_GELU_GRAD || act == VB_ACT_MUL_AUX) && !aux_in) return VB_ERR_ARG;
    if (act == VB_ACT_GELU_SAVE_GRAD && !aux_out) return VB_ERR_ARG;
    if (act < VB_ACT_NONE || act > VB_ACT_MUL_AUX) return VB_ERR_ARG;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.addend = addend; g.ld_addend = ld_addend; g.aux_in = aux_in; g.aux_out = aux_out;
    g.ld_aux = ld_aux; g.alpha = alpha; g.alpha_dev = alpha_dev; g.colsum = colsum_out; g.act = act; g.accumulate = accumulate;
    g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
    g.debug = g_debug; g.trace = g_trace;
    g.split_out = split_out ? 1 : 0;
    g.nt_store = (!x3 && dtype == VB_BF16 && K <= 1024 && !(g_debug & (1 << 26))) ? 1 : 0;      // developer library, debug bit 26: plain stores (the A/B and bit-compare arm)
    g.x3 = x3 ? 1 : 0; g.a_lo = x3 ? (int)(lda / 2) : 0; g.b_lo = x3 ? (int)(ldb / 2) : 0; g.kseg = x3 ? K / 64 : 0;
    g.stripe = 0;
    g.sk_splits = 1; g.sk_kps = 0; g.sk_slabs = nullptr; g.sk_cnt = nullptr;
    g.drop_thresh = p_drop > 0.f ? vb_drop_thresh16(p_drop) : 0u; g.drop_scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    g.drop_seed = drop_seed; g.drop_stream = drop_stream;
    const int bk = dtype == VB_F32 ? 32 : 64;
    const int nk = x3 ? 3 * (K / 64) : (K + bk - 1) / bk;
    // LDS-direct copies need whole K tiles (a masked lane would leave stale LDS behind)
    g.fast_a = (a_layout == VB_KCONTIG && (K % bk) == 0) ? 1 : 0;
    g.fast_b = (b_layout == VB_KCONTIG && (K % bk) == 0) ? 1 : 0;
    // split-K only where the result is an fp32 accumulator without an element-wise epilogue
    g.splits = 1; g.kt_per_split = nk;
    const int tiles = g.tiles_m * g.tiles_n;
    if (accumulate && out_dtype == VB_F32 && !bias && !addend && !colsum_out && act == VB_ACT_NONE && tiles < 256 && nk >= 16) {
        int want = (320 + tiles - 1) / tiles;
        int maxs = nk / 8;
        if (want > maxs) want = maxs;
        if (want > 1) {
            g.kt_per_split = (nk + want - 1) / want;
            g.splits = (nk + g.kt_per_split - 1) / g.kt_per_split;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    const int of32 = (out_dtype == VB_F32 || split_out) ? 1 : 0;
    // weight-gradient form: the persistent TN kernel reduces over whole 64-token K tiles; a ragged token count (tokens % 64 != 0:
    // any batch whose B x S is not a multiple of 64) sends the last < 64 tokens through the generic kernel, accumulating onto the same C
    const int k_main = K & ~63;
    if (dtype == VB_BF16 && of32 && a_layout == VB_KSTRIDED && b_layout == VB_KSTRIDED && accumulate && !bias && !addend &&
        !colsum_out && act == VB_ACT_NONE && t_opts.nt_kernel != 1 && k_main >= 64 &&
        tn_eligible(A, lda, B, ldb, (const float*)C, ldc, M, N, k_main)) {
        TnArgs tg;
        tg.nprob = 1; tg.alpha = alpha; tg.alpha_dev = alpha_dev;
        tg.p[0].A = A; tg.p[0].B = B; tg.p[0].C = (float*)C; tg.p[0].lda = lda; tg.p[0].ldb = ldb; tg.p[0].ldc = ldc;
        tg.p[0].Mo = M; tg.p[0].Ni = N;
        if (tn_small_eligible(tg, K)) return launch_tn_small(tg, K, s);      // few tokens (the MLM decoder's weight gradient over the labelled rows, small batches)
        const int rc = launch_tn_group(tg, k_main, s);
        if (rc != VB_OK || k_main == K) return rc;
        return vb_gemm(dtype, out_dtype, a_layout, b_layout, (const bf16*)A + (long)k_main * lda, lda, (const bf16*)B + (long)k_main * ldb, ldb,
                       C, ldc, M, N, K - k_main, alpha, alpha_dev, nullptr, nullptr, 0, VB_ACT_NONE, nullptr, nullptr, 0, 1, nullptr, stream);
    }
    if (dtype == VB_BF16 || x3) return dispatch<bf16>(of32 && true, a_layout, b_layout, g, s);
    return dispatch<float>(1, a_layout, b_layout, g, s);
}

extern "C" int vb_wgrad_grouped(int dtype, int n, const void* const* dy, const int64_t* ld_dy, const void* const* x,
                                const int64_t* ld_x, void* const* dw, const int64_t* ld_dw, const int* n_out,
                                const int* n_in, int tokens, float alpha, const float* alpha_dev, void* stream) {
    if (n <= 0 || n > VB_TN_MAX || !dy || !ld_dy || !x || !ld_x || !dw || !ld_dw || !n_out || !n_in || tokens <= 0)
        return VB_ERR_ARG;
    if (dtype != VB_F32 && dtype != VB_BF16 && dtype != VB_BF16X3) return VB_ERR_ARG;
    if (dtype == VB_BF16X3) {
        // split operands [tokens][hi | lo]: dW += dy_hi^T x_hi + dy_lo^T x_hi + dy_hi^T x_lo -- three passes of the bf16 path
        // over the planes (the accumulation into the fp32 dW is what the kernel does anyway)
        const void* dyp[VB_TN_MAX]; const void* xp[VB_TN_MAX];
        struct ProfScope {                                   // launch records: algorithmic FLOPs (the three passes sum to 2 M N K), tagged
            ProfScope() { t_vb_prof_scale = 1.0 / 3.0; t_vb_prof_key_or = 256; }
            ~ProfScope() { t_vb_prof_scale = 1.0; t_vb_prof_key_or = 0; }
        } prof_scope;
        for (int pass = 0; pass < 3; ++pass) {
            for (int i = 0; i < n; ++i) {
                if ((ld_dy[i] % 16) || (ld_x[i] % 16) || n_out[i] > ld_dy[i] / 2 || n_in[i] > ld_x[i] / 2) return VB_ERR_UNSUPPORTED;
                dyp[i] = (const bf16*)dy[i] + (pass == 1 ? ld_dy[i] / 2 : 0);
                xp[i] = (const bf16*)x[i] + (pass == 2 ? ld_x[i] / 2 : 0);
            }
            const int rc = vb_wgrad_grouped(VB_BF16, n, dyp, ld_dy, xp, ld_x, dw, ld_dw, n_out, n_in, tokens, alpha, alpha_dev, stream);
            if (rc != VB_OK) return rc;
        }
        return VB_OK;
    }
    t_opts = vb_opts_for(stream);
    vb_prof_select(stream);
    // the grouped kernel takes whole 64-token K tiles; the last tokens % 64 rows (ragged B x S) go through the generic kernel below
    const int main_tok = tokens & ~63;
    bool fast = dtype == VB_BF16 && t_opts.nt_kernel != 1;
    for (int i = 0; i < n && fast; ++i)          // alignment / range checks (they do not depend on the token count)
        fast = tn_eligible(dy[i], ld_dy[i], x[i], ld_x[i], (const float*)dw[i], ld_dw[i], n_out[i], n_in[i], 64);
    int done = 0;
    if (fast) {
        TnArgs tg;
        tg.nprob = n; tg.alpha = alpha; tg.alpha_dev = alpha_dev;
        for (int i = 0; i < n; ++i) {
            tg.p[i].A = dy[i]; tg.p[i].B = x[i]; tg.p[i].C = (float*)dw[i];
            tg.p[i].lda = ld_dy[i]; tg.p[i].ldb = ld_x[i]; tg.p[i].ldc = ld_dw[i];
            tg.p[i].Mo = n_out[i]; tg.p[i].Ni = n_in[i];
        }
        if (tn_small_eligible(tg, tokens)) return launch_tn_small(tg, tokens, (hipStream_t)stream);     // few tokens: one launch, no atomics
        if (main_tok >= 64) {
            const int rc = launch_tn_group(tg, main_tok, (hipStream_t)stream);
            if (rc != VB_OK || main_tok == tokens) return rc;
            return launch_tn_tail(tg, main_tok, tokens - main_tok, (hipStream_t)stream);   // the ragged rows of every problem: one launch
        }
    }
    const size_t es = dtype == VB_BF16 ? 2 : 4;
    for (int i = 0; i < n; ++i) {
        int rc = vb_gemm(dtype, VB_F32, VB_KSTRIDED, VB_KSTRIDED, (const char*)dy[i] + (size_t)done * ld_dy[i] * es, ld_dy[i],
                         (const char*)x[i] + (size_t)done * ld_x[i] * es, ld_x[i], dw[i], ld_dw[i],
                         n_out[i], n_in[i], tokens - done, alpha, alpha_dev, nullptr, nullptr, 0, VB_ACT_NONE, nullptr, nullptr, 0,
                         1, nullptr, stream);
        if (rc != VB_OK) return rc;
    }
    return VB_OK;
}

extern "C" int vb_stream_profile(void* stream, int enable) { return vb_prof_enable(stream, enable); }

extern "C" int64_t vb_stream_profile_read(void* stream, double* ms, double* flops, int* key, int64_t max_records) {
    return vb_prof_read(stream, ms, flops, key, max_records);
}

// developer library only (never part of the simulator build)
#ifdef VB_DEV_KNOBS
extern "C" int vb_gemm_set_debug(int bits) { g_debug = bits; return VB_OK; }

// ---- measurement aid: issue-rate ceiling of the two bf16 MFMA shapes at the clocks this chip really holds ---
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ void __launch_bounds__(512) mfma_peak_kernel(float* out, int iters) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (bf16)(0.001f * (threadIdx.x + j)); b[j] = (bf16)(0.002f * (threadIdx.x - j)); }
    if (KIND == 2) {
        // like kind 0, but the operands CHANGE from instruction to instruction (8 pseudo-random register pairs per lane):
        // the switching activity of a real GEMM's operand stream, still without any memory traffic
        bf16x8 av[8], bv[8];
        uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
        for (int q = 0; q < 8; ++q)
            for (int j = 0; j < 8; ++j) {
                h = h * 1664525u + 1013904223u; av[q][j] = (bf16)(((int)(h >> 9) & 0xFFFF) * (1.0f / 32768.0f) - 1.0f);
                h = h * 1664525u + 1013904223u; bv[q][j] = (bf16)(((int)(h >> 9) & 0xFFFF) * (1.0f / 32768.0f) - 1.0f);
            }
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i & 7], bv[(i + 3) & 7], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[(i + 5) & 7], bv[i & 7], acc[i], 0, 0, 0);
        }
        float s = 0; for (int i = 0; i < 16; ++i) s += acc[i][0];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else if (KIND == 0) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0; for (int i = 0; i < 16; ++i) s += acc[i][0];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}
// kind 0: 32 x v_mfma_f32_16x16x32_bf16 per iteration, kind 1: 16 x v_mfma_f32_32x32x16_bf16 (same FLOPs: 524288 per wave-iter),
// kind 2: kind 0 with operands that change every instruction
extern "C" int vb_mfma_peak(int kind, int iters, int blocks, float* out, void* stream) {
    if (kind == 2) hipLaunchKernelGGL(mfma_peak_kernel<2>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, out, iters);
    else if (kind == 0) hipLaunchKernelGGL(mfma_peak_kernel<0>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, out, iters);
    else hipLaunchKernelGGL(mfma_peak_kernel<1>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, out, iters);
    return vb_check_launch();
}

// ---- measurement aid: ceiling of the global -> LDS (LDS-direct) path per CU ---------------------------------
// every wave of every 512-thread block streams `iters` x 1 KiB pieces from an `span`-byte window of src (L2- or
// HBM-resident depending on span) into an LDS ring, keeping DEPTH pieces in flight (counted vmcnt).
template <int DEPTH>
__global__ void __launch_bounds__(512) glds_stream_kernel(const unsigned char* src, long span, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* myring = ring + wave * DEPTH * 1024;
    // each block owns a window; consecutive pieces walk through it (wraps), each lane 16 B
    long off = ((long)blockIdx.x * 8 + wave) * 65536 % span;
    for (int i = 0; i < iters; ++i) {
        const unsigned char* g = src + (off + (long)i * 1024) % span + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(myring + (i % DEPTH) * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (sink && threadIdx.x == 0) sink[blockIdx.x] = *(float*)ring;
}
extern "C" int vb_glds_stream(int depth, const void* src, int64_t span, int iters, int blocks, float* sink, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const unsigned char* p = (const unsigned char*)src;
    switch (depth) {
        case 1: hipLaunchKernelGGL(glds_stream_kernel<1>, dim3(blocks), dim3(512), 8 * 1 * 1024, s, p, (long)span, iters, sink); break;
        case 2: hipLaunchKernelGGL(glds_stream_kernel<2>, dim3(blocks), dim3(512), 8 * 2 * 1024, s, p, (long)span, iters, sink); break;
        case 4: hipLaunchKernelGGL(glds_stream_kernel<4>, dim3(blocks), dim3(512), 8 * 4 * 1024, s, p, (long)span, iters, sink); break;
        case 8: hipLaunchKernelGGL(glds_stream_kernel<8>, dim3(blocks), dim3(512), 8 * 8 * 1024, s, p, (long)span, iters, sink); break;
        case 16: hipLaunchKernelGGL(glds_stream_kernel<16>, dim3(blocks), dim3(512), 8 * 16 * 1024, s, p, (long)span, iters, sink); break;
        default: return VB_ERR_ARG;
    }
    return vb_check_launch();
}

extern "C" int vb_gemm_set_trace(void* device_u64x1024) { g_trace = (unsigned long long*)device_u64x1024; return VB_OK; }
#endif  // VB_DEV_KNOBS

// =================================================================================================
// PROTOTYPE (developer library and simulator only): the split-operand product with its two cross terms on the
// block-scaled fp8 pipe.  Operand image row (ld bf16 elements = 4 K bytes, as in VB_BF16X3): [ hi: K bf16 | hi8: K e4m3 | lo8: K e4m3 ]
// with ONE power-of-two scale per row and fp8 plane (vb_split_f8; per-row scales are as good as per-32-element ones here:
// profiles/r04_x3_cross_term_bits.txt -- and leave the K loop free of scale traffic).  The persistent 256x256 kernel walks
//   K / 64 tiles  hi . hi     on v_mfma_f32_16x16x32_bf16            (128-byte rows of 64 bf16)
//   K / 128 tiles lo8 . hi8   on v_mfma_scale_f32_16x16x128_f8f6f4   (128-byte rows of 128 e4m3)
//   K / 128 tiles hi8 . lo8
// into the same fp32 accumulators: K / 32 tile slots instead of 3 K / 64, and the LDS stage, the copy stream and the fragment reads
// are byte for byte those of the bf16 kernel -- an fp8 operand of the K = 128 instruction IS the pair of 16-byte chunks (lg, 4 + lg)
// of a row that the two bf16 K steps read (vb_rt.h: vb_mma_f8).  Plain epilogue (+ bias), fp32 result, M, N multiples of 256.
// =================================================================================================
#if defined(VB_DEV_KNOBS) || defined(VB_EMU)
struct F8Scales { const unsigned char *a_hi, *a_lo, *b_hi, *b_lo; };     // E8M0 byte per row and plane

VB_KERNEL VB_LAUNCH_BOUNDS(512) gemm_nt_8ph_f8_kernel(GemmArgs g, F8Scales sc) {
    typedef bf16 T;
    constexpr int HALF = 128 * 128, BUF = 4 * HALF;
    constexpr int SLOT_A0 = 0, SLOT_A1 = 1, SLOT_B0 = 2, SLOT_B1 = 3;
    VB_DYN_SMEM(smem);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = vb_uniform(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int li = lane & 15, lg = lane >> 4;
    const int ntiles = g.tiles_m * g.tiles_n;
    const int G = (int)gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    const unsigned char* A = (const unsigned char*)g.A;
    const unsigned char* B = (const unsigned char*)g.B;
    unsigned char* slab = smem + 2 * BUF + wave * EPI8_BYTES_PER_WAVE;
    const int kb = g.K / 64, kf = g.K / 128, nk = kb + 2 * kf;      // tile slots per output tile: bf16 | lo8.hi8 | hi8.lo8

    f32x4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int l3 = lane >> 3;
    const int rowA = (wave >> 2) * 128 + (wave & 3) * 16 + l3;
    const int rowB = (wave >> 1) * 64 + (wave & 1) * 16 + l3;
    const int csrc0 = ((lane & 7) ^ swz(wave * 16 + l3)) * 16;
    const int csrc1 = csrc0 ^ 64;
    unsigned offA[2][2], offB[2][2];
    int ld_j = 0, ld_t = 0;
    auto origin = [&](int j, int& m0, int& n0) {
        const int tile = xcd_remap((int)blockIdx.x + G * j, ntiles);
        m0 = (tile / g.tiles_n) * 256; n0 = (tile % g.tiles_n) * 256;
    };
    auto set_load_tile = [&](int j) {
        int m0, n0;
        origin(j, m0, n0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int a0 = m0 + rowA + h * 64, a1 = a0 + 8, b0 = n0 + rowB + h * 32, b1 = b0 + 8;
            a0 = a0 < g.M ? a0 : g.M - 1; a1 = a1 < g.M ? a1 : g.M - 1;
            b0 = b0 < g.N ? b0 : g.N - 1; b1 = b1 < g.N ? b1 : g.N - 1;
            offA[h][0] = (unsigned)(a0 * (int)g.lda) * 2u + csrc0;
            offA[h][1] = (unsigned)(a1 * (int)g.lda) * 2u + csrc1;
            offB[h][0] = (unsigned)(b0 * (int)g.ldb) * 2u + csrc0;
            offB[h][1] = (unsigned)(b1 * (int)g.ldb) * 2u + csrc1;
        }
    };
    set_load_tile(0);
    // byte offset of tile slot v inside an operand row: the hi plane, then the fp8 planes (hi8 at 2 K, lo8 at 3 K bytes)
    auto off_a = [&](int v) { return v < kb ? v * 128 : (v < kb + kf ? 3 * g.K + (v - kb) * 128 : 2 * g.K + (v - kb - kf) * 128); };
    auto off_b = [&](int v) { return v < kb ? v * 128 : (v < kb + kf ? 2 * g.K + (v - kb) * 128 : 3 * g.K + (v - kb - kf) * 128); };
    auto issueA = [&](int mh, int par) {
        unsigned char* dst = smem + par * BUF + (mh ? SLOT_A1 : SLOT_A0) * HALF + wave * 2048;
        const unsigned char* src = A + off_a(ld_t);
        vb_glds16(src + offA[mh][0], dst);
        vb_glds16(src + offA[mh][1], dst + 1024);
    };
    auto issueB = [&](int nh, int par) {
        unsigned char* dst = smem + par * BUF + (nh ? SLOT_B1 : SLOT_B0) * HALF + wave * 2048;
        const unsigned char* src = B + off_b(ld_t);
        vb_glds16(src + offB[nh][0], dst);
        vb_glds16(src + offB[nh][1], dst + 1024);
    };
    auto ld_advance = [&]() {
        if (++ld_t == nk) { ld_t = 0; ++ld_j; set_load_tile(ld_j); }
    };

    // a fragment = the two 16-byte chunks (lg, 4 + lg) of a 128-byte row, kept as ONE 8-register tuple: its halves are the operands of
    // the two bf16 K steps, the whole tuple is the operand of the K = 128 fp8 instruction (no register copies between the two forms)
    typedef int i32x4v __attribute__((ext_vector_type(4)));
    i32x8 fa[4], fb0[2], fb1[2];
    auto frag8 = [&](const unsigned char* half, int row) {
        const bf16x8 lo = load_frag(half, row, 0, lg, T()), hi = load_frag(half, row, 1, lg, T());
        const i32x4v a = *(const i32x4v*)&lo, b = *(const i32x4v*)&hi;
        return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto readA = [&](const unsigned char* half) {
#pragma unroll
        for (int f = 0; f < 4; ++f) fa[f] = frag8(half, wr * 64 + f * 16 + li);
    };
    auto readB = [&](i32x8 (&fb)[2], const unsigned char* half) {
#pragma unroll
        for (int f = 0; f < 2; ++f) fb[f] = frag8(half, wc * 32 + f * 16 + li);
    };
    auto part = [](const i32x8& v, auto kstag) {            // K step ks of a fragment, as the bf16 MFMA takes it
        constexpr int ks = decltype(kstag)::value;
        const i32x4v h = ks == 0 ? __builtin_shufflevector(v, v, 0, 1, 2, 3) : __builtin_shufflevector(v, v, 4, 5, 6, 7);
        return *(const bf16x8*)&h;
    };
    // per-row scales of the CURRENT output tile, four fragments per register (the instruction selects the byte):
    //   sAh / sAl [mh]: byte f = scale of rows m0 + wr 128 + mh 64 + f 16 + li of the hi8 / lo8 plane of A
    //   sBh / sBl     : byte nh 2 + q = scale of rows n0 + wc 64 + nh 32 + q 16 + li of B
    int sAh[2], sAl[2], sBh, sBl;
    // (vb_split_f8 stores the scale of row r at byte (r & ~63) | ((r & 15) << 2) | ((r >> 4) & 3): the four scales a lane needs for
    //  the fragments f = 0..3 of a 64-row block are one aligned dword -- six loads per output tile instead of 48 byte loads, whose
    //  48 result registers at the tile boundary cost spills that were reloaded inside the K loop)
    auto load_scales = [&](int cj) {
        int m0, n0;
        origin(cj, m0, n0);
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            const int r0 = m0 + wr * 128 + mh * 64 + li * 4;
            sAh[mh] = *(const int*)(sc.a_hi + r0); sAl[mh] = *(const int*)(sc.a_lo + r0);
        }
        const int c0 = n0 + wc * 64 + li * 4;
        sBh = *(const int*)(sc.b_hi + c0); sBl = *(const int*)(sc.b_lo + c0);
    };
    // one quadrant (64 x 32 outputs of this wave) from the fragments in registers; SEG: 0 = bf16 hi.hi, 1 = lo8.hi8, 2 = hi8.lo8
    auto quad = [&](auto mhtag, auto nhtag, i32x8 (&fb)[2], auto segtag) {
        constexpr int mh = decltype(mhtag)::value, nh = decltype(nhtag)::value, SEG = decltype(segtag)::value;
        vb_setprio<1>();
        if constexpr (SEG == 0) {
            vb_static_for<0, 2>([&](auto kstag) {
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc[mh * 4 + f][nh * 2 + q] = vb_mma(part(fa[f], kstag), part(fb[q], kstag), acc[mh * 4 + f][nh * 2 + q]);
            });
        } else {
            const int sa = SEG == 1 ? sAl[mh] : sAh[mh], sb = SEG == 1 ? sBh : sBl;
            vb_static_for<0, 4>([&](auto ftag) {
                constexpr int f = decltype(ftag)::value;
                vb_static_for<0, 2>([&](auto qtag) {
                    constexpr int q = decltype(qtag)::value;
                    acc[mh * 4 + f][nh * 2 + q] = vb_mma_f8_op<f, nh * 2 + q>(fa[f], fb[q], acc[mh * 4 + f][nh * 2 + q], sa, sb);
                });
            });
        }
        vb_setprio<0>();
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    const int GK = my_tiles * nk;
    issueA(0, 0); issueB(0, 0); issueB(1, 0); issueA(1, 0);
    if (GK > 1) { ld_advance(); issueA(0, 1); issueB(0, 1); issueB(1, 1); vb_wait_vmcnt<6>(); }
    else vb_wait_vmcnt<0>();
    load_scales(0);
    vb_phase_barrier();
    if (wr == 1) vb_phase_barrier();               // waves 4-7 run one barrier behind waves 0-3
    typedef std::integral_constant<int, 2> I2;
    // The fp8 MFMAs are plain register operations to the compiler: nothing ties them to the barriers (asm statements that mention no
    // accumulator), and it SANK the first phase's 16 below the phase barrier, the next fragment reads and the next barrier, next to the
    // second phase's 16 -- both wave groups then compute at the same time and read at the same time, and an fp8 slot cost twice a bf16
    // slot (second measurement: 0.92 - 1.01 of the bf16 form's time instead of 2/3).  Redefining a phase's accumulators in front of
    // its closing barrier pins the phase's MFMAs where they are written.
    auto pin_rows = [&](auto mhtag, auto segtag) {
        if constexpr (decltype(segtag)::value != 0) {
            constexpr int mh = decltype(mhtag)::value;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int n = 0; n < 4; ++n) vb_pin(acc[mh * 4 + f][n]);
        }
    };
    int gk = 0;
    // one tile slot held in buffer gk & 1; SEG is a compile-time property of the loop it runs in (three loops per output tile): with the
    // segment as a run-time branch around the two kinds of MFMA blocks the kernel needed 244 bytes of scratch, reloaded behind
    // s_waitcnt vmcnt(0) inside the K loop -- which drains the copy stream: 4.5x SLOWER than the bf16 form (first measurement)
    auto slot = [&](auto segtag) {
        const int par = gk & 1;
        const unsigned char* buf = smem + par * BUF;
        const bool n1 = gk + 1 < GK, n2 = gk + 2 < GK;
        // ---- E
        readB(fb0, buf + SLOT_B0 * HALF);
        readA(buf + SLOT_A0 * HALF);
        readB(fb1, buf + SLOT_B1 * HALF);
        if (n1) { issueA(1, par ^ 1); vb_wait_vmcnt<8>(); } else vb_wait_vmcnt<0>();
        vb_raw_barrier();
        vb_sched_fence();
        quad(I0(), I0(), fb0, segtag);
        quad(I0(), I1(), fb1, segtag);
        pin_rows(I0(), segtag);
        vb_phase_barrier();
        // ---- O
        readA(buf + SLOT_A1 * HALF);
        if (n2) { ld_advance(); issueA(0, par); issueB(0, par); issueB(1, par); vb_wait_vmcnt<8>(); }
        else if (n1) vb_wait_vmcnt<2>();
        else vb_wait_vmcnt<0>();
        vb_raw_barrier();
        vb_sched_fence();
        quad(I1(), I1(), fb1, segtag);
        quad(I1(), I0(), fb0, segtag);
        pin_rows(I1(), segtag);
        vb_phase_barrier();
        ++gk;
    };
    for (int cj = 0; cj < my_tiles; ++cj) {
        for (int c = 0; c < kb; ++c) slot(I0());
        for (int c = 0; c < kf; ++c) slot(I1());
        for (int c = 0; c < kf; ++c) slot(I2());
        int m0, n0;
        origin(cj, m0, n0);
        gemm_epilogue_private<float, float, VB_ACT_NONE, 0>(acc, slab, g, m0 + wr * 128, n0 + wc * 64, lane);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (cj + 1 < my_tiles) load_scales(cj + 1);
    }
    if (wr == 0) vb_phase_barrier();               // balance the stagger barrier
}

extern "C" int vb_gemm_x3f8(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K,
                            const float* bias, const void* sa_hi, const void* sa_lo, const void* sb_hi, const void* sb_lo, void* stream) {
    if (!A || !B || !C || !sa_hi || !sa_lo || !sb_hi || !sb_lo || M <= 0 || N <= 0 || K <= 0) return VB_ERR_ARG;
    if ((M % 256) || (N % 256) || (K % 128) || lda < 2 * (int64_t)K || ldb < 2 * (int64_t)K || (lda % 8) || (ldb % 8) || ldc < N || (ldc % 4))
        return VB_ERR_UNSUPPORTED;
    if ((long)M * lda >= (1L << 30) || (long)N * ldb >= (1L << 30)) return VB_ERR_UNSUPPORTED;
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias; g.alpha = 1.f;
    g.act = VB_ACT_NONE; g.tiles_m = M / 256; g.tiles_n = N / 256; g.x3 = 1; g.kseg = K / 64;
    const int ntiles = g.tiles_m * g.tiles_n;
    int wgs = vb_num_cus();
    if (wgs >= ntiles) wgs = ntiles;
    else if (wgs >= 8) wgs &= ~7;
    constexpr int SM = 2 * 4 * 128 * 128 + 8 * EPI8_BYTES_PER_WAVE;
    F8Scales sc{(const unsigned char*)sa_hi, (const unsigned char*)sa_lo, (const unsigned char*)sb_hi, (const unsigned char*)sb_lo};
    hipStream_t s = (hipStream_t)stream;
    return vb_prof_launch(2.0 * M * N * K, 4 | 16 | 256 | 512, s, [&]() { VB_LAUNCH(gemm_nt_8ph_f8_kernel, dim3((unsigned)wgs), dim3(512), SM, s, g, sc); });
}
#endif
