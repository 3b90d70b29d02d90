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
//   VB_KCONTIG : stored [rows][K], K contiguous   (x, dy as A; W as B in forward)
//   VB_KSTRIDED: stored [K][rows], rows contiguous (W as B in dgrad; dy and x in wgrad)
// K-strided tiles are transposed in registers on their way into LDS, so no transposed copies of
// weights or activations ever exist in HBM.
//
// Tile: 128x128 per workgroup of 4 waves (2x2, 64x64 per wave = 4x4 MFMA 16x16 fragments, 64 fp32
// accumulators per lane).  LDS rows are always 128 B (64 bf16 / 32 fp32 of K) with a 16-byte-chunk XOR
// swizzle; fragments are read with ds_read_b128.  LDS is double-buffered (2 x 32 KB): one barrier
// per K tile, the next tile's loads fly under the current tile's MFMAs.
//   * K-contiguous operand, K a multiple of the tile: global_load_lds_dwordx4 straight into LDS (no
//     VGPR staging, no ds_write); the swizzle is applied to the per-lane SOURCE address because the
//     LDS image of such a load is lane-linear (guide rule 21).
//   * K-strided operand (or a ragged K tail): staged through registers, transposed 8x8 on the way.
// Split-K (wgrad only, fp32 accumulate): the token dimension is the reduction, the output is only
// [out,in] -- 36..144 tiles for BERT-base -- so the reduction is cut into `splits` slices that add
// their partial tile with fire-and-forget fp32 atomics (the output is an accumulator anyway).
// The accumulators leave through LDS so that the epilogue (bias, GELU, GELU', residual addend, fp32
// accumulate) works on 8 consecutive columns per lane and every global store is a full 128-byte row
// segment.
#include "vb_rt.h"
#include "../../include/visualbert_hip.h"
#include <vector>

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
    float* colsum;                // optional fp32 [N]: += column sums of the stored values (bias gradient)
    int debug;                    // ablation bits (measurement only): 1 skip tile loads, 2 skip fragment reads, 4 skip MFMAs
};

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

#ifndef VB_EMU
struct ProfRec { hipEvent_t e0, e1; double flops; int key; };
static std::vector<ProfRec>* g_prof = nullptr;
#endif

// XCD-aware, bijective remap of the linear workgroup id: hardware places workgroup b on XCD b % 8
// (observed, speed only); give each XCD a contiguous run of logical tiles so that tiles sharing an
// A row-panel hit the same L2 (guide T1, bijective form for nwg % 8 != 0).
VB_DEVICE int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// ---------------- epilogue: accumulators -> LDS (wave-private slab) -> row-contiguous vectors.
// mw0 / nw0: first row / column of this wave's 64x64 sub-tile.  Uses __syncthreads(): every wave of the
// workgroup must call it.  (A register-direct epilogue with swapped MFMA operand roles -- 4 consecutive
// columns per lane, no LDS -- was measured 8-15 % SLOWER on MI355X: 8-byte stores and scattered atomics
// lose more than the LDS round trip costs; gpurun_out/gemm_bench_c.txt.)
template <typename T, typename TO>
VB_DEVICE void gemm_epilogue(f32x4 (&acc)[4][4], unsigned char* smem, const GemmArgs& g, int mw0, int nw0,
                             int wave, int lane) {
    const int li = lane & 15, lg = lane >> 4;
    unsigned char* slab = smem + wave * EPI_BYTES_PER_WAVE;
    TO* C = (TO*)g.C;
    const float alpha = g.alpha_dev ? g.alpha * g.alpha_dev[0] : g.alpha;
    float cs[8];                                      // column sums of this lane's 8 columns (bias gradient)
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = 0.f;
    // a lane always owns the same 8 columns: bias is loaded once, ahead of the first barrier
    const int cc = lane & 7;
    const int ncol = nw0 + cc * 8;
    const bool full = (ncol + 8 <= g.N);
    const int nv = full ? 8 : (ncol < g.N ? g.N - ncol : 0);
    float bb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bb[j] = 0.f;
    if (g.bias && ncol < g.N) {
        if (full && (((uintptr_t)(g.bias + ncol)) & 15) == 0) load8(bb, g.bias + ncol);
        else for (int j = 0; j < nv; ++j) bb[j] = g.bias[ncol + j];
    }
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
                        vb_atomic_add_noret((float*)C + (long)m * g.ldc + n, alpha * *(const float*)(slab + row * EPI_PITCH + lane * 4));
                }
            }
            continue;
        }
        // per-row operands of this pass (4 rows per lane): all global loads are issued back to back BEFORE any
        // store of the pass -- interleaved with the stores they would each pay a full memory round trip,
        // because the compiler must assume C may alias them (measured: ~8 us of a 12 us K=64 launch)
        float xa[4][8], xd[4][8], xc[4][8];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int m = mw0 + pass * 32 + it * 8 + (lane >> 3);
            const bool ok = m < g.M && ncol < g.N;
            if (g.act == VB_ACT_GELU_GRAD) {
                const T* ai = (const T*)g.aux_in + (long)m * g.ld_aux + ncol;
                if (ok && full && (g.ld_aux & 7) == 0) load8(xa[it], ai);
                else for (int j = 0; j < 8; ++j) xa[it][j] = (ok && j < nv) ? to_f32(ai[j]) : 0.f;
            }
            if (g.addend) {
                const T* ad = (const T*)g.addend + (long)m * g.ld_addend + ncol;
                if (ok && full && (g.ld_addend & 7) == 0) load8(xd[it], ad);
                else for (int j = 0; j < 8; ++j) xd[it][j] = (ok && j < nv) ? to_f32(ad[j]) : 0.f;
            }
            if (g.accumulate) {
                const TO* cp = C + (long)m * g.ldc + ncol;
                if (ok && full && (g.ldc & 7) == 0) load8(xc[it], cp);
                else for (int j = 0; j < 8; ++j) xc[it][j] = (ok && j < nv) ? to_f32(cp[j]) : 0.f;
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3);
            const int m = mw0 + pass * 32 + row;
            const int n = ncol;
            if (m >= g.M || n >= g.N) continue;
            float v[8];
            {
                f32x4 lo = *(const f32x4*)(slab + row * EPI_PITCH + cc * 32);
                f32x4 hi = *(const f32x4*)(slab + row * EPI_PITCH + cc * 32 + 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * alpha + bb[j];
            if (g.act == VB_ACT_GELU) {
                if (g.aux_out) {                                  // pre-activation, kept for backward
                    T* ao = (T*)g.aux_out + (long)m * g.ld_aux + n;
                    if (full && (g.ld_aux & 7) == 0) store8(ao, v);
                    else for (int j = 0; j < nv; ++j) ao[j] = from_f32<T>(v[j]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
            } else if (g.act == VB_ACT_TANH) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = tanhf(v[j]);
            } else if (g.act == VB_ACT_GELU_GRAD) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_f(xa[it][j]);
            }
            if (g.addend) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += xd[it][j];
            }
            if (g.accumulate) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += xc[it][j];
            }
            TO* cp = C + (long)m * g.ldc + n;
            if (g.debug & 64) { if (v[0] == 123.456f) store8(cp, v); }     // ablation: no global stores
            else if (full && (g.ldc & 7) == 0) store8(cp, v);
            else for (int j = 0; j < nv; ++j) cp[j] = from_f32<TO>(v[j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j < nv) cs[j] += v[j];
        }
    }
    if (g.colsum && g.splits == 1) {
        // lanes with equal (lane & 7) own the same 8 columns on different rows: reduce over lane bits 3..5,
        // then one fp32 atomic per column per wave
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            cs[j] += __shfl_xor(cs[j], 8);
            cs[j] += __shfl_xor(cs[j], 16);
            cs[j] += __shfl_xor(cs[j], 32);
        }
        if (lane < 8) {
            const int n = nw0 + lane * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (n + j < g.N) vb_atomic_add_noret(g.colsum + n + j, cs[j]);
        }
    }
}

template <typename T, typename TO, int AL, int BL>
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

    const int nk_all = (g.K + BK - 1) / BK;
    const int kt0 = split * g.kt_per_split;
    const int kt1 = (kt0 + g.kt_per_split < nk_all) ? kt0 + g.kt_per_split : nk_all;

    // issue(): start moving K tile `kt` towards LDS buffer `buf`; commit(): finish it for the
    // register-staged operands (LDS-direct copies need no commit, only the barrier)
    auto issue = [&](int kt, int buf) {
        const int k0 = kt * BK;
        unsigned char* la = smem + buf * 2 * TILE_BYTES;
        unsigned char* lb = la + TILE_BYTES;
        if constexpr (AL == VB_KCONTIG) {
            if (glds_a) glds_kcontig<T>(la, A, g.lda, g.M, m0, k0, wave, lane);
            else gload_kcontig<T>(ra, A, g.lda, g.M, g.K, m0, k0, t);
        } else {
            if (a_active) gload_kstrided<T>(ra, A, g.lda, g.M, g.K, m0, k0, t);
        }
        if constexpr (BL == VB_KCONTIG) {
            if (glds_b) glds_kcontig<T>(lb, B, g.ldb, g.N, n0, k0, wave, lane);
            else gload_kcontig<T>(rb, B, g.ldb, g.N, g.K, n0, k0, t);
        } else {
            if (b_active) gload_kstrided<T>(rb, B, g.ldb, g.N, g.K, n0, k0, t - B_TBASE);
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

    gemm_epilogue<T, TO>(acc, smem, g, m0 + wm * 64, n0 + wn * 64, wave, lane);
}

// ---- optional per-launch HIP-event timing (bench.py's roofline leg) -----------------------------------
// Events are recorded on the stream the kernel is launched on, immediately around the launch, so the
// elapsed time is the kernel's own duration even when the host is the bottleneck.

template <typename T, typename TO, int AL, int BL>
int launch_gemm(const GemmArgs& g, hipStream_t stream) {
    dim3 grid((unsigned)(g.tiles_m * g.tiles_n * g.splits)), block(NT);
#ifndef VB_EMU
    if (g_prof) {
        ProfRec r;
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return VB_ERR_LAUNCH;
        r.flops = 2.0 * g.M * g.N * g.K;
        r.key = (sizeof(T) == 2 ? 0 : 8) | (sizeof(TO) == 4 ? 4 : 0) | (AL << 1) | BL;
        (void)hipEventRecord(r.e0, stream);
        VB_LAUNCH((gemm_kernel<T, TO, AL, BL>), grid, block, SMEM_BYTES, stream, g);
        (void)hipEventRecord(r.e1, stream);
        g_prof->push_back(r);
        return vb_check_launch();
    }
#endif
    VB_LAUNCH((gemm_kernel<T, TO, AL, BL>), grid, block, SMEM_BYTES, stream, g);
    return vb_check_launch();
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
    const T* a[A_INSTR];
    const T* b[B_INSTR];
};
template <typename T, int WM>
VB_DEVICE void fast_setup(FastPtrs<T, WM>& p, const T* A, const T* B, const GemmArgs& g, int m0, int n0, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < FastPtrs<T, WM>::A_INSTR; ++i) {
        const int row = (wave * FastPtrs<T, WM>::A_INSTR + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz(row);
        int grow = m0 + row;
        grow = grow < g.M ? grow : g.M - 1;
        p.a[i] = A + (long)grow * g.lda + c * TT<T>::EPC;
    }
#pragma unroll
    for (int i = 0; i < FastPtrs<T, WM>::B_INSTR; ++i) {
        const int row = (wave * FastPtrs<T, WM>::B_INSTR + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz(row);
        int grow = n0 + row;
        grow = grow < g.N ? grow : g.N - 1;
        p.b[i] = B + (long)grow * g.ldb + c * TT<T>::EPC;
    }
}
template <typename T, int WM>
VB_DEVICE void fast_issue(unsigned char* stage, const FastPtrs<T, WM>& p, int k0, int wave) {
    unsigned char* la = stage;
    unsigned char* lb = stage + FastPtrs<T, WM>::BMX * 128;
#pragma unroll
    for (int i = 0; i < FastPtrs<T, WM>::A_INSTR; ++i)
        vb_glds16(p.a[i] + k0, la + (wave * FastPtrs<T, WM>::A_INSTR + i) * 8 * 128);
#pragma unroll
    for (int i = 0; i < FastPtrs<T, WM>::B_INSTR; ++i)
        vb_glds16(p.b[i] + k0, lb + (wave * FastPtrs<T, WM>::B_INSTR + i) * 8 * 128);
}

template <typename T, typename TO, int WM, int STAGES, int DBG = 0>
VB_KERNEL VB_LAUNCH_BOUNDS(WM * 128) gemm_nt_pipe_kernel(GemmArgs g) {
    constexpr int NW = WM * 2, BMX = WM * 64;
    constexpr int BK = TT<T>::BK, KSTEPS = TT<T>::KSTEPS;
    constexpr int STAGE_BYTES = (BMX + 128) * 128;
    constexpr int PER_TILE = (BMX / 8) / NW + (128 / 8) / NW;       // LDS-direct instructions per wave per tile
    VB_DYN_SMEM(smem);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int nwg = g.tiles_m * g.tiles_n;
    const int tile = xcd_remap((int)blockIdx.x, nwg);
    const int m0 = (tile / g.tiles_n) * BMX, n0 = (tile % g.tiles_n) * BN;
    const T* A = (const T*)g.A;
    const T* B = (const T*)g.B;

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    FastPtrs<T, WM> ptrs;
    fast_setup<T, WM>(ptrs, A, B, g, m0, n0, wave, lane);
    typename VecOf<T>::v8 fa[4], fb[4];
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) fast_issue<T, WM>(smem + s * STAGE_BYTES, ptrs, s * BK, wave);

    for (int kt = 0; kt < nk; ++kt) {
        // tile kt must have landed; tiles kt+1 .. kt+STAGES-2 (if they exist) may stay in flight
        if (STAGES >= 3 && kt + STAGES - 2 < nk) vb_wait_vmcnt<(STAGES - 2) * PER_TILE>();
        else if (STAGES >= 4 && kt + STAGES - 3 < nk) vb_wait_vmcnt<(STAGES >= 4 ? STAGES - 3 : 0) * PER_TILE>();
        else vb_wait_vmcnt<0>();
        if (!(DBG & 32) || (kt & 3) == 0)
            vb_raw_barrier(); // (a) everyone's part of tile kt is in LDS, (b) everyone finished reading tile kt-1
        // DBG is a compile-time ablation / experiment mask (0 in production): 1 skip tile loads, 2 skip fragment
        // reads, 4 skip MFMAs, 8 raise wave priority around the MFMA block, 16 issue the next tile's copies
        // after the first K step instead of before it
        if (!(DBG & 16) && kt + STAGES - 1 < nk && !(DBG & 1))
            fast_issue<T, WM>(smem + ((kt + STAGES - 1) % STAGES) * STAGE_BYTES, ptrs, (kt + STAGES - 1) * BK, wave);
        const unsigned char* ldsA = smem + (kt % STAGES) * STAGE_BYTES;
        const unsigned char* ldsB = ldsA + BMX * 128;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            if (!(DBG & 2) || kt == 0) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) fa[mi] = load_frag(ldsA, wm * 64 + mi * 16 + li, ks, lg, T());
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) fb[ni] = load_frag(ldsB, wn * 64 + ni * 16 + li, ks, lg, T());
            }
            if (!(DBG & 4)) {
#ifndef VB_EMU
                if (DBG & 8) __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = vb_mma(fa[mi], fb[ni], acc[mi][ni]);
#ifndef VB_EMU
                if (DBG & 8) __builtin_amdgcn_s_setprio(0);
#endif
            } else {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[mi][0][0] += to_f32(fa[mi][0]) + to_f32(fb[mi][0]);   // keep the reads live
            }
            if ((DBG & 16) && ks == 0 && kt + STAGES - 1 < nk)
                fast_issue<T, WM>(smem + ((kt + STAGES - 1) % STAGES) * STAGE_BYTES, ptrs, (kt + STAGES - 1) * BK, wave);
        }
    }
    gemm_epilogue<T, TO>(acc, smem, g, m0 + wm * 64, n0 + wn * 64, wave, lane);
}

// =================================================================================================
// Interleaved variant (256x128 tile, 8 waves, 3 LDS stages).  Same data flow as the pipelined kernel, but
// inside every K tile the next tile-but-one's LDS-direct copies and the second K step's fragment reads
// are woven BETWEEN the first K step's MFMAs (an MFMA occupies the matrix pipe for 16 cycles = ~4 issue
// slots; the other 3 are free for LDS / VMEM instructions of the same wave).  The copies are
// unconditional (a clamped, redundant tile near the end) so the whole K tile is one basic block the
// scheduler hints (sched_group_barrier) can order.
// =================================================================================================
template <typename T, typename TO>
VB_KERNEL VB_LAUNCH_BOUNDS(512) gemm_nt_weave_kernel(GemmArgs g) {
    constexpr int WM = 4, BMX = 256, STAGES = 3;
    constexpr int BK = TT<T>::BK, KSTEPS = TT<T>::KSTEPS;
    constexpr int STAGE_BYTES = (BMX + 128) * 128;
    constexpr int PER_TILE = (BMX / 8) / 8 + (128 / 8) / 8;        // 6 LDS-direct instructions per wave per tile
    VB_DYN_SMEM(smem);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int nwg = g.tiles_m * g.tiles_n;
    const int tile = xcd_remap((int)blockIdx.x, nwg);
    const int m0 = (tile / g.tiles_n) * BMX, n0 = (tile % g.tiles_n) * BN;
    const T* A = (const T*)g.A;
    const T* B = (const T*)g.B;

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    FastPtrs<T, WM> ptrs;
    fast_setup<T, WM>(ptrs, A, B, g, m0, n0, wave, lane);
    fast_issue<T, WM>(smem, ptrs, 0, wave);
    fast_issue<T, WM>(smem + STAGE_BYTES, ptrs, (nk > 1 ? 1 : 0) * BK, wave);

    for (int kt = 0; kt < nk; ++kt) {
        vb_wait_vmcnt<PER_TILE>();            // tile kt has landed; tile kt+1 may still be in flight
        vb_raw_barrier();                     // tile kt visible to all; everyone done with tile kt-1 (stage (kt+2)%3)
        const unsigned char* ldsA = smem + (kt % STAGES) * STAGE_BYTES;
        const unsigned char* ldsB = ldsA + BMX * 128;
        typename VecOf<T>::v8 fa0[4], fb0[4], fa1[4], fb1[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) fa0[mi] = load_frag(ldsA, wm * 64 + mi * 16 + li, 0, lg, T());
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) fb0[ni] = load_frag(ldsB, wn * 64 + ni * 16 + li, 0, lg, T());
        // tile kt+2 (clamped) into the stage tile kt-1 occupied
        const int kn = kt + 2 < nk ? kt + 2 : nk - 1;
        fast_issue<T, WM>(smem + ((kt + 2) % STAGES) * STAGE_BYTES, ptrs, kn * BK, wave);
        if (KSTEPS == 2) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) fa1[mi] = load_frag(ldsA, wm * 64 + mi * 16 + li, 1, lg, T());
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) fb1[ni] = load_frag(ldsB, wn * 64 + ni * 16 + li, 1, lg, T());
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = vb_mma(fa0[mi], fb0[ni], acc[mi][ni]);
        if (KSTEPS == 2) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = vb_mma(fa1[mi], fb1[ni], acc[mi][ni]);
        }
        // order: 8 fragment reads (K step 0) first, then weave {MFMA, VMEM} x 6, {MFMA, DS read} x 8, the rest MFMAs
        VB_SCHED_GROUP(0x100, 8);
#pragma unroll
        for (int i = 0; i < 6; ++i) { VB_SCHED_GROUP(0x8, 1); VB_SCHED_GROUP(0x20, 1); }
#pragma unroll
        for (int i = 0; i < 8; ++i) { VB_SCHED_GROUP(0x8, 1); VB_SCHED_GROUP(0x100, 1); }
        VB_SCHED_GROUP(0x8, 18);
    }
    vb_wait_vmcnt<0>();                        // the clamped tail copies must not land on the epilogue's slabs
    gemm_epilogue<T, TO>(acc, smem, g, m0 + wm * 64, n0 + wn * 64, wave, lane);
}

template <typename T, typename TO>
int launch_weave(GemmArgs g, hipStream_t stream) {
    constexpr int SM = 3 * (256 + 128) * 128;
    g.tiles_m = (g.M + 255) / 256;
    dim3 grid((unsigned)(g.tiles_m * g.tiles_n)), block(512);
#ifndef VB_EMU
    if (g_prof) {
        ProfRec r;
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return VB_ERR_LAUNCH;
        r.flops = 2.0 * g.M * g.N * g.K;
        r.key = (sizeof(T) == 2 ? 0 : 8) | (sizeof(TO) == 4 ? 4 : 0);
        (void)hipEventRecord(r.e0, stream);
        VB_LAUNCH((gemm_nt_weave_kernel<T, TO>), grid, block, SM, stream, g);
        (void)hipEventRecord(r.e1, stream);
        g_prof->push_back(r);
        return vb_check_launch();
    }
#endif
    VB_LAUNCH((gemm_nt_weave_kernel<T, TO>), grid, block, SM, stream, g);
    return vb_check_launch();
}

// =================================================================================================
// Ping-pong variant of the 256x128 / 8-wave kernel.  Waves w and w+4 share a SIMD; the two wave groups
// (waves 0-3, waves 4-7) run ONE SEGMENT out of phase, so on every SIMD one wave is in an MFMA segment
// (16 MFMAs = one K step of its 64x64 tile) while its partner is in a fragment-read segment
// (8 ds_read_b128).  Segments are separated by raw barriers (4 per K tile); the matrix pipe of each SIMD
// then sees back-to-back MFMA segments instead of "both waves read, both waves compute".
//     segment 4kt   : G0 reads (kt, k-step 0)   | G1 MFMAs (kt-1, k-step 1)      + issue tile kt+1
//     segment 4kt+1 : G0 MFMAs (kt, 0)          | G1 reads (kt, 0)
//     segment 4kt+2 : G0 reads (kt, 1)          | G1 MFMAs (kt, 0)
//     segment 4kt+3 : G0 MFMAs (kt, 1)          | G1 reads (kt, 1)
// =================================================================================================
template <typename T, typename TO>
VB_KERNEL VB_LAUNCH_BOUNDS(512) gemm_nt_pingpong_kernel(GemmArgs g) {
    constexpr int WM = 4, BMX = 256;
    constexpr int BK = TT<T>::BK, KSTEPS = TT<T>::KSTEPS;
    constexpr int STAGE_BYTES = (BMX + 128) * 128;
    VB_DYN_SMEM(smem);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int grp = wave >> 2;
    const int nwg = g.tiles_m * g.tiles_n;
    const int tile = xcd_remap((int)blockIdx.x, nwg);
    const int m0 = (tile / g.tiles_n) * BMX, n0 = (tile % g.tiles_n) * BN;
    const T* A = (const T*)g.A;
    const T* B = (const T*)g.B;

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    typename VecOf<T>::v8 fa[4], fb[4];

    const int nk = g.K / BK;
    FastPtrs<T, WM> ptrs;
    fast_setup<T, WM>(ptrs, A, B, g, m0, n0, wave, lane);
    fast_issue<T, WM>(smem, ptrs, 0, wave);

    auto rd = [&](int kt, int ks) {
        const unsigned char* ldsA = smem + (kt & 1) * STAGE_BYTES;
        const unsigned char* ldsB = ldsA + BMX * 128;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) fa[mi] = load_frag(ldsA, wm * 64 + mi * 16 + li, ks, lg, T());
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) fb[ni] = load_frag(ldsB, wn * 64 + ni * 16 + li, ks, lg, T());
    };
    auto mm = [&]() {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = vb_mma(fa[mi], fb[ni], acc[mi][ni]);
    };
    static_assert(KSTEPS == 2 || KSTEPS == 1, "");

    for (int kt = 0; kt <= nk; ++kt) {
        // ---- segment 4kt
        if (kt < nk) vb_wait_vmcnt<0>();           // my share of tile kt (issued one K tile ago) has landed
        vb_raw_barrier();
        if (kt + 1 < nk) fast_issue<T, WM>(smem + ((kt + 1) & 1) * STAGE_BYTES, ptrs, (kt + 1) * BK, wave);
        if (grp == 0) { if (kt < nk) rd(kt, 0); }
        else { if (kt > 0) mm(); }
        if (kt == nk) break;
        // ---- segment 4kt+1
        vb_raw_barrier();
        if (grp == 0) mm(); else rd(kt, 0);
        if (KSTEPS == 2) {
            // ---- segment 4kt+2
            vb_raw_barrier();
            if (grp == 0) rd(kt, 1); else mm();
            // ---- segment 4kt+3
            vb_raw_barrier();
            if (grp == 0) mm(); else rd(kt, 1);
        }
    }
    gemm_epilogue<T, TO>(acc, smem, g, m0 + wm * 64, n0 + wn * 64, wave, lane);
}

template <typename T, typename TO>
int launch_pingpong(GemmArgs g, hipStream_t stream) {
    constexpr int SM = 2 * (256 + 128) * 128;
    g.tiles_m = (g.M + 255) / 256;
    dim3 grid((unsigned)(g.tiles_m * g.tiles_n)), block(512);
#ifndef VB_EMU
    if (g_prof) {
        ProfRec r;
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return VB_ERR_LAUNCH;
        r.flops = 2.0 * g.M * g.N * g.K;
        r.key = (sizeof(T) == 2 ? 0 : 8) | (sizeof(TO) == 4 ? 4 : 0);
        (void)hipEventRecord(r.e0, stream);
        VB_LAUNCH((gemm_nt_pingpong_kernel<T, TO>), grid, block, SM, stream, g);
        (void)hipEventRecord(r.e1, stream);
        g_prof->push_back(r);
        return vb_check_launch();
    }
#endif
    VB_LAUNCH((gemm_nt_pingpong_kernel<T, TO>), grid, block, SM, stream, g);
    return vb_check_launch();
}

// =================================================================================================
// 256x256 tile, 8 waves as 2 (M) x 4 (N), 128x64 per wave (8x4 fragments, 128 fp32 accumulators per lane),
// two 64 KB LDS stages.  Per flop it moves 2/3 of the L2->LDS bytes of the 256x128 tile (measured wall:
// ~15-19 TB/s of tile traffic saturates the load path, profiles/r01_gemm_ablation.txt) and reads 25 %
// fewer fragment bytes per MFMA.  Used when the grid still fills the chip (>= ~200 tiles).
// =================================================================================================
template <typename T, typename TO>
VB_KERNEL VB_LAUNCH_BOUNDS(512) gemm_nt_big_kernel(GemmArgs g) {
    constexpr int BK = TT<T>::BK, KSTEPS = TT<T>::KSTEPS, EPC = TT<T>::EPC;
    constexpr int STAGE_BYTES = 2 * 256 * 128;
    VB_DYN_SMEM(smem);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 15, lg = lane >> 4;
    const int nwg = g.tiles_m * g.tiles_n;
    const int tile = xcd_remap((int)blockIdx.x, nwg);
    const int m0 = (tile / g.tiles_n) * 256, n0 = (tile % g.tiles_n) * 256;
    const T* A = (const T*)g.A;
    const T* B = (const T*)g.B;

    f32x4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // LDS-direct copies: 32 + 32 one-KiB instructions per K tile, 4 + 4 per wave
    const T* pa[4];
    const T* pb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz(row);
        int ga = m0 + row; ga = ga < g.M ? ga : g.M - 1;
        int gb = n0 + row; gb = gb < g.N ? gb : g.N - 1;
        pa[i] = A + (long)ga * g.lda + c * EPC;
        pb[i] = B + (long)gb * g.ldb + c * EPC;
    }
    auto issue = [&](int kt) {
        unsigned char* la = smem + (kt & 1) * STAGE_BYTES;
        unsigned char* lb = la + 256 * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) vb_glds16(pa[i] + kt * BK, la + (wave * 4 + i) * 8 * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i) vb_glds16(pb[i] + kt * BK, lb + (wave * 4 + i) * 8 * 128);
    };

    const int nk = g.K / BK;
    issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        vb_wait_vmcnt<0>();
        vb_raw_barrier();
        if (kt + 1 < nk) issue(kt + 1);
        const unsigned char* ldsA = smem + (kt & 1) * STAGE_BYTES;
        const unsigned char* ldsB = ldsA + 256 * 128;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            typename VecOf<T>::v8 fb[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) fb[ni] = load_frag(ldsB, wn * 64 + ni * 16 + li, ks, lg, T());
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                typename VecOf<T>::v8 fa[4];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) fa[mi] = load_frag(ldsA, wm * 128 + (mh * 4 + mi) * 16 + li, ks, lg, T());
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc[mh * 4 + mi][ni] = vb_mma(fa[mi], fb[ni], acc[mh * 4 + mi][ni]);
            }
        }
    }
    // two 64x64 sub-tiles per wave through the shared epilogue
    f32x4(&lo)[4][4] = *reinterpret_cast<f32x4(*)[4][4]>(&acc[0]);
    f32x4(&hi)[4][4] = *reinterpret_cast<f32x4(*)[4][4]>(&acc[4]);
    gemm_epilogue<T, TO>(lo, smem, g, m0 + wm * 128, n0 + wn * 64, wave, lane);
    gemm_epilogue<T, TO>(hi, smem, g, m0 + wm * 128 + 64, n0 + wn * 64, wave, lane);
}

template <typename T, typename TO>
int launch_big(GemmArgs g, hipStream_t stream) {
    constexpr int SM = 2 * 2 * 256 * 128;
    static_assert(8 * EPI_BYTES_PER_WAVE <= SM, "epilogue slabs must fit");
    g.tiles_m = (g.M + 255) / 256;
    g.tiles_n = (g.N + 255) / 256;
    dim3 grid((unsigned)(g.tiles_m * g.tiles_n)), block(512);
#ifndef VB_EMU
    if (g_prof) {
        ProfRec r;
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return VB_ERR_LAUNCH;
        r.flops = 2.0 * g.M * g.N * g.K;
        r.key = (sizeof(T) == 2 ? 0 : 8) | (sizeof(TO) == 4 ? 4 : 0);
        (void)hipEventRecord(r.e0, stream);
        VB_LAUNCH((gemm_nt_big_kernel<T, TO>), grid, block, SM, stream, g);
        (void)hipEventRecord(r.e1, stream);
        g_prof->push_back(r);
        return vb_check_launch();
    }
#endif
    VB_LAUNCH((gemm_nt_big_kernel<T, TO>), grid, block, SM, stream, g);
    return vb_check_launch();
}

template <typename T, typename TO, int WM, int STAGES>
int launch_pipe(GemmArgs g, hipStream_t stream, double* flops_key_unused = nullptr) {
    (void)flops_key_unused;
    constexpr int BMX = WM * 64;
    constexpr int SM = STAGES * (BMX + 128) * 128;
    static_assert(WM * 2 * EPI_BYTES_PER_WAVE <= SM, "epilogue slabs must fit");
    g.tiles_m = (g.M + BMX - 1) / BMX;
    dim3 grid((unsigned)(g.tiles_m * g.tiles_n)), block(WM * 128);
#ifndef VB_EMU
    if (g_prof) {
        ProfRec r;
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return VB_ERR_LAUNCH;
        r.flops = 2.0 * g.M * g.N * g.K;
        r.key = (sizeof(T) == 2 ? 0 : 8) | (sizeof(TO) == 4 ? 4 : 0);
        (void)hipEventRecord(r.e0, stream);
        VB_LAUNCH((gemm_nt_pipe_kernel<T, TO, WM, STAGES>), grid, block, SM, stream, g);
        (void)hipEventRecord(r.e1, stream);
        g_prof->push_back(r);
        return vb_check_launch();
    }
#endif
    if constexpr (sizeof(T) == 2 && sizeof(TO) == 2 && WM == 4 && STAGES == 2) {
        switch (g.debug & 63) {                  // ablation / experiment builds of the production kernel
            case 0: break;
#define VB_DBG_CASE(D) case D: VB_LAUNCH((gemm_nt_pipe_kernel<T, TO, WM, STAGES, D>), grid, block, SM, stream, g); return vb_check_launch();
            VB_DBG_CASE(1) VB_DBG_CASE(2) VB_DBG_CASE(3) VB_DBG_CASE(4) VB_DBG_CASE(5) VB_DBG_CASE(6) VB_DBG_CASE(8) VB_DBG_CASE(16) VB_DBG_CASE(24) VB_DBG_CASE(35) VB_DBG_CASE(32) VB_DBG_CASE(37) VB_DBG_CASE(38)
#undef VB_DBG_CASE
            default: return VB_ERR_ARG;
        }
    }
    VB_LAUNCH((gemm_nt_pipe_kernel<T, TO, WM, STAGES>), grid, block, SM, stream, g);
    return vb_check_launch();
}

// variant of the pipelined kernel: tens = waves in M (2 -> 128-row tile, 4 -> 256-row tile), units = stages.
// 0 = use the generic kernel.
static int g_nt_variant = 1;     // 1 = auto: 256x256 tile where the grid still fills the chip and K or N is large, else 256x128
static int g_debug = 0;

template <typename T, typename TO>
int dispatch_pipe(const GemmArgs& g, hipStream_t s) {
    int variant = g_nt_variant;
    if (variant == 1) {
        // measured on MI355X (profiles/r01_gemm_variant_sweep_b128.txt): the 256x256 tile wins on long-K and
        // wide-N shapes once it yields >= ~200 workgroups; the 256x128 tile wins everywhere else
        const long t256 = (long)((g.M + 255) / 256) * ((g.N + 255) / 256);
        variant = (t256 >= 200 && (g.K >= 1536 || g.N >= 3072) && g.N <= 8192) ? 88 : 42;
    }
    switch (variant) {
        case 22: return launch_pipe<T, TO, 2, 2>(g, s);
        case 23: return launch_pipe<T, TO, 2, 3>(g, s);
        case 24: return launch_pipe<T, TO, 2, 4>(g, s);
        case 42: return launch_pipe<T, TO, 4, 2>(g, s);
        case 43: return launch_pipe<T, TO, 4, 3>(g, s);
        case 44: return launch_pingpong<T, TO>(g, s);
        case 88: return launch_big<T, TO>(g, s);
        case 53: return launch_weave<T, TO>(g, s);
        default: return VB_ERR_UNSUPPORTED;
    }
}

template <typename T>
int dispatch(int out_dtype_is_f32, int al, int bl, const GemmArgs& g, hipStream_t s) {
    if (al == VB_KCONTIG && bl == VB_KCONTIG) {
        if (g.fast_a && g.fast_b && g.splits == 1 && g_nt_variant != 0)
            return out_dtype_is_f32 ? dispatch_pipe<T, float>(g, s) : dispatch_pipe<T, T>(g, s);
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
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return VB_ERR_ARG;
    if (dtype != VB_F32 && dtype != VB_BF16) return VB_ERR_ARG;
    if (out_dtype != VB_F32 && out_dtype != dtype) return VB_ERR_ARG;
    const int epc = dtype == VB_BF16 ? 8 : 4;
    // vector loads are 16 bytes: leading dimensions and base pointers must keep them aligned
    if ((lda % 8) || (ldb % 8) || (((uintptr_t)A | (uintptr_t)B) & 15)) return VB_ERR_ARG;
    (void)epc;
    if (act == VB_ACT_GELU_GRAD && !aux_in) return VB_ERR_ARG;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.addend = addend; g.ld_addend = ld_addend; g.aux_in = aux_in; g.aux_out = aux_out;
    g.ld_aux = ld_aux; g.alpha = alpha; g.alpha_dev = alpha_dev; g.colsum = colsum_out; g.act = act; g.accumulate = accumulate;
    g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
    g.debug = g_debug;
    const int bk = dtype == VB_BF16 ? 64 : 32;
    const int nk = (K + bk - 1) / bk;
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
    const int of32 = (out_dtype == VB_F32) ? 1 : 0;
    if (dtype == VB_BF16) return dispatch<bf16>(of32 && true, a_layout, b_layout, g, s);
    return dispatch<float>(1, a_layout, b_layout, g, s);
}

extern "C" int vb_gemm_profile(int enable) {
#ifndef VB_EMU
    if (g_prof) {
        for (auto& r : *g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
        delete g_prof;
        g_prof = nullptr;
    }
    if (enable) g_prof = new std::vector<ProfRec>();
#else
    (void)enable;
#endif
    return VB_OK;
}

extern "C" int64_t vb_gemm_profile_read(double* ms, double* flops, int* key, int64_t max_records) {
#ifndef VB_EMU
    if (!g_prof) return 0;
    int64_t n = 0;
    for (auto& r : *g_prof) {
        if (n >= max_records) break;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return -1;   // caller must synchronise first
        if (ms) ms[n] = t;
        if (flops) flops[n] = r.flops;
        if (key) key[n] = r.key;
        ++n;
    }
    return n;
#else
    (void)ms; (void)flops; (void)key; (void)max_records;
    return 0;
#endif
}

extern "C" int vb_gemm_set_variant(int variant) {
    if (variant != 0 && variant != 1 && variant != 22 && variant != 23 && variant != 24 && variant != 42 && variant != 43 && variant != 44 && variant != 88 && variant != 53) return VB_ERR_ARG;
    g_nt_variant = variant;
    return VB_OK;
}

extern "C" int vb_gemm_set_debug(int bits) { g_debug = bits; return VB_OK; }

// ---- measurement aid: issue-rate ceiling of the two bf16 MFMA shapes at the clocks this chip really holds ---
#ifndef VB_EMU
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ void __launch_bounds__(512) mfma_peak_kernel(float* out, int iters) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (bf16)(0.001f * (threadIdx.x + j)); b[j] = (bf16)(0.002f * (threadIdx.x - j)); }
    if (KIND == 0) {
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
#endif
// kind 0: 32 x v_mfma_f32_16x16x32_bf16 per iteration, kind 1: 16 x v_mfma_f32_32x32x16_bf16 (same FLOPs: 524288 per wave-iter)
extern "C" int vb_mfma_peak(int kind, int iters, int blocks, float* out, void* stream) {
#ifndef VB_EMU
    if (kind == 0) hipLaunchKernelGGL(mfma_peak_kernel<0>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, out, iters);
    else hipLaunchKernelGGL(mfma_peak_kernel<1>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, out, iters);
    return vb_check_launch();
#else
    (void)kind; (void)iters; (void)blocks; (void)out; (void)stream;
    return VB_ERR_UNSUPPORTED;
#endif
}
