// attention.hip -- fused multi-head self-attention over the concatenated [text ; visual] sequence.
//
// Replaces BertSelfAttention.forward after the Q/K/V Linears
//   (pytorch_pretrained_bert/modeling.py:236-256: transpose_for_scores, QK^T / sqrt(d), + additive
//    mask, softmax, dropout on the probabilities, .V, merge heads) and its autograd backward.
// The S x S score / probability tensors never exist in HBM: what is saved for backward is the
// per-row log-sum-exp (fp32) and, when dropout is on, one keep-bit per probability.
//
// Layout in HBM: qkv is token-major [B*S, 3H] exactly as the packed Q|K|V Linear writes it (head h of
// Q at columns h*64.., K at H + h*64.., V at 2H + h*64..: one 128-byte line per (token, head) in bf16);
// ctx / dctx are [B*S, H]; dqkv mirrors qkv.  Head size is 64 (BERT-base 768/12, config-1 128/2).
//
// MFMA formulation (16x16x32 fragments, wave = 64 lanes).  S = 164 fits on chip, so one workgroup
// owns one (batch, head) and stages K / V (or Q / dO) in LDS once.
//   forward  : S^T = K Q^T  (lane <-> query, registers <-> keys)  -> softmax reductions are in-lane +
//              2 shuffles; P^T feeds O^T = V^T P^T straight from registers (no LDS round trip) because
//              the MFMA K index may be any permutation as long as A and B agree (guide T12).
//   backward A (per query block): S^T, dP^T = V dO^T, dS^T, dQ^T = K^T dS^T, and D = rowsum(P o dP).
//   backward B (per key block)  : S = Q K^T, dP = dO V^T (lane <-> key, registers <-> queries), then
//              dV^T += dO^T P and dK^T += Q^T dS with the query index as the MFMA K index.
// Transposed operands (V^T, K^T, Q^T, dO^T) are built while staging into LDS.
#include "vb_rt.h"
#include "../../include/visualbert_hip.h"
#include "vb_opts.h"

struct xf32 { float v; };             // element tag of the split mode (see LT<xf32> below): fp32 in memory
struct x3frag { bf16x8 hi, lo; };     // its MFMA operand
template <> struct VecOf<xf32> { typedef x3frag v8; typedef f32x4 v4; };
// the three-product MFMA of the split mode, small terms first (global scope: overloads the vb_mma set of vb_rt.h)
VB_DEVICE f32x4 vb_mma(const x3frag& a, const x3frag& b, f32x4 c) {
    c = vb_mma(a.lo, b.hi, c);
    c = vb_mma(a.hi, b.lo, c);
    return vb_mma(a.hi, b.hi, c);
}

namespace {

constexpr int NT = 256;
constexpr int D = 64;

template <typename T> struct LT;   // LDS tile geometry
// TPAD: padding of a transposed tile's rows, in bytes.  Round 6 ("17 % LDS bank conflicts, no commit has changed the
// tile layouts") tried the pitch the documented bank model asks for -- 8-byte fragment reads, one row per lane of a 16-lane group, two column
// groups per half-wave on 64 banks: conflict-free at a pitch of 4 x odd dwords, i.e. 16 bytes of padding for every tile width, where
// 8 bytes (2 x odd dwords) leave one 2-way conflict per read -- and it LOST on every kernel (profiles/r06_attn_pitch_ab.txt, three product
// builds on one box, B = 1024, S = 164, p = 0.1): one-pass backward + bias gradient 842 -> 954 us alone (1 059 vs 959 us in the step),
// forward 293 -> 313 us, step 117.5 -> 119.5 ms; with only the backward's dS tile and K^T image re-pitched (VB_ATTN_BPAD) 842 -> 902 us.
// The counter's conflict cycles are not where these kernels lose time, and whatever the wider rows cost (the 2- and 4-byte staging stores
// land 8-way instead of 4-way) outweighs them.  8 bytes stay; the macros keep the arms buildable (tools/build_variant.sh).
#ifndef VB_ATTN_TPAD
#define VB_ATTN_TPAD 8
#endif
template <> struct LT<bf16> {
    static constexpr int RB = 128, CPR = 8, KPT = 2, TPAD = VB_ATTN_TPAD;
    static constexpr bool SPLIT = false;
    VB_DEVICE int sw(int row) { return (row >> 1) & 7; }    // conflict-free for ds_read_b128 fragment reads (see gemm.hip)
};
template <> struct LT<float> {
    static constexpr int RB = 256, CPR = 16, KPT = 1, TPAD = 16;
    static constexpr bool SPLIT = false;
    VB_DEVICE int sw(int row) { return (row & 7) << 1; }
};
// Split mode (VB_BF16X3): the tensors in HBM are fp32, every MFMA operand is the pair (hi, lo) of bf16 planes with
// hi = bf16(x), lo = bf16(x - hi), and a product is the three MFMAs hi*hi + lo*hi + hi*lo (the lo*lo term, 2^-18
// relative, is dropped as in gemm.hip).  The split is made ONCE per element while staging: a row-major LDS row is
// [64 hi | 64 lo] (256 B, the fp32 row's size), a transposed tile is a hi tile followed by a lo tile of bf16 pitch, so
// the LDS budgets -- and with them the sequence limits of every kernel form -- are those of the fp32 instantiation.
template <> struct LT<xf32> {
    static constexpr int RB = 256, CPR = 16, TPAD = VB_ATTN_TPAD;
    static constexpr bool SPLIT = true;
    VB_DEVICE int sw(int row) { return (row & 7) << 1; }
};
template <typename T> VB_DEVICE int rm_off(int row, int c) { return row * LT<T>::RB + ((c ^ LT<T>::sw(row)) << 4); }
template <typename T> constexpr int tr_pitch(int nk) { return nk * (int)sizeof(T) + LT<T>::TPAD; }
template <typename T> constexpr int rm_bytes(int nrows) { return nrows * LT<T>::RB; }
template <typename T> constexpr int tr_bytes(int nk) { return D * tr_pitch<T>(nk); }
template <> constexpr int tr_pitch<xf32>(int nk) { return nk * 2 + LT<xf32>::TPAD; }      // pitch of ONE bf16 plane
template <> constexpr int tr_bytes<xf32>(int nk) { return 2 * D * tr_pitch<xf32>(nk); }   // hi tile, then lo tile

VB_DEVICE void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[j] = (bf16)a[j]; lo[j] = (bf16)(a[j] - (float)hi[j]);
        hi[4 + j] = (bf16)b[j]; lo[4 + j] = (bf16)(b[j] - (float)hi[4 + j]);
    }
}
// ---- staging, bf16 ------------------------------------------------------------------------------------------------
// A thread first issues ALL its loads of a [NROWS][64] tile -- pair-item i = rows (2j, 2j+1) x 16-byte chunk dc -- and only
// then stores, row-major and / or transposed, from the same registers.  (A rolled "load, wait, store" loop, which is what
// this replaced, pays one HBM round trip per trip: 9 serial round trips per workgroup in the forward kernel.)
// zero fill as a select on a value that was loaded UNCONDITIONALLY (see frag_g)
VB_DEVICE u32x4 zsel(bool ok, const u32x4& x) { return u32x4{ok ? x[0] : 0u, ok ? x[1] : 0u, ok ? x[2] : 0u, ok ? x[3] : 0u}; }
// NTH: threads of the workgroup that stage the tile together (256; 512 for the long-sequence forms whose tiles leave one workgroup per CU)
template <int NROWS, int NTH = NT>
struct PairTile {
    static constexpr int ITEMS = (NROWS / 2) * 8, PER = (ITEMS + NTH - 1) / NTH;
    u32x4 x0[PER], x1[PER];
};
template <int NROWS, int NTH = NT>
VB_DEVICE void pair_load(PairTile<NROWS, NTH>& p, const bf16* X, long ldx, long row0, int c0, int S, int t) {
    // unconditional loads from clamped rows first, the zero fill of rows >= S as selects afterwards (see frag_g)
#pragma unroll
    for (int k = 0; k < PairTile<NROWS, NTH>::PER; ++k) {
        const int idx = t + k * NTH;
        const int dc = idx & 7, r = (idx >> 3) * 2;
        p.x0[k] = *(const u32x4*)(X + (row0 + (r < S ? r : S - 1)) * ldx + c0 + dc * 8);
        p.x1[k] = *(const u32x4*)(X + (row0 + (r + 1 < S ? r + 1 : S - 1)) * ldx + c0 + dc * 8);
    }
#pragma unroll
    for (int k = 0; k < PairTile<NROWS, NTH>::PER; ++k) {
        const int idx = t + k * NTH;
        const int r = (idx >> 3) * 2;
        const bool in = idx < PairTile<NROWS, NTH>::ITEMS;
        p.x0[k] = zsel(in && r < S, p.x0[k]);
        p.x1[k] = zsel(in && r + 1 < S, p.x1[k]);
    }
}
template <int NROWS, int NTH = NT>
VB_DEVICE void pair_store_rm(const PairTile<NROWS, NTH>& p, unsigned char* lds, int t) {
#pragma unroll
    for (int k = 0; k < PairTile<NROWS, NTH>::PER; ++k) {
        const int idx = t + k * NTH;
        if (idx >= PairTile<NROWS, NTH>::ITEMS) continue;
        const int dc = idx & 7, r = (idx >> 3) * 2;
        *(u32x4*)(lds + rm_off<bf16>(r, dc)) = p.x0[k];
        *(u32x4*)(lds + rm_off<bf16>(r + 1, dc)) = p.x1[k];
    }
}
template <int NROWS, int NTH = NT>
VB_DEVICE void pair_store_tr(const PairTile<NROWS, NTH>& p, unsigned char* lds, int t) {
    const int pitch = tr_pitch<bf16>(NROWS);
#pragma unroll
    for (int k = 0; k < PairTile<NROWS, NTH>::PER; ++k) {
        const int idx = t + k * NTH;
        if (idx >= PairTile<NROWS, NTH>::ITEMS) continue;
        const int dc = idx & 7, r = (idx >> 3) * 2;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t a = p.x0[k][w], b = p.x1[k][w];
            *(uint32_t*)(lds + (dc * 8 + 2 * w) * pitch + r * 2) = (a & 0xFFFFu) | (b << 16);
            *(uint32_t*)(lds + (dc * 8 + 2 * w + 1) * pitch + r * 2) = (a >> 16) | (b & 0xFFFF0000u);
        }
    }
}

// ---- staging, fp32 sources (fp32 and split modes) ------------------------------------------------------------------------
// The same for the 4-byte element types -- where it matters more: their tiles fill the LDS, so ONE workgroup per CU has nothing
// to hide a round trip behind.  Item i = rows (2j, 2j+1) x 8-element chunk dc: four 16-byte loads, all issued before the first store.
// NTH: threads of the workgroup that stage the tile together (the 4-byte kernels run with up to 768: see attn_fwd_kernel)
template <int NROWS, int NTH = NT>
struct PairTile32 {
    static constexpr int ITEMS = (NROWS / 2) * 8, PER = (ITEMS + NTH - 1) / NTH;
    f32x4 a0[PER], b0[PER], a1[PER], b1[PER];              // row r: elements 0-3 / 4-7 of the chunk; row r + 1 likewise
};
VB_DEVICE f32x4 zsel4(bool ok, const f32x4& x) { return f32x4{ok ? x[0] : 0.f, ok ? x[1] : 0.f, ok ? x[2] : 0.f, ok ? x[3] : 0.f}; }
template <int NROWS, typename T, int NTH = NT>
VB_DEVICE void pair_load32(PairTile32<NROWS, NTH>& p, const T* X, long ldx, long row0, int c0, int S, int t) {
#pragma unroll
    for (int k = 0; k < PairTile32<NROWS, NTH>::PER; ++k) {     // unconditional loads from clamped rows (see frag_g)
        const int idx = t + k * NTH;
        const int dc = idx & 7, r = (idx >> 3) * 2;
        const float* p0 = (const float*)(X + (row0 + (r < S ? r : S - 1)) * ldx + c0 + dc * 8);
        const float* p1 = (const float*)(X + (row0 + (r + 1 < S ? r + 1 : S - 1)) * ldx + c0 + dc * 8);
        p.a0[k] = *(const f32x4*)p0; p.b0[k] = *(const f32x4*)(p0 + 4);
        p.a1[k] = *(const f32x4*)p1; p.b1[k] = *(const f32x4*)(p1 + 4);
    }
#pragma unroll
    for (int k = 0; k < PairTile32<NROWS, NTH>::PER; ++k) {
        const int r = ((t + k * NTH) >> 3) * 2;
        p.a0[k] = zsel4(r < S, p.a0[k]); p.b0[k] = zsel4(r < S, p.b0[k]);
        p.a1[k] = zsel4(r + 1 < S, p.a1[k]); p.b1[k] = zsel4(r + 1 < S, p.b1[k]);
    }
}
template <int NROWS, typename T, int NTH = NT>
VB_DEVICE void pair_store_rm32(const PairTile32<NROWS, NTH>& p, unsigned char* lds, int t) {
#pragma unroll
    for (int k = 0; k < PairTile32<NROWS, NTH>::PER; ++k) {
        const int idx = t + k * NTH;
        if (idx >= PairTile32<NROWS, NTH>::ITEMS) continue;
        const int dc = idx & 7, r = (idx >> 3) * 2;
        if constexpr (LT<T>::SPLIT) {
            bf16x8 h0, l0, h1, l1;
            split8(p.a0[k], p.b0[k], h0, l0);
            split8(p.a1[k], p.b1[k], h1, l1);
            *(bf16x8*)(lds + rm_off<T>(r, dc)) = h0;     *(bf16x8*)(lds + rm_off<T>(r, 8 + dc)) = l0;
            *(bf16x8*)(lds + rm_off<T>(r + 1, dc)) = h1; *(bf16x8*)(lds + rm_off<T>(r + 1, 8 + dc)) = l1;
        } else {
            *(f32x4*)(lds + rm_off<T>(r, 2 * dc)) = p.a0[k];     *(f32x4*)(lds + rm_off<T>(r, 2 * dc + 1)) = p.b0[k];
            *(f32x4*)(lds + rm_off<T>(r + 1, 2 * dc)) = p.a1[k]; *(f32x4*)(lds + rm_off<T>(r + 1, 2 * dc + 1)) = p.b1[k];
        }
    }
}
template <int NROWS, typename T, int NTH = NT>
VB_DEVICE void pair_store_tr32(const PairTile32<NROWS, NTH>& p, unsigned char* lds, int t) {
    constexpr int pitch = tr_pitch<T>(NROWS);
#pragma unroll
    for (int k = 0; k < PairTile32<NROWS, NTH>::PER; ++k) {
        const int idx = t + k * NTH;
        if (idx >= PairTile32<NROWS, NTH>::ITEMS) continue;
        const int dc = idx & 7, r = (idx >> 3) * 2;
        if constexpr (LT<T>::SPLIT) {
            typedef bf16 pair_t __attribute__((ext_vector_type(2)));
            unsigned char* ldl = lds + D * pitch;
            bf16x8 h0, l0, h1, l1;
            split8(p.a0[k], p.b0[k], h0, l0);
            split8(p.a1[k], p.b1[k], h1, l1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                *(pair_t*)(lds + (dc * 8 + j) * pitch + r * 2) = pair_t{h0[j], h1[j]};
                *(pair_t*)(ldl + (dc * 8 + j) * pitch + r * 2) = pair_t{l0[j], l1[j]};
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *(f32x2*)(lds + (dc * 8 + j) * pitch + r * 4) = f32x2{p.a0[k][j], p.a1[k][j]};
                *(f32x2*)(lds + (dc * 8 + 4 + j) * pitch + r * 4) = f32x2{p.b0[k][j], p.b1[k][j]};
            }
        }
    }
}
// one tile type for every element type: the loads of a [NROWS][64] tile in registers, stored row-major and / or transposed
template <typename T, int NROWS> struct TileOf { typedef PairTile32<NROWS> type; };
template <int NROWS> struct TileOf<bf16, NROWS> { typedef PairTile<NROWS> type; };
template <int NROWS, typename T>
VB_DEVICE void tile_load(typename TileOf<T, NROWS>::type& p, const T* X, long ldx, long row0, int c0, int S, int t) {
    if constexpr (sizeof(T) == 2) pair_load<NROWS>(p, X, ldx, row0, c0, S, t);
    else pair_load32<NROWS, T>(p, X, ldx, row0, c0, S, t);
}
template <int NROWS, typename T>
VB_DEVICE void tile_store_rm(const typename TileOf<T, NROWS>::type& p, unsigned char* lds, int t) {
    if constexpr (sizeof(T) == 2) pair_store_rm<NROWS>(p, lds, t);
    else pair_store_rm32<NROWS, T>(p, lds, t);
}
template <int NROWS, typename T>
VB_DEVICE void tile_store_tr(const typename TileOf<T, NROWS>::type& p, unsigned char* lds, int t) {
    if constexpr (sizeof(T) == 2) pair_store_tr<NROWS>(p, lds, t);
    else pair_store_tr32<NROWS, T>(p, lds, t);
}

// ---- fragments -------------------------------------------------------------------------------
// 8 consecutive d (d0 = ks*32 + g*8) of row `row` of a row-major LDS tile
VB_DEVICE bf16x8 frag_rm(const unsigned char* lds, int row, int ks, int g, bf16) {
    return *(const bf16x8*)(lds + rm_off<bf16>(row, ks * 4 + g));
}
VB_DEVICE f32x8 frag_rm(const unsigned char* lds, int row, int ks, int g, float) {
    const int c = 2 * (ks * 4 + g);
    f32x4 lo = *(const f32x4*)(lds + rm_off<float>(row, c));
    f32x4 hi = *(const f32x4*)(lds + rm_off<float>(row, c + 1));
    return f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
VB_DEVICE x3frag frag_rm(const unsigned char* lds, int row, int ks, int g, xf32) {
    return x3frag{*(const bf16x8*)(lds + rm_off<xf32>(row, ks * 4 + g)), *(const bf16x8*)(lds + rm_off<xf32>(row, 8 + ks * 4 + g))};
}
// same 8 elements straight from global memory (row pointer already offset to the head's column 0)
// rowp must be readable even when !ok (callers clamp the row): the load is UNCONDITIONAL and the zeroing a select on its
// result.  A load under `if (ok)` compiles to an exec-masked branch with a zero-fill of the destination registers, and the
// compiler then serialises neighbouring loads with s_waitcnt vmcnt(0) (write-after-write on those registers).
VB_DEVICE bf16x8 frag_g(const bf16* rowp, int ks, int g, bool ok) {
    const u32x4 x = zsel(ok, *(const u32x4*)(rowp + ks * 32 + g * 8));
    return *(const bf16x8*)&x;
}
VB_DEVICE f32x8 frag_g(const float* rowp, int ks, int g, bool ok) {
    f32x4 lo = *(const f32x4*)(rowp + ks * 32 + g * 8), hi = *(const f32x4*)(rowp + ks * 32 + g * 8 + 4);
    return f32x8{ok ? lo[0] : 0.f, ok ? lo[1] : 0.f, ok ? lo[2] : 0.f, ok ? lo[3] : 0.f,
                 ok ? hi[0] : 0.f, ok ? hi[1] : 0.f, ok ? hi[2] : 0.f, ok ? hi[3] : 0.f};
}
VB_DEVICE x3frag frag_g(const xf32* rowp, int ks, int g, bool ok) {
    const float* p = (const float*)rowp + ks * 32 + g * 8;
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    x3frag f;
    split8(f32x4{ok ? a[0] : 0.f, ok ? a[1] : 0.f, ok ? a[2] : 0.f, ok ? a[3] : 0.f},
           f32x4{ok ? b[0] : 0.f, ok ? b[1] : 0.f, ok ? b[2] : 0.f, ok ? b[3] : 0.f}, f.hi, f.lo);
    return f;
}
// A fragment from a transposed tile: row d, MFMA k index (g, j) <-> r = 32*ks + 16*(j>>2) + 4*g + (j&3)
VB_DEVICE bf16x8 frag_tr(const unsigned char* lds, int pitch, int d, int ks, int g, bf16) {
    const unsigned char* p = lds + d * pitch + (32 * ks + 4 * g) * 2;
    bf16x4 lo = *(const bf16x4*)p, hi = *(const bf16x4*)(p + 32);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
VB_DEVICE f32x8 frag_tr(const unsigned char* lds, int pitch, int d, int ks, int g, float) {
    const unsigned char* p = lds + d * pitch + (32 * ks + 4 * g) * 4;
    f32x4 lo = *(const f32x4*)p, hi = *(const f32x4*)(p + 64);
    return f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
VB_DEVICE x3frag frag_tr(const unsigned char* lds, int pitch, int d, int ks, int g, xf32) {
    return x3frag{frag_tr(lds, pitch, d, ks, g, bf16()), frag_tr(lds + D * pitch, pitch, d, ks, g, bf16())};
}
// B fragment from two C-layout fragments (regs of frag 2ks -> j 0..3, frag 2ks+1 -> j 4..7)
VB_DEVICE void pack_b(bf16x8& o, const f32x4& a, const f32x4& b) {
    o = bf16x8{(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
}
VB_DEVICE void pack_b(f32x8& o, const f32x4& a, const f32x4& b) {
    o = f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
VB_DEVICE void store4(bf16* p, const f32x4& v) { *(bf16x4*)p = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]}; }
VB_DEVICE void store4(float* p, const f32x4& v) { *(f32x4*)p = v; }
VB_DEVICE void pack_b(x3frag& o, const f32x4& a, const f32x4& b) { split8(a, b, o.hi, o.lo); }
VB_DEVICE void store4(xf32* p, const f32x4& v) { *(f32x4*)p = v; }
// Two adjacent 16-column blocks (df = 2 j, 2 j + 1) of one row of a [.., 64]-wide head: lane (li, lg) holds columns 4 lg .. 4 lg + 3 of
// each block (ve, vo); row32 = this lane's row + 32 j.  bf16 with a 16-byte-aligned row (`wide`, wave-uniform): ONE 16-byte store per
// lane after a v_permlane16_swap pair instead of two 8-byte stores (store instructions are issue-bound, ~64 cycles each whatever
// their width: guide T21).  The swaps run on ALL lanes (they exchange data between the four lane groups of the same row); only the
// store is predicated.  Other element types: the two plain stores.
VB_DEVICE void store4x2(bf16* row32, int lg, const f32x4& ve, const f32x4& vo, bool ok, bool wide) {
    if (wide) {
        const bf16x4 e = bf16x4{(bf16)ve[0], (bf16)ve[1], (bf16)ve[2], (bf16)ve[3]}, o = bf16x4{(bf16)vo[0], (bf16)vo[1], (bf16)vo[2], (bf16)vo[3]};
        u32x2 eu = __builtin_bit_cast(u32x2, e), ou = __builtin_bit_cast(u32x2, o);
        uint32_t e0 = eu[0], e1 = eu[1], o0 = ou[0], o1 = ou[1];
        vb_permlane16_swap(e0, o0);
        vb_permlane16_swap(e1, o1);
        if (ok) *(u32x4*)(row32 + vb_wide_col(lg)) = u32x4{e0, e1, o0, o1};
    } else if (ok) {
        store4(row32 + lg * 4, ve);
        store4(row32 + 16 + lg * 4, vo);
    }
}
template <typename T> VB_DEVICE void store4x2(T* row32, int lg, const f32x4& ve, const f32x4& vo, bool ok, bool) {
    if (ok) { store4(row32 + lg * 4, ve); store4(row32 + 16 + lg * 4, vo); }
}
// is a bf16 row pointer family (base, pitch in elements) 16-byte aligned at every head's column 0?  (heads are 64 elements = 128 B wide)
VB_DEVICE bool wide_ok(const void* base, long ld_elems) { return ((((uintptr_t)base) | (uintptr_t)(ld_elems * 2)) & 15) == 0; }

struct AttnArgs {
    const void* qkv; const float* mask_add; void* ctx; float* lse; uint64_t* keepbits;   // forward
    const void* dctx; void* dqkv; float* dsum; const void* ctx_fwd;                       // backward
    float* bias_ws;                  // one-pass backward: per-sample column sums of dQ | dK | dV, fp32 [B][3H] (or NULL)
    // split-operand mode, self-attention only (vb_attn_*_sp): bf16 hi | lo images of the fp32 results written next to them --
    // ctx_sp [B*S, 2H] (half = H), dqkv_sp [B*S, 6H] (half = 3H; dQ | dK | dV column blocks like dqkv) -- or NULL
    bf16* ctx_sp; bf16* dqkv_sp;
    int sp_only;                     // with an image: do not write the fp32 result at all (only GEMMs read it, and they read the image)
    // General form (self- AND cross-attention; the forward and the two-pass backward kernels read only these): queries come
    // from q [B*Sq rows, pitch ldq], keys / values from k, v [B*S rows, pitch ldk / ldv] -- for self-attention three column
    // blocks of the packed qkv matrix, for cross-attention (LXRT: language attends to vision and back) different tensors
    // with different sequence lengths.  Pointers are to column 0 of head 0; pitches in elements.
    const void *q, *k, *v; void *dq, *dk, *dv;
    long ldq, ldk, ldv, ldc, lddo, lddq, lddk, lddv;
    int Sq;                          // queries per sample (S = keys per sample)
    int B, S, nh; float scale; float p; float inv_keep; uint32_t thresh; uint32_t stream; uint64_t seed;
};

// keep-bits: one uint64 per (b, h, q, g = key&15>>2, word w): nibble (kf & 15) of word (kf >> 4)
// holds keys kf*16 + g*4 + {0..3}.  Written by forward, read by both backward passes.
VB_DEVICE long keep_index(const AttnArgs& a, int bh, int q, int g, int w, int nw) {
    return (((long)bh * a.Sq + q) * 4 + g) * nw + w;
}

// =================================================================================================
// forward
// =================================================================================================
// NKF = key fragments of 16 that hold real keys (scores, softmax and dropout run over exactly these: S = 164 -> 11, where
// rounding up to an even 12 spent 1/12 of the per-probability VALU work -- the bound of this kernel -- on padding);
// the P.V MFMAs consume keys 32 at a time, so V^T (and the packed probabilities) are padded to NKV = even(NKF) fragments
// with an all-zero upper half.
// PF (bf16): the Q fragment of a wave's next query block is fetched while the current block computes and pinned until the
// top of the next iteration (vb_pin) -- one HBM round trip per block leaves the wave's critical path.
// EXACT: every one of the NKF key fragments (and NKS 32-key steps) holds at least one real key -- the launcher's promise for
// S = 161..176 -- so the per-fragment "is it inside the sequence" branches (46 scalar branches per query block) go, and a
// chain's first MFMA takes the literal 0 as its C operand instead of four zeroed registers.
// NTH: threads per workgroup = 64 x waves.  bf16 tiles are small enough for three 4-wave workgroups per CU; the 4-byte element
// types (fp32, split mode) fill the LDS with ONE workgroup's K / V, so that workgroup brings the waves itself: 12 waves (3 per
// SIMD, the same 168-register budget) walk the query blocks side by side over the same tiles instead of 4 waves in 3 rounds.
template <typename T, int NKF, bool PF = false, bool EXACT = false, int NTH = NT>
VB_KERNEL VB_LAUNCH_BOUNDS2(NTH, (NTH == NT ? 3 : 1)) attn_fwd_kernel(AttnArgs a) {
    constexpr int NKV = (NKF + 1) / 2 * 2;
    constexpr int NK = NKF * 16, NKVK = NKV * 16, NKS = NKV / 2, NW = (NKF + 15) / 16;
    VB_DYN_SMEM(smem);
    unsigned char* ldsK = smem;
    unsigned char* ldsVT = ldsK + rm_bytes<T>(NK);
    float* ldsMask = (float*)(ldsVT + tr_bytes<T>(NKVK));
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4;
    const int bh = blockIdx.x, b = bh / a.nh, h = bh % a.nh;
    const int S = a.S, Sq = a.Sq;                          // keys / queries per sample
    const long rowk = (long)b * S, rowq = (long)b * Sq;
    const T* Qp = (const T*)a.q;
    const T* Kp = (const T*)a.k;
    const T* Vp = (const T*)a.v;
    const bool wide_c = sizeof(T) == 2 && wide_ok(a.ctx, a.ldc);     // wave-uniform: 16-byte context stores (store4x2)

    if constexpr (sizeof(T) == 2) {
        PairTile<NK, NTH> tk;
        PairTile<NKVK, NTH> tv;
        pair_load<NK, NTH>(tk, Kp, a.ldk, rowk, h * D, S, t);
        pair_load<NKVK, NTH>(tv, Vp, a.ldv, rowk, h * D, S, t);
        pair_store_rm<NK, NTH>(tk, ldsK, t);
        pair_store_tr<NKVK, NTH>(tv, ldsVT, t);             // rows >= S arrive as zeros: the padded key columns are 0
    } else {
        PairTile32<NK, NTH> tk;
        PairTile32<NKVK, NTH> tv;
        pair_load32<NK, T, NTH>(tk, Kp, a.ldk, rowk, h * D, S, t);
        pair_load32<NKVK, T, NTH>(tv, Vp, a.ldv, rowk, h * D, S, t);
        pair_store_rm32<NK, T, NTH>(tk, ldsK, t);
        pair_store_tr32<NKVK, T, NTH>(tv, ldsVT, t);
    }
    for (int k = t; k < NK; k += NTH) ldsMask[k] = k < S ? a.mask_add[(long)b * S + k] : -INFINITY;
    __syncthreads();

    const int nqf = (Sq + 15) / 16;
    u32x4 qn[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
    auto fetch_q = [&](int qf) {                           // unconditional, from a clamped row
        const int q = qf * 16 + li;
        const T* qrow = Qp + (rowq + (q < Sq ? q : Sq - 1)) * a.ldq + h * D;
        qn[0] = *(const u32x4*)(qrow + lg * 8);
        qn[1] = *(const u32x4*)(qrow + 32 + lg * 8);
    };
    if constexpr (PF) fetch_q(wave);
    for (int qf = wave; qf < nqf; qf += NTH / 64) {
        const int q = qf * 16 + li;
        const bool qok = q < Sq;
        typename VecOf<T>::v8 qb[2];
        if constexpr (PF) {
            vb_pin(qn[0]); vb_pin(qn[1]);
            const u32x4 z0 = zsel(qok, qn[0]), z1 = zsel(qok, qn[1]);
            qb[0] = *(const typename VecOf<T>::v8*)&z0;
            qb[1] = *(const typename VecOf<T>::v8*)&z1;
            fetch_q(qf + NTH / 64);                               // past the last block: a clamped (valid, unused) row
        } else {
            const T* qrow = Qp + (rowq + (qok ? q : 0)) * a.ldq + h * D;
            qb[0] = frag_g(qrow, 0, lg, qok);
            qb[1] = frag_g(qrow, 1, lg, qok);
        }

        f32x4 st[NKF];
        float m = -INFINITY;
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            if (EXACT || kf * 16 < S) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) acc = vb_mma(frag_rm(ldsK, kf * 16 + li, ks, lg, T()), qb[ks], acc);
            }
            const f32x4 mk = *(const f32x4*)(ldsMask + kf * 16 + lg * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[r] = acc[r] * a.scale + mk[r];      // -inf for keys >= S
                m = fmaxf(m, acc[r]);
            }
            st[kf] = acc;
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float sum = 0.f;
        // exp(s - m) = exp2(s * log2(e) - m * log2(e)): one packed fma per two scores instead of a subtract and a multiply each
        const f32x2 l2e = vb_splat2(1.44269504088896340736f), negm = vb_splat2(-m * 1.44269504088896340736f);
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f32x2 x = vb_fma2(f32x2{st[kf][r], st[kf][r + 1]}, l2e, negm);
                st[kf][r] = fast_exp2(x[0]); st[kf][r + 1] = fast_exp2(x[1]);
                sum += st[kf][r]; sum += st[kf][r + 1];
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        const float invk = inv * a.inv_keep;               // normalisation and 1/(1-p) in one factor
        if (a.lse && lg == 0 && qok) a.lse[(long)bh * Sq + q] = m + logf(sum);

        // normalise, dropout (keep-bits recorded), round to T as the MFMA B operand
        uint64_t bits[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) bits[w] = 0;
#pragma unroll
        for (int kp = 0; kp < NKS; ++kp) {
            if (a.p > 0.f) {
                // one generator call covers this lane's 8 probabilities of fragments 2kp, 2kp+1
                const uint64_t grp = ((((uint64_t)bh * Sq + q) * 4 + lg) << 5) + kp;
                Rand8 rnd = vb_dropout_bits8(a.seed, grp, a.stream);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int kf = 2 * kp + (e >> 2), r = e & 3;
                    if (kf >= NKF) continue;                // the padding half of an odd fragment count
                    const bool keep = rand8_keep(rnd, e, a.thresh);
                    st[kf][r] *= keep ? invk : 0.f;         // (select the factor, then one -- packable -- multiply)
                    if (keep) bits[kf >> 4] |= (uint64_t)1 << ((kf & 15) * 4 + r);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (2 * kp + (e >> 2) < NKF) st[2 * kp + (e >> 2)][e & 3] *= inv;
            }
        }
        if (a.p > 0.f && a.keepbits && qok) {
#pragma unroll
            for (int w = 0; w < NW; ++w) a.keepbits[keep_index(a, bh, q, lg, w, NW)] = bits[w];
        }
        typename VecOf<T>::v8 pb[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (2 * ks + 1 < NKF) pack_b(pb[ks], st[2 * ks], st[2 * ks + 1]);
            else pack_b(pb[ks], st[2 * ks], f32x4{0.f, 0.f, 0.f, 0.f});
        }

        T* crow = (T*)a.ctx + (rowq + (qok ? q : 0)) * a.ldc + h * D;
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {                     // two 16-column blocks at a time: one 16-byte store per lane (store4x2)
            f32x4 acc2[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int df = 2 * dj + e;
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    if (EXACT || ks * 32 < S)
                        acc = vb_mma(frag_tr(ldsVT, tr_pitch<T>(NKVK), df * 16 + li, ks, lg, T()), pb[ks], acc);
                }
                acc2[e] = acc;                               // lane: query q, d = df*16 + lg*4 + 0..3
                if constexpr (std::is_same<T, xf32>::value) {
                    if (a.ctx_sp && qok) store_split4(a.ctx_sp + (rowq + q) * (2 * a.ldc) + h * D + df * 16 + lg * 4, a.ldc, acc);
                }
            }
            store4x2(crow + dj * 32, lg, acc2[0], acc2[1], qok && !a.sp_only, wide_c);
        }
    }
}

// =================================================================================================
// backward, pass A: dQ and D = rowsum(P o dP)   (per query block, all keys)
// =================================================================================================
template <typename T, int NKF, int NTH = NT>
VB_KERNEL VB_LAUNCH_BOUNDS(NTH) attn_bwd_dq_kernel(AttnArgs a) {
    constexpr int NK = NKF * 16, NKS = NKF / 2, NW = (NKF + 15) / 16;
    VB_DYN_SMEM(smem);
    unsigned char* ldsK = smem;
    unsigned char* ldsV = ldsK + rm_bytes<T>(NK);
    unsigned char* ldsKT = ldsV + rm_bytes<T>(NK);
    float* ldsMask = (float*)(ldsKT + tr_bytes<T>(NK));
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4;
    const int bh = blockIdx.x, b = bh / a.nh, h = bh % a.nh;
    const int S = a.S, Sq = a.Sq;                          // keys / queries per sample
    const long rowk = (long)b * S, rowq = (long)b * Sq;
    const T* Qp = (const T*)a.q;
    const T* Kp = (const T*)a.k;
    const T* Vp = (const T*)a.v;

    if constexpr (sizeof(T) == 2) {
        PairTile<NK, NTH> tk, tv;                          // K is fetched once for both of its LDS images
        pair_load<NK, NTH>(tk, Kp, a.ldk, rowk, h * D, S, t);
        pair_load<NK, NTH>(tv, Vp, a.ldv, rowk, h * D, S, t);
        pair_store_rm<NK, NTH>(tk, ldsK, t);
        pair_store_tr<NK, NTH>(tk, ldsKT, t);
        pair_store_rm<NK, NTH>(tv, ldsV, t);
    } else {
        PairTile32<NK, NTH> tk, tv;
        pair_load32<NK, T, NTH>(tk, Kp, a.ldk, rowk, h * D, S, t);
        pair_load32<NK, T, NTH>(tv, Vp, a.ldv, rowk, h * D, S, t);
        pair_store_rm32<NK, T, NTH>(tk, ldsK, t);
        pair_store_tr32<NK, T, NTH>(tk, ldsKT, t);
        pair_store_rm32<NK, T, NTH>(tv, ldsV, t);
    }
    for (int k = t; k < NK; k += NTH) ldsMask[k] = k < S ? a.mask_add[(long)b * S + k] : -INFINITY;
    __syncthreads();

    const int nqf = (Sq + 15) / 16;
    for (int qf = wave; qf < nqf; qf += NTH / 64) {
        const int q = qf * 16 + li;
        const bool qok = q < Sq;
        const T* qrow = Qp + (rowq + (qok ? q : 0)) * a.ldq + h * D;
        const T* dorow = (const T*)a.dctx + (rowq + (qok ? q : 0)) * a.lddo + h * D;
        typename VecOf<T>::v8 qb[2], dob[2];
        qb[0] = frag_g(qrow, 0, lg, qok); qb[1] = frag_g(qrow, 1, lg, qok);
        dob[0] = frag_g(dorow, 0, lg, qok); dob[1] = frag_g(dorow, 1, lg, qok);
        const float lse = qok ? a.lse[(long)bh * Sq + q] : 0.f;
        uint64_t bits[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w)
            bits[w] = (a.p > 0.f && qok) ? a.keepbits[keep_index(a, bh, q, lg, w, NW)] : ~(uint64_t)0;

        f32x4 pt[NKF], dpt[NKF];
        float dsum = 0.f;
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf) {
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = s;
            if (kf * 16 < S) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    s = vb_mma(frag_rm(ldsK, kf * 16 + li, ks, lg, T()), qb[ks], s);
                    dp = vb_mma(frag_rm(ldsV, kf * 16 + li, ks, lg, T()), dob[ks], dp);
                }
            }
            const f32x4 mk = *(const f32x4*)(ldsMask + kf * 16 + lg * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = fast_exp(s[r] * a.scale + mk[r] - lse);            // 0 for keys >= S
                const bool keep = (bits[kf >> 4] >> ((kf & 15) * 4 + r)) & 1;
                const float dpv = keep ? dp[r] * a.inv_keep : 0.f;
                dsum += p * dpv;
                s[r] = p; dp[r] = dpv;
            }
            pt[kf] = s; dpt[kf] = dp;
        }
        dsum += __shfl_xor(dsum, 16);
        dsum += __shfl_xor(dsum, 32);
        if (a.dsum && lg == 0 && qok) a.dsum[(long)bh * Sq + q] = dsum;
        // dS^T = P o (dP - D) * scale, rounded to T as the MFMA B operand
        typename VecOf<T>::v8 dsb[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            f32x4 x0, x1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x0[r] = pt[2 * ks][r] * (dpt[2 * ks][r] - dsum) * a.scale;
                x1[r] = pt[2 * ks + 1][r] * (dpt[2 * ks + 1][r] - dsum) * a.scale;
            }
            pack_b(dsb[ks], x0, x1);
        }
        T* dqrow = (T*)a.dq + (rowq + (qok ? q : 0)) * a.lddq + h * D;
#pragma unroll
        for (int df = 0; df < 4; ++df) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks * 32 < S)
                    acc = vb_mma(frag_tr(ldsKT, tr_pitch<T>(NK), df * 16 + li, ks, lg, T()), dsb[ks], acc);
            }
            if (qok && !a.sp_only) store4(dqrow + df * 16 + lg * 4, acc);
            if constexpr (std::is_same<T, xf32>::value) {
                if (a.dqkv_sp && qok) store_split4(a.dqkv_sp + (rowq + q) * (2 * a.lddq) + h * D + df * 16 + lg * 4, a.lddq, acc);
            }
        }
    }
}

// =================================================================================================
// KEY-TILED forward and dQ pass: the kernels above keep a (sample, head)'s whole K / V in LDS, which holds S <= 256 (forward)
// and S <= 192 (dQ pass) in fp32 and S <= 416 (dQ pass) in bf16.  The reference trains up to max_position_embeddings = 512
// (modeling.py:83, :183; NLVR2 as it ships is S = 416): beyond those limits the keys stream through LDS in chunks of 128.
//   forward : online softmax -- running maximum m and sum l per query; the context accumulators are rescaled by
//             exp(m_old - m_new) when a chunk raises the maximum; dropout keep-bits and the generator's group index use the
//             ABSOLUTE key position, so the masks are the ones the untiled kernels would draw; 1 / (l (1 - p)) at the end.
//   dQ pass : the softmax statistics are known (lse from forward), so chunks are independent; D = rowsum(P o dP) needs every
//             key before any dS can be formed: sweep 1 accumulates D over the chunks, sweep 2 recomputes P, dP per chunk and
//             accumulates dQ += dS K.  (Twice the score work of the untiled form: this is the long-sequence parity path.)
// Per workgroup one (sample, head); per pass over the chunks four query blocks (one per wave); K / V of a (sample, head) are
// re-read from L2 once per group of 64 queries.  NW = keep-bit words per (query, key quad): the launcher's keepbits_nw(S).
// The dK / dV pass below is already independent of S.
// =================================================================================================
constexpr int TCF = 8, TCK = TCF * 16, TCS = TCF / 2;       // key fragments, keys, 32-key steps per chunk

template <typename T, int NW>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) attn_fwd_tiled_kernel(AttnArgs a) {
    VB_DYN_SMEM(smem);
    unsigned char* ldsK = smem;
    unsigned char* ldsVT = ldsK + rm_bytes<T>(TCK);
    float* ldsMask = (float*)(ldsVT + tr_bytes<T>(TCK));
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4;
    const int bh = blockIdx.x, b = bh / a.nh, h = bh % a.nh;
    const int S = a.S, Sq = a.Sq;
    const long rowk = (long)b * S, rowq = (long)b * Sq;
    const T* Qp = (const T*)a.q;
    const T* Kp = (const T*)a.k;
    const T* Vp = (const T*)a.v;
    const int nqf = (Sq + 15) / 16;
    for (int qg = 0; qg < nqf; qg += 4) {                   // every wave runs every trip: the staging barriers are workgroup-wide
        const int q = (qg + wave) * 16 + li;
        const bool qok = q < Sq;
        const T* qrow = Qp + (rowq + (qok ? q : 0)) * a.ldq + h * D;
        typename VecOf<T>::v8 qb[2];
        qb[0] = frag_g(qrow, 0, lg, qok); qb[1] = frag_g(qrow, 1, lg, qok);
        float m = -INFINITY, l = 0.f;
        f32x4 acc[4];
#pragma unroll
        for (int df = 0; df < 4; ++df) acc[df] = f32x4{0.f, 0.f, 0.f, 0.f};
        uint64_t bits[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) bits[w] = 0;
        for (int c0 = 0; c0 < S; c0 += TCK) {
            __syncthreads();                                // the previous chunk is consumed
            {
                typename TileOf<T, TCK>::type tk, tv;
                tile_load<TCK, T>(tk, Kp, a.ldk, rowk + c0, h * D, S - c0, t);
                tile_load<TCK, T>(tv, Vp, a.ldv, rowk + c0, h * D, S - c0, t);
                tile_store_rm<TCK, T>(tk, ldsK, t);
                tile_store_tr<TCK, T>(tv, ldsVT, t);
            }
            for (int k = t; k < TCK; k += NT) ldsMask[k] = c0 + k < S ? a.mask_add[(long)b * S + c0 + k] : -INFINITY;
            __syncthreads();
            f32x4 st[TCF];
            float mc = -INFINITY;
#pragma unroll
            for (int kf = 0; kf < TCF; ++kf) {
                f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f};
                if (c0 + kf * 16 < S) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) sc = vb_mma(frag_rm(ldsK, kf * 16 + li, ks, lg, T()), qb[ks], sc);
                }
                const f32x4 mk = *(const f32x4*)(ldsMask + kf * 16 + lg * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) { sc[r] = sc[r] * a.scale + mk[r]; mc = fmaxf(mc, sc[r]); }
                st[kf] = sc;
            }
            mc = fmaxf(mc, __shfl_xor(mc, 16));
            mc = fmaxf(mc, __shfl_xor(mc, 32));
            const float mn = fmaxf(m, mc);                   // finite: every chunk that starts below S holds a real key
            const float alpha = fast_exp(m - mn);            // 0 on the first chunk (m = -inf)
            float sum = 0.f;
#pragma unroll
            for (int kf = 0; kf < TCF; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) { st[kf][r] = fast_exp(st[kf][r] - mn); sum += st[kf][r]; }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            l = l * alpha + sum;
            m = mn;
#pragma unroll
            for (int df = 0; df < 4; ++df)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[df][r] *= alpha;
            if (a.p > 0.f) {
#pragma unroll
                for (int kp = 0; kp < TCS; ++kp) {
                    const uint64_t grp = ((((uint64_t)bh * Sq + q) * 4 + lg) << 5) + (uint64_t)(c0 / 32 + kp);
                    Rand8 rnd = vb_dropout_bits8(a.seed, grp, a.stream);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int kf = 2 * kp + (e >> 2), r = e & 3;
                        const bool keep = rand8_keep(rnd, e, a.thresh);
                        st[kf][r] = keep ? st[kf][r] : 0.f;
                        const int kfa = c0 / 16 + kf;        // absolute key fragment: word kfa >> 4, nibble kfa & 15
                        if (keep) bits[(kfa >> 4) < NW ? (kfa >> 4) : NW - 1] |= (uint64_t)1 << ((kfa & 15) * 4 + r);
                    }
                }
            }
            typename VecOf<T>::v8 pb[TCS];
#pragma unroll
            for (int ks = 0; ks < TCS; ++ks) pack_b(pb[ks], st[2 * ks], st[2 * ks + 1]);
#pragma unroll
            for (int df = 0; df < 4; ++df)
#pragma unroll
                for (int ks = 0; ks < TCS; ++ks)
                    if (c0 + ks * 32 < S)
                        acc[df] = vb_mma(frag_tr(ldsVT, tr_pitch<T>(TCK), df * 16 + li, ks, lg, T()), pb[ks], acc[df]);
        }
        const float inv = a.inv_keep / l;                    // normalisation and 1 / (1 - p) in one factor
        if (a.lse && lg == 0 && qok) a.lse[(long)bh * Sq + q] = m + logf(l);
        if (a.p > 0.f && a.keepbits && qok) {
#pragma unroll
            for (int w = 0; w < NW; ++w) a.keepbits[keep_index(a, bh, q, lg, w, NW)] = bits[w];
        }
        T* crow = (T*)a.ctx + (rowq + (qok ? q : 0)) * a.ldc + h * D;
#pragma unroll
        for (int df = 0; df < 4; ++df) {
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = acc[df][r] * inv;
            if (qok && !a.sp_only) store4(crow + df * 16 + lg * 4, o);
            if constexpr (std::is_same<T, xf32>::value) {
                if (a.ctx_sp && qok) store_split4(a.ctx_sp + (rowq + q) * (2 * a.ldc) + h * D + df * 16 + lg * 4, a.ldc, o);
            }
        }
    }
}

template <typename T, int NW>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) attn_bwd_dq_tiled_kernel(AttnArgs a) {
    VB_DYN_SMEM(smem);
    unsigned char* ldsK = smem;
    unsigned char* ldsV = ldsK + rm_bytes<T>(TCK);
    unsigned char* ldsKT = ldsV + rm_bytes<T>(TCK);
    float* ldsMask = (float*)(ldsKT + tr_bytes<T>(TCK));
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4;
    const int bh = blockIdx.x, b = bh / a.nh, h = bh % a.nh;
    const int S = a.S, Sq = a.Sq;
    const long rowk = (long)b * S, rowq = (long)b * Sq;
    const T* Qp = (const T*)a.q;
    const T* Kp = (const T*)a.k;
    const T* Vp = (const T*)a.v;
    const int nqf = (Sq + 15) / 16;
    for (int qg = 0; qg < nqf; qg += 4) {
        const int q = (qg + wave) * 16 + li;
        const bool qok = q < Sq;
        const T* qrow = Qp + (rowq + (qok ? q : 0)) * a.ldq + h * D;
        const T* dorow = (const T*)a.dctx + (rowq + (qok ? q : 0)) * a.lddo + h * D;
        typename VecOf<T>::v8 qb[2], dob[2];
        qb[0] = frag_g(qrow, 0, lg, qok); qb[1] = frag_g(qrow, 1, lg, qok);
        dob[0] = frag_g(dorow, 0, lg, qok); dob[1] = frag_g(dorow, 1, lg, qok);
        const float lse = qok ? a.lse[(long)bh * Sq + q] : 0.f;
        uint64_t bits[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w)
            bits[w] = (a.p > 0.f && qok) ? a.keepbits[keep_index(a, bh, q, lg, w, NW)] : ~(uint64_t)0;
        // P and the dropped dP of one chunk (both sweeps); keys >= S have mask -inf: P = 0
        auto chunk_p_dp = [&](int c0, f32x4 (&pt)[TCF], f32x4 (&dpt)[TCF]) {
#pragma unroll
            for (int kf = 0; kf < TCF; ++kf) {
                f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f}, dp = sc;
                if (c0 + kf * 16 < S) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        sc = vb_mma(frag_rm(ldsK, kf * 16 + li, ks, lg, T()), qb[ks], sc);
                        dp = vb_mma(frag_rm(ldsV, kf * 16 + li, ks, lg, T()), dob[ks], dp);
                    }
                }
                const f32x4 mk = *(const f32x4*)(ldsMask + kf * 16 + lg * 4);
                const int kfa = c0 / 16 + kf;
                const uint64_t word = bits[(kfa >> 4) < NW ? (kfa >> 4) : NW - 1];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = fast_exp(sc[r] * a.scale + mk[r] - lse);
                    const bool keep = (word >> ((kfa & 15) * 4 + r)) & 1;
                    sc[r] = pv;
                    dp[r] = keep ? dp[r] * a.inv_keep : 0.f;
                }
                pt[kf] = sc; dpt[kf] = dp;
            }
        };
        // ---- sweep 1: D = rowsum(P o dP) over every key
        float dsum = 0.f;
        for (int c0 = 0; c0 < S; c0 += TCK) {
            __syncthreads();
            {
                typename TileOf<T, TCK>::type tk, tv;
                tile_load<TCK, T>(tk, Kp, a.ldk, rowk + c0, h * D, S - c0, t);
                tile_load<TCK, T>(tv, Vp, a.ldv, rowk + c0, h * D, S - c0, t);
                tile_store_rm<TCK, T>(tk, ldsK, t);
                tile_store_rm<TCK, T>(tv, ldsV, t);
            }
            for (int k = t; k < TCK; k += NT) ldsMask[k] = c0 + k < S ? a.mask_add[(long)b * S + c0 + k] : -INFINITY;
            __syncthreads();
            f32x4 pt[TCF], dpt[TCF];
            chunk_p_dp(c0, pt, dpt);
#pragma unroll
            for (int kf = 0; kf < TCF; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) dsum += pt[kf][r] * dpt[kf][r];
        }
        dsum += __shfl_xor(dsum, 16);
        dsum += __shfl_xor(dsum, 32);
        if (a.dsum && lg == 0 && qok) a.dsum[(long)bh * Sq + q] = dsum;
        // ---- sweep 2: dQ += dS K, dS = P o (dP - D) / sqrt(d)
        f32x4 acc[4];
#pragma unroll
        for (int df = 0; df < 4; ++df) acc[df] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < S; c0 += TCK) {
            __syncthreads();
            {
                typename TileOf<T, TCK>::type tk, tv;
                tile_load<TCK, T>(tk, Kp, a.ldk, rowk + c0, h * D, S - c0, t);
                tile_load<TCK, T>(tv, Vp, a.ldv, rowk + c0, h * D, S - c0, t);
                tile_store_rm<TCK, T>(tk, ldsK, t);
                tile_store_tr<TCK, T>(tk, ldsKT, t);
                tile_store_rm<TCK, T>(tv, ldsV, t);
            }
            for (int k = t; k < TCK; k += NT) ldsMask[k] = c0 + k < S ? a.mask_add[(long)b * S + c0 + k] : -INFINITY;
            __syncthreads();
            f32x4 pt[TCF], dpt[TCF];
            chunk_p_dp(c0, pt, dpt);
            typename VecOf<T>::v8 dsb[TCS];
#pragma unroll
            for (int ks = 0; ks < TCS; ++ks) {
                f32x4 x0, x1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    x0[r] = pt[2 * ks][r] * (dpt[2 * ks][r] - dsum) * a.scale;
                    x1[r] = pt[2 * ks + 1][r] * (dpt[2 * ks + 1][r] - dsum) * a.scale;
                }
                pack_b(dsb[ks], x0, x1);
            }
#pragma unroll
            for (int df = 0; df < 4; ++df)
#pragma unroll
                for (int ks = 0; ks < TCS; ++ks)
                    if (c0 + ks * 32 < S)
                        acc[df] = vb_mma(frag_tr(ldsKT, tr_pitch<T>(TCK), df * 16 + li, ks, lg, T()), dsb[ks], acc[df]);
        }
        T* dqrow = (T*)a.dq + (rowq + (qok ? q : 0)) * a.lddq + h * D;
#pragma unroll
        for (int df = 0; df < 4; ++df) {
            if (qok && !a.sp_only) store4(dqrow + df * 16 + lg * 4, acc[df]);
            if constexpr (std::is_same<T, xf32>::value) {
                if (a.dqkv_sp && qok) store_split4(a.dqkv_sp + (rowq + q) * (2 * a.lddq) + h * D + df * 16 + lg * 4, a.lddq, acc[df]);
            }
        }
    }
}

// =================================================================================================
// backward, pass B: dK and dV.  grid = (B*nh, ceil(key fragments / 4)): a wave owns ONE 16-key
// fragment and sweeps all queries in chunks of QC; Q / dO (row-major and transposed), lse, D and the
// keep-bits of a chunk are staged in LDS per chunk, so LDS use is independent of S.
// =================================================================================================
constexpr int QC = 64;

template <typename T, int NKF>
VB_KERNEL VB_LAUNCH_BOUNDS(NT) attn_bwd_dkv_kernel(AttnArgs a) {
    constexpr int NW = (NKF + 15) / 16;
    VB_DYN_SMEM(smem);
    unsigned char* ldsQ = smem;
    unsigned char* ldsDO = ldsQ + rm_bytes<T>(QC);
    unsigned char* ldsQT = ldsDO + rm_bytes<T>(QC);
    unsigned char* ldsDOT = ldsQT + tr_bytes<T>(QC);
    float* ldsLse = (float*)(ldsDOT + tr_bytes<T>(QC));
    float* ldsD = ldsLse + QC;
    uint64_t* ldsBits = (uint64_t*)(ldsD + QC);           // [QC][4][NW]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4;
    const int bh = blockIdx.x, b = bh / a.nh, h = bh % a.nh;
    const int S = a.S, Sq = a.Sq;                          // keys / queries per sample
    const long rowk = (long)b * S, rowq = (long)b * Sq;
    const T* Qp = (const T*)a.q;
    const T* Kp = (const T*)a.k;
    const T* Vp = (const T*)a.v;
    const T* dctx = (const T*)a.dctx;

    const int kf = blockIdx.y * 4 + wave;
    const int key = kf * 16 + li;
    const bool wave_on = kf * 16 < S;                      // wave-uniform
    const bool kok = key < S;
    const T* krow = Kp + (rowk + (kok ? key : 0)) * a.ldk + h * D;
    const T* vrow = Vp + (rowk + (kok ? key : 0)) * a.ldv + h * D;
    typename VecOf<T>::v8 kb[2], vb[2];
    kb[0] = frag_g(krow, 0, lg, kok); kb[1] = frag_g(krow, 1, lg, kok);
    vb[0] = frag_g(vrow, 0, lg, kok); vb[1] = frag_g(vrow, 1, lg, kok);
    const float mk = kok ? a.mask_add[(long)b * S + key] : -INFINITY;
    f32x4 dkT[4], dvT[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) { dkT[df] = f32x4{0.f, 0.f, 0.f, 0.f}; dvT[df] = dkT[df]; }

    // A query chunk's global data (Q, dO rows, lse, D, keep-bits) is fetched into REGISTERS one chunk ahead and written to
    // LDS in both layouts (row-major + transposed) from the same registers: the HBM trip of chunk c+1 runs under the
    // MFMAs of chunk c, and Q / dO are read once instead of twice (bf16; the fp32 parity path stages as before).
    constexpr bool PIPE = sizeof(T) == 2;
    constexpr int BPT = (QC * 4 * NW + NT - 1) / NT;       // keep-bit words per thread per chunk
    u32x4 cq0 = u32x4{0u, 0u, 0u, 0u}, cq1 = cq0, cd0 = cq0, cd1 = cq0;
    float c_lse = INFINITY, c_d = 0.f;
    uint64_t c_bits[BPT];
    auto load_chunk = [&](int q0) {
        const int dc = t & 7, r = (t >> 3) * 2;             // rows q0 + r, q0 + r + 1; 16-byte column chunk dc
        const u32x4 z = u32x4{0u, 0u, 0u, 0u};
        const bool ok0 = q0 + r < Sq, ok1 = q0 + r + 1 < Sq;
        cq0 = ok0 ? *(const u32x4*)(Qp + (rowq + q0 + r) * a.ldq + h * D + dc * 8) : z;
        cq1 = ok1 ? *(const u32x4*)(Qp + (rowq + q0 + r + 1) * a.ldq + h * D + dc * 8) : z;
        cd0 = ok0 ? *(const u32x4*)(dctx + (rowq + q0 + r) * a.lddo + h * D + dc * 8) : z;
        cd1 = ok1 ? *(const u32x4*)(dctx + (rowq + q0 + r + 1) * a.lddo + h * D + dc * 8) : z;
        if (t < QC) {
            const int q = q0 + t;
            c_lse = q < Sq ? a.lse[(long)bh * Sq + q] : INFINITY;   // exp(x - inf) = 0 for padded queries
            c_d = q < Sq ? a.dsum[(long)bh * Sq + q] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
            const int i = t + j * NT;
            const int q = q0 + i / (4 * NW);
            c_bits[j] = (a.p > 0.f && i < QC * 4 * NW && q < Sq) ? a.keepbits[((long)bh * Sq + q0) * 4 * NW + i] : ~(uint64_t)0;
        }
    };
    auto store_tr2 = [&](unsigned char* lds, const u32x4& x0, const u32x4& x1) {
        const int dc = t & 7, r = (t >> 3) * 2;
        const int pitch = tr_pitch<bf16>(QC);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t a0 = x0[w], b0 = x1[w];
            *(uint32_t*)(lds + (dc * 8 + 2 * w) * pitch + r * 2) = (a0 & 0xFFFFu) | (b0 << 16);
            *(uint32_t*)(lds + (dc * 8 + 2 * w + 1) * pitch + r * 2) = (a0 >> 16) | (b0 & 0xFFFF0000u);
        }
    };
    auto store_chunk = [&]() {
        const int dc = t & 7, r = (t >> 3) * 2;
        *(u32x4*)(ldsQ + rm_off<T>(r, dc)) = cq0;
        *(u32x4*)(ldsQ + rm_off<T>(r + 1, dc)) = cq1;
        *(u32x4*)(ldsDO + rm_off<T>(r, dc)) = cd0;
        *(u32x4*)(ldsDO + rm_off<T>(r + 1, dc)) = cd1;
        store_tr2(ldsQT, cq0, cq1);
        store_tr2(ldsDOT, cd0, cd1);
        if (t < QC) { ldsLse[t] = c_lse; ldsD[t] = c_d; }
#pragma unroll
        for (int j = 0; j < BPT; ++j) if (t + j * NT < QC * 4 * NW) ldsBits[t + j * NT] = c_bits[j];
    };
    // fp32 / split modes: the same pipeline with the chunk's Q and dO tiles held in registers (pair tiles for 4-byte sources)
    PairTile32<QC> tq32, td32;
    auto load_chunk32 = [&](int q0) {
        pair_load32<QC, T>(tq32, Qp, a.ldq, rowq + q0, h * D, Sq - q0, t);
        pair_load32<QC, T>(td32, dctx, a.lddo, rowq + q0, h * D, Sq - q0, t);
    };
    if constexpr (PIPE) load_chunk(0);
    else load_chunk32(0);

    for (int q0 = 0; q0 < Sq; q0 += QC) {
        __syncthreads();                                   // previous chunk fully consumed
        if constexpr (PIPE) {
            store_chunk();
            __syncthreads();
            if (q0 + QC < Sq) load_chunk(q0 + QC);
        } else {
            pair_store_rm32<QC, T>(tq32, ldsQ, t);
            pair_store_tr32<QC, T>(tq32, ldsQT, t);
            pair_store_rm32<QC, T>(td32, ldsDO, t);
            pair_store_tr32<QC, T>(td32, ldsDOT, t);
            for (int k = t; k < QC; k += NT) {
                const int q = q0 + k;
                ldsLse[k] = q < Sq ? a.lse[(long)bh * Sq + q] : INFINITY;   // exp(x - inf) = 0 for padded queries
                ldsD[k] = q < Sq ? a.dsum[(long)bh * Sq + q] : 0.f;
            }
            for (int i = t; i < QC * 4 * NW; i += NT) {
                const int q = q0 + i / (4 * NW);
                ldsBits[i] = (a.p > 0.f && q < Sq) ? a.keepbits[((long)bh * Sq + q0) * 4 * NW + i] : ~(uint64_t)0;
            }
            __syncthreads();
            if (q0 + QC < Sq) load_chunk32(q0 + QC);       // in flight under this chunk's MFMAs
        }
        if (!wave_on) continue;
#pragma unroll
        for (int qc = 0; qc < QC / 32; ++qc) {
            if (q0 + qc * 32 >= Sq) continue;
            f32x4 pd[2], dsv[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int qf = 2 * qc + hf;                // fragment index inside the chunk
                if constexpr (!std::is_same<T, bf16>::value) {          // (bf16: +6 ... 8 VGPRs across the 128 mark; S <= 192 runs the one-pass kernel anyway)
                    if (hf == 1 && q0 + qf * 16 >= Sq) {   // wave-uniform: the second half of the last 32-query block is all padding
                        pd[1] = f32x4{0.f, 0.f, 0.f, 0.f}; dsv[1] = pd[1];
                        continue;
                    }
                }
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    s = vb_mma(frag_rm(ldsQ, qf * 16 + li, ks, lg, T()), kb[ks], s);
                    dp = vb_mma(frag_rm(ldsDO, qf * 16 + li, ks, lg, T()), vb[ks], dp);
                }
                // lane: key = kf*16 + li (column), queries q0 + qf*16 + lg*4 + r (rows)
                const f32x4 lse4 = *(const f32x4*)(ldsLse + qf * 16 + lg * 4);
                const f32x4 d4 = *(const f32x4*)(ldsD + qf * 16 + lg * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ql = qf * 16 + lg * 4 + r;
                    const float p = fast_exp(s[r] * a.scale + mk - lse4[r]);
                    const uint64_t w = ldsBits[(ql * 4 + (li >> 2)) * NW + (kf >> 4)];
                    const bool keep = (w >> ((kf & 15) * 4 + (li & 3))) & 1;
                    const float pdrop = keep ? p * a.inv_keep : 0.f;
                    const float dpv = keep ? dp[r] * a.inv_keep : 0.f;
                    pd[hf][r] = pdrop;
                    dsv[hf][r] = p * (dpv - d4[r]) * a.scale;
                }
            }
            typename VecOf<T>::v8 pb, dsb;
            pack_b(pb, pd[0], pd[1]);
            pack_b(dsb, dsv[0], dsv[1]);
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                dvT[df] = vb_mma(frag_tr(ldsDOT, tr_pitch<T>(QC), df * 16 + li, qc, lg, T()), pb, dvT[df]);
                dkT[df] = vb_mma(frag_tr(ldsQT, tr_pitch<T>(QC), df * 16 + li, qc, lg, T()), dsb, dkT[df]);
            }
        }
    }
    {   // every lane runs the stores' lane exchanges (store4x2); only the stores themselves are predicated on the key being real
        const int keyr = kok ? key : 0;
        T* dkrow = (T*)a.dk + (rowk + keyr) * a.lddk + h * D;
        T* dvrow = (T*)a.dv + (rowk + keyr) * a.lddv + h * D;
        const bool wide_k = sizeof(T) == 2 && wide_ok(a.dk, a.lddk), wide_v = sizeof(T) == 2 && wide_ok(a.dv, a.lddv);
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            store4x2(dkrow + dj * 32, lg, dkT[2 * dj], dkT[2 * dj + 1], kok && !a.sp_only, wide_k);
            store4x2(dvrow + dj * 32, lg, dvT[2 * dj], dvT[2 * dj + 1], kok && !a.sp_only, wide_v);
        }
        if constexpr (std::is_same<T, xf32>::value) {
            if (kok && a.dqkv_sp) {                         // self-attention: dK | dV are column blocks H.. and 2H.. of the dqkv image
#pragma unroll
                for (int df = 0; df < 4; ++df) {
                    bf16* sp = a.dqkv_sp + (rowk + key) * (2 * a.lddk) + h * D + df * 16 + lg * 4;
                    store_split4(sp + a.nh * D, a.lddk, dkT[df]);
                    store_split4(sp + 2 * a.nh * D, a.lddk, dvT[df]);
                }
            }
        }
    }
}


// =================================================================================================
// backward in ONE pass (bf16, S <= 192, needs the forward output O = ctx): the two-pass form above computes every
// score, probability and dP twice (once per pass) and its dQ pass alone is 45 % of the backward time -- per-probability
// VALU work, not MFMAs, bounds these kernels.  Here one workgroup of 12 waves owns a (batch, head).  Per chunk of 64 queries:
//   phase A  wave w owns key fragment w exactly like the dK/dV pass (lane <-> key): S, P, dP, dS once; dV^T += dO^T P,
//            dK^T += Q^T dS; its dS fragment is also written to a [64 queries][192 keys] tile in LDS.
//            D = rowsum(P o dP) comes from dO . O (same value, dropout included: O = P_drop V), computed while staging.
//   phase B  the 16 (query fragment, d block) outputs of dQ = dS K for the chunk are split over the waves: full-depth MFMAs
//            with dS from the tile and K^T from an LDS image built once per (batch, head) -- no atomics, no second exp.
// (A first version added per-wave dQ partials into an LDS tile with ds_add_f32: 7x slower than two passes.)
// Round 5, measured and removed (profiles/r05_attn_bench_b1024_prefetch_ab.txt): an L2 prefetch of the lines workgroup b + 256 (same XCD)
// opens with, issued by this workgroup once its own fetches were consumed -- 845 -> 907 us alone, 956 -> 1015 us in the step: the four
// registers it cost spilled (the kernel sits at 168), and scratch reloads wait on vmcnt.  Also measured and removed
// (profiles/r05_attn_persistent_ab.txt; the kernel is in the git history): the PERSISTENT form -- one workgroup per CU walking the pairs,
// the next pair's K / V fragments, mask and chunk 0 fetched into registers that are dead by then and its K rows into spare LDS by
// LDS-direct copies, under the current pair's last phase B and epilogue.  Bit-identical, no VGPR spill after five compiler
// work-arounds, and at best PARITY: 862 against 853 us.
// =================================================================================================
constexpr int FWPB = 12, FNT = FWPB * 64, FNK = FWPB * 16;  // 192 keys
// Pitches of the two LDS tiles phase B reads: rows of FNK keys + VB_ATTN_BPAD bytes (see LT<bf16>::TPAD above for the round-6
// experiment with 16: it lost)
#ifndef VB_ATTN_BPAD
#define VB_ATTN_BPAD 8
#endif
constexpr int TSP = FNK * 2 + VB_ATTN_BPAD;                // dS tile pitch in bytes (pad: 16 rows -> distinct banks)
constexpr int KTP = FNK * 2 + VB_ATTN_BPAD;                // K^T image pitch in bytes
constexpr int KT_BYTES = D * KTP;

template <int NKF, int CQ>
VB_KERNEL VB_LAUNCH_BOUNDS(FNT) attn_bwd_fused_kernel(AttnArgs a) {
    typedef bf16 T;
    constexpr int NW = (NKF + 15) / 16;
    VB_DYN_SMEM(smem);
    // two sets of chunk images: chunk c+1 is written to LDS while the slower waves still compute on chunk c
    constexpr int SETB = 2 * rm_bytes<T>(CQ) + 2 * tr_bytes<T>(CQ) + 2 * CQ * 4 + CQ * 4 * NW * 8;
    unsigned char *ldsQ, *ldsDO, *ldsQT, *ldsDOT;
    float *ldsLse, *ldsD;
    uint64_t* ldsBits;                                     // [CQ][4][NW]
    auto use_set = [&](int set) {
        ldsQ = smem + set * SETB;
        ldsDO = ldsQ + rm_bytes<T>(CQ);
        ldsQT = ldsDO + rm_bytes<T>(CQ);
        ldsDOT = ldsQT + tr_bytes<T>(CQ);
        ldsLse = (float*)(ldsDOT + tr_bytes<T>(CQ));
        ldsD = ldsLse + CQ;
        ldsBits = (uint64_t*)(ldsD + CQ);
    };
    unsigned char* ldsKT = smem + 2 * SETB;                // K^T of the whole sequence: [64 d][FNK keys]
    unsigned char* ldsDS = ldsKT + KT_BYTES;               // dS of the chunk: [CQ queries][FNK keys], pitch TSP
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4;
    const int bh = blockIdx.x, b = bh / a.nh, h = bh % a.nh;
    const int S = a.S, H = a.nh * D;
    const long ldx = 3L * H, row0 = (long)b * S;
    const T* qkv = (const T*)a.qkv;
    const T* dctx = (const T*)a.dctx;
    const T* octx = (const T*)a.ctx_fwd;

    // Every global load below is UNCONDITIONAL from a clamped (always readable) row, and the zero / -inf / all-ones fill of the
    // rows past S is a select at the point of use: the loads carry no branches, nothing waits between them, and a chunk's
    // fetch really runs a chunk ahead (with `ok ? load : fill` the compiler put s_waitcnt vmcnt(0) between the loads of the
    // "prefetch" -- the registers' fill is a write-after-write on a pending load -- and the workgroup, alone on its compute
    // unit, sat through four HBM round trips before its first MFMA and one more per chunk).
    const int kf = wave;
    const int key = kf * 16 + li;
    const bool wave_on = kf * 16 < S;                      // wave-uniform
    const bool kok = key < S;
    const int keyc = kok ? key : S - 1;
    // K^T image source: one row pair x 16-byte chunk per thread (96 pairs x 8 chunks = 768 items = FNT)
    const int kdc = t & 7, kr = (t >> 3) * 2;
    const u32x4 kx0 = *(const u32x4*)(qkv + (row0 + (kr < S ? kr : S - 1)) * ldx + H + h * D + kdc * 8);
    const u32x4 kx1 = *(const u32x4*)(qkv + (row0 + (kr + 1 < S ? kr + 1 : S - 1)) * ldx + H + h * D + kdc * 8);
    const T* krow = qkv + (row0 + keyc) * ldx + H + h * D;
    const T* vrow = qkv + (row0 + keyc) * ldx + 2 * H + h * D;
    bf16x8 kb[2], vb[2];
    kb[0] = frag_g(krow, 0, lg, kok); kb[1] = frag_g(krow, 1, lg, kok);
    vb[0] = frag_g(vrow, 0, lg, kok); vb[1] = frag_g(vrow, 1, lg, kok);
    const float mk_raw = a.mask_add[(long)b * S + keyc];
    f32x4 dkT[4], dvT[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) { dkT[df] = f32x4{0.f, 0.f, 0.f, 0.f}; dvT[df] = dkT[df]; }

    constexpr int BPT = (CQ * 4 * NW + FNT - 1) / FNT;
    constexpr int SI = CQ * 4;                             // staging items per tensor (row pair x 16-byte chunk)
    static_assert(SI % 64 == 0 && CQ == 64, "the staging role must be wave-uniform; wave 0 carries the chunk's lse values");
    // staging roles (wave-uniform): role 0 (waves 0-3) fetches dO and O for D = dO . O and carries the chunk's lse and keep-bit
    // words, role 1 (waves 4-7) stages dO, role 2 (waves 8-11) stages Q.  The heaviest staging goes to the OLDEST waves: the
    // SIMD arbiter favours them in phase A (an in-kernel cycle trace shows waves 0-3 done with a chunk's phase A after ~3700
    // cycles, waves 8-11 after ~6000 -- three waves per SIMD share its VALU), so they have the slack.
    const int st = t % SI, role = vb_uniform(t / SI);
    const int sdc = st & 7, sr = (st >> 3) * 2;            // rows q0 + sr, q0 + sr + 1; 16-byte column chunk sdc
    // addresses: byte offsets of this thread's row q0 = 0 from wave-uniform bases, 32 bits (launcher: tokens x pitch x 2 below
    // 2^32); a chunk adds a small signed multiple of the pitch, which also expresses the clamp of rows past S
    const unsigned char* sbase = (const unsigned char*)((role == 2 ? qkv : dctx) + h * D);   // Q rows | dO rows (roles 0 and 1)
    const int sldb = (int)((role == 2 ? ldx : (long)H) * 2);                 // row pitch in bytes
    const unsigned char* obase = (const unsigned char*)(octx + h * D);       // O rows: same pitch and offsets as dO
    const unsigned off_s = (unsigned)((row0 + sr) * (long)sldb) + sdc * 16;
    u32x4 c0 = u32x4{0u, 0u, 0u, 0u}, c1 = c0, o0 = c0, o1 = c0;
    float c_lse = 0.f;
    uint64_t c_bits[BPT];
#pragma unroll
    for (int j = 0; j < BPT; ++j) c_bits[j] = 0;
    static_assert(BPT == 1 && CQ * 4 * NW == SI, "keep-bit words of a chunk: one per thread of waves 0-3");
    auto load_chunk = [&](int q0) {
        // rows q0 + sr (+ 1), clamped to S - 1, as a row delta from this thread's chunk-0 row sr
        const int d0r = q0 + sr < S ? q0 : S - 1 - sr, d1r = q0 + sr + 1 < S ? q0 + 1 : S - 1 - sr;
        // ONE pair of loads for every role (role 0 reads the same dO rows as role 1): two branches filling the same registers
        // made the compiler put a vmcnt wait in front of the second branch's loads
        c0 = *(const u32x4*)(sbase + (off_s + (unsigned)(d0r * sldb)));
        c1 = *(const u32x4*)(sbase + (off_s + (unsigned)(d1r * sldb)));
        if (role == 0) {                                   // wave-uniform: a scalar branch
            o0 = *(const u32x4*)(obase + (off_s + (unsigned)(d0r * sldb)));
            o1 = *(const u32x4*)(obase + (off_s + (unsigned)(d1r * sldb)));
            if (wave == 0) c_lse = a.lse[(long)bh * S + (q0 + lane < S ? q0 + lane : S - 1)];      // CQ = 64 queries = one wave
            if (a.p > 0.f) {
                const long i = ((long)bh * S + q0) * 4 * NW + t, last = ((long)bh * S + S) * 4 * NW - 1;
                c_bits[0] = a.keepbits[i < last ? i : last];
            }
        }
    };
    auto store_tr2 = [&](unsigned char* lds, const u32x4& x0, const u32x4& x1) {
        const int pitch = tr_pitch<bf16>(CQ);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t a0 = x0[w], b0 = x1[w];
            *(uint32_t*)(lds + (sdc * 8 + 2 * w) * pitch + sr * 2) = (a0 & 0xFFFFu) | (b0 << 16);
            *(uint32_t*)(lds + (sdc * 8 + 2 * w + 1) * pitch + sr * 2) = (a0 >> 16) | (b0 & 0xFFFF0000u);
        }
    };
    auto store_chunk = [&](int q0) {
        // the fetched registers are consumed from here on, not earlier (the compiler would hoist the selects and the
        // D products -- and with them the wait for the fetch -- above the phase the fetch is supposed to run under)
        vb_pin(c0); vb_pin(c1); vb_pin(o0); vb_pin(o1); vb_pin(c_lse);
#pragma unroll
        for (int j = 0; j < BPT; ++j) vb_pin(c_bits[j]);
        const u32x4 z0 = zsel(q0 + sr < S, c0), z1 = zsel(q0 + sr + 1 < S, c1);        // rows past S are zeros
        if (role > 0) {
            unsigned char* rm = role == 2 ? ldsQ : ldsDO;
            *(u32x4*)(rm + rm_off<T>(sr, sdc)) = z0;
            *(u32x4*)(rm + rm_off<T>(sr + 1, sdc)) = z1;
            store_tr2(role == 2 ? ldsQT : ldsDOT, z0, z1);
        } else {                                           // D[q] = dO[q] . O[q]: 8 products per thread, 8 threads per row
            const bf16x8 d0 = *(const bf16x8*)&z0, d1 = *(const bf16x8*)&z1, p0 = *(const bf16x8*)&o0, p1 = *(const bf16x8*)&o1;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { s0 += (float)d0[j] * (float)p0[j]; s1 += (float)d1[j] * (float)p1[j]; }
            s0 = oct_sum(s0); s1 = oct_sum(s1);
            if (sdc == 0) { ldsD[sr] = s0; ldsD[sr + 1] = s1; }
            // lse in units of log 2 (phase A computes exp2 of one packed fma); exp2(x - inf) = 0 for padded queries
            if (wave == 0) ldsLse[lane] = q0 + lane < S ? c_lse * 1.44269504088896340736f : INFINITY;
            // keep words, re-laid for phase A: a lane there needs the SAME 32-bit half (its wave's key fragment) and the same
            // key group g of the four queries lg*4 + 0..3 -- stored as [qf][lg][g][half][r] they are one 16-byte read instead
            // of four address computations and four 4-byte reads
            const int i = t, ql = i >> 2, g = i & 3;
            const uint64_t kw = (a.p > 0.f && q0 + ql < S) ? c_bits[0] : ~(uint64_t)0;
            uint32_t* bw = (uint32_t*)ldsBits + (((ql >> 2) * 4 + g) * 8 + (ql & 3));
            bw[0] = (uint32_t)kw; bw[4] = (uint32_t)(kw >> 32);
        }
    };
    load_chunk(0);
    const float mk = kok ? mk_raw : -INFINITY;
    const float mk2 = mk * 1.44269504088896340736f, sc2 = a.scale * 1.44269504088896340736f;
    {   // K^T image and a zeroed dS tile
        const u32x4 x0 = zsel(kr < S, kx0), x1 = zsel(kr + 1 < S, kx1);
        const int pitch = KTP;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t lo = x0[w], hi = x1[w];
            *(uint32_t*)(ldsKT + (kdc * 8 + 2 * w) * pitch + kr * 2) = (lo & 0xFFFFu) | (hi << 16);
            *(uint32_t*)(ldsKT + (kdc * 8 + 2 * w + 1) * pitch + kr * 2) = (lo >> 16) | (hi & 0xFFFF0000u);
        }
        for (int i = t; i < CQ * TSP / 8; i += FNT) *(uint64_t*)(ldsDS + i * 8) = 0;
    }
    use_set(0);
    store_chunk(0);
    __syncthreads();                                       // chunk 0, the K^T image and the zeroed tile are in LDS
    if (CQ < S) load_chunk(CQ);

    // q/k/v bias gradients = column sums of dqkv over tokens.  This kernel holds dK^T, dV^T (and, block by block, dQ^T) in
    // fp32 registers with the token index along the LANES (li): a 16-lane butterfly per value, once per workgroup, gives the
    // per-sample sums, which leave through a [B][3H] workspace (one slot per workgroup: no same-address global atomics --
    // those cost +140 us per layer in round 1) and a tiny reduction kernel.  Replaces an 85 us pass over dqkv per layer.
    f32x4 dqsum = f32x4{0.f, 0.f, 0.f, 0.f};               // this wave's phase-B blocks all have df = wave & 3 (12 % 4 == 0)
    int cur = 0;
    for (int q0 = 0; q0 < S; q0 += CQ, cur ^= 1) {
        use_set(cur);
        // ---- phase A
        if (wave_on) {
#pragma unroll
            for (int qc = 0; qc < CQ / 32; ++qc) {
                if (q0 + qc * 32 >= S) continue;
                f32x4 pd[2], dsv[2];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int qf = 2 * qc + hf;            // fragment index inside the chunk
                    if (hf == 1 && q0 + qf * 16 >= S) {    // wave-uniform: the second half of the LAST 32-query block is all padding
                        pd[1] = f32x4{0.f, 0.f, 0.f, 0.f}; dsv[1] = pd[1];      // (S = 164: one fragment in twelve); phase B skips its rows too
                        continue;
                    }
                    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        s = vb_mma(frag_rm(ldsQ, qf * 16 + li, ks, lg, T()), kb[ks], s);
                        dp = vb_mma(frag_rm(ldsDO, qf * 16 + li, ks, lg, T()), vb[ks], dp);
                    }
                    // lane: key = kf*16 + li (column), queries q0 + qf*16 + lg*4 + r (rows)
                    const f32x4 lse4 = *(const f32x4*)(ldsLse + qf * 16 + lg * 4);         // already times log2(e)
                    const f32x4 d4 = *(const f32x4*)(ldsD + qf * 16 + lg * 4);
                    // this wave's nibble of the keep words lies in ONE 32-bit half (kf is wave-uniform); the four queries'
                    // halves are adjacent (see store_chunk); one select per probability yields the factor for both products
                    const u32x4 w4 = *(const u32x4*)((const uint32_t*)ldsBits + (((qf * 4 + lg) * 4 + (li >> 2)) * 2 + ((kf & 15) >> 3)) * 4);
                    // p = exp(s * scale + mask - lse) = exp2(s * (scale log2 e) + (mask log2 e - lse log2 e)): two packed
                    // instructions per two probabilities instead of an fma, a subtract and a multiply each
                    f32x4 xe;
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f32x2 c = vb_splat2(mk2) - f32x2{lse4[r], lse4[r + 1]};
                        const f32x2 x = vb_fma2(f32x2{s[r], s[r + 1]}, vb_splat2(sc2), c);
                        xe[r] = x[0]; xe[r + 1] = x[1];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ql = qf * 16 + lg * 4 + r;
                        const float p = fast_exp2(xe[r]);
                        const float kscale = ((w4[r] >> ((kf & 7) * 4 + (li & 3))) & 1u) ? a.inv_keep : 0.f;
                        const float pdrop = p * kscale;
                        const float dpv = dp[r] * kscale;
                        pd[hf][r] = pdrop;
                        dsv[hf][r] = p * (dpv - d4[r]);             // dS / scale: the factor goes onto dK^T and dQ^T once (below)
                        *(bf16*)(ldsDS + ql * TSP + (kf * 16 + li) * 2) = (bf16)dsv[hf][r];      // dS / scale as [query][key]
                    }
                }
                bf16x8 pb, dsb;
                pack_b(pb, pd[0], pd[1]);
                pack_b(dsb, dsv[0], dsv[1]);
#pragma unroll
                for (int df = 0; df < 4; ++df) {
                    dvT[df] = vb_mma(frag_tr(ldsDOT, tr_pitch<T>(CQ), df * 16 + li, qc, lg, T()), pb, dvT[df]);
                    dkT[df] = vb_mma(frag_tr(ldsQT, tr_pitch<T>(CQ), df * 16 + li, qc, lg, T()), dsb, dkT[df]);
                }
            }
        }
        if (q0 + CQ < S) {                                 // the next chunk's images go to the other set meanwhile
            use_set(cur ^ 1);
            store_chunk(q0 + CQ);
        }
        __syncthreads();                                   // the chunk's dS tile is complete, the next chunk is staged
        if (q0 + 2 * CQ < S) load_chunk(q0 + 2 * CQ);
        // ---- phase B: dQ^T block (d rows df*16.., query columns qf*16..) = K^T dS^T over all keys.  16 blocks, 12 waves:
        // wave w owns block w (qf = w >> 2, df = w & 3) and waves 0..3 also block w + 12 (qf = 3, the SAME df): the two are
        // computed together -- one K^T fragment feeds both, and their MFMA chains (six dependent instructions each) overlap
        // instead of running back to back on the four waves the whole workgroup then waits for
        {
            static_assert(CQ == 64 && FWPB == 12, "block ownership below assumes 16 blocks on 12 waves");
            const int df = wave & 3, qf0 = wave >> 2, qf1 = 3;
            const bool v0 = q0 + qf0 * 16 < S, v1 = wave < 4 && q0 + qf1 * 16 < S;      // wave-uniform
            if (v0 || v1) {
                f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
                for (int ks = 0; ks < FNK / 32; ++ks) {
                    if (ks * 32 >= S) continue;
                    // B operand: lane (n = query li, g = lg) holds dS[q][key(g, j)], key(g, j) = 32 ks + 16 (j >> 2) + 4 g + (j & 3)
                    const bf16x8 kt = frag_tr(ldsKT, KTP, df * 16 + li, ks, lg, T());
                    if (v0) {
                        const unsigned char* src = ldsDS + (qf0 * 16 + li) * TSP + (32 * ks + 4 * lg) * 2;
                        const bf16x4 lo = *(const bf16x4*)src, hi = *(const bf16x4*)(src + 32);
                        acc0 = vb_mma(kt, bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}, acc0);
                    }
                    if (v1) {
                        const unsigned char* src = ldsDS + (qf1 * 16 + li) * TSP + (32 * ks + 4 * lg) * 2;
                        const bf16x4 lo = *(const bf16x4*)src, hi = *(const bf16x4*)(src + 32);
                        acc1 = vb_mma(kt, bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}, acc1);
                    }
                }
                // dQ rows: 32-bit byte offsets from the scalar base (the launcher checks tokens x pitch x 2 < 2^32) -- a 64-bit lane
                // address here is a register pair the kernel (at its 168-register budget) spills, and its scratch reload's
                // vmcnt(0) also waits for the chunk fetch in flight
                const unsigned dq_col = (unsigned)(h * D + df * 16 + lg * 4) * 2u;
                unsigned char* const dq_base = (unsigned char*)a.dqkv;
                if (v0) {
                    const int q = q0 + qf0 * 16 + li;
                    acc0 *= a.scale;                        // the score scale left out of dS in phase A
                    if (q < S) store4((T*)(dq_base + ((unsigned)((int)row0 + q) * (unsigned)((int)ldx * 2) + dq_col)), acc0);
                    dqsum += acc0;                          // columns of padded queries are exactly 0 (their dS rows are)
                }
                if (v1) {
                    const int q = q0 + qf1 * 16 + li;
                    acc1 *= a.scale;
                    if (q < S) store4((T*)(dq_base + ((unsigned)((int)row0 + q) * (unsigned)((int)ldx * 2) + dq_col)), acc1);
                    dqsum += acc1;
                }
            }
        }
        __syncthreads();                                   // phase B is done with the tile before the next phase A writes it
    }
#pragma unroll
    for (int df = 0; df < 4; ++df) dkT[df] *= a.scale;     // dK^T was accumulated from dS / scale (phase A): one multiply per output
    {   // 16-byte stores (store4x2: every lane takes part in the lane exchange, the store is predicated on a real key)
        T* dkrow = (T*)a.dqkv + (row0 + keyc) * ldx + H + h * D;
        T* dvrow = (T*)a.dqkv + (row0 + keyc) * ldx + 2 * H + h * D;
        const bool wide = wide_ok(a.dqkv, ldx);
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            store4x2(dkrow + dj * 32, lg, dkT[2 * dj], dkT[2 * dj + 1], kok, wide);
            store4x2(dvrow + dj * 32, lg, dvT[2 * dj], dvT[2 * dj + 1], kok, wide);
        }
    }
    if (a.bias_ws) {
        // every LDS image is idle now (the loop ended on a barrier): each lane parks its 36 accumulators (4 dQ sums, 16 dK^T,
        // 16 dV^T) as raw[wave][value][lane] -- 36 conflict-free ds_write_b32 -- and 192 threads add up, per output column d,
        // the 16 lanes (keys / queries) x the waves that hold it.  (The first form reduced inside each wave with 36 DPP
        // row sums per lane before going to LDS: ~5000 of the workgroup's ~45000 cycles, by the in-kernel cycle trace.)
        float* raw = (float*)smem;                          // 12 x 36 x 64 floats = 108 KB of the 124 KB
        constexpr int NV = 4 + 2 * 16;
        static_assert((size_t)FWPB * NV * 64 * 4 <= 2 * (size_t)SETB + KT_BYTES + (size_t)CQ * TSP, "raw partials fit the idle images");
        float* mine = raw + ((long)wave * NV) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[r * 64] = dqsum[r];
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int r = 0; r < 4; ++r) {                   // padded keys contribute exact zeros
                mine[(4 + df * 4 + r) * 64] = dkT[df][r];
                mine[(20 + df * 4 + r) * 64] = dvT[df][r];
            }
        __syncthreads();
        if (t < 3 * D) {
            const int which = t / D, dd = t % D, df = dd >> 4, glg = (dd >> 2) & 3, r = dd & 3;
            float sum = 0.f;
            if (which == 0) {                               // dQ^T block df lives in the waves with (wave & 3) == df
                for (int w = df; w < FWPB; w += 4) {
                    const f32x4* src = (const f32x4*)(raw + ((long)w * NV + r) * 64 + glg * 16);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const f32x4 x = src[i]; sum += (x[0] + x[1]) + (x[2] + x[3]); }
                }
            } else {
                const int v = (which == 1 ? 4 : 20) + df * 4 + r;
                for (int w = 0; w < FWPB; ++w) {
                    const f32x4* src = (const f32x4*)(raw + ((long)w * NV + v) * 64 + glg * 16);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const f32x4 x = src[i]; sum += (x[0] + x[1]) + (x[2] + x[3]); }
                }
            }
            a.bias_ws[((long)b * 3 + which) * H + h * D + dd] = sum;
        }
    }
}

// out[c] += sum over samples of ws[b][c]  (c < 3H): 256 columns x a slice of the batch per workgroup
VB_KERNEL VB_LAUNCH_BOUNDS(256) attn_bias_reduce_kernel(const float* ws, float* out, int B, int C, int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int b0 = blockIdx.y * rows_per_block;
    const int b1 = b0 + rows_per_block < B ? b0 + rows_per_block : B;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int bi = b0;
    for (; bi + 3 < b1; bi += 4) {
        s0 += ws[(long)bi * C + c]; s1 += ws[(long)(bi + 1) * C + c];
        s2 += ws[(long)(bi + 2) * C + c]; s3 += ws[(long)(bi + 3) * C + c];
    }
    for (; bi < b1; ++bi) s0 += ws[(long)bi * C + c];
    vb_atomic_add_noret(out + c, (s0 + s1) + (s2 + s3));
}
template <int NKF, int CQ> size_t fused_smem() {
    return 2 * (2 * rm_bytes<bf16>(CQ) + 2 * tr_bytes<bf16>(CQ) + 2 * CQ * 4 + (size_t)CQ * 4 * ((NKF + 15) / 16) * 8) +
           KT_BYTES + (size_t)CQ * TSP;
}

template <typename T, int NKF> size_t fwd_smem() { return rm_bytes<T>(NKF * 16) + tr_bytes<T>((NKF + 1) / 2 * 32) + NKF * 16 * 4; }
template <typename T, int NKF> size_t dq_smem() { return 2 * rm_bytes<T>(NKF * 16) + tr_bytes<T>(NKF * 16) + NKF * 16 * 4; }
template <typename T, int NKF> size_t dkv_smem() {
    return 2 * rm_bytes<T>(QC) + 2 * tr_bytes<T>(QC) + 2 * QC * 4 + (size_t)QC * 4 * ((NKF + 15) / 16) * 8;
}

template <typename T> size_t fwd_tiled_smem() { return rm_bytes<T>(TCK) + tr_bytes<T>(TCK) + TCK * 4; }
template <typename T> size_t dq_tiled_smem() { return 2 * rm_bytes<T>(TCK) + tr_bytes<T>(TCK) + TCK * 4; }

constexpr size_t kMaxLds = 160 * 1024;


template <typename T, int NKF>
int launch_all(int which, const AttnArgs& a, hipStream_t s) {
    dim3 grid((unsigned)(a.B * a.nh)), block(NT);
    constexpr int NW = (NKF + 15) / 16;                     // keep-bit words: the layout every kernel of this NKF family assumes
    if (which == 0) {
        const size_t sm = fwd_smem<T, NKF>();
        if (sm > kMaxLds) {                                 // K / V of the whole sequence do not fit: stream the keys in chunks
            VB_LAUNCH((attn_fwd_tiled_kernel<T, NW>), grid, block, fwd_tiled_smem<T>(), s, a);
            return vb_check_launch();
        }
        if constexpr (sizeof(T) == 4 && NKF <= 12 && NKF > 4) {      // 4-byte tiles: one workgroup per CU, so it brings 12 / 8 waves itself
            constexpr int NTHF = NKF > 8 ? 768 : 512;
            VB_LAUNCH((attn_fwd_kernel<T, NKF, false, false, NTHF>), grid, dim3(NTHF), sm, s, a);
        } else if constexpr (sizeof(T) == 2 && NKF >= 20) {
            // long bf16 sequences (NLVR2 as the reference runs it, S = 416: 106 KB of K / V^T): ONE workgroup per CU here too, and the
            // 256-thread form was compiled for three (168 VGPRs: 340 bytes of scratch at 26 key fragments) -- eight waves at 256
            // registers keep all scores in registers and give every SIMD a second wave to hide LDS / HBM latency behind
            VB_LAUNCH((attn_fwd_kernel<T, NKF, false, false, 512>), grid, dim3(512), sm, s, a);
        } else {
            VB_LAUNCH((attn_fwd_kernel<T, NKF>), grid, block, sm, s, a);
        }
    } else {
        // (S <= 64: only 4 of the one-pass kernel's 12 waves own a key fragment -- the two-pass form plus the separate
        //  bias-gradient pass is faster there: 349 + ~85 us against 483 us per layer at 86 k tokens, S = 56)
        if constexpr (sizeof(T) == 2 && NKF <= 12 && NKF > 4) {
            if (a.ctx_fwd && a.qkv && a.Sq == a.S && a.S <= FWPB * 16 && vb_opts_for((void*)s).attn_two_pass != 1 &&
                (long)a.B * a.S * 3 * a.nh * D * 2 < (1L << 32)) {   // one-pass backward (needs the forward output; 32-bit byte offsets)
                // 64-query chunks (86 KB of LDS, one workgroup per CU): 528-541 us per layer at B=512; 32-query chunks (two
                // workgroups per CU, twice the barriers): 575 us
                VB_LAUNCH((attn_bwd_fused_kernel<NKF, 64>), grid, dim3(FNT), (fused_smem<NKF, 64>()), s, a);
                return vb_check_launch() == VB_OK ? 1 : VB_ERR_LAUNCH;      // 1: the bias workspace (if any) was filled
            }
        }
        const size_t sm1 = dq_smem<T, NKF>(), sm2 = dkv_smem<T, NKF>();
        if (sm2 > kMaxLds) return VB_ERR_UNSUPPORTED;
        if (sm1 > kMaxLds) VB_LAUNCH((attn_bwd_dq_tiled_kernel<T, NW>), grid, block, dq_tiled_smem<T>(), s, a);
        // (the long bf16 forms, NKF >= 20, stay at four waves: the dQ pass holds P and dP of ALL keys -- 208 registers at 26 fragments --
        //  and needs the 512-register budget of one wave per SIMD: eight waves spill 840 bytes per lane)
        else if constexpr (sizeof(T) == 4 && NKF <= 12 && NKF > 4) VB_LAUNCH((attn_bwd_dq_kernel<T, NKF, 512>), grid, dim3(512), sm1, s, a);
        else VB_LAUNCH((attn_bwd_dq_kernel<T, NKF>), grid, block, sm1, s, a);
        dim3 grid2((unsigned)(a.B * a.nh), (unsigned)(((a.S + 15) / 16 + 3) / 4));
        VB_LAUNCH((attn_bwd_dkv_kernel<T, NKF>), grid2, block, sm2, s, a);
    }
    return vb_check_launch();
}

template <typename T>
int dispatch_nkf(int which, const AttnArgs& a, hipStream_t s) {
    const int nkf = ((a.S + 31) / 32) * 2;          // key fragments, padded to an even count
    if constexpr (sizeof(T) == 2) {
        // forward with EXACTLY ceil(S / 16) key fragments of work for the sequence lengths of BASELINE's configurations
        // (S = 164 pre-training: 11 fragments; NLVR2 S = 112: 7, where the padded-to-even form computed 8; VQA S = 56: 4) -- the
        // prefetching, branch-free instantiation
        const int nf = (a.S + 15) / 16;
        if (which == 0 && (nf == 11 || nf == 7 || nf == 4)) {
            const dim3 grid((unsigned)(a.B * a.nh));
            if (nf == 11) VB_LAUNCH((attn_fwd_kernel<T, 11, true, true>), grid, dim3(NT), (fwd_smem<T, 11>()), s, a);
            else if (nf == 7) VB_LAUNCH((attn_fwd_kernel<T, 7, true, true>), grid, dim3(NT), (fwd_smem<T, 7>()), s, a);
            else VB_LAUNCH((attn_fwd_kernel<T, 4, true, true>), grid, dim3(NT), (fwd_smem<T, 4>()), s, a);
            return vb_check_launch();
        }
    }
    if (which == 0 && (a.S + 15) / 16 == 11) {      // forward at S = 161..176 without the prefetch (fp32)
        const size_t sm = fwd_smem<T, 11>();
        // (without the prefetch the branch-free form lets the scheduler hoist every fragment read: 612-1448 bytes of scratch, 4x slower)
        if constexpr (sizeof(T) == 4) VB_LAUNCH((attn_fwd_kernel<T, 11, false, false, 768>), dim3((unsigned)(a.B * a.nh)), dim3(768), sm, s, a);
        else VB_LAUNCH((attn_fwd_kernel<T, 11>), dim3((unsigned)(a.B * a.nh)), dim3(NT), sm, s, a);
        return vb_check_launch();
    }
    if (nkf <= 4) return launch_all<T, 4>(which, a, s);
    if (nkf <= 8) return launch_all<T, 8>(which, a, s);
    if (nkf <= 12) return launch_all<T, 12>(which, a, s);
    if (nkf <= 16) return launch_all<T, 16>(which, a, s);
    if constexpr (sizeof(T) == 2) {
        // long sequences (NLVR2 as the reference really runs it: 2 x 144 regions + 128 tokens, S = 416).  The dQ pass keeps
        // K, V and K^T of the whole sequence in LDS: 26 key fragments is what 160 KB holds; beyond that only forward fits.
        if (nkf <= 20) return launch_all<T, 20>(which, a, s);
        if (nkf <= 24) return launch_all<T, 24>(which, a, s);
        if (nkf <= 26) return launch_all<T, 26>(which, a, s);
    }
    if (nkf <= 32) return launch_all<T, 32>(which, a, s);
    return VB_ERR_UNSUPPORTED;
}

int fill_args(AttnArgs& a, int B, int S, int nh, int head_dim, float p, uint64_t seed, uint32_t stream_id) {
    if (B <= 0 || S <= 0 || nh <= 0 || head_dim != D || p < 0.f || p >= 1.f) return VB_ERR_ARG;
    a.B = B; a.S = S; a.Sq = S; a.nh = nh; a.scale = 0.125f; // 1/sqrt(64), modeling.py:242
    a.p = p; a.inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    a.thresh = vb_drop_thresh16(p); a.stream = stream_id; a.seed = seed;
    return VB_OK;
}

// self-attention over the packed qkv matrix expressed in the general form: three column blocks, one pitch
void set_self(AttnArgs& a, int dtype, const void* qkv, void* dqkv) {
    const long H = (long)a.nh * D, es = dtype == VB_BF16 ? 2 : 4;
    a.Sq = a.S;
    a.q = qkv; a.k = (const char*)qkv + H * es; a.v = (const char*)qkv + 2 * H * es;
    a.ldq = a.ldk = a.ldv = 3 * H;
    a.ldc = a.lddo = H;
    a.dq = dqkv;
    a.dk = dqkv ? (char*)dqkv + H * es : nullptr;
    a.dv = dqkv ? (char*)dqkv + 2 * H * es : nullptr;
    a.lddq = a.lddk = a.lddv = 3 * H;
}

int keepbits_nw(int Sk) {
    const int nkf = ((Sk + 31) / 32) * 2;
    const int nkft = nkf <= 4 ? 4 : nkf <= 8 ? 8 : nkf <= 12 ? 12 : nkf <= 16 ? 16 : 32;   // 17..32 fragments: 2 words either way
    return (nkft + 15) / 16;
}

}  // namespace

// ---- cross-attention (LXRT: unsupervised_visualbert/src/lxrt/modeling.py:347-411 BertAttention with context != hidden_states) -----
extern "C" int64_t vb_attn_cross_keepbits_words(int Sq, int Sk) { return (int64_t)Sq * 4 * keepbits_nw(Sk); }

extern "C" int vb_attn_cross_fwd(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                 const float* mask_add, void* ctx, int64_t ldctx, float* lse, uint64_t* keepbits,
                                 int B, int Sq, int Sk, int nh, int head_dim, float p_drop, uint64_t seed, uint32_t stream_id,
                                 void* stream) {
    AttnArgs a{};
    int rc = fill_args(a, B, Sk, nh, head_dim, p_drop, seed, stream_id);
    if (rc) return rc;
    if (Sq <= 0 || !q || !k || !v || !mask_add || !ctx || (p_drop > 0.f && !keepbits)) return VB_ERR_ARG;
    if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (ldctx % 4)) return VB_ERR_ARG;          // 16-byte vector loads per (row, head)
    a.Sq = Sq; a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldc = ldctx;
    a.mask_add = mask_add; a.ctx = ctx; a.lse = lse; a.keepbits = keepbits;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VB_BF16) return dispatch_nkf<bf16>(0, a, s);
    if (dtype == VB_F32) return dispatch_nkf<float>(0, a, s);
    if (dtype == VB_BF16X3) return dispatch_nkf<xf32>(0, a, s);
    return VB_ERR_ARG;
}

extern "C" int vb_attn_cross_bwd(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                 const float* mask_add, const void* dctx, int64_t lddctx, const float* lse,
                                 const uint64_t* keepbits, float* dsum_ws, void* dq, int64_t lddq, void* dk, int64_t lddk,
                                 void* dv, int64_t lddv, int B, int Sq, int Sk, int nh, int head_dim, float p_drop,
                                 uint64_t seed, uint32_t stream_id, void* stream) {
    AttnArgs a{};
    int rc = fill_args(a, B, Sk, nh, head_dim, p_drop, seed, stream_id);
    if (rc) return rc;
    if (Sq <= 0 || !q || !k || !v || !mask_add || !dctx || !lse || !dsum_ws || !dq || !dk || !dv || (p_drop > 0.f && !keepbits))
        return VB_ERR_ARG;
    if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (lddctx % 8) || (lddq % 4) || (lddk % 4) || (lddv % 4)) return VB_ERR_ARG;
    a.Sq = Sq; a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.lddo = lddctx;
    a.dq = dq; a.dk = dk; a.dv = dv; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    a.mask_add = mask_add; a.dctx = dctx; a.lse = (float*)lse; a.keepbits = (uint64_t*)keepbits; a.dsum = dsum_ws;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VB_BF16) rc = dispatch_nkf<bf16>(1, a, s);
    else if (dtype == VB_F32) rc = dispatch_nkf<float>(1, a, s);
    else if (dtype == VB_BF16X3) rc = dispatch_nkf<xf32>(1, a, s);
    else return VB_ERR_ARG;
    return rc < 0 ? rc : VB_OK;
}

extern "C" int64_t vb_attn_keepbits_words(int S) {
    return (int64_t)S * 4 * keepbits_nw(S);                   // uint64 words per (batch, head)
}

extern "C" int vb_attn_fwd(int dtype, const void* qkv, const float* mask_add, void* ctx, float* lse,
                           uint64_t* keepbits, int B, int S, int nh, int head_dim,
                           float p_drop, uint64_t seed, uint32_t stream_id, void* stream) {
    return vb_attn_fwd_sp(dtype, qkv, mask_add, ctx, lse, keepbits, B, S, nh, head_dim, p_drop, seed, stream_id, nullptr, 0, stream);
}

int vb_attn_fwd_sp(int dtype, const void* qkv, const float* mask_add, void* ctx, float* lse,
                   uint64_t* keepbits, int B, int S, int nh, int head_dim,
                   float p_drop, uint64_t seed, uint32_t stream_id, void* ctx_split, int split_only, void* stream) {
    AttnArgs a{};
    int rc = fill_args(a, B, S, nh, head_dim, p_drop, seed, stream_id);
    if (rc) return rc;
    if (!qkv || !mask_add || (!ctx && !(ctx_split && split_only)) || (p_drop > 0.f && !keepbits)) return VB_ERR_ARG;
    if (ctx_split && (dtype != VB_BF16X3 || (((uintptr_t)ctx_split) & 7))) return VB_ERR_ARG;
    a.qkv = qkv; a.mask_add = mask_add; a.ctx = ctx; a.lse = lse; a.keepbits = keepbits;
    a.ctx_sp = (bf16*)ctx_split; a.sp_only = (ctx_split && split_only) ? 1 : 0;
    set_self(a, dtype, qkv, nullptr);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VB_BF16) return dispatch_nkf<bf16>(0, a, s);
    if (dtype == VB_F32) return dispatch_nkf<float>(0, a, s);
    if (dtype == VB_BF16X3) return dispatch_nkf<xf32>(0, a, s);
    return VB_ERR_ARG;
}


extern "C" int64_t vb_attn_bwd_ws_floats(int B, int S, int nh) {
    const int64_t d = (int64_t)B * nh * S, bias = (int64_t)B * 3 * nh * D;
    return d > bias ? d : bias;
}

extern "C" int vb_attn_bwd(int dtype, const void* qkv, const float* mask_add, const void* dctx, const float* lse,
                           const uint64_t* keepbits, float* dsum_ws, void* dqkv, const void* ctx_fwd, float* dqkv_bias,
                           int B, int S, int nh, int head_dim,
                           float p_drop, uint64_t seed, uint32_t stream_id, void* stream) {
    return vb_attn_bwd_sp(dtype, qkv, mask_add, dctx, lse, keepbits, dsum_ws, dqkv, ctx_fwd, dqkv_bias, B, S, nh, head_dim, p_drop, seed,
                          stream_id, nullptr, 0, stream);
}

int vb_attn_bwd_sp(int dtype, const void* qkv, const float* mask_add, const void* dctx, const float* lse,
                   const uint64_t* keepbits, float* dsum_ws, void* dqkv, const void* ctx_fwd, float* dqkv_bias,
                   int B, int S, int nh, int head_dim,
                   float p_drop, uint64_t seed, uint32_t stream_id, void* dqkv_split, int split_only, void* stream) {
    AttnArgs a{};
    int rc = fill_args(a, B, S, nh, head_dim, p_drop, seed, stream_id);
    if (rc) return rc;
    if (!qkv || !mask_add || !dctx || !lse || !dsum_ws || (!dqkv && !(dqkv_split && split_only)) || (p_drop > 0.f && !keepbits)) return VB_ERR_ARG;
    if (dqkv_split && (dtype != VB_BF16X3 || (((uintptr_t)dqkv_split) & 7))) return VB_ERR_ARG;
    a.qkv = qkv; a.mask_add = mask_add; a.dctx = dctx; a.lse = (float*)lse; a.keepbits = (uint64_t*)keepbits;
    a.dsum = dsum_ws; a.dqkv = dqkv; a.ctx_fwd = ctx_fwd;
    a.dqkv_sp = (bf16*)dqkv_split; a.sp_only = (dqkv_split && split_only) ? 1 : 0;
    set_self(a, dtype, qkv, dqkv);
    a.bias_ws = dqkv_bias ? dsum_ws : nullptr;              // the one-pass kernel does not need D in memory: same scratch
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VB_BF16) rc = dispatch_nkf<bf16>(1, a, s);
    else if (dtype == VB_F32) rc = dispatch_nkf<float>(1, a, s);
    else if (dtype == VB_BF16X3) rc = dispatch_nkf<xf32>(1, a, s);
    else return VB_ERR_ARG;
    if (rc < 0 || !dqkv_bias) return rc < 0 ? rc : VB_OK;
    const int C = 3 * nh * D;
    if (rc == 1) {                                          // per-sample partial sums are in the workspace
        // (deferred form: 32 row groups per workgroup already split the samples -- row slices only when there are many)
        if (vb_reduce_defer((const float*)dsum_ws, dqkv_bias, B, C, C, B >= 256 ? 4 : 1)) return vb_check_launch();
        const int slices = B >= 64 ? 16 : (B >= 8 ? 4 : 1);
        const int rows = (B + slices - 1) / slices;
        VB_LAUNCH(attn_bias_reduce_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)((B + rows - 1) / rows)), dim3(256), 0, s,
                  (const float*)dsum_ws, dqkv_bias, B, C, rows);
        return vb_check_launch();
    }
    // two-pass kernels (fp32 parity mode, long sequences): one column-sum pass over dqkv
    if (a.sp_only) {
        // no fp32 dqkv exists: the sums of the image's two planes (bf16 [B S, 2 C]: hi | lo) -- the same bytes as one fp32 pass
        return vb_colsum_image(dqkv_split, 2 * C, dqkv_bias, B * S, C, stream);
    }
    return vb_colsum(dtype == VB_BF16 ? VB_BF16 : VB_F32, dqkv, C, dqkv_bias, nullptr, B * S, C, stream);
}
