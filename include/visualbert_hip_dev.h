/* visualbert_hip_dev.h -- developer-only entry points of libvisualbert_hip_dev.so (built with -DVB_DEV_KNOBS).
 *
 * NOT part of the product ABI (include/visualbert_hip.h) and not exported by libvisualbert_hip.so: ablation switches that
 * make results WRONG, in-kernel timelines and pure measurement kernels.  tools/*.py and bench.py's MFMA-ceiling
 * measurement bind this library; the visualbert_amd package never needs it.  These knobs are process-wide state. */
#ifndef VISUALBERT_HIP_DEV_H
#define VISUALBERT_HIP_DEV_H

#include "visualbert_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* vb_stream_opts.nt_kernel values accepted by the DEVELOPER library only (the product returns VB_ERR_ARG for them):
 *   80  = persistent 256x256 tile, eight slots per K tile (81 is the four-slot form that ships)
 *   82  = kernel 81 with the REGISTER-DIRECT epilogue (round 5: swapped MFMA operand roles, v_permlane16_swap pairs, one 16-byte store per
 *         lane and no LDS transposition) in the specialised instantiations; measured 3-12 % slower than the LDS route on every shape
 *   91  = the two-workgroups-per-CU kernel with the copies issued ahead of the fragment reads
 *   92  = kernel 90 with the register-direct epilogue (as 82)
 *   100 = persistent 256x256 tile with four waves, 128x128 outputs each, inline-asm K loop (K / 64 even, else 90)
 *   101 = the same tile with B fetched straight into fragment registers (1 x 4 waves; K / 64 % 4 == 0 and N % 256 == 0, else 90)
 *   200 = the vendor yardstick: plain GEMMs (bias only, or "+ addend") are handed to hipBLASLt (csrc/vendor_gemm.hip: dlopen'ed
 *         on first use, one 64 MB workspace per (device, stream)); fused epilogues and whatever the library declines stay ours */

/* ablation switch for kernel analysis (results are WRONG when non-zero): 1 skip tile loads, 2 skip fragment
 * reads, 4 skip MFMAs in the pipelined kernel, 128 predicate the epilogue's global stores off */
/* bit 26 (1 << 26): the bf16 epilogues' streaming (`nt`) stores replaced by plain stores -- results are IDENTICAL (this bit does not
 * make them wrong): the bit-compare arm of tests/test_kernels.py::test_streaming_stores_equal_plain_stores */
/* bit 27 (1 << 27): the round-4 dispatch rule for "+ addend" GEMMs with K >= 2048 (two-workgroup kernel instead of the persistent one);
 * results identical up to summation order */
/* bit 29 (1 << 29): arms the dropout form of the "+ residual" GEMM epilogue (csrc/gemm.hip: vb_gemm_dropres / EPI_DROP) that
 * vb_bert_layer_fwd asks for when hidden dropout is on -- results are CORRECT (tests/test_bench_shape.py replays the mask); it lost its
 * A/B against dropout + residual in the LayerNorm launch (profiles/r06_dropres_epilogue_ab.txt), so the product library declines it */
int vb_gemm_set_debug(int bits);
/* debug bit 64 (256x128 pipelined kernel, bf16): waves 0 and 4 of workgroup 0 write per-K-tile shader-clock stamps
 * {landed, barrier, copies issued, frags0, mfma0, frags1, mfma1} to this device buffer (uint64[2][64][8]) */
int vb_gemm_set_trace(void* device_u64x1024);
/* MFMA issue-rate ceiling micro-kernel (measurement aid): kind 0 = 16x16x32 bf16, 1 = 32x32x16 bf16, 2 = kind 0 with
 * operands that change every instruction; each wave of each 512-thread block issues iters x 524288 FLOP; out: fp32[blocks*512] sink */
int vb_mfma_peak(int kind, int iters, int blocks, float* out, void* stream);
/* global -> LDS (LDS-direct) streaming ceiling (measurement aid): each wave of each 512-thread block streams iters
 * 1-KiB pieces from a span-byte window with `depth` (1,2,4,8,16) pieces in flight */
int vb_glds_stream(int depth, const void* src, int64_t span, int iters, int blocks, float* sink, void* stream);

/* groundwork for the block-scaled fp8 cross terms of the split-operand product; not used by the product.
 * vb_mma_f8_probe: ONE v_mfma_scale_f32_16x16x128_f8f6f4 through csrc/vb_rt.h's vb_mma_f8 with the documented lane layout:
 *   A, B: [16][128] OCP e4m3 bytes (K contiguous); scale_a, scale_b: [16][4] E8M0 bytes per (row, 32-element K block);
 *   D[16][16] fp32 = sum_k A[i][k] 2^(scale_a[i][k/32] - 127) B[j][k] 2^(scale_b[j][k/32] - 127)
 * vb_cvt_fp8_probe: y[n] e4m3 bytes = round-to-nearest-even of x[n] (|x| <= 448, n % 4 == 0) through vb_cvt4_fp8 */
int vb_mma_f8_probe(const void* A, const void* B, const void* scale_a, const void* scale_b, float* D, void* stream);
int vb_cvt_fp8_probe(const float* x, void* y, int n, void* stream);
/* PROTOTYPE of the split-operand GEMM with its two cross terms on the fp8 pipe (csrc/gemm.hip, end of file).
 * vb_split_f8: fp32 x[rows, cols] -> image[rows, ld_img bf16 elements = 4 K bytes, K = ld_img / 2 >= cols, K % 128 == 0 for the GEMM]:
 *   [ hi = bf16(x): K bf16 | hi8: K e4m3 bytes | lo8 = (x - hi): K e4m3 bytes ], each fp8 plane times ONE power of two per row;
 *   scale_hi / scale_lo [round_up(rows, 64)]: the E8M0 bytes (value = byte-decoded x 2^(scale - 127)), row r at byte
 *   (r & ~63) | ((r & 15) << 2) | ((r >> 4) & 3) -- the four fragments of a 64-row block a GEMM lane needs are one dword
 * vb_gemm_x3f8: C[M, N] fp32 = A B^T (+ bias) from two such images; M, N multiples of 256, K of 128 */
int vb_split_f8(const float* x, int64_t ldx, void* image, int64_t ld_img, int rows, int cols, void* scale_hi, void* scale_lo, void* stream);
int vb_gemm_x3f8(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K, const float* bias,
                 const void* sa_hi, const void* sa_lo, const void* sb_hi, const void* sb_lo, void* stream);

#ifdef __cplusplus
}
#endif
#endif
