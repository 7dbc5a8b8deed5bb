/*
 * h2y_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, one thread) of the hdr2yuv in-memory
 * convert path, written from the behaviour of the reference functions cited
 * at each entry.  It is the checker for the HIP path: only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it.  The product
 * library (hdr2yuv_amd/csrc) never includes, links or calls anything here.
 *
 * Pinned: byte-for-byte against the reference's own convert.cpp / common.cpp
 * compiled into oracle/_ref (tests/test_oracle_vs_ref.py, container only) and
 * against the committed fixtures in tests/golden/ which that build produced.
 * The write_yuv() arithmetic (tiff.cpp:457-550) cannot be compiled here
 * (tiff.cpp needs libtiff headers that this image lacks), so that last stage
 * is pinned only by the full-frame md5 known answers recorded in
 * SURVEY.md section 8c (tests/golden/known_md5.json).
 *
 * Build: gcc -O2 -ffp-contract=off (no -march, no -ffast-math): the reference
 * is built without FMA contraction and contraction changes output bytes.
 */
#ifndef H2Y_ORACLE_H
#define H2Y_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/hdr2yuv_hip.h" /* h2y_desc and the enum values only */

#ifdef __cplusplus
extern "C" {
#endif

/* clip_limits_t, hdr.h:345-356, filled as set_pic_clip() does, common.cpp:300-327 */
typedef struct h2y_oracle_clip {
    uint64_t minCV, maxCV;
    uint16_t minVR, maxVR, minVRC, maxVRC, Half;
} h2y_oracle_clip;

void h2y_oracle_set_clip(int bit_depth, int full_range, h2y_oracle_clip *clip);

/* PQ10000_r, convert.cpp:56-63 */
float h2y_oracle_pq10000_r(float L);

/* source transfer -> linear -> destination transfer for one sample, convert.cpp:1024-1109 */
float h2y_oracle_transfer_chain(int src_transfer, int dst_transfer, float x);

/* pic_stats F32 branch, common.cpp:116-136. mm = {min0,max0,min1,max1,min2,max2} */
void h2y_oracle_stats_f32(const float *const planes[3], size_t n, float mm[6],
                          int32_t floor_[3], int32_t ceil_[3]);
/* pic_stats U16 branch, common.cpp:74-106 */
void h2y_oracle_stats_u16(const uint16_t *const planes[3], size_t n, int bit_depth,
                          uint16_t mm[6], int32_t floor_[3], int32_t ceil_[3]);

/* matrix_convert, convert.cpp:879-1221 (U16-out branch). in planes are float
 * (F16 input is widened by the caller, exr.cpp:233) or uint16_t. */
int h2y_oracle_matrix_convert(const h2y_desc *d, const void *const in_planes[3],
                              const int32_t floor_[3], const int32_t ceil_[3],
                              int tmp_bit_depth, uint16_t *const out444[3]);

/* Subsample444to420_box, convert.cpp:91-172 */
void h2y_oracle_sub420_box(uint16_t *dst, const uint16_t *src, int width, int height);
/* Subsample444to420_FIR, convert.cpp:261-383 */
void h2y_oracle_sub420_fir(uint16_t *dst, const uint16_t *src, int width, int height,
                           uint64_t minCV, uint64_t maxCV);

/* Subsample420to444, convert.cpp:1869-1986: (width/2 x height/2) -> (width x height), row-major planes;
 * algorithm 0 = replication, else the 6-tap / 6-tap FIR pair */
void h2y_oracle_up444(uint16_t *dst, const uint16_t *src, int width, int height, int algorithm,
                      unsigned minCV, unsigned maxCV);

/* write_yuv per-sample arithmetic, tiff.cpp:394,457-550, in place on a plane */
void h2y_oracle_yuv_clamp(uint16_t *plane, size_t n, int down_shift, int full_range,
                          unsigned lo, unsigned hi, uint64_t maxCV);

/* Whole path as main() strings it together, hdr2yuv.cpp:797-928.
 * in_planes: float* (F32), uint16_t* holding IEEE half bits (F16) or uint16_t*
 * samples (U16).  out_yuv: h2y_oracle_frame_bytes(d) bytes. Returns 0 / H2Y_E*. */
int h2y_oracle_convert_frame(const h2y_desc *d, const void *const in_planes[3],
                             uint16_t *out_yuv);

size_t h2y_oracle_frame_bytes(const h2y_desc *d);

/* matrix_inverse, convert.cpp:1320-1867, U16 4:4:4 in -> U16 out (the .yuv -> .tiff flow, hdr2yuv.cpp:818-819).
 * Returns 1 where the reference calls exit() (matrix_coeffs 0). */
int h2y_oracle_matrix_inverse(int width, int height, int in_bit_depth, int in_full_range, int in_matrix, int out_bit_depth,
                              const uint16_t *const in_planes[3], uint16_t *const out_planes[3]);

/* The synthetic frame of SURVEY.md 8c/8d: LCG seeded per frame, planted 0.0
 * and 1.0.  sample_type F32 -> float planes; F16 -> half bit patterns. */
void h2y_oracle_synth_plane_f32(float *plane, size_t n, uint32_t *lcg_state);
uint16_t h2y_oracle_f32_to_f16(float f);
float h2y_oracle_f16_to_f32(uint16_t h);

#ifdef __cplusplus
}
#endif
#endif
