/*
 * hdr2yuv_hip.h -- C-ABI of the MI355X (gfx950) conversion hot path.
 *
 * This is the drop-in boundary for the in-memory convert path of hdr2yuv:
 *
 *     pic_stats()       /root/reference/common.cpp:66   (hdr.h:439)
 *  -> matrix_convert()  /root/reference/convert.cpp:879 (hdr.h:422)
 *  -> convert()         /root/reference/convert.cpp:513 (hdr.h:420)
 *  -> write_yuv()       /root/reference/tiff.cpp:368    (hdr.h:421), the
 *                       per-sample shift + range clamp only; the file write
 *                       stays with the caller.
 *
 * The reference calls those four functions back to back from main()
 * (hdr2yuv.cpp:797-928) on one planar picture.  h2y_convert_frame() replaces
 * that whole sequence; the per-stage entry points below it exist so that a
 * maintainer can swap one stage at a time (the reference's own precedent for
 * that is the compile-time OPENCV_ENABLED switch at hdr2yuv.cpp:892-896).
 *
 * Plain C: pointers, sizes and one POD descriptor.  No C++ or torch types.
 * The library never calls exit(); every entry returns 0 on success and a
 * non-zero H2Y_E* code otherwise (the reference returns 0/1 and exit()s,
 * convert.cpp:1196, common.cpp:231); h2y_last_error() gives the text.
 */
#ifndef HDR2YUV_HIP_H
#define HDR2YUV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define H2Y_ABI_VERSION 1
/* set in h2y_abi_version() by a timing-experiment build of the library (-DH2Y_EXPERIMENT: its kernels may write wrong bytes) */
#define H2Y_ABI_EXPERIMENT 0x40000000

/* ---- enums: integer values are the reference's CLI contract ------------ */

/* pic_t.pic_buffer_type, hdr.h:297-298; F16 is the EXR case (exr.cpp:233,
 * half widened to float before anything else happens). */
#define H2Y_SAMPLE_U16 1
#define H2Y_SAMPLE_F32 2
#define H2Y_SAMPLE_F16 3

/* chroma_format_idc, hdr.h:14-17 */
#define H2Y_CHROMA_420 1
#define H2Y_CHROMA_444 3

/* transfer_characteristics, hdr.h:108-138 */
#define H2Y_TRANSFER_LINEAR 8
#define H2Y_TRANSFER_PQ 16

/* matrix_coeffs, hdr.h:172-193 */
#define H2Y_MATRIX_GBR 0
#define H2Y_MATRIX_BT709 1
#define H2Y_MATRIX_BT2020NC 9
#define H2Y_MATRIX_YDZDX 11
#define H2Y_MATRIX_YDZDX_Y500 12
#define H2Y_MATRIX_YDZDX_Y100 13

/* error codes */
#define H2Y_OK 0
#define H2Y_EINVAL 1       /* descriptor / argument rejected            */
#define H2Y_EUNSUPPORTED 2 /* valid for the reference, not on this path */
#define H2Y_EHIP 3         /* a HIP runtime call failed                 */
#define H2Y_ENOMEM 4

/*
 * Picture-pair descriptor: the attribute set of pic_t (hdr.h:363-378) for the
 * source and destination pictures plus user_args_t.chroma_resampler_type
 * (hdr.h:275).  Source and destination have the same width/height (the
 * reference's convert() never resizes either; resizing is cv.cpp, compiled
 * out).  The source is always 4:4:4 (matrix_convert() refuses anything else,
 * convert.cpp:886).
 */
typedef struct h2y_desc {
    int32_t width;
    int32_t height;
    int32_t in_sample_type;        /* H2Y_SAMPLE_*                              */
    int32_t src_bit_depth;         /* used for U16 input only (hdr2yuv.cpp:805) */
    int32_t dst_bit_depth;         /* 8..16                                     */
    int32_t src_transfer;          /* --src_transfer_characteristics            */
    int32_t dst_transfer;          /* --dst_transfer_characteristics            */
    int32_t src_matrix;            /* --src_matrix_coeffs                       */
    int32_t dst_matrix;            /* --dst_matrix_coeffs                       */
    int32_t src_primaries;         /* --src_colour_primaries                    */
    int32_t dst_primaries;         /* --dst_colour_primaries                    */
    int32_t dst_full_range;        /* --dst_video_full_range_flag               */
    int32_t dst_chroma_format_idc; /* H2Y_CHROMA_420 or H2Y_CHROMA_444          */
    int32_t chroma_resampler_type; /* 0 = 2x2 box, non-zero = FIR (convert.cpp:807) */
    /* 0: take floor/ceiling from the frame like pic_stats() does
     * (common.cpp:135-136); 1: use the values below instead (the caller
     * already knows them). */
    int32_t stats_override;
    int32_t floor[3];
    int32_t ceiling[3];
} h2y_desc;

typedef struct h2y_ctx h2y_ctx;

/* Bytes of one output frame in .yuv layout: Y plane, then Cb, then Cr, each
 * row-major little-endian uint16 with no padding (tiff.cpp:457-551).
 * Returns 0 for an invalid descriptor. */
size_t h2y_frame_bytes(const h2y_desc *d);
/* Bytes of one input plane (width*height*sizeof(sample)). */
size_t h2y_plane_bytes(const h2y_desc *d);

/* Validate a descriptor for this path. H2Y_OK / H2Y_EINVAL / H2Y_EUNSUPPORTED.
 * `why` (may be NULL) receives a static string. */
int h2y_desc_check(const h2y_desc *d, const char **why);

int h2y_abi_version(void);

/* Create a context on HIP device `device`. Owns streams, device scratch and
 * the PQ coefficient table. Fails (H2Y_EHIP) when no device is present:
 * there is no CPU fallback in this library. */
int h2y_ctx_create(int device, h2y_ctx **out);
void h2y_ctx_destroy(h2y_ctx *ctx);
const char *h2y_last_error(const h2y_ctx *ctx); /* ctx may be NULL: global error */

/* Tuning and test knobs of one context, as strings (the library reads nothing from the environment):
 *   "t1" "0"|"1"|"always" (binary32 first tier; "always": never steered away from it), "groups" "0"|"1".."64" (frame groups; 0: by the frame's size), "cols8" "0"|"1"
 *   (8-column tiles for half input), "balance" "adaptive"|"xcd"|"off"|"<xcd mask>,<ratio>" (slices by measured block / XCD speed),
 *   "tail" "auto"|"on"|"off" (the last frame of a frame group dealt dynamically), "fir" "auto"|"twopass"|"fused" (how chroma_resampler_type != 0 runs), "firsync" "auto"|"0".."1024" (the one-pass
 *   FIR kernel's waves meet at a barrier every so many steps).  None changes a byte of output.
 * The reference has no counterpart (its only knobs are the command-line flags in h2y_desc). */
int h2y_ctx_set_option(h2y_ctx *ctx, const char *name, const char *value);

/* Use the caller's HIP stream (hipStream_t passed as void*) for every launch
 * of this context; NULL restores the context's own stream. */
int h2y_ctx_set_stream(h2y_ctx *ctx, void *hip_stream);

/*
 * Whole path on HOST buffers: replaces pic_stats + matrix_convert + convert +
 * write_yuv's arithmetic (hdr2yuv.cpp:797-928).
 *   in_planes[3]  planar source samples in the reference's plane order
 *                 0=G/Y, 1=B/Z, 2=R/X (convert.cpp:980-982)
 *   out_yuv       h2y_frame_bytes() bytes, .yuv layout
 * Copies in, runs the device path, copies out, synchronises.
 */
int h2y_convert_frame(h2y_ctx *ctx, const h2y_desc *d,
                      const void *const in_planes[3], uint16_t *out_yuv);

/*
 * Whole path on DEVICE buffers, n_frames independent frames in one call
 * (the reference runs one process per frame and appends, tiff.cpp:440).
 *   d_in[f*3 + c]  device pointer to plane c of frame f
 *   d_out[f]       device pointer to h2y_frame_bytes() bytes for frame f
 * The arrays of pointers themselves are host memory.  Asynchronous on the
 * context's stream except for one small status read-back at the end, after
 * which every frame is final.
 */
int h2y_convert_batch(h2y_ctx *ctx, const h2y_desc *d, int n_frames,
                      const void *const *d_in, uint16_t *const *d_out);

/* As h2y_convert_batch but enqueue-only: no host synchronisation and no
 * status read-back.  h2y_batch_finish() must be called before the outputs
 * are consumed: it waits for the OLDEST batch in flight, checks the per-frame
 * floor/ceiling the kernels measured against the ones they assumed, and
 * re-runs any frame where they differ.  Returns the number of frames re-run
 * through *n_redone.
 * Up to TWO batches may be in flight: enqueue k+1 before finishing k and the
 * next launch is already queued behind the running one (no idle gap between
 * them).  A batch's input and output buffers belong to the library until its
 * own h2y_batch_finish() returns; batches finish in the order enqueued.
 * h2y_convert_batch() finishes everything in flight, its own batch last. */
int h2y_convert_batch_enqueue(h2y_ctx *ctx, const h2y_desc *d, int n_frames,
                              const void *const *d_in, uint16_t *const *d_out);
int h2y_batch_finish(h2y_ctx *ctx, int *n_redone);

/* ---- per-stage entries (device buffers), for stage-at-a-time swaps ------ */

/* pic_stats() F32/F16/U16 branch (common.cpp:74-139): per-plane min/max and
 * the derived estimated_floor/ceiling.  fminmax[6] = {min0,max0,min1,...} as
 * float (integers for U16), floor_ceiling[6] likewise as int. */
int h2y_pic_stats(h2y_ctx *ctx, const h2y_desc *d, const void *const d_in[3],
                  float fminmax[6], int32_t floor_ceiling[6]);

/* matrix_convert() (convert.cpp:879-1221), F32/F16/U16 in -> U16 4:4:4 out.
 * d->floor/ceiling are used (stats_override is ignored: caller ran stats). */
int h2y_matrix_convert(h2y_ctx *ctx, const h2y_desc *d, const void *const d_in[3],
                       uint16_t *const d_out444[3]);

/* convert() chroma part (convert.cpp:802-859): one U16 plane 4:4:4 -> 4:2:0,
 * box (resampler 0) or FIR. bit_depth gives the clamp (clip->maxCV). */
int h2y_subsample_420(h2y_ctx *ctx, int width, int height, int bit_depth,
                      int chroma_resampler_type, const uint16_t *d_src,
                      uint16_t *d_dst);

/* matrix_inverse() (hdr.h:423, convert.cpp:1320-1867): the .yuv -> .tiff flow of
 * hdr2yuv.cpp:818-819.  U16 4:4:4 planes Y', Cb/Dz, Cr/Dx in, U16 planes G, B, R out, on the device,
 * 8-byte aligned.  Byte-exact with the compiled reference, including its oddities: Half/Full are those
 * of 12 bits at any bit depth, only matrix_coeffs 1 (BT.709) takes the Y'CbCr equations -- every other
 * value, BT.2020 included, the Y'DzDx ones --, matrix_coeffs 0 is refused (the reference exits), the
 * result is clamped to the INPUT picture's video (or full) range and shifted to out_bit_depth.
 * The function indexes all three planes at full resolution: 4:2:0 input goes through h2y_upsample_444 first
 * (h2y_inverse_420 does both). */
int h2y_matrix_inverse(h2y_ctx *ctx, int width, int height, int in_bit_depth, int in_full_range,
                       int in_matrix_coeffs, int out_bit_depth, const uint16_t *const d_in[3],
                       uint16_t *const d_out[3]);

/* Subsample420to444() (convert.cpp:1869-1986; the same function is yuv2tiff.cpp:575-692, called at :341-342;
 * in convert.cpp only its call site :1576-1577 is under #if 0): one U16 chroma plane of (width/2) x (height/2)
 * samples -> width x height, on the device.  algorithm 0 replicates samples (:1871-1881); any other value runs
 * the FIR pair (:1882-1983): vertical (3 -16 67 227 -32 7)/256 per row parity into a U16 intermediate, then
 * even samples copied and odd samples (21 -52 159 159 -52 21)/256; edges replicate; every stage clamps to
 * [min_cv, max_cv] and truncates.  width and height even (for odd sizes the reference reads rows of its
 * intermediate that it never wrote); d_dst 4-byte aligned. */
int h2y_upsample_444(h2y_ctx *ctx, int width, int height, int algorithm, unsigned min_cv, unsigned max_cv,
                     const uint16_t *d_src, uint16_t *d_dst);

/* The .yuv 4:2:0 -> RGB flow (yuv2tiff.cpp:341-342 followed by its pixel loop = matrix_inverse()):
 * d_in = Y (width x height), Cb/Dz and Cr/Dx (width/2 x height/2); both chroma planes are upsampled with
 * minCV 0 / maxCV 2^in_bit_depth - 1 (yuv2tiff.cpp:92-93,142-154) into scratch the context owns, then
 * h2y_matrix_inverse() runs on the three full planes.  width a multiple of 4, height even. */
int h2y_inverse_420(h2y_ctx *ctx, int width, int height, int in_bit_depth, int in_full_range, int in_matrix_coeffs,
                    int out_bit_depth, int algorithm, const uint16_t *const d_in[3], uint16_t *const d_out[3]);

/* The same flow on HOST buffers, one frame: what main() does between read_planar_integer_file() (hdr2yuv.cpp:582-656) and
 * write_tiff() (tiff.cpp:559-652: the `<< (out depth - in depth)` of its sample loop is part of out_bit_depth here) when a
 * .yuv is read for a .tiff (hdr2yuv.cpp:818-819).  in_planes = Y, Cb/Dz, Cr/Dx as the file holds them: all three width x height
 * for in_chroma_format_idc 3 (h2y_matrix_inverse), chroma at half size each way for 1 (h2y_inverse_420 with `algorithm`);
 * out_planes = G, B, R, width x height each.  Copies in, runs the device path, copies out, synchronises. */
int h2y_inverse_frame(h2y_ctx *ctx, int width, int height, int in_chroma_format_idc, int in_bit_depth, int in_full_range,
                      int in_matrix_coeffs, int out_bit_depth, int algorithm, const uint16_t *const in_planes[3],
                      uint16_t *const out_planes[3]);

/* ---- host <-> device pipeline --------------------------------------------
 * The reference reads a frame, converts it and appends it to the .yuv, one after the other
 * (hdr2yuv.cpp:582-656 reader, :797-928, tiff.cpp:457-551 writer).  Here the upload of frame
 * k+1, the conversion of frame k and the download of frame k-1 overlap, through a ring of
 * `depth` pinned host slots that the caller fills and drains in place:
 *
 *     h2y_stream_open(ctx, &desc, 3);
 *     for each frame:  h2y_stream_input(ctx, planes);   read the file into planes[0..2]
 *                      h2y_stream_submit(ctx);
 *                      if (frames in flight == depth - 1) { h2y_stream_output(ctx, &yuv); write yuv; }
 *     drain:           h2y_stream_output(ctx, &yuv) for the frames still in flight
 *     h2y_stream_close(ctx);
 *
 * planes[c] hold h2y_plane_bytes() each, yuv h2y_frame_bytes(); the pointer h2y_stream_output
 * returns stays valid until the next h2y_stream_output / h2y_stream_close.  Frames come out in
 * submission order.  No other entry of the context may be used while a stream is open. */
int h2y_stream_open(h2y_ctx *ctx, const h2y_desc *d, int depth /* 2..16 slots */);
int h2y_stream_input(h2y_ctx *ctx, void *planes[3]);
int h2y_stream_submit(h2y_ctx *ctx);
int h2y_stream_output(h2y_ctx *ctx, const uint16_t **yuv);
int h2y_stream_close(h2y_ctx *ctx);

/* Timing of the last h2y_convert_batch*() call measured with HIP events on
 * the stream the kernels ran on: total ms over the main kernels and how many
 * launches that covered. */
int h2y_last_kernel_ms(const h2y_ctx *ctx, float *ms, int *launches);

/* Name of the kernel those launches ran ("k_fused", "k_fused_t1", "k_fused_lut16",
 * "k_fused_narrow"): the name to look for in a rocprofv3 kernel trace. */
const char *h2y_last_kernel_name(const h2y_ctx *ctx);
/* The same with its template arguments and launch shape, e.g. "k_fused_t1<F32,420BOX,YCBCR,PQ_IDENT> groups=8 xcd=1";
 * "+k_fir420" after the '>' when the chroma went through the two-pass FIR form.  Tests assert on it: which
 * variant a call took must not depend on what ran before. */
const char *h2y_last_kernel_variant(const h2y_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* HDR2YUV_HIP_H */
