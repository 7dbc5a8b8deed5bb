/*
 * ref_shim.cpp -- TEST INFRASTRUCTURE ONLY, container only.
 *
 * Thin extern "C" driver around the REFERENCE's own object code
 * (/root/reference/convert.cpp and common.cpp compiled where they lie by
 * oracle/Makefile into oracle/_ref/).  It walks the same call sequence as the
 * reference's main() (hdr2yuv.cpp:797-928) on caller-supplied planes so that
 * tests can compare the restatement in h2y_oracle.c with the real thing and
 * generate the fixtures in tests/golden/.
 *
 * Not built from reference source: this file is ours; it only #includes the
 * reference's hdr.h for the pic_t/hdr_t layouts and calls its functions.
 *
 * write_yuv() lives in tiff.cpp, which cannot be compiled in this image
 * (absolute #include of libtiff's header, absent here) -- so the final
 * shift+clamp below is OUR restatement (h2y_oracle_yuv_clamp), not reference
 * code; everything before it is the reference's object code.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <fcntl.h>

#include "hdr.h" /* from -I/root/reference */
#include "h2y_oracle.h"

namespace {
/* The reference printf()s from inside every hot function; keep test logs
 * readable by parking stdout on /dev/null while it runs. */
struct MuteStdout {
    int saved;
    MuteStdout() {
        fflush(stdout);
        saved = dup(1);
        int nul = open("/dev/null", O_WRONLY);
        dup2(nul, 1);
        close(nul);
    }
    ~MuteStdout() {
        fflush(stdout);
        dup2(saved, 1);
        close(saved);
    }
};
} // namespace

extern "C" {

float h2y_ref_pq10000_r(float L);
int h2y_ref_convert_frame(const h2y_desc *d, const void *const in_planes[3], uint16_t *out_yuv,
                          uint16_t *tmp444_out /* optional: 3 planes of matrix_convert output */);
int h2y_ref_sub420(const uint16_t *src, uint16_t *dst, int width, int height, int bit_depth, int fir);
int h2y_ref_matrix_inverse(int width, int height, int in_bit_depth, int in_full_range, int in_matrix, int out_bit_depth,
                           const uint16_t *const in_planes[3], uint16_t *const out_planes[3]);
int h2y_ref_up444(const uint16_t *src, uint16_t *dst, int width, int height, int algorithm, unsigned minCV, unsigned maxCV);

} // extern "C"

float PQ10000_r(float L); /* convert.cpp:56 (not declared in hdr.h) */
void Subsample444to420_FIR(unsigned short *, unsigned short *, short, short, unsigned long, unsigned long);
void Subsample444to420_box(unsigned short *, unsigned short *, short, short, unsigned long, unsigned long);

void Subsample420to444(unsigned short **src, unsigned short **dst, short width, short height, short algorithmn, unsigned short minCV,
                       unsigned short maxCV); /* convert.cpp:1869: defined and compiled, only its call site (:1576) is under #if 0 */

float h2y_ref_pq10000_r(float L) { return PQ10000_r(L); }

/* The reference's Subsample420to444() on row-major planes: it takes arrays of COLUMN pointers
 * (src[col][row], dst[col][row]), which are built here and copied back. */
int h2y_ref_up444(const uint16_t *src, uint16_t *dst, int width, int height, int algorithm, unsigned minCV, unsigned maxCV)
{
    MuteStdout mute;
    const int w2 = width >> 1, h2 = height >> 1;
    unsigned short **scol = (unsigned short **)malloc(sizeof(unsigned short *) * (w2 > 0 ? w2 : 1));
    unsigned short **dcol = (unsigned short **)malloc(sizeof(unsigned short *) * (width > 0 ? width : 1));
    for (int i = 0; i < w2; i++) {
        scol[i] = (unsigned short *)malloc(sizeof(unsigned short) * (h2 > 0 ? h2 : 1));
        for (int j = 0; j < h2; j++) scol[i][j] = src[(size_t)j * w2 + i];
    }
    for (int i = 0; i < width; i++) {
        dcol[i] = (unsigned short *)malloc(sizeof(unsigned short) * height);
        for (int j = 0; j < height; j++) dcol[i][j] = dst[(size_t)j * width + i]; /* what the function leaves untouched stays the caller's */
    }
    Subsample420to444(scol, dcol, (short)width, (short)height, (short)algorithm, (unsigned short)minCV, (unsigned short)maxCV);
    for (int i = 0; i < width; i++) {
        for (int j = 0; j < height; j++) dst[(size_t)j * width + i] = dcol[i][j];
        free(dcol[i]);
    }
    for (int i = 0; i < w2; i++) free(scol[i]);
    free(scol);
    free(dcol);
    return 0;
}

int h2y_ref_sub420(const uint16_t *src, uint16_t *dst, int width, int height, int bit_depth, int fir)
{
    MuteStdout mute;
    unsigned long maxCV = (1ul << bit_depth) - 1;
    if (fir) Subsample444to420_FIR(dst, const_cast<uint16_t *>(src), (short)width, (short)height, 0, maxCV);
    else Subsample444to420_box(dst, const_cast<uint16_t *>(src), (short)width, (short)height, 0, maxCV);
    return 0;
}

int h2y_ref_convert_frame(const h2y_desc *d, const void *const in_planes[3], uint16_t *out_yuv,
                          uint16_t *tmp444_out)
{
    MuteStdout mute;
    static hdr_t h; /* large; zeroed like main() zeroes in_pic/out_pic */
    memset(&h, 0, sizeof(h));
    h.user_args.chroma_resampler_type = d->chroma_resampler_type; /* uninitialised in the reference unless passed (Q14) */
    h.user_args.verbose_level = 0;

    pic_t *in_pic = &h.in_pic, *out_pic = &h.out_pic;
    pic_t tmp_storage;
    memset(&tmp_storage, 0, sizeof(tmp_storage));
    pic_t *tmp_pic = &tmp_storage;

    const int W = d->width, H = d->height;
    const size_t n = (size_t)W * H;
    const bool u16in = d->in_sample_type == H2Y_SAMPLE_U16;

    /* read_file() equivalent: an in-memory 4:4:4 picture with the source attributes */
    init_pic(in_pic, W, H, CHROMA_444, u16in ? d->src_bit_depth : 32, 1, d->src_primaries, d->src_transfer,
             d->src_matrix, 0, u16in ? PIC_TYPE_U16 : PIC_TYPE_F32, 0, 0, "in_pic");
    for (int c = 0; c < 3; c++) {
        if (u16in) memcpy(in_pic->buf[c], in_planes[c], n * sizeof(uint16_t));
        else if (d->in_sample_type == H2Y_SAMPLE_F16) {
            const uint16_t *hp = (const uint16_t *)in_planes[c];
            for (size_t i = 0; i < n; i++) in_pic->fbuf[c][i] = h2y_oracle_f16_to_f32(hp[i]); /* exr.cpp:233 widening */
        } else memcpy(in_pic->fbuf[c], in_planes[c], n * sizeof(float));
    }

    /* parse_options() results for the destination */
    out_pic->width = W;
    out_pic->height = H;
    out_pic->chroma_format_idc = d->dst_chroma_format_idc;
    out_pic->bit_depth = d->dst_bit_depth;
    out_pic->video_full_range_flag = d->dst_full_range;
    out_pic->colour_primaries = d->dst_primaries;
    out_pic->transfer_characteristics = d->dst_transfer;
    out_pic->matrix_coeffs = d->dst_matrix;
    out_pic->chroma_sample_loc_type = 0;
    out_pic->pic_buffer_type = PIC_TYPE_U16; /* .yuv output, hdr2yuv.cpp:419-440 */

    /* hdr2yuv.cpp:797 */
    pic_stats(in_pic, &in_pic->stats, 1);
    if (d->stats_override)
        for (int c = 0; c < 3; c++) {
            in_pic->stats.estimated_floor[c] = d->floor[c];
            in_pic->stats.estimated_ceiling[c] = d->ceiling[c];
        }

    /* hdr2yuv.cpp:803-812 */
    int tmp_bit_depth = (out_pic->pic_buffer_type == in_pic->pic_buffer_type) ? in_pic->bit_depth : out_pic->bit_depth;
    init_pic(tmp_pic, W, H, in_pic->chroma_format_idc, tmp_bit_depth, out_pic->video_full_range_flag,
             out_pic->colour_primaries, out_pic->transfer_characteristics, out_pic->matrix_coeffs, 0,
             out_pic->pic_buffer_type, 0, 0, "tmp_pic");

    /* hdr2yuv.cpp:821 */
    int rc = matrix_convert(tmp_pic, &h, in_pic);
    if (tmp444_out)
        for (int c = 0; c < 3; c++) memcpy(tmp444_out + c * n, tmp_pic->buf[c], n * sizeof(uint16_t));

    /* hdr2yuv.cpp:865 */
    init_pic(out_pic, W, H, out_pic->chroma_format_idc, out_pic->bit_depth, out_pic->video_full_range_flag,
             out_pic->colour_primaries, out_pic->transfer_characteristics, out_pic->matrix_coeffs, 0, PIC_TYPE_U16, 0,
             0, "out_pic");

    /* hdr2yuv.cpp:868-922 */
    if (out_pic->chroma_format_idc != in_pic->chroma_format_idc) rc |= convert(out_pic, &h, tmp_pic);
    else
        for (int c = 0; c < 3; c++) memcpy(out_pic->buf[c], tmp_pic->buf[c], n * sizeof(uint16_t));

    /* write_yuv(), tiff.cpp:457-550 -- our restatement, see header comment */
    const int down_shift = tmp_pic->bit_depth - out_pic->bit_depth;
    const clip_limits_t *clip = &out_pic->clip;
    size_t off = 0;
    for (int c = 0; c < 3; c++) {
        size_t pn = (size_t)out_pic->plane[c].width * out_pic->plane[c].height;
        memcpy(out_yuv + off, out_pic->buf[c], pn * sizeof(uint16_t));
        h2y_oracle_yuv_clamp(out_yuv + off, pn, down_shift, out_pic->video_full_range_flag,
                             c == 0 ? clip->minVR : clip->minVRC, c == 0 ? clip->maxVR : clip->maxVRC, clip->maxCV);
        off += pn;
    }

    deinit_pic(in_pic);
    deinit_pic(tmp_pic);
    deinit_pic(out_pic);
    return down_shift < 0 ? 1 : rc;
}

/* matrix_inverse() as main() reaches it for a .yuv -> .tiff run (hdr2yuv.cpp:803-819): in_pic is the
 * 4:4:4 U16 picture read from the .yuv, tmp_pic has the input's buffer type and bit depth
 * (hdr2yuv.cpp:805-808: same buffer types => tmp_bit_depth = in_pic->bit_depth) unless the caller asks
 * for another output depth (the shift at convert.cpp:1800-1814 is what that exercises). */
extern "C" int h2y_ref_matrix_inverse(int width, int height, int in_bit_depth, int in_full_range, int in_matrix, int out_bit_depth,
                                      const uint16_t *const in_planes[3], uint16_t *const out_planes[3])
{
    MuteStdout mute;
    static hdr_t h;
    memset(&h, 0, sizeof(h));
    pic_t *in_pic = &h.in_pic;
    pic_t tmp_storage;
    memset(&tmp_storage, 0, sizeof(tmp_storage));
    pic_t *tmp_pic = &tmp_storage;
    const size_t n = (size_t)width * height;
    init_pic(in_pic, width, height, CHROMA_444, in_bit_depth, in_full_range, 1, 1, in_matrix, 0, PIC_TYPE_U16, 0, 0, "in_pic");
    for (int c = 0; c < 3; c++) memcpy(in_pic->buf[c], in_planes[c], n * sizeof(uint16_t));
    init_pic(tmp_pic, width, height, CHROMA_444, out_bit_depth, 0, 1, 1, 0, 0, PIC_TYPE_U16, 0, 0, "tmp_pic");
    int rc = matrix_inverse(tmp_pic, &h, in_pic);
    for (int c = 0; c < 3; c++) memcpy(out_planes[c], tmp_pic->buf[c], n * sizeof(uint16_t));
    for (int c = 0; c < 3; c++) {
        free(in_pic->buf[c]);
        free(tmp_pic->buf[c]);
    }
    return rc;
}
