/*
 * hdr2yuv (MI355X build) -- host program with the reference's command-line
 * surface (hdr2yuv.cpp:73-263) for the in-memory convert path.  It reads raw
 * planar input, hands the planes to the C-ABI (include/hdr2yuv_hip.h) and
 * appends the .yuv frame exactly as write_yuv() does (tiff.cpp:440: the file is
 * opened in append mode; planes Y, Cb, Cr, little-endian 16-bit).
 *
 * File decoding stays where the reference has it (exr.cpp / tiff.cpp / dpx.cpp
 * need OpenEXR and libtiff): this binary takes the formats that need no codec:
 *   .yuv / .rgb  16-bit planar integer (hdr2yuv.cpp:582-656; .rgb is R,G,B in
 *                the file, planes 2,0,1 in memory)
 *   .f32 / .f16  raw planar float / half in G,B,R plane order -- what read_exr()
 *                (exr.cpp:233-235) or dpx_read() leave in memory
 *   --synthetic N  the seeded test frame of SURVEY 8c (no input file)
 * Unknown flags warn and are skipped, as in the reference (hdr2yuv.cpp:258).
 * Numeric codes only for the enum flags (the reference's name lookup indexes
 * its table with the wrong variable).
 */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hdr2yuv_hip.h"

struct args {
    const char *src = nullptr, *dst = nullptr;
    int src_w = 0, src_h = 0, dst_w = 0, dst_h = 0;
    int src_depth = 0, dst_depth = 0;
    int src_half = 0;
    int src_chroma = -1, dst_chroma = -1;
    int start_frame = 0, n_frames = 1, verbose = 0;
    int src_prim = -1, dst_prim = -1, src_mat = -1, dst_mat = -1, src_tr = -1, dst_tr = -1;
    int src_full = -1, dst_full = -1;
    int resampler = 1; /* the reference leaves this uninitialised; FIR as in make.sh's example */
    int synthetic = -1;
    int device = 0;
};

static void help()
{
    printf("hdr2yuv (gfx950): --src_filename F --dst_filename F.yuv --src_pic_width W --src_pic_height H\n"
           "  [--src_bit_depth N] [--dst_bit_depth N] [--src_half_float_flag 0|1] [--src_chroma_format_idc 3]\n"
           "  [--dst_chroma_format_idc 1|3] [--src_start_frame K] [--n_frames N] [--verbose_level L]\n"
           "  [--src_colour_primaries P] [--dst_colour_primaries P] [--src_matrix_coeffs M] [--dst_matrix_coeffs M]\n"
           "  [--src_transfer_characteristics T] [--dst_transfer_characteristics T]\n"
           "  [--src_video_full_range_flag 0|1] [--dst_video_full_range_flag 0|1] [--chroma_resampler_type 0|1]\n"
           "  extra: [--synthetic SEEDFRAME] [--device D]\n"
           "input by extension: .yuv .rgb (16-bit planar), .f32 .f16 (raw planar float, plane order G,B,R)\n");
}

static const char *ext_of(const char *fn)
{
    const char *dot = fn ? strrchr(fn, '.') : nullptr;
    return dot ? dot + 1 : "";
}

static uint16_t f32_to_f16(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    int32_t e = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
    uint32_t m = x & 0x7FFFFFu;
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        int shift = 14 - e;
        uint32_t r = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)e << 10) | (m >> 13), rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) r++;
    return (uint16_t)(sign | r);
}

int main(int argc, char **argv)
{
    args a;
    for (int i = 1; i < argc; i++) {
        auto is = [&](const char *n) { return !strcmp(argv[i], n); };
        auto val = [&]() -> const char * { return (i + 1 < argc) ? argv[++i] : "0"; };
        if (is("--help")) { help(); return 0; }
        else if (is("--src_filename")) a.src = val();
        else if (is("--dst_filename")) a.dst = val();
        else if (is("--ref_filename") || is("--sigma_compare") || is("--alpha_channel") || is("--cutout_hd") || is("--cutout_qhd") ||
                 is("--dst_half_float_flag")) (void)val(); /* parsed by the reference, not on this path */
        else if (is("--src_pic_width")) a.src_w = atoi(val());
        else if (is("--src_pic_height")) a.src_h = atoi(val());
        else if (is("--dst_pic_width")) a.dst_w = atoi(val());
        else if (is("--dst_pic_height")) a.dst_h = atoi(val());
        else if (is("--src_bit_depth")) a.src_depth = atoi(val());
        else if (is("--dst_bit_depth")) a.dst_depth = atoi(val());
        else if (is("--src_half_float_flag")) a.src_half = atoi(val());
        else if (is("--src_chroma_format_idc")) a.src_chroma = atoi(val());
        else if (is("--dst_chroma_format_idc")) a.dst_chroma = atoi(val());
        else if (is("--src_start_frame")) a.start_frame = atoi(val());
        else if (is("--n_frames")) a.n_frames = atoi(val());
        else if (is("--verbose_level")) a.verbose = atoi(val());
        else if (is("--src_colour_primaries")) a.src_prim = atoi(val());
        else if (is("--dst_colour_primaries")) a.dst_prim = atoi(val());
        else if (is("--src_matrix_coeffs")) a.src_mat = atoi(val());
        else if (is("--dst_matrix_coeffs")) a.dst_mat = atoi(val());
        else if (is("--src_transfer_characteristics")) a.src_tr = atoi(val());
        else if (is("--dst_transfer_characteristics")) a.dst_tr = atoi(val());
        else if (is("--src_video_full_range_flag")) a.src_full = atoi(val());
        else if (is("--dst_video_full_range_flag")) a.dst_full = atoi(val());
        else if (is("--chroma_resampler_type")) a.resampler = atoi(val());
        else if (is("--synthetic")) a.synthetic = atoi(val());
        else if (is("--device")) a.device = atoi(val());
        else printf("WARNING: unrecognized argument: %s\n", argv[i]);
    }
    /* destination defaults copy the source (hdr2yuv.cpp:265-318) */
    if (!a.dst_w) a.dst_w = a.src_w;
    if (!a.dst_h) a.dst_h = a.src_h;
    if (a.src_chroma < 0) a.src_chroma = H2Y_CHROMA_444;
    if (a.dst_chroma < 0) a.dst_chroma = a.src_chroma;
    if (a.src_prim < 0) a.src_prim = 9;
    if (a.dst_prim < 0) a.dst_prim = a.src_prim;
    if (a.src_mat < 0) a.src_mat = H2Y_MATRIX_GBR;
    if (a.dst_mat < 0) a.dst_mat = a.src_mat;
    if (a.src_tr < 0) a.src_tr = H2Y_TRANSFER_LINEAR;
    if (a.dst_tr < 0) a.dst_tr = a.src_tr;
    if (a.src_full < 0) a.src_full = 1;
    if (a.dst_full < 0) a.dst_full = a.src_full;

    if (!a.dst || (!a.src && a.synthetic < 0)) { help(); return 1; }
    if (strcasecmp(ext_of(a.dst), "yuv")) {
        printf("ERROR: only .yuv output is on this path (TIFF/EXR/DPX writers stay with the reference host code)\n");
        return 1;
    }
    if (a.dst_w != a.src_w || a.dst_h != a.src_h) {
        printf("ERROR: resizing is not part of convert() (cv.cpp is compiled out in the reference)\n");
        return 1;
    }
    if (a.src_chroma != H2Y_CHROMA_444) {
        printf("ERROR, matrix_convert(): input picture must be 4:4:4\n"); /* convert.cpp:886 */
        return 1;
    }

    h2y_desc d;
    memset(&d, 0, sizeof d);
    d.width = a.src_w;
    d.height = a.src_h;
    const char *ext = a.src ? ext_of(a.src) : "f32";
    bool rgb_order = false;
    if (a.synthetic >= 0) d.in_sample_type = a.src_half ? H2Y_SAMPLE_F16 : H2Y_SAMPLE_F32;
    else if (!strcasecmp(ext, "f32")) d.in_sample_type = H2Y_SAMPLE_F32;
    else if (!strcasecmp(ext, "f16")) d.in_sample_type = H2Y_SAMPLE_F16;
    else if (!strcasecmp(ext, "yuv")) d.in_sample_type = H2Y_SAMPLE_U16;
    else if (!strcasecmp(ext, "rgb")) { d.in_sample_type = H2Y_SAMPLE_U16; rgb_order = true; }
    else {
        printf("WARNING: input file (%s) type extension (%s) is either not recognized or not supported\n"
               "         (.exr/.tiff/.dpx decoding stays with the reference's host I/O; this path takes raw planes)\n", a.src, ext);
        return 1;
    }
    if (d.in_sample_type == H2Y_SAMPLE_U16 && !a.src_depth) a.src_depth = 16;
    if (!a.dst_depth) a.dst_depth = d.in_sample_type == H2Y_SAMPLE_U16 ? a.src_depth : 10;
    d.src_bit_depth = d.in_sample_type == H2Y_SAMPLE_U16 ? a.src_depth : 32;
    d.dst_bit_depth = a.dst_depth;
    d.src_transfer = a.src_tr;
    d.dst_transfer = a.dst_tr;
    d.src_matrix = a.src_mat;
    d.dst_matrix = a.dst_mat;
    d.src_primaries = a.src_prim;
    d.dst_primaries = a.dst_prim;
    d.dst_full_range = a.dst_full;
    d.dst_chroma_format_idc = a.dst_chroma;
    d.chroma_resampler_type = a.resampler;

    const char *why = nullptr;
    int rc = h2y_desc_check(&d, &why);
    if (rc) { printf("ERROR: %s\n", why); return 1; }

    h2y_ctx *ctx = nullptr;
    rc = h2y_ctx_create(a.device, &ctx);
    if (rc) { printf("ERROR: %s\n", h2y_last_error(nullptr)); return 1; }

    const size_t n = (size_t)d.width * d.height, pb = h2y_plane_bytes(&d), ob = h2y_frame_bytes(&d);
    FILE *fin = nullptr;
    if (a.synthetic < 0) {
        fin = fopen(a.src, "rb");
        if (!fin) { printf("ERROR: unable to open file %s\n", a.src); return 1; }
        if (fseek(fin, (long)((size_t)3 * pb * a.start_frame), SEEK_SET)) { printf("ERROR: seek failed\n"); return 1; }
    }
    FILE *fout = fopen(a.dst, "ab"); /* tiff.cpp:440: ios::ate | ios::app */
    if (!fout) { printf("ERROR: unable to open %s\n", a.dst); return 1; }

    /* The reader fills the pinned slot of the pipeline directly, the writer appends what comes out of
     * it two frames later: upload, conversion and download of neighbouring frames overlap. */
    const int depth = 3;
    rc = h2y_stream_open(ctx, &d, depth);
    if (rc) { printf("ERROR: %s\n", h2y_last_error(ctx)); return 1; }
    int in_flight = 0, written = 0;
    auto drain_one = [&]() -> int {
        const uint16_t *yuv = nullptr;
        if (h2y_stream_output(ctx, &yuv)) { printf("ERROR: %s\n", h2y_last_error(ctx)); return 1; }
        if (fwrite(yuv, 1, ob, fout) != ob) { printf("ERROR: short write to %s\n", a.dst); return 1; }
        if (a.verbose > 0) printf("frame %d: %zu bytes appended to %s\n", written, ob, a.dst);
        written++;
        in_flight--;
        return 0;
    };
    for (int f = 0; f < (a.n_frames > 0 ? a.n_frames : 1); f++) {
        void *planes[3];
        if (h2y_stream_input(ctx, planes)) { printf("ERROR: %s\n", h2y_last_error(ctx)); return 1; }
        if (fin) {
            /* file plane order -> memory planes (0=G/Y, 1=B/Cb, 2=R/Cr) */
            const int order_rgb[3] = {2, 0, 1}, order_nat[3] = {0, 1, 2};
            const int *ord = rgb_order ? order_rgb : order_nat;
            size_t got = 0;
            for (int k = 0; k < 3; k++) got += fread(planes[ord[k]], 1, pb, fin);
            if (got != 3 * pb) {
                if (f == 0) { printf("ERROR: only %zu bytes read from %s, expecting %zu\n", got, a.src, 3 * pb); return 1; }
                break;
            }
        } else {
            uint32_t s = 12345u + (uint32_t)(a.synthetic + f);
            for (int c = 0; c < 3; c++) {
                for (size_t i = 0; i < n; i++) {
                    s = s * 1664525u + 1013904223u;
                    float v = (float)(s >> 8) * (1.0f / 16777216.0f);
                    if (i == 0) v = 0.0f;
                    if (i == 1) v = 1.0f;
                    if (d.in_sample_type == H2Y_SAMPLE_F16) ((uint16_t *)planes[c])[i] = f32_to_f16(v);
                    else ((float *)planes[c])[i] = v;
                }
            }
        }
        if (h2y_stream_submit(ctx)) { printf("ERROR: %s\n", h2y_last_error(ctx)); return 1; }
        in_flight++;
        if (in_flight == depth - 1 && drain_one()) return 1;
    }
    while (in_flight > 0)
        if (drain_one()) return 1;
    h2y_stream_close(ctx);
    if (fin) fclose(fin);
    fclose(fout);
    h2y_ctx_destroy(ctx);
    return 0;
}
