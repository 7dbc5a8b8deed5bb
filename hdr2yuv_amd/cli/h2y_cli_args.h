/*
 * h2y_cli_args.h -- the command line of hdr2yuv and how its values are resolved, restated from the reference's
 * behaviour (no HIP here: `hdr2yuv --dry_run` prints the resolved attributes without touching a device, and
 * tests/test_host_logic.py checks them on the CPU against the rules below).
 *
 * What the reference does, in its order (hdr2yuv.cpp):
 *   :765-766  in_pic and out_pic are zeroed: every source attribute that is not given is 0 -- range 0 (video), primaries 0,
 *             matrix 0 (GBR), transfer 0, chroma_format_idc 0, bit depth 0, width/height 0
 *   :61-62    the destination's chroma format, range, transfer, matrix and primaries start at -1
 *   :73-263   flags overwrite (unknown flags warn and are skipped, :258)
 *   :265-318  what the destination leaves unset is copied from the source AS PARSED (depth, width, height: 0 means unset)
 *   :321-372  input type by extension; float types (.exr/.dpx) force the INPUT chroma format to 4:4:4 -- after the copy above
 *   :386-440  output type by extension; integer depth outside [10,16] only warns
 *   :519-572  sanity checks: width/height in [2,10000], src depth in [8,32], input 4:4:4, dst likewise; any failure:
 *             "TOO MANY ARGUMENT ERRORS", exit
 *   read_file :662-756: .rgb forces the input matrix to GBR (:677-680); .dpx forces GBR, 4:4:4, 32 bits (:729-734); .exr forces
 *             4:4:4, 32 bits, GBR and FULL RANGE on the input picture (exr.cpp:172-183).  The destination keeps what :265-318 gave it.
 *   read_planar_integer_file :592-610: integer input needs a depth in [10,16]
 * Only user_args_t.chroma_resampler_type has no defined default there (never initialised): FIR here, as in
 * make.sh's example.  The reference calls exit(0) on its argument errors; this program returns 1.
 */
#ifndef H2Y_CLI_ARGS_H
#define H2Y_CLI_ARGS_H

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <strings.h>
#include <string>
#include <vector>

#include "../../include/hdr2yuv_hip.h"

enum { CLI_IN_NONE = 0, CLI_IN_YUV, CLI_IN_RGB, CLI_IN_F32, CLI_IN_F16, CLI_IN_SYNTH, CLI_IN_CODEC };
enum { CLI_OUT_NONE = 0, CLI_OUT_YUV, CLI_OUT_RGB, CLI_OUT_CODEC };

struct cli_pic { /* the attribute set of pic_t that the command line fills (hdr.h:363-378) */
    int width, height, bit_depth, half_float_flag, chroma_format_idc, video_full_range_flag;
    int colour_primaries, transfer_characteristics, matrix_coeffs;
};

struct cli_args {
    const char *src = nullptr, *dst = nullptr;
    cli_pic in{}, out{};
    int start_frame = 0, n_frames = 1, verbose = 0;
    int resampler = 1;
    /* additional flags of this build */
    int synthetic = -1, device = 0, gpus = 1, dry_run = 0, help = 0;
    std::vector<int> devices;
    /* resolved */
    int in_type = CLI_IN_NONE, out_type = CLI_OUT_NONE;
    bool inverse = false; /* .yuv -> RGB: matrix_inverse() instead of matrix_convert() (hdr2yuv.cpp:818-819) */
};

static inline const char *cli_ext_of(const char *fn)
{
    const char *dot = fn ? strrchr(fn, '.') : nullptr;
    return dot ? dot + 1 : "";
}

static inline void cli_help()
{
    printf("hdr2yuv (gfx950): --src_filename F --dst_filename F.yuv --src_pic_width W --src_pic_height H --src_bit_depth N\n"
           "  [--dst_bit_depth N] [--src_half_float_flag 0|1] [--src_chroma_format_idc 3] [--dst_chroma_format_idc 1|3]\n"
           "  [--src_start_frame K] [--n_frames N] [--verbose_level L]\n"
           "  [--src_colour_primaries P] [--dst_colour_primaries P] [--src_matrix_coeffs M] [--dst_matrix_coeffs M]\n"
           "  [--src_transfer_characteristics T] [--dst_transfer_characteristics T]\n"
           "  [--src_video_full_range_flag 0|1] [--dst_video_full_range_flag 0|1] [--chroma_resampler_type 0|1]\n"
           "  unset source attributes are 0, unset destination attributes take the source's (as the reference resolves them)\n"
           "  additional: [--synthetic SEEDFRAME] [--device D] [--gpus N [--devices d0,d1,..]] [--dry_run 1]\n"
           "input by extension: .yuv .rgb (16-bit planar), .f32 .f16 (raw planar float / half, plane order G,B,R: what\n"
           "  dpx_read() / read_exr() leave in memory); output: .yuv, or .rgb (planar R,G,B) from .yuv input = the .yuv -> .tiff flow\n");
}

/* hdr2yuv.cpp:73-263 */
static inline void cli_parse(cli_args &a, int argc, char **argv)
{
    memset(&a.in, 0, sizeof a.in);   /* :765 */
    memset(&a.out, 0, sizeof a.out); /* :766 */
    a.out.chroma_format_idc = a.out.video_full_range_flag = -1; /* :61 */
    a.out.transfer_characteristics = a.out.matrix_coeffs = a.out.colour_primaries = -1; /* :62 */
    for (int i = 1; i < argc; i++) {
        auto is = [&](const char *n) { return !strcmp(argv[i], n); };
        auto val = [&]() -> const char * { return (i + 1 < argc) ? argv[++i] : "0"; };
        if (is("--help")) { cli_help(); a.help = 1; } /* :82-85: prints and carries on */
        else if (is("--src_filename")) a.src = val();
        else if (is("--dst_filename")) a.dst = val();
        else if (is("--ref_filename") || is("--sigma_compare") || is("--alpha_channel") || is("--cutout_hd") || is("--cutout_qhd")) (void)val();
        else if (is("--src_pic_width")) a.in.width = atoi(val());
        else if (is("--src_pic_height")) a.in.height = atoi(val());
        else if (is("--dst_pic_width")) a.out.width = atoi(val());
        else if (is("--dst_pic_height")) a.out.height = atoi(val());
        else if (is("--src_bit_depth")) a.in.bit_depth = atoi(val());
        else if (is("--dst_bit_depth")) a.out.bit_depth = atoi(val());
        else if (is("--src_half_float_flag")) a.in.half_float_flag = atoi(val());
        else if (is("--dst_half_float_flag")) a.out.half_float_flag = atoi(val());
        else if (is("--src_chroma_format_idc")) a.in.chroma_format_idc = atoi(val());
        else if (is("--dst_chroma_format_idc")) a.out.chroma_format_idc = atoi(val());
        else if (is("--src_start_frame")) a.start_frame = atoi(val());
        else if (is("--n_frames")) a.n_frames = atoi(val());
        else if (is("--verbose_level")) a.verbose = atoi(val());
        else if (is("--src_colour_primaries")) a.in.colour_primaries = atoi(val());
        else if (is("--dst_colour_primaries")) a.out.colour_primaries = atoi(val());
        else if (is("--src_matrix_coeffs")) a.in.matrix_coeffs = atoi(val());
        else if (is("--dst_matrix_coeffs")) a.out.matrix_coeffs = atoi(val());
        else if (is("--src_transfer_characteristics")) a.in.transfer_characteristics = atoi(val());
        else if (is("--dst_transfer_characteristics")) a.out.transfer_characteristics = atoi(val());
        else if (is("--src_video_full_range_flag")) a.in.video_full_range_flag = atoi(val());
        else if (is("--dst_video_full_range_flag")) a.out.video_full_range_flag = atoi(val());
        else if (is("--chroma_resampler_type")) a.resampler = atoi(val());
        else if (is("--synthetic")) a.synthetic = atoi(val());
        else if (is("--device")) a.device = atoi(val());
        else if (is("--gpus")) a.gpus = atoi(val());
        else if (is("--dry_run")) a.dry_run = atoi(val());
        else if (is("--devices")) {
            a.devices.clear();
            for (const char *p = val(); *p;) {
                a.devices.push_back(atoi(p));
                p = strchr(p, ',');
                if (!p) break;
                p++;
            }
        } else printf("WARNING: argument (%s) unrecongized\n", argv[i]);
    }
}

/* hdr2yuv.cpp:265-572 and the attribute overrides of read_file(); returns the number of argument errors */
static inline int cli_resolve(cli_args &a)
{
    int arg_errors = 0;
    /* :265-318: unset destination attributes <- the source's, as parsed */
    if (a.out.bit_depth == 0) a.out.bit_depth = a.in.bit_depth;
    if (a.out.width == 0) a.out.width = a.in.width;
    if (a.out.height == 0) a.out.height = a.in.height;
    if (a.out.chroma_format_idc == -1) a.out.chroma_format_idc = a.in.chroma_format_idc;
    if (a.out.video_full_range_flag == -1) a.out.video_full_range_flag = a.in.video_full_range_flag;
    if (a.out.colour_primaries == -1) a.out.colour_primaries = a.in.colour_primaries;
    if (a.out.transfer_characteristics == -1) a.out.transfer_characteristics = a.in.transfer_characteristics;
    if (a.out.matrix_coeffs == -1) a.out.matrix_coeffs = a.in.matrix_coeffs;

    /* :321-372 input type */
    const char *ext = cli_ext_of(a.src);
    if (a.synthetic >= 0) a.in_type = CLI_IN_SYNTH;
    else if (!strcasecmp(ext, "yuv")) a.in_type = CLI_IN_YUV;
    else if (!strcasecmp(ext, "rgb")) a.in_type = CLI_IN_RGB;
    else if (!strcasecmp(ext, "f32")) a.in_type = CLI_IN_F32;
    else if (!strcasecmp(ext, "f16")) a.in_type = CLI_IN_F16;
    else if (!strcasecmp(ext, "exr") || !strcasecmp(ext, "dpx") || !strcasecmp(ext, "tiff")) a.in_type = CLI_IN_CODEC;
    if (a.in_type == CLI_IN_NONE) {
        printf("WARNING: input file (%s) type extension (%s) is either not recongized or not supported\n", a.src ? a.src : "(none)", ext);
        arg_errors++;
    } else if (a.in_type == CLI_IN_CODEC) {
        printf("WARNING: input file (%s): .%s decoding stays with the reference's host I/O (exr.cpp / dpx.cpp / tiff.cpp);\n"
               "         this program takes the planes they leave in memory as .f16 / .f32 / .rgb\n", a.src, ext);
        arg_errors++;
    }
    const bool int_in = a.in_type == CLI_IN_YUV || a.in_type == CLI_IN_RGB;
    if (int_in) {
        if (a.in.bit_depth < 10 || a.in.bit_depth > 16)
            printf("WARNING: src bit_depth(%d) outside range [10,16] for integer input file type(%s)\n", a.in.bit_depth, ext);
    } else if (a.in.chroma_format_idc != H2Y_CHROMA_444) {
        printf("file-type is 4:4:4.  Settig chroma_format_idc(%d) to  %d.\n", a.in.chroma_format_idc, H2Y_CHROMA_444);
        a.in.chroma_format_idc = H2Y_CHROMA_444; /* :351-355: after the destination took its copy */
    }

    /* :386-440 output type */
    ext = cli_ext_of(a.dst);
    if (!strcasecmp(ext, "yuv")) a.out_type = CLI_OUT_YUV;
    else if (!strcasecmp(ext, "rgb")) a.out_type = CLI_OUT_RGB;
    else if (!strcasecmp(ext, "exr") || !strcasecmp(ext, "dpx") || !strcasecmp(ext, "tiff")) a.out_type = CLI_OUT_CODEC;
    if (a.out_type == CLI_OUT_NONE) {
        printf("WARNING: output file (%s) type extension (%s) is either not recongized or not supported\n", a.dst ? a.dst : "(none)", ext);
        arg_errors++;
    } else if (a.out_type == CLI_OUT_CODEC) {
        printf("WARNING: output file (%s): the .%s writers stay with the reference's host I/O; this program writes .yuv, and the\n"
               "         .yuv -> .tiff flow's samples as planar .rgb\n", a.dst, ext);
        arg_errors++;
    } else if (a.out.bit_depth < 10 || a.out.bit_depth > 16)
        printf("WARNING: dst bit_depth(%d) outside range [10,16] for integer input file type(%s)\n", a.out.bit_depth, ext);
    /* hdr2yuv.cpp:818: a .yuv read for a .tiff goes through matrix_inverse(); planar .rgb stands in for the .tiff's samples */
    a.inverse = a.in_type == CLI_IN_YUV && a.out_type == CLI_OUT_RGB;
    if (a.out_type == CLI_OUT_RGB && !a.inverse) {
        printf("WARNING: .rgb output is the .yuv -> RGB flow's (matrix_inverse); the reference writes no .rgb either\n");
        arg_errors++;
    }

    if (a.start_frame != 0 && !int_in && a.in_type != CLI_IN_F32 && a.in_type != CLI_IN_F16 && a.in_type != CLI_IN_SYNTH)
        printf("WARNING: start_frame(%d) only makes sense when file type is .yuv, .rgb, or .y4m\n", a.start_frame);

    /* :472-507: what was resolved */
    printf("src_filename: %s\n", a.src ? a.src : "(synthetic)");
    printf("src_pic_width: %d\nsrc_pic_height: %d\nsrc_chroma_format_idc: %d\nsrc_bit_depth: %d\nsrc_half_float_flag: %d\n",
           a.in.width, a.in.height, a.in.chroma_format_idc, a.in.bit_depth, a.in.half_float_flag);
    printf("src_full_range_video_flag: %d\nsrc_colour_primaries: %d\nsrc_transfer_characteristics: %d\nsrc_matrix_coeffs: %d\n",
           a.in.video_full_range_flag, a.in.colour_primaries, a.in.transfer_characteristics, a.in.matrix_coeffs);
    printf("dst_filename: %s\n", a.dst ? a.dst : "(none)");
    printf("dst_pic_width: %d\ndst_pic_height: %d\ndst_chroma_format_idc: %d\ndst_bit_depth: %d\ndst_half_float_flag: %d\n",
           a.out.width, a.out.height, a.out.chroma_format_idc, a.out.bit_depth, a.out.half_float_flag);
    printf("dst_video_full_range_flag: %d\ndst_colour_primaries: %d\ndst_transfer_characteristics: %d\ndst_matrix_coeffs: %d\n",
           a.out.video_full_range_flag, a.out.colour_primaries, a.out.transfer_characteristics, a.out.matrix_coeffs);
    printf("verbose_level: %d\nsrc_start_frame: %d\nn_frames: %d\nchroma_resampler_type: %d\n", a.verbose, a.start_frame, a.n_frames, a.resampler);

    /* :519-572 sanity checks */
    if (a.in.width < 2 || a.in.width > 10000) { printf("WARNING: pic_width(%d) outside range [0,10000]\n", a.in.width); arg_errors++; }
    if (a.in.height < 2 || a.in.height > 10000) { printf("WARNING: pic_height(%d) outside range [0,10000]\n", a.in.height); arg_errors++; }
    if (a.in.bit_depth < 8 || a.in.bit_depth > 32) { printf("WARNING: src bit_depth(%d) outside range [8,32]\n", a.in.bit_depth); arg_errors++; }
    if (a.in.chroma_format_idc != H2Y_CHROMA_444 && !(a.inverse && a.in.chroma_format_idc == H2Y_CHROMA_420)) {
        /* (4:2:0 input is taken on the inverse flow only, the yuv2tiff.cpp:341-342 order: upsample, then matrix_inverse) */
        printf("WARNING: chroma_format_idc(%d) not %d, Only 4:4:4 input supported at this moment..\n", a.in.chroma_format_idc, H2Y_CHROMA_444);
        arg_errors++;
    }
    if (a.out.width < 2 || a.out.width > 10000) { printf("WARNING: pic_width(%d) outside range [0,10000]\n", a.out.width); arg_errors++; }
    if (a.out.height < 2 || a.out.height > 10000) { printf("WARNING: pic_height(%d) outside range [0,10000]\n", a.out.height); arg_errors++; }
    if (a.out.bit_depth < 8 || a.out.bit_depth > 32) { printf("WARNING: dst bit_depth(%d) outside range [32]\n", a.out.bit_depth); arg_errors++; }
    if (arg_errors) return arg_errors;

    /* read_file(): what the readers force on the INPUT picture (the destination's copies were taken above) */
    if (a.in_type == CLI_IN_RGB && a.in.matrix_coeffs != H2Y_MATRIX_GBR) {
        printf("WARNING: RGB src matrix_coefs(%d) being overriden to MATRIX_GBR (%d)\n", a.in.matrix_coeffs, H2Y_MATRIX_GBR); /* :677-680 */
        a.in.matrix_coeffs = H2Y_MATRIX_GBR;
    }
    if (!int_in) { /* .dpx :729-734, .exr exr.cpp:172-183 */
        if (a.in.matrix_coeffs != H2Y_MATRIX_GBR) printf("overriding matrix_coeffs(%d) to MATRIX_GBR(%d)\n", a.in.matrix_coeffs, H2Y_MATRIX_GBR);
        a.in.matrix_coeffs = H2Y_MATRIX_GBR;
        a.in.chroma_format_idc = H2Y_CHROMA_444;
        a.in.bit_depth = 32;
        if (a.in_type != CLI_IN_F32) a.in.video_full_range_flag = 1; /* read_exr() only; dpx keeps the flag (:712-713 prints, does not set) */
    }
    if (int_in && (a.in.bit_depth < 10 || a.in.bit_depth > 16)) { /* :592-610 */
        printf("read_planar_integer_file(), WARNING: bit_depth(%d) outside supported range [10,16]\n", a.in.bit_depth);
        return 1;
    }
    return 0;
}

/* the picture pair as the C-ABI takes it */
static inline void cli_make_desc(const cli_args &a, h2y_desc *d)
{
    memset(d, 0, sizeof *d);
    d->width = a.in.width;
    d->height = a.in.height;
    switch (a.in_type) {
    case CLI_IN_F32: d->in_sample_type = H2Y_SAMPLE_F32; break;
    case CLI_IN_F16: d->in_sample_type = H2Y_SAMPLE_F16; break;
    case CLI_IN_SYNTH: d->in_sample_type = a.in.half_float_flag ? H2Y_SAMPLE_F16 : H2Y_SAMPLE_F32; break;
    default: d->in_sample_type = H2Y_SAMPLE_U16; break;
    }
    d->src_bit_depth = a.in.bit_depth;
    d->dst_bit_depth = a.out.bit_depth;
    d->src_transfer = a.in.transfer_characteristics;
    d->dst_transfer = a.out.transfer_characteristics;
    d->src_matrix = a.in.matrix_coeffs;
    d->dst_matrix = a.out.matrix_coeffs;
    d->src_primaries = a.in.colour_primaries;
    d->dst_primaries = a.out.colour_primaries;
    d->dst_full_range = a.out.video_full_range_flag;
    d->dst_chroma_format_idc = a.out.chroma_format_idc;
    d->chroma_resampler_type = a.resampler;
}

#endif /* H2Y_CLI_ARGS_H */
