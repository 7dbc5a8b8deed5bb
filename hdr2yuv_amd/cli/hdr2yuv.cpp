/*
 * hdr2yuv (MI355X build) -- host program with the reference's command-line surface (hdr2yuv.cpp:73-263) and its
 * resolution of unset attributes (h2y_cli_args.h) for the in-memory convert path.  It reads raw planar input, hands the
 * planes to the C-ABI (include/hdr2yuv_hip.h) and writes the .yuv frames where write_yuv() would append them
 * (tiff.cpp:440: the file is opened in append mode; planes Y, Cb, Cr, little-endian 16-bit).
 *
 * File decoding stays where the reference has it (exr.cpp / tiff.cpp / dpx.cpp need OpenEXR and libtiff): this binary
 * takes the formats that need no codec:
 *   .yuv / .rgb  16-bit planar integer (hdr2yuv.cpp:582-656; .rgb is R,G,B in the file, planes 2,0,1 in memory)
 *   .f32 / .f16  raw planar float / half in G,B,R plane order -- what dpx_read() or read_exr() (exr.cpp:233-235) leave in
 *                memory; the attributes those readers force on the input picture are forced here too
 *   --synthetic N  the seeded test frame of SURVEY 8c (no input file), treated as an .exr-like float input
 * and, from .yuv input, .rgb output: the .yuv -> .tiff flow (hdr2yuv.cpp:818-819, matrix_inverse) with the samples
 * write_tiff() would interleave written as planes R, G, B instead (no libtiff here).
 *
 * Several GPUs (--gpus N, an addition: the reference converts one frame per process): one host thread per GPU, each with
 * its own context and pinned ring; thread r takes a contiguous block of the frame indices (the split of
 * hdr2yuv_amd/shard.py), reads frame k at its offset in the source (hdr2yuv.cpp:624) and writes it at
 * `size of the file at start + k x frame bytes` -- the bytes N appending runs in frame order would have left (tiff.cpp:440).
 */
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "h2y_cli_args.h"

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

/* the seeded frame of SURVEY 8c: frame k of a run uses seed 12345 + k */
static void synth_fill(const h2y_desc &d, void *const planes[3], uint32_t seed)
{
    const size_t n = (size_t)d.width * d.height;
    uint32_t s = seed;
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

static bool write_at(int fd, const void *buf, size_t n, off_t at)
{
    const char *p = (const char *)buf;
    while (n) {
        ssize_t w = pwrite(fd, p, n, at);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += w;
        at += w;
        n -= (size_t)w;
    }
    return true;
}

struct block { /* one thread's share: frames [first, first + count) of the run, on `device` */
    int device = 0;
    long first = 0, count = 0;
    std::string err;
    long done = 0;
};

/* forward path: frames [first, first+count) through one context's pinned ring */
static void run_block(const cli_args &a, const h2y_desc &d, int fd_out, off_t base, block *b)
{
    h2y_ctx *ctx = nullptr;
    FILE *fin = nullptr;
    auto fail = [&](const std::string &m) {
        b->err = m;
        if (ctx) { h2y_stream_close(ctx); h2y_ctx_destroy(ctx); }
        if (fin) fclose(fin);
    };
    if (b->count < 1) return;
    if (h2y_ctx_create(b->device, &ctx)) return fail(h2y_last_error(nullptr));
    const size_t pb = h2y_plane_bytes(&d), ob = h2y_frame_bytes(&d);
    if (a.in_type != CLI_IN_SYNTH) {
        fin = fopen(a.src, "rb");
        if (!fin) return fail(std::string("unable to open file ") + a.src);
        if (fseeko(fin, (off_t)(3 * pb) * (off_t)(a.start_frame + b->first), SEEK_SET)) return fail("seek failed"); /* hdr2yuv.cpp:624 */
    }
    /* The reader fills the pinned slot of the pipeline directly, the writer drains what comes out of it two frames
     * later: upload, conversion and download of neighbouring frames overlap. */
    const int depth = 3;
    if (h2y_stream_open(ctx, &d, depth)) return fail(h2y_last_error(ctx));
    long in_flight = 0;
    auto drain_one = [&]() -> bool {
        const uint16_t *yuv = nullptr;
        if (h2y_stream_output(ctx, &yuv)) { fail(h2y_last_error(ctx)); return false; }
        const long k = b->first + b->done;
        if (!write_at(fd_out, yuv, ob, base + (off_t)k * (off_t)ob)) { fail(std::string("short write to ") + a.dst); return false; }
        if (a.verbose > 0) printf("frame %ld: %zu bytes written to %s (device %d)\n", k, ob, a.dst, b->device);
        b->done++;
        in_flight--;
        return true;
    };
    for (long f = 0; f < b->count; f++) {
        void *planes[3];
        if (h2y_stream_input(ctx, planes)) return fail(h2y_last_error(ctx));
        if (fin) {
            /* file plane order -> memory planes (0=G/Y, 1=B/Cb, 2=R/Cr); .rgb holds R,G,B (hdr2yuv.cpp:635-637) */
            const int order_rgb[3] = {2, 0, 1}, order_nat[3] = {0, 1, 2};
            const int *ord = a.in_type == CLI_IN_RGB ? order_rgb : order_nat;
            size_t got = 0;
            for (int k = 0; k < 3; k++) got += fread(planes[ord[k]], 1, pb, fin);
            if (got != 3 * pb) return fail("only " + std::to_string(got) + " bytes read from " + a.src + ", expecting " + std::to_string(3 * pb));
        } else synth_fill(d, planes, 12345u + (uint32_t)(a.synthetic + a.start_frame + b->first + f));
        if (h2y_stream_submit(ctx)) return fail(h2y_last_error(ctx));
        in_flight++;
        if (in_flight == depth - 1 && !drain_one()) return;
    }
    while (in_flight > 0)
        if (!drain_one()) return;
    h2y_stream_close(ctx);
    if (fin) fclose(fin);
    h2y_ctx_destroy(ctx);
}

/* .yuv -> RGB (matrix_inverse), frame by frame on the host-buffer entry */
static void run_block_inverse(const cli_args &a, int fd_out, off_t base, block *b)
{
    h2y_ctx *ctx = nullptr;
    FILE *fin = nullptr;
    auto fail = [&](const std::string &m) {
        b->err = m;
        if (ctx) h2y_ctx_destroy(ctx);
        if (fin) fclose(fin);
    };
    if (b->count < 1) return;
    if (h2y_ctx_create(b->device, &ctx)) return fail(h2y_last_error(nullptr));
    const size_t n = (size_t)a.in.width * a.in.height;
    const bool sub = a.in.chroma_format_idc == H2Y_CHROMA_420;
    const size_t nc = sub ? (size_t)(a.in.width / 2) * (a.in.height / 2) : n, in_frame = (n + 2 * nc) * 2, out_frame = 3 * n * 2;
    std::vector<uint16_t> in(n + 2 * nc), out(3 * n);
    fin = fopen(a.src, "rb");
    if (!fin) return fail(std::string("unable to open file ") + a.src);
    if (fseeko(fin, (off_t)in_frame * (off_t)(a.start_frame + b->first), SEEK_SET)) return fail("seek failed");
    for (long f = 0; f < b->count; f++) {
        if (fread(in.data(), 1, in_frame, fin) != in_frame) return fail(std::string("short read from ") + a.src);
        const uint16_t *ip[3] = {in.data(), in.data() + n, in.data() + n + nc};
        uint16_t *op[3] = {out.data() + n, out.data() + 2 * n, out.data()}; /* planes G,B,R -> file order R,G,B (write_tiff: R,G,B per pixel) */
        if (h2y_inverse_frame(ctx, a.in.width, a.in.height, a.in.chroma_format_idc, a.in.bit_depth, a.in.video_full_range_flag,
                              a.in.matrix_coeffs, a.out.bit_depth, a.resampler, ip, op))
            return fail(h2y_last_error(ctx));
        const long k = b->first + f;
        if (!write_at(fd_out, out.data(), out_frame, base + (off_t)k * (off_t)out_frame)) return fail(std::string("short write to ") + a.dst);
        b->done++;
    }
    fclose(fin);
    h2y_ctx_destroy(ctx);
}

int main(int argc, char **argv)
{
    cli_args a;
    cli_parse(a, argc, argv);
    if (!a.dst || (!a.src && a.synthetic < 0)) {
        if (!a.help) cli_help();
        return a.help ? 0 : 1;
    }
    if (cli_resolve(a)) {
        printf("TOO MANY ARGUMENT ERRORS. ABORTING PROGRAM. --help to show options\n\n"); /* hdr2yuv.cpp:567-572 (which exits 0) */
        return 1;
    }
    if (a.out.width != a.in.width || a.out.height != a.in.height) {
        printf("ERROR: resizing is not part of convert() (cv.cpp is compiled out in the reference)\n");
        return 1;
    }
    h2y_desc d;
    cli_make_desc(a, &d);
    size_t in_frame_bytes, out_frame_bytes;
    if (a.inverse) {
        if (a.out.bit_depth > 16 || a.in.bit_depth > 16) { printf("ERROR: bit depths must be 8..16 on the inverse flow\n"); return 1; }
        if (a.out.bit_depth < a.in.bit_depth) { /* tiff.cpp:564: SR = dst - src depth, then `R << SR` */
            printf("ERROR: dst bit_depth(%d) < src bit_depth(%d): write_tiff() would shift by a negative count (undefined in the reference)\n", a.out.bit_depth, a.in.bit_depth);
            return 1;
        }
        const size_t n = (size_t)a.in.width * a.in.height;
        const size_t nc = a.in.chroma_format_idc == H2Y_CHROMA_420 ? (size_t)(a.in.width / 2) * (a.in.height / 2) : n;
        in_frame_bytes = (n + 2 * nc) * 2;
        out_frame_bytes = 3 * n * 2;
    } else {
        const char *why = nullptr;
        if (h2y_desc_check(&d, &why)) { printf("ERROR: %s\n", why); return 1; }
        in_frame_bytes = 3 * h2y_plane_bytes(&d);
        out_frame_bytes = h2y_frame_bytes(&d);
    }

    /* how many frames there are to do: --n_frames, or what the file holds from --src_start_frame on if that is fewer */
    long frames = a.n_frames > 0 ? a.n_frames : 1;
    struct stat st;
    if (a.in_type != CLI_IN_SYNTH && !(a.dry_run && stat(a.src, &st))) { /* (a dry run may name a file that is not there) */
        if (stat(a.src, &st)) { printf("ERROR: unable to open file %s\n", a.src); return 1; }
        const long have = (long)((st.st_size - (off_t)in_frame_bytes * a.start_frame) / (off_t)in_frame_bytes);
        if (st.st_size < (off_t)in_frame_bytes * (a.start_frame + 1)) {
            printf("ERROR: only %lld bytes in %s, expecting %zu from frame %d on\n", (long long)st.st_size, a.src, in_frame_bytes, a.start_frame);
            return 1;
        }
        if (frames > have) frames = have;
    }
    if (a.gpus < 1) a.gpus = 1;
    if (a.devices.empty()) {
        if (a.gpus == 1) a.devices.push_back(a.device);
        else for (int r = 0; r < a.gpus; r++) a.devices.push_back(r);
    }
    if ((int)a.devices.size() != a.gpus) { printf("ERROR: --devices names %zu devices, --gpus %d\n", a.devices.size(), a.gpus); return 1; }
    printf("gpus: %d (devices", a.gpus);
    for (int dv : a.devices) printf(" %d", dv);
    printf(")\nframes: %ld\nframe_bytes: %zu\n", frames, out_frame_bytes);
    if (a.dry_run) return 0;

    /* tiff.cpp:440 opens ios::ate | ios::app: what is in the file stays, frames go behind it */
    int fd = open(a.dst, O_WRONLY | O_CREAT, 0644);
    if (fd < 0) { printf("ERROR: unable to open %s\n", a.dst); return 1; }
    const off_t base = lseek(fd, 0, SEEK_END);

    /* contiguous blocks of frame indices, the first `frames % gpus` one longer (hdr2yuv_amd/shard.py) */
    std::vector<block> blocks(a.gpus);
    long at = 0;
    for (int r = 0; r < a.gpus; r++) {
        blocks[r].device = a.devices[r];
        blocks[r].first = at;
        blocks[r].count = frames / a.gpus + (r < frames % a.gpus ? 1 : 0);
        at += blocks[r].count;
    }
    auto work = [&](block *b) { a.inverse ? run_block_inverse(a, fd, base, b) : run_block(a, d, fd, base, b); };
    if (a.gpus == 1) work(&blocks[0]);
    else {
        std::vector<std::thread> th;
        for (int r = 0; r < a.gpus; r++) th.emplace_back(work, &blocks[r]);
        for (auto &t : th) t.join();
    }
    close(fd);
    int rc = 0;
    for (int r = 0; r < a.gpus; r++)
        if (!blocks[r].err.empty()) {
            printf("ERROR (device %d, frames %ld..%ld): %s\n", blocks[r].device, blocks[r].first, blocks[r].first + blocks[r].count - 1, blocks[r].err.c_str());
            rc = 1;
        }
    return rc;
}
