/*
 * h2y_api.hip -- the C-ABI shim declared in include/hdr2yuv_hip.h.
 *
 * Host side of the drop-in boundary: descriptor validation, the scalar setup
 * the reference does in init_pic()/set_pic_clip() (common.cpp:172-327), device
 * buffer ownership, and kernel launches.  No pixel is ever computed on the
 * host: without a HIP device h2y_ctx_create() fails and nothing else works.
 */
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/hdr2yuv_hip.h"
#include "h2y_kernels.h"
#include "h2y_math.h"
#include "h2y_walk.h"

using namespace h2y;

namespace {

thread_local std::string g_err;

struct clip_limits { /* clip_limits_t, hdr.h:345-356 */
    uint32_t minCV, maxCV, minVR, maxVR, minVRC, maxVRC, Half;
};

/* set_pic_clip(), common.cpp:300-327 */
clip_limits make_clip(int bit_depth, int full_range)
{
    clip_limits c;
    c.minCV = 0;
    c.maxCV = (1u << bit_depth) - 1;
    c.Half = 1u << (bit_depth - 1);
    if (!full_range) {
        uint32_t D = 1u << (bit_depth - 8);
        c.minVR = 16 * D;
        c.maxVR = 219 * D + c.minVR; /* = 235*D, kept as the reference has it */
        c.minVRC = c.minVR;
        c.maxVRC = 224 * D + c.minVRC;
    } else {
        c.minVR = 0;
        c.maxVR = c.maxCV;
        c.minVRC = 0;
        c.maxVRC = c.maxCV;
    }
    return c;
}

const int kMaxEvents = 64;
const size_t kRangeWords = 1025; /* a table of slice ranges: up to 1024 blocks of a group + 1 */
/* Share of a batch's pixels (counted in tiles of eight) the first tier passed on, above which the next batches go to the binary64
 * tier's kernels.  Both first-tier kernels now take that tier inside their loops, a wave at a time: with 0.1-0.2 % of the pixels
 * passed on they are 11-16 % ahead of the binary64 tier's kernels, with 1.3-1.5 % 8-19 % behind (tools/densebench.sh: u^2 and u^3
 * of the uniform picture); the lines cross near 0.7 %.  (Rounds 1-2: 8 % -- of tiles, one unsettled pixel making a tile.) */
const double kT1DenseShare = 0.007;
/* A probe of the first tier on dense content is dear (letterboxed 4K, a quarter of the tiles flagged: 8.6 ms per 64-frame launch
 * against k_fused2's 1.6), staying on the binary64 tier too long is cheap (2-10 % slower than the first tier on content that
 * suits it): probe rarely -- after 32 batches, then 64, ... 1024. */
const double kFirSyncMaxFlagged = 0.004; /* k_fir_fused: tiles-of-eight share of unsettled pixels above which its waves are left out of step */
const int kTailMinFrames = 8; /* frames per group from which its last one is drawn dynamically (h2y_walk.h): the plain loop that takes it is
                                 slower than the prefetching one, and a frame is 1/8 of the group's work at most */
const int kT1SkipBatches = 32;
const int kT1SkipBatchesMax = 1024;
const int kFirSubBatch = 32; /* frames per fused launch on the FIR path: every launch pays its table staging and its last redo pass */

} // namespace

/* Everything one batch in flight owns: two of them let h2y_convert_batch_enqueue() queue batch k+1 behind batch k
 * before h2y_batch_finish() has looked at k (the 35 us between two launches -- the statistics kernel, one copy, the
 * host's turn-around -- disappear behind the running kernel). */
struct batch_state {
    /* per-batch device arrays */
    frame_io *d_frames = nullptr, *h_frames = nullptr;
    size_t frames_cap = 0;
    std::vector<frame_io> dev_frames; /* what d_frames holds (size frames_cap once anything was copied; cleared when d_frames is reallocated) */
    float *d_partial = nullptr;
    size_t partial_cap = 0;
    uint32_t *d_redo = nullptr; /* k_fused_t1: per-wave counts of redone tiles */
    size_t redo_cap = 0;
    uint32_t *d_low = nullptr;  /* k_fused_t1: per-frame flag "a sample <= -1 was seen" (zero between launches) */
    size_t low_cap = 0;
    bool approx_min = false;    /* the batch's statistics hold a subsampled minimum (exact only where they match) */
    unsigned long long *d_clock = nullptr;
    size_t clock_cap = 0;
    int bal_slot = 0;                         /* the eight run times travel in the frame_stats entry after the batch's last */
    bool bal_pending = false;                 /* h_fstats[bal_slot] will hold the times of a launch dealt with bal_work */
    double bal_work[8] = {1, 1, 1, 1, 1, 1, 1, 1}; /* relative work a block of XCD x had in that launch */
    /* fused_args.slice_ranges ([blocks of a group + 1]): two tables in pinned host memory that the kernels read in place (a
     * block reads two words of it, once) -- no copy command between two launches.  Two, because the launches of one batch may
     * need different tables (the last one, when it holds fewer frames) while the earlier ones have not run yet. */
    uint32_t *h_ranges = nullptr, *hd_ranges = nullptr; /* host and device address of the same 2 x kRangeWords words */
    uint32_t *d_tail = nullptr; /* the dynamic last frame's counters: [16 groups][H2Y_TAIL_WORDS]: counters and exhausted bits, zero between launches (k_stats_final) */
    float *h_btime = nullptr, *hd_btime = nullptr; /* every block's run time of a timed launch (pinned, written by k_stats_final) */
    int bal_grid = 0, bal_groups = 0;              /* the launch those times (and bal_bwork) belong to; 0: none */
    std::vector<double> bal_bwork;                 /* relative work each block of the grid had in that launch */
    std::vector<uint32_t> range_slot[2];      /* what the two tables hold */
    bool slot_busy[2] = {false, false};       /* a launch of the batch being queued reads it */
    /* k_fir_fused: the rows of every unit (frame, segment, strip), cut by XCD speed */
    uint32_t *d_unit_rows = nullptr, *h_unit_rows = nullptr;
    size_t unit_rows_cap = 0;                 /* in units */
    std::vector<uint32_t> dev_unit_rows;      /* what d_unit_rows holds */
    bool ffb_pending = false;                 /* h_fstats[bal_slot] will hold the XCD run times of a k_fir_fused launch ... */
    double ffb_work[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* ... in which a block of XCD x had this much work (steps, mean) */
    frame_stats *d_fstats = nullptr, *h_fstats = nullptr;
    frame_stats *m_fstats = nullptr; /* h_fstats as the device sees it (pinned host memory): k_stats_final of a batch writes there, no copy command */
    frame_stats *fs_out = nullptr;   /* where run_frames() has the statistics written: d_fstats, or m_fstats for an enqueued batch */
    assumed_stats *d_assumed = nullptr, *h_assumed = nullptr; /* [2]: [0] batch, [1] redo */
    assumed_stats dev_assumed;       /* what d_assumed[0] holds when dev_assumed_ok (one small copy command less per batch) */
    bool dev_assumed_ok = false;
    /* the batch itself, between enqueue and finish */
    h2y_desc p_desc;
    int p_n = 0;
    bool p_check = false;
    bool was_t1 = false;
    std::vector<frame_io> p_frames;
    hipEvent_t ev_done = nullptr; /* after the batch's last operation on the stream (the copy of its statistics) */
    /* timing of the main kernels */
    hipEvent_t ev[kMaxEvents][2];
    int n_ev = 0;
};

struct h2y_ctx {
    int device = 0;
    batch_state bs[2];
    batch_state *b = &bs[0]; /* the batch the shim is working on (enqueue: the newest; finish: the oldest) */
    int q_head = 0, q_count = 0; /* batches in flight: bs[q_head] is the oldest */
    int n_cu = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    /* FIR pass runs on its own stream so that it overlaps the next sub-batch's fused kernel */
    hipStream_t fir_stream = nullptr;
    hipEvent_t ev_fused[2] = {nullptr, nullptr}, ev_fir[2] = {nullptr, nullptr};
    bool fir_used[2] = {false, false};
    void *d_table = nullptr;
    void *d_table1 = nullptr; /* binary32 first-tier records */
    void *d_table_ext = nullptr; /* pq_build_table_ext(): the binary64 table below 2^-24, read from global memory by pq_slow() */
    void *d_tfn[H2Y_TFN_COUNT] = {}; /* the other transfer functions' tables (tfn_build_table), built when first needed */
    void *d_tfn_ext[H2Y_TFN_COUNT] = {}; /* and their full-range tables in global memory (tfn_build_ext; PQ10000_r's is d_table_ext) */
    float *d_lut16 = nullptr; /* PQ10000_r of every half in [0,2), built on the device at creation */
    /* The first tier is slow on pictures with many exactly-zero samples (black bars: every such tile is done twice).
     * The kernel counts the tiles it had to redo; when their share in a batch exceeds kT1DenseShare the next
     * kT1SkipBatches batches go to k_fused2 (the binary64 tier answers zero by itself), then the first tier is
     * tried again. */
    /* Balancing across XCDs (frame_walk in h2y_kernels.hip): the loop-form kernels leave the mean run time of the blocks
     * of each XCD; the shares of the next launch follow the speeds seen (balance_update()). */
    bool bal_have = false;
    double bal_speed[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    std::vector<double> bal_bspeed; /* per block of the grid (round 3): what is left between blocks once their XCDs are level */
    int bal_bgrid = 0, bal_bgroups = 0; /* the grid shape bal_bspeed is for */
    bool ffb_have = false;                    /* k_fir_fused has its own speeds: it is vector-issue bound, the XCDs differ more on it */
    double ffb_speed[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    int t1_skip = 0, t1_skip_len = 0;
    bool cur_skip_t1 = false;
    /* h2y_ctx_set_option(): tuning / test knobs, per context (nothing is read from the environment) */
    bool opt_t1 = true;        /* "t1": binary32 first tier on */
    bool opt_t1_steer = true;  /* ... and left for the binary64 tier's kernels while the pictures keep it busy passing pixels on */
    int opt_groups = 0;        /* "groups": at most this many frame groups (power of two; 1 = off); 0 = by the frame's size (groups_cap()) */
    bool opt_cols8 = true;     /* "cols8": 8-column tiles for half input where the planes allow */
    int opt_bal_mode = 0;      /* "balance": 0 adaptive, 1 off, 2 fixed */
    int opt_tail = 2;           /* "tail": 0 auto (groups of at least kTailMinFrames frames), 1 on (two frames suffice), 2 off (the default: measured
                                   neutral on 64 x 4K -- the blocks' finish times close up from +-30 us to +-15 us of a 1.5 ms launch, and the
                                   frame's own dealing costs what that saves) */
    bool opt_bal_blocks = true; /* adaptive: by the speed of every block ("adaptive"), or of the XCDs only ("xcd") */
    uint32_t opt_bal_mask = 0xFFu;
    double opt_bal_rho = 1.0;
    int opt_fir = 0;           /* "fir": 0 auto, 1 two-pass (4:4:4 scratch + k_fir420), 2 fused single pass where it applies */
    int opt_fir_sync = -1;     /* "firsync": k_fir_fused's blocks meet at a barrier every so many steps (power of two; 0 = never);
                                  -1 = by the pictures: every step, never while the first tier passes many pixels on */
    double fir_flag_share = 0.0; /* share of the last k_fir_fused batch's pixels (in tiles of eight) the first tier could not settle */
    uint16_t *d_tmp = nullptr;
    size_t tmp_cap = 0;

    /* staging for the host-buffer entry */
    void *d_in = nullptr;
    size_t in_cap = 0;
    uint16_t *d_out = nullptr;
    size_t out_cap = 0;
    /* floor/ceiling of the last frame seen: the assumption for the next batch */
    bool have_hint = false;
    int hint_kind = -1;
    int32_t hint_floor[3] = {0, 0, 0}, hint_ceil[3] = {0, 0, 0};
    /* streaming pipeline (h2y_stream_*): a ring of pinned host slots with device twins */
    struct stream_slot {
        char *h_in = nullptr;      /* pinned: three planes, each plane_al bytes apart */
        uint16_t *h_out = nullptr; /* pinned: one .yuv frame */
        char *d_in = nullptr;
        uint16_t *d_out = nullptr;
        hipEvent_t ev_h2d = nullptr, ev_conv = nullptr, ev_done = nullptr;
        int state = 0; /* 0 free, 1 handed out for filling, 2 submitted, 3 output lent to the caller */
    };
    std::vector<stream_slot> ss;
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    h2y_desc s_desc;
    size_t s_plane_al = 0;
    int s_head = 0, s_tail = 0, s_lent = -1;
    bool streaming = false;
    int slot_base = 0; /* run_frames(): first entry of d_frames/h_frames to use (one per stream slot) */
    float last_ms = 0.f;
    const char *last_name = "";
    std::string last_variant; /* last_name with its template arguments and launch shape, e.g. "k_fused_t1<F32,420BOX,YCBCR,PQ_IDENT> groups=8 xcd=1" */
    int last_launches = 0;
    std::string err;
};

namespace {

int fail(h2y_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    if (ctx) ctx->err = buf;
    return code;
}

#define HIP_TRY(ctx, call)                                                                                  \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) return fail(ctx, H2Y_EHIP, "%s: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <typename T> int ensure(h2y_ctx *ctx, T *&p, size_t &cap, size_t need_bytes)
{
    if (cap >= need_bytes) return 0;
    if (p) HIP_TRY(ctx, hipFree(p));
    p = nullptr;
    cap = 0;
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, need_bytes);
    if (e != hipSuccess) return fail(ctx, H2Y_ENOMEM, "hipMalloc(%zu): %s", need_bytes, hipGetErrorString(e));
    p = static_cast<T *>(q);
    cap = need_bytes;
    return 0;
}

/* transfer_characteristics code -> what matrix_convert() does with it (convert.cpp:1024-1109);
 * -1: the reference only prints a warning for every pixel */
int tf_class(int t)
{
    switch (t) {
    case 8: return H2Y_TF_LINEAR;
    case 16: return H2Y_TF_PQ;
    case 18: return H2Y_TF_RHO_GAMMA;
    case 1: case 6: case 14: case 15: return H2Y_TF_BT1886; /* BT709, BT601, BT2020_10bit, BT2020_12bit */
    default: return -1;
    }
}

int in_kind_of(const h2y_desc *d)
{
    return d->in_sample_type == H2Y_SAMPLE_F32 ? H2Y_IN_F32 : d->in_sample_type == H2Y_SAMPLE_F16 ? H2Y_IN_F16 : H2Y_IN_U16;
}
size_t sample_bytes(const h2y_desc *d) { return d->in_sample_type == H2Y_SAMPLE_F32 ? 4 : 2; }

/* hdr2yuv.cpp:803-808: the matrix_convert() target takes the input's depth
 * when both pictures are U16, else the output's */
int tmp_depth_of(const h2y_desc *d) { return d->in_sample_type == H2Y_SAMPLE_U16 ? d->src_bit_depth : d->dst_bit_depth; }

/* Scalar setup for the kernels: everything matrix_convert()/convert()/
 * write_yuv() derive from the picture attributes before their pixel loops. */
void derive_params(const h2y_desc *d, pix_params *pp, bool stage_matrix_only)
{
    memset(pp, 0, sizeof *pp);
    const int tmp_depth = tmp_depth_of(d);
    const clip_limits tc = make_clip(tmp_depth, d->dst_full_range);
    const clip_limits oc = make_clip(d->dst_bit_depth, d->dst_full_range);
    pp->src_tf = tf_class(d->src_transfer);
    pp->dst_tf = tf_class(d->dst_transfer);
    if (d->src_transfer == d->dst_transfer) pp->convert_transfer = 0; /* convert.cpp:930 */
    else pp->convert_transfer = (pp->src_tf == H2Y_TF_LINEAR && pp->dst_tf == H2Y_TF_PQ) ? 1 : 2;
    /* the two stages of a generic pair (tables in h2y_math.h); -1 until run_frames() has the tables on the device */
    pp->src_fn = pp->dst_fn = -1;
    /* convert.cpp:1123-1145 (full range: multiply only; add stays 0.0f) */
    if (d->dst_full_range) {
        pp->mulY = pp->mulC = (float)tc.maxCV;
    } else if (d->dst_matrix == H2Y_MATRIX_GBR) {
        pp->mulY = pp->mulC = (float)(int)tc.maxVR;
        pp->addY = pp->addC = (float)(int)tc.minVR;
    } else {
        pp->mulY = (float)(int)tc.maxVR;
        pp->addY = (float)(int)tc.minVR;
        pp->mulC = (float)(int)tc.maxVRC;
        pp->addC = (float)(int)tc.minVRC;
    }
    /* convert.cpp:1159-1198 */
    if (d->dst_matrix == d->src_matrix && d->dst_primaries == d->src_primaries) pp->mode = H2Y_MODE_IDENTITY;
    else if (d->dst_matrix == H2Y_MATRIX_YDZDX) pp->mode = H2Y_MODE_YDZDX;
    else if (d->dst_matrix == H2Y_MATRIX_BT2020NC) {
        pp->mode = H2Y_MODE_YCBCR;
        pp->kr = 0.2627; pp->kg = 0.6780; pp->kb = 0.0593; pp->dcb = 1.8814; pp->dcr = 1.4746;
    } else if (d->dst_matrix == H2Y_MATRIX_BT709) {
        pp->mode = H2Y_MODE_YCBCR;
        pp->kr = 0.2126; pp->kg = 0.7152; pp->kb = 0.0722; pp->dcb = 1.8556; pp->dcr = 1.5748;
    } else {
        pp->mode = H2Y_MODE_YPQRS; /* convert.cpp:913-925 */
        if (d->dst_matrix == H2Y_MATRIX_YDZDX_Y100) { pp->P = -0.5f; pp->Q = 0.491722f; pp->RR = 0.5f; pp->S = -0.49495f; }
        else { pp->P = -0.5f; pp->Q = 0.493393f; pp->RR = 0.5f; pp->S = -0.49602f; }
    }
    if (pp->mode == H2Y_MODE_YCBCR) {
        pp->inv_dcb = 1.0 / pp->dcb;
        pp->inv_dcr = 1.0 / pp->dcr;
    }
    pp->half_m1 = tc.Half - 1;
    pp->maxCV = tc.maxCV;
    pp->fir_max = (float)tc.maxCV;
    if (stage_matrix_only) { /* identity clamp: values are already <= maxCV <= 65535 */
        pp->down_shift = 0;
        pp->ylo = pp->clo = 0;
        pp->yhi = pp->chi = 0xFFFFu;
    } else {
        pp->down_shift = tmp_depth - d->dst_bit_depth; /* tiff.cpp:394 */
        if (d->dst_full_range) { /* tiff.cpp:476: only "> maxCV" */
            pp->ylo = pp->clo = 0;
            pp->yhi = pp->chi = oc.maxCV;
        } else {
            pp->ylo = oc.minVR; pp->yhi = oc.maxVR; pp->clo = oc.minVRC; pp->chi = oc.maxVRC;
        }
    }
    pix_limits_finish(pp);
}

struct geom {
    bool narrow;
    uint32_t wq, wq_magic, tiles, chunks;
};
geom make_geom(const h2y_desc *d, int threads, int cols = 4 /* columns of a thread tile (8: k_fused_lut16 on wide-aligned pictures) */)
{
    geom g;
    g.narrow = (d->width % 4) != 0;
    g.wq = g.narrow ? (uint32_t)d->width : (uint32_t)d->width / (uint32_t)cols;
    g.wq_magic = (uint32_t)(0x100000000ull / g.wq);
    if (g.wq == 1) g.wq_magic = 0xFFFFFFFFu;
    g.tiles = g.wq * (uint32_t)((d->height + 1) / 2);
    g.chunks = (g.tiles + threads - 1) / threads;
    return g;
}

/* Frame groups of a launch (frame_walk in h2y_kernels.hip), unless the caller set a number: as few as leave every block
 * kMinSlicesPerBlock 64-tile slices of a frame -- a 4K frame on 256 blocks: two groups, 1080p: eight, 8K: one.  Few, because
 * with g groups g frames are read and written at equal offsets at any moment, and whether those streams meet in the same DRAM
 * banks depends on where the frames happen to lie: eight groups ran the same 64 x 4K launch in 1.42 ... 1.72 ms from one set of
 * buffers to the next, two in 1.42 ... 1.51, one in 1.44 ... 1.49 (tools/layoutbench.py).  Not fewer, because a block pays for
 * every frame it visits (its waves' tickets, statistics records, the run-in of its prefetch): 1080p at one group runs at 0.47 of
 * the bandwidth it reaches at eight (0.61). */
const int kMinSlicesPerBlock = 100;
int groups_cap(const h2y_ctx *ctx, uint32_t tiles_per_frame, int grid)
{
    if (ctx->opt_groups) return ctx->opt_groups;
    const uint64_t slices = (tiles_per_frame + 63u) / 64u;
    int ng = 1;
    while (ng < 8 && slices * (uint64_t)ng < (uint64_t)kMinSlicesPerBlock * (uint64_t)grid) ng *= 2;
    return ng;
}

int grid_for(const h2y_ctx *ctx, const fused_variant &v, uint64_t total_chunks)
{
    /* persistent grid: exactly the blocks the chip holds at once */
    uint64_t g = (uint64_t)ctx->n_cu * h2y_fused_blocks_per_cu(v);
    if (g > total_chunks) g = total_chunks;
    if (g < 1) g = 1;
    return (int)g;
}

/* start of a batch: is the first tier to be skipped this time? */
void t1_begin_batch(h2y_ctx *ctx)
{
    ctx->cur_skip_t1 = ctx->t1_skip > 0;
    if (ctx->cur_skip_t1) ctx->t1_skip--;
}
/* end of a batch that ran k_fused_t1: how many of its tiles had to be redone */
void t1_end_batch(h2y_ctx *ctx, const h2y_desc *d, const frame_stats *fs, int n)
{
    if (!ctx->b->was_t1 || n < 1) return;
    uint64_t redone = 0;
    for (int f = 0; f < n; f++) redone += fs[f].redone;
    const uint64_t tiles = (uint64_t)n * make_geom(d, 1024).tiles;
    if (!strcmp(ctx->last_name, "k_fir_fused")) ctx->fir_flag_share = tiles ? (double)redone / (double)tiles : 0.0;
    { /* for whoever asks h2y_last_kernel_variant(): the share of tiles the first tier passed on */
        const size_t at = ctx->last_variant.find(" flagged=");
        if (at != std::string::npos) ctx->last_variant.erase(at);
        char note[48];
        snprintf(note, sizeof note, " flagged=%.5f", tiles ? (double)redone / (double)tiles : 0.0);
        ctx->last_variant += note;
    }
    if (ctx->opt_t1_steer && (double)redone > kT1DenseShare * (double)tiles) {
        /* still dense at the next probe: stay away twice as long */
        ctx->t1_skip_len = ctx->t1_skip_len ? (ctx->t1_skip_len < kT1SkipBatchesMax ? 2 * ctx->t1_skip_len : kT1SkipBatchesMax) : kT1SkipBatches;
        ctx->t1_skip = ctx->t1_skip_len;
    } else ctx->t1_skip_len = 0;
}

/* known: the floor/ceiling the kernels will assume, when the HOST knows them (hint or
 * override); NULL when they only exist in device memory (stats pre-pass). */
fused_variant pick_variant(const h2y_ctx *ctx, const h2y_desc *d, const pix_params &pp, int out_kind, const assumed_stats *known, t1_sens *sn)
{
    memset(sn, 0, sizeof *sn);
    fused_variant v;
    v.in_kind = in_kind_of(d);
    v.out_kind = out_kind;
    v.mode = pp.mode;
    v.narrow = (d->width % 4) != 0;
    v.even_h = (d->height & 1) == 0;
    v.pipe = 0;
    /* equal transfers (the 16-bit .tiff / .yuv flows): samples straight into the matrix */
    if (!pp.convert_transfer && !v.narrow && v.even_h && (pp.mode == H2Y_MODE_YCBCR || pp.mode == H2Y_MODE_YDZDX)) v.pipe = 6;
    if (pp.convert_transfer && !v.narrow) {
        bool ident = known != nullptr;
        for (int c = 0; c < 3 && ident; c++) ident = known->floor_[c] == 0 && known->ceil_[c] == 1;
        v.pipe = ident ? 1 : 2; /* 2 is always valid: (x - 0) / 1 == x exactly */
        if (pp.convert_transfer == 2) /* generic transfer pair: its two stages' tables, in the loop form where that exists */
            v.pipe = (v.even_h && (pp.mode == H2Y_MODE_YCBCR || pp.mode == H2Y_MODE_YDZDX)) ? 7 /* H2Y_PIPE_TFN */ : 0;
        /* binary32 first tier where few pixels would fall through it (moderate bit depths) */
        if ((v.pipe == 1 || v.pipe == 2) && v.in_kind != H2Y_IN_U16 && (d->height & 1) == 0 && ctx->opt_t1 && t1_bounds(pp, sn)) {
            v.pipe += 3;
            v.t1_ok = true;
        }
        /* half input with the identity normalisation: the whole transfer is a 64 KB table */
        if (pp.convert_transfer == 1 && ident && v.in_kind == H2Y_IN_F16 && v.even_h && (pp.mode == H2Y_MODE_YCBCR || pp.mode == H2Y_MODE_YDZDX)) v.pipe = 3;
    }
    return v;
}

constexpr int kMaxFramesPerLaunch = 128;

int out_kind_of(const h2y_desc *d)
{
    if (d->dst_chroma_format_idc == H2Y_CHROMA_444) return H2Y_OUT_444;
    return d->chroma_resampler_type == 0 ? H2Y_OUT_420BOX : H2Y_OUT_444TMP;
}

/* after a launch whose block clocks came back: speed of each XCD = the share its blocks had / the time they took, and the
 * same for every block by itself; the next launch's slice ranges follow the speeds (run_frames()) */
void balance_update(h2y_ctx *ctx)
{
    if (!ctx->b->bal_pending) return;
    ctx->b->bal_pending = false;
    double sp[8], mean = 0.0;
    for (int x = 0; x < 8; x++) {
        const double t = reinterpret_cast<const float *>(ctx->b->h_fstats + ctx->b->bal_slot)[x];
        if (!(t > 0.0)) return; /* grid smaller than a round of XCDs, or nothing measured */
        sp[x] = ctx->b->bal_work[x] / t;
        mean += sp[x] / 8.0;
    }
    for (int x = 0; x < 8; x++) {
        double v = sp[x] / mean;
        if (v < 0.75) v = 0.75;
        if (v > 1.25) v = 1.25;
        ctx->bal_speed[x] = ctx->bal_have ? 0.5 * ctx->bal_speed[x] + 0.5 * v : v;
    }
    ctx->bal_have = true;
    /* per block: a launch ends with its slowest BLOCK, and blocks of one XCD differ too (+-0.5 % of a launch, half of it the
     * same blocks from launch to launch).  Lighter smoothing than for the XCDs: one block's time is noisier than the mean of 32 */
    const int grid = ctx->b->bal_grid;
    if (grid > 0 && grid <= 1024 && ctx->b->h_btime && (int)ctx->b->bal_bwork.size() == grid) {
        std::vector<double> bs((size_t)grid);
        double bmean = 0.0;
        for (int b = 0; b < grid; b++) {
            const double t = ctx->b->h_btime[b];
            if (!(t > 0.0)) return;
            bs[(size_t)b] = ctx->b->bal_bwork[(size_t)b] / t;
            bmean += bs[(size_t)b] / grid;
        }
        const bool have = ctx->bal_bgrid == grid && ctx->bal_bgroups == ctx->b->bal_groups && (int)ctx->bal_bspeed.size() == grid;
        if (!have) ctx->bal_bspeed.assign((size_t)grid, 1.0);
        for (int b = 0; b < grid; b++) {
            double v = bs[(size_t)b] / bmean;
            if (v < 0.75) v = 0.75;
            if (v > 1.25) v = 1.25;
            ctx->bal_bspeed[(size_t)b] = have ? 0.65 * ctx->bal_bspeed[(size_t)b] + 0.35 * v : v;
        }
        ctx->bal_bgrid = grid;
        ctx->bal_bgroups = ctx->b->bal_groups;
    }
}

/* the table of transfer function fn on the device (built on the host the first time it is asked for) */
int ensure_tfn(h2y_ctx *ctx, int fn)
{
    if (fn <= H2Y_TFN_NONE || fn >= H2Y_TFN_COUNT || ctx->d_tfn[fn]) return 0;
    std::vector<pq_recA> A(H2Y_PQ_NREC);
    std::vector<pq_recB> B(H2Y_PQ_NREC);
    (void)tfn_build_table(fn, A.data(), B.data());
    char *t = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&t, H2Y_PQ_TABLE_BYTES));
    ctx->d_tfn[fn] = t;
    HIP_TRY(ctx, hipMemcpy(t, A.data(), H2Y_PQ_NREC * 16, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(t + H2Y_PQ_NREC * 16, B.data(), H2Y_PQ_NREC * 16, hipMemcpyHostToDevice));
    if (fn == H2Y_TFN_PQ_R) ctx->d_tfn_ext[fn] = ctx->d_table_ext; /* the same function, the same layout */
    else {
        std::vector<pq_ext_rec> X(H2Y_PQX_NSEG);
        (void)tfn_build_ext(fn, X.data());
        void *x = nullptr;
        HIP_TRY(ctx, hipMalloc(&x, H2Y_PQX_TABLE_BYTES));
        ctx->d_tfn_ext[fn] = x;
        HIP_TRY(ctx, hipMemcpy(x, X.data(), H2Y_PQX_TABLE_BYTES, hipMemcpyHostToDevice));
    }
    return 0;
}

/* after a k_fir_fused launch whose block clocks came back: speed of each XCD = steps a wave of it had / time it took */
void ffb_update(h2y_ctx *ctx)
{
    if (!ctx->b->ffb_pending) return;
    ctx->b->ffb_pending = false;
    double sp[8], mean = 0.0;
    for (int x = 0; x < 8; x++) {
        const double t = reinterpret_cast<const float *>(ctx->b->h_fstats + ctx->b->bal_slot)[x];
        if (!(t > 0.0) || !(ctx->b->ffb_work[x] > 0.0)) return;
        sp[x] = ctx->b->ffb_work[x] / t;
        mean += sp[x] / 8.0;
    }
    for (int x = 0; x < 8; x++) {
        double v = sp[x] / mean;
        if (v < 0.75) v = 0.75;
        if (v > 1.25) v = 1.25;
        ctx->ffb_speed[x] = ctx->ffb_have ? 0.5 * ctx->ffb_speed[x] + 0.5 * v : v;
    }
    ctx->ffb_have = true;
}

/* launch fused (+FIR) over frames [0,n) whose frame_io entries are in h_frames */
int run_frames(h2y_ctx *ctx, const h2y_desc *d, const frame_io *frames, int n, const assumed_stats *d_assumed,
               const assumed_stats *known, bool check, int fstats_offset, bool time_it)
{
    pix_params pp;
    derive_params(d, &pp, false);
    pp.pq_ext = ctx->d_table_ext;
    const int out_kind = out_kind_of(d);
    t1_sens sn;
    fused_variant var = pick_variant(ctx, d, pp, out_kind, known, &sn);
    /* k_fused_t1's redo list numbers tiles as frame * tiles + tile in 32 bits */
    if ((var.pipe == 4 || var.pipe == 5) && (uint64_t)n * make_geom(d, h2y_fused_threads(var)).tiles >= 0xFFFFFFFFull) var.pipe -= 3;
    if ((var.pipe == 4 || var.pipe == 5) && ctx->cur_skip_t1) var.pipe -= 3; /* dense zeros lately: binary64 tier for now */
    ctx->b->was_t1 = var.pipe == 4 || var.pipe == 5;
    if (var.pipe == 3 && d->width % 8 == 0) { /* half input through the table: 8-column tiles when every plane allows 16-byte accesses */
        bool ok = ctx->opt_cols8;
        for (int i = 0; i < n && ok; i++) {
            for (int c = 0; c < 3; c++) ok = ok && (reinterpret_cast<uintptr_t>(frames[i].in[c]) & 15u) == 0;
            ok = ok && (reinterpret_cast<uintptr_t>(frames[i].out) & 15u) == 0;
        }
        var.cols8 = ok;
    }
    if (out_kind == H2Y_OUT_444TMP && ctx->opt_fir != 1 && var.t1_ok && !ctx->cur_skip_t1 && tmp_depth_of(d) <= H2Y_FIR_INT_MAX_DEPTH) {
        /* The FIR resampler in one pass (k_fir_fused): a wave's unit of work is (frame, segment of chroma rows, strip of
         * 240 columns).  Segments: as few as give every wave of the chip a unit, never shorter than 64 rows (each cut
         * costs six recomputed row pairs).  "auto" keeps short batches, which cannot fill the chip that way, on the
         * two-pass form. */
        const uint32_t wq = (uint32_t)d->width / 4u, h2 = (uint32_t)d->height / 2u;
        const uint32_t ns = (wq + H2Y_FF_OWN_LANES - 1u) / H2Y_FF_OWN_LANES, gw = (uint32_t)ctx->n_cu * 16u;
        const uint32_t max_seg = h2 / 64u > 0u ? h2 / 64u : 1u;
        uint32_t want = (gw + (uint32_t)n * ns - 1u) / ((uint32_t)n * ns);
        if (want > max_seg) want = max_seg;
        if (want < 1u) want = 1u;
        const uint32_t seg_rows = (h2 + want - 1u) / want, nseg = (h2 + seg_rows - 1u) / seg_rows;
        const uint64_t units = (uint64_t)n * ns * nseg;
        if (ctx->opt_fir == 2 || 2u * units >= gw) {
            const bool ident = var.pipe == 4 || var.pipe == 3; /* assumed floor 0 / ceiling 1 (pipe 3: half input, the table kernel's case) */
            const uint32_t upf = ns * nseg;
            ctx->b->was_t1 = true;
            bool on_device = ctx->b->dev_frames.size() == ctx->b->frames_cap;
            if (!on_device) ctx->b->dev_frames.assign(ctx->b->frames_cap, frame_io{});
            for (int i = 0; i < n; i++) {
                const size_t idx = (size_t)ctx->slot_base + i;
                on_device = on_device && memcmp(&ctx->b->dev_frames[idx], &frames[i], sizeof(frame_io)) == 0;
                ctx->b->h_frames[idx] = frames[i];
            }
            if (!on_device) {
                HIP_TRY(ctx, hipMemcpyAsync(ctx->b->d_frames + ctx->slot_base, ctx->b->h_frames + ctx->slot_base, n * sizeof(frame_io), hipMemcpyHostToDevice, ctx->stream));
                for (int i = 0; i < n; i++) ctx->b->dev_frames[(size_t)ctx->slot_base + i] = ctx->b->h_frames[(size_t)ctx->slot_base + i];
            }
            int rc = ensure(ctx, ctx->b->d_partial, ctx->b->partial_cap, (size_t)n * upf * 6 * sizeof(float));
            if (rc) return rc;
            rc = ensure(ctx, ctx->b->d_redo, ctx->b->redo_cap, (size_t)n * upf * sizeof(uint32_t));
            if (rc) return rc;
            if (ident) {
                const size_t need = (size_t)(n > 64 ? n : 64) * sizeof(uint32_t);
                if (ctx->b->low_cap < need) {
                    rc = ensure(ctx, ctx->b->d_low, ctx->b->low_cap, need);
                    if (rc) return rc;
                    HIP_TRY(ctx, hipMemsetAsync(ctx->b->d_low, 0, need, ctx->stream));
                }
            }
            if (check) ctx->b->approx_min = ident;
            firf_args fa;
            fa.frames = ctx->b->d_frames + ctx->slot_base;
            fa.n_frames = n;
            fa.width = (uint32_t)d->width;
            fa.height = (uint32_t)d->height;
            fa.wq = wq;
            fa.n_strips = ns;
            fa.n_seg = nseg;
            fa.seg_rows = seg_rows;
            fa.units_per_frame = upf;
            fa.total_units = (uint32_t)units;
            /* In step (k_fir_fused, "In step"): every step (round 2's kernel: every second; with a fifth of the step's instructions
             * gone since, meeting every step is 0.4-1.7 % ahead, tools/firsyncbench.sh) -- unless the pictures keep sending pixels
             * to the exact tiers (each such pixel holds its wave for a microsecond, and in step all sixteen wait with it: a
             * picture with 0.02 % of its samples below the tables ran in 2.75 ms in step, 2.37 out of step; the usual picture
             * 1.74 and 1.93) */
            const int fsync = ctx->opt_fir_sync >= 0 ? ctx->opt_fir_sync : (ctx->fir_flag_share > kFirSyncMaxFlagged ? 0 : 1);
            fa.sync_mask = fsync > 0 ? (uint32_t)fsync - 1u : ~0u;
            fa.table = ctx->d_table;
            fa.table1 = ctx->d_table1;
            fa.lut16 = ctx->d_lut16;
            fa.sn = sn;
            fa.partial = ctx->b->d_partial;
            fa.redo_count = ctx->b->d_redo;
            fa.low_flag = ident ? ctx->b->d_low : nullptr;
            fa.assumed = d_assumed;
            fa.pp = pp;
            const uint32_t blocks_needed = (uint32_t)((units + 15u) / 16u);
            const int grid = (int)(blocks_needed < (uint32_t)ctx->n_cu ? blocks_needed : (uint32_t)ctx->n_cu);
            /* Rows by XCD speed.  The XCDs of a card are not equally fast on this kernel (measured: the odd ones finish
             * 11 % later on equal shares), a wave's units are fixed, and a launch ends with its slowest wave.  Strips are
             * independent, so every (frame, strip) column is cut into its segments in proportion to the speeds of the XCDs
             * its units will run on (unit u -> wave u % GW -> block / 16 -> XCD block % 8), lead-in steps included.  The
             * speeds come from the block clocks of earlier launches (ffb_update()). */
            fa.unit_rows = nullptr;
            fa.block_clock = nullptr;
            const bool full = grid == ctx->n_cu && grid % 8 == 0;
            fa.mix_xcds = grid % 8 == 0 ? 1u : 0u;
            const bool clocks = full && time_it;
            double sp[8];
            bool weigh = full && nseg >= 2 && ctx->opt_bal_mode != 1 && (ctx->opt_bal_mode == 2 || ctx->ffb_have);
            for (int x = 0; x < 8; x++)
                sp[x] = ctx->opt_bal_mode == 2 ? (((ctx->opt_bal_mask >> x) & 1u) ? ctx->opt_bal_rho : 1.0) : ctx->ffb_speed[x];
            double work[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (weigh || clocks) {
                const uint32_t gwaves = (uint32_t)grid * 16u;
                if (weigh) {
                    if (ctx->b->unit_rows_cap < units) {
                        if (ctx->b->d_unit_rows) HIP_TRY(ctx, hipFree(ctx->b->d_unit_rows));
                        if (ctx->b->h_unit_rows) HIP_TRY(ctx, hipHostFree(ctx->b->h_unit_rows));
                        ctx->b->d_unit_rows = ctx->b->h_unit_rows = nullptr;
                        ctx->b->unit_rows_cap = 0;
                        ctx->b->dev_unit_rows.clear();
                        HIP_TRY(ctx, hipMalloc((void **)&ctx->b->d_unit_rows, units * sizeof(uint32_t)));
                        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->b->h_unit_rows, units * sizeof(uint32_t), hipHostMallocDefault));
                        ctx->b->unit_rows_cap = units;
                    }
                }
                std::vector<uint32_t> rows((size_t)units);
                std::vector<int> xs;
                for (uint32_t f = 0; f < (uint32_t)n; f++)
                    for (uint32_t st = 0; st < ns; st++) {
                        double ssum = 0.0, csum = 0.0;
                        xs.resize(nseg);
                        for (uint32_t i = 0; i < nseg; i++) {
                            const uint32_t u = (f * nseg + i) * ns + st;
                            xs[i] = (int)(h2y_firf_vblock((u % gwaves) / 16u) % 8u); /* the block that works as virtual block (u % GW) / 16 */
                            ssum += weigh ? sp[xs[i]] : 1.0;
                            csum += i == 0 ? 3.0 : 6.0;
                        }
                        const double T = ((double)h2 + csum) / ssum;
                        uint32_t j0 = 0;
                        for (uint32_t i = 0; i < nseg; i++) {
                            const uint32_t u = (f * nseg + i) * ns + st;
                            uint32_t j1;
                            if (!weigh) j1 = (i + 1u) * seg_rows < h2 ? (i + 1u) * seg_rows : h2;
                            else if (i + 1u == nseg) j1 = h2;
                            else {
                                double r = (weigh ? sp[xs[i]] : 1.0) * T - (i == 0 ? 3.0 : 6.0);
                                const uint32_t left = nseg - 1u - i; /* segments after this one: eight rows each at least */
                                if (r < 8.0) r = 8.0;
                                j1 = j0 + (uint32_t)(r + 0.5);
                                if (j1 + 8u * left > h2) j1 = h2 - 8u * left;
                                if (j1 <= j0) j1 = j0 + 1u;
                            }
                            rows[u] = j0 | (j1 << 16);
                            work[xs[i]] += (double)(j1 - j0) + 3.0 + (j0 < 3u ? (double)j0 : 3.0);
                            j0 = j1;
                        }
                    }
                for (int x = 0; x < 8; x++) work[x] /= (double)(gwaves / 8u); /* steps per wave of that XCD */
                if (weigh) {
                    if (ctx->b->dev_unit_rows != rows) {
                        memcpy(ctx->b->h_unit_rows, rows.data(), units * sizeof(uint32_t));
                        HIP_TRY(ctx, hipMemcpyAsync(ctx->b->d_unit_rows, ctx->b->h_unit_rows, units * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
                        ctx->b->dev_unit_rows = rows;
                    }
                    fa.unit_rows = ctx->b->d_unit_rows;
                }
            }
            if (clocks) {
                const size_t need = (size_t)2 * grid * sizeof(unsigned long long);
                if (ctx->b->clock_cap < need) {
                    rc = ensure(ctx, ctx->b->d_clock, ctx->b->clock_cap, need);
                    if (rc) return rc;
                    HIP_TRY(ctx, hipMemsetAsync(ctx->b->d_clock, 0, need, ctx->stream)); /* k_stats_final clears the finish entries from here on */
                }
                fa.block_clock = ctx->b->d_clock;
            }
            const bool ev = time_it && ctx->b->n_ev < kMaxEvents;
            if (ev) {
                HIP_TRY(ctx, hipEventRecord(ctx->b->ev[ctx->b->n_ev][0], ctx->stream));
                ctx->last_name = "k_fir_fused";
                char buf[192];
                snprintf(buf, sizeof buf, "k_fir_fused<%s,420FIR,%s,%s%s> strips=%u segments=%u rows=%u", var.in_kind == H2Y_IN_F16 ? "F16" : "F32",
                         var.mode == H2Y_MODE_YCBCR ? "YCBCR" : "YDZDX", ident ? "PQ_IDENT" : "PQ_NORM", var.pipe == 3 ? ",LUT16" : "", ns, nseg, seg_rows);
                ctx->last_variant = buf;
            }
            HIP_TRY(ctx, h2y_launch_fir_fused(var.in_kind, var.mode, ident, var.pipe == 3 /* the 16 384-entry table applies */, grid, ctx->stream, fa));
            if (ev) {
                HIP_TRY(ctx, hipEventRecord(ctx->b->ev[ctx->b->n_ev][1], ctx->stream));
                ctx->b->n_ev++;
            }
            final_args fin;
            fin.partial = ctx->b->d_partial;
            fin.nblk = (int)upf;
            fin.redo_count = ctx->b->d_redo;
            fin.low_flag = ident ? ctx->b->d_low : nullptr;
            fin.out = ctx->b->fs_out + fstats_offset;
            fin.is_u16 = 0;
            fin.src_bit_depth = d->src_bit_depth;
            fin.check = check ? 1 : 0;
            fin.assumed = d_assumed;
            fin.publish = nullptr;
            fin.block_clock = clocks ? ctx->b->d_clock : nullptr;
            fin.grid = grid;
            fin.xcd_time = reinterpret_cast<float *>(ctx->b->fs_out + fstats_offset + n); /* the caller's copy of the statistics takes one entry more */
            HIP_TRY(ctx, h2y_launch_stats_final(n, ctx->stream, fin));
            if (clocks) {
                ctx->b->bal_slot = fstats_offset + n;
                ctx->b->ffb_pending = true;
                for (int x = 0; x < 8; x++) ctx->b->ffb_work[x] = work[x];
            }
            return 0;
        }
    }
    const geom g = make_geom(d, h2y_fused_threads(var), var.cols8 ? 8 : 4);
    const size_t npix = (size_t)d->width * d->height;
    /* one launch covers at most kMaxFramesPerLaunch frames: k_fused_t1's waves draw their tiles from one LDS counter
     * per frame of their group (H2Y_CLAIM_FRAMES of them) */
    /* (with frame groups the bound is per GROUP: a launch of g groups takes up to g x 128 frames -- sub_batch() below) */
    auto sub_batch = [&](int left) -> int {
        if (out_kind == H2Y_OUT_444TMP) return left < kFirSubBatch ? left : kFirSubBatch;
        if (left <= kMaxFramesPerLaunch || !h2y_fused_grouped(var)) return left < kMaxFramesPerLaunch ? left : kMaxFramesPerLaunch;
        const int gridf = grid_for(ctx, var, (uint64_t)g.chunks * left);
        for (int ng = groups_cap(ctx, g.tiles, gridf); ng > 1; ng >>= 1)
            if (gridf % ng == 0) {
                int cand = left < kMaxFramesPerLaunch * ng ? left : kMaxFramesPerLaunch * ng;
                cand -= cand % ng; /* whole groups; what is left over goes into the next launch */
                return cand > kMaxFramesPerLaunch ? cand : kMaxFramesPerLaunch;
            }
        return kMaxFramesPerLaunch;
    };
    /* the 4:4:4 chroma scratch of the two-pass FIR form: as many frames as a sub-batch holds, twice over when the batch
     * has more than one sub-batch (sub-batch i writes half i % 2 while the FIR pass still reads the other).  A single
     * frame (h2y_convert_frame, the CLI's ring) takes 33 MB at 4K, not the 2.1 GB of a full double sub-batch. */
    const int fir_sub = n < kFirSubBatch ? n : kFirSubBatch;
    if (out_kind == H2Y_OUT_444TMP) {
        /* earlier calls may have laid their halves out differently: nothing of theirs may still be reading */
        for (int hlf = 0; hlf < 2; hlf++)
            if (ctx->fir_used[hlf]) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_fir[hlf], 0));
        int rc = ensure(ctx, ctx->d_tmp, ctx->tmp_cap, (size_t)(n > kFirSubBatch ? 2 : 1) * fir_sub * 2 * npix * sizeof(uint16_t));
        if (rc) return rc;
    }
    int sub = 0;
    for (int f0 = 0, nf = 0; f0 < n; f0 += nf, sub++) {
        nf = sub_batch(n - f0);
        const int half = sub & 1;
        if (out_kind == H2Y_OUT_444TMP && ctx->fir_used[half]) /* scratch half still being read by an earlier FIR pass? */
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_fir[half], 0));
        /* frame descriptors: host -> device (tiny) -- unless the device already holds exactly these (a caller
         * cycling through the same buffers): one stream operation less in front of the kernel */
        bool on_device = ctx->b->dev_frames.size() == ctx->b->frames_cap;
        if (!on_device) ctx->b->dev_frames.assign(ctx->b->frames_cap, frame_io{});
        for (int i = 0; i < nf; i++) {
            frame_io io = frames[f0 + i];
            if (out_kind == H2Y_OUT_444TMP) {
                io.tmp_cb = ctx->d_tmp + ((size_t)half * fir_sub + i) * 2 * npix;
                io.tmp_cr = io.tmp_cb + npix;
            }
            const size_t idx = (size_t)ctx->slot_base + f0 + i;
            on_device = on_device && memcmp(&ctx->b->dev_frames[idx], &io, sizeof io) == 0;
            ctx->b->h_frames[idx] = io;
        }
        if (!on_device) {
            HIP_TRY(ctx, hipMemcpyAsync(ctx->b->d_frames + ctx->slot_base + f0, ctx->b->h_frames + ctx->slot_base + f0, nf * sizeof(frame_io), hipMemcpyHostToDevice,
                                        ctx->stream));
            for (int i = 0; i < nf; i++) ctx->b->dev_frames[(size_t)ctx->slot_base + f0 + i] = ctx->b->h_frames[(size_t)ctx->slot_base + f0 + i];
        }
        const int grid = grid_for(ctx, var, (uint64_t)g.chunks * nf);
        const int waves = h2y_fused_threads(var) / 64; /* the fused kernels leave one min/max record per wave */
        /* frame groups (frame_walk in h2y_kernels.hip): as many as divide both the batch and the grid, up to groups_cap() */
        int groups = 1;
        if (h2y_fused_grouped(var))
            for (int ng = groups_cap(ctx, g.tiles, grid); ng > 1; ng >>= 1)
                if (nf % ng == 0 && grid % ng == 0) {
                    groups = ng;
                    break;
                }
        if (nf / groups > kMaxFramesPerLaunch) return fail(ctx, H2Y_EINVAL, "internal: %d frames in %d groups exceed the per-group bound", nf, groups);
        int rc = ensure(ctx, ctx->b->d_partial, ctx->b->partial_cap, (size_t)nf * grid * waves * 6 * sizeof(float));
        if (rc) return rc;
        const bool t1 = var.pipe == 4 || var.pipe == 5;
        if (t1) {
            rc = ensure(ctx, ctx->b->d_redo, ctx->b->redo_cap, (size_t)nf * grid * waves * sizeof(uint32_t));
            if (rc) return rc;
        }
        const bool approx = var.pipe == 4; /* first tier, assumed floor 0 / ceiling 1: subsampled minimum */
        if (approx) {
            const size_t need = (size_t)(nf > 64 ? nf : 64) * sizeof(uint32_t);
            if (ctx->b->low_cap < need) {
                rc = ensure(ctx, ctx->b->d_low, ctx->b->low_cap, need);
                if (rc) return rc;
                HIP_TRY(ctx, hipMemsetAsync(ctx->b->d_low, 0, need, ctx->stream)); /* the kernels keep it zero from here on */
            }
        }
        if (check) ctx->b->approx_min = approx;
        /* XCD-aware rounds and their weights; the block clocks of timed launches feed balance_update() */
        const bool xcd_layout = h2y_fused_grouped(var) && grid % (8 * groups) == 0;
        /* Slices by XCD speed: block i of a group takes one contiguous run of every frame's 64-tile slices, as long as
         * the measured speed of its XCD says (block i of a group runs on XCD i % 8 under xcd_layout).  "off": the even
         * round-robin dealing of frame_walk. */
        const uint32_t *d_slice_ranges = nullptr;
        uint32_t range_stride = 0;
        bool tail_on = false;
        uint32_t tail_slices = 0;
        std::vector<double> bwork;
        double work[8] = {1, 1, 1, 1, 1, 1, 1, 1};
        if (xcd_layout && ctx->opt_bal_mode != 1) {
            const uint32_t G = (uint32_t)grid / (uint32_t)groups, nslices = (g.tiles + 63u) / 64u;
            double sp[8], mean = 0.0;
            for (int x = 0; x < 8; x++) {
                sp[x] = ctx->opt_bal_mode == 2 ? (((ctx->opt_bal_mask >> x) & 1u) ? ctx->opt_bal_rho : 1.0) : (ctx->bal_have ? ctx->bal_speed[x] : 1.0);
                mean += sp[x] / 8.0;
            }
            for (int x = 0; x < 8; x++) work[x] = sp[x] / mean;
            /* per block when this grid shape has been measured (adaptive mode), else per XCD: one table for every group */
            const bool per_block = ctx->opt_bal_mode == 0 && ctx->opt_bal_blocks && ctx->bal_bgrid == grid && ctx->bal_bgroups == groups &&
                                   (int)ctx->bal_bspeed.size() == grid && (size_t)groups * (G + 1u) <= kRangeWords;
            std::vector<uint32_t> r(per_block ? (size_t)groups * (G + 1u) : (size_t)G + 1u);
            bwork.assign((size_t)grid, 1.0);
            if (per_block) {
                std::vector<double> w(G);
                for (uint32_t gi = 0; gi < (uint32_t)groups; gi++) {
                    for (uint32_t i = 0; i < G; i++) w[i] = ctx->bal_bspeed[walk_block_of(gi, i, (uint32_t)groups)];
                    slice_ranges_w(w.data(), G, nslices, r.data() + (size_t)gi * (G + 1u)); /* h2y_walk.h */
                }
                range_stride = G + 1u;
            } else slice_ranges(sp, G, nslices, r.data());
            /* the dynamic last frame (k_fused_t1): its eight shards' boundaries ride behind the ranges */
            const int per_group = nf / groups;
            tail_on = t1 && ctx->opt_tail != 2 && nf % groups == 0 && per_group >= (ctx->opt_tail == 1 ? 2 : kTailMinFrames) && groups <= 16 &&

                      /* a block holds 64 chunks of H2Y_TAIL_CHUNK slices at most (H2Y_TAIL_QLEN): the group's G blocks must be able to take the
                       * whole frame with room to spare, however unevenly they draw (any block may end up in the common pool) */
                      (uint64_t)(nslices / H2Y_TAIL_CHUNK + 96u) * 2u <= 64ull * G && G >= 8u;
            tail_slices = nslices;
            for (uint32_t gi = 0; gi < (uint32_t)groups; gi++) {
                const uint32_t *rg = r.data() + (size_t)(per_block ? gi : 0u) * (G + 1u);
                for (uint32_t i = 0; i < G; i++) bwork[walk_block_of(gi, i, (uint32_t)groups)] = (double)(rg[i + 1] - rg[i]) * (double)G / (double)nslices;
            }
            if (!ctx->b->h_ranges) {
                HIP_TRY(ctx, hipHostMalloc((void **)&ctx->b->h_ranges, 2 * kRangeWords * sizeof(uint32_t), hipHostMallocMapped));
                HIP_TRY(ctx, hipHostGetDevicePointer((void **)&ctx->b->hd_ranges, ctx->b->h_ranges, 0));
                HIP_TRY(ctx, hipHostMalloc((void **)&ctx->b->h_btime, 1024 * sizeof(float), hipHostMallocMapped));
                HIP_TRY(ctx, hipHostGetDevicePointer((void **)&ctx->b->hd_btime, ctx->b->h_btime, 0));
            }
            if (r.size() > kRangeWords) return fail(ctx, H2Y_EINVAL, "internal: %zu slice ranges", r.size());
            /* The kernels read these tables IN PLACE from mapped pinned memory: a slot may only be rewritten once every launch that
             * reads it has finished.  A batch_state is handed out again only after its batch was finished (enqueue / finish,
             * h2y_convert_frame), so both slots are free at a batch's first launch -- except on the stream pipeline, which calls
             * run_frames() back to back on one state without synchronising: there the slots stay busy until the third-table
             * path below has waited for the stream. */
            if (sub == 0 && !ctx->streaming) ctx->b->slot_busy[0] = ctx->b->slot_busy[1] = false;
            int slot = -1;
            for (int k = 0; k < 2 && slot < 0; k++)
                if (ctx->b->range_slot[k] == r) slot = k;
            if (slot < 0) {
                for (int k = 0; k < 2 && slot < 0; k++)
                    if (!ctx->b->slot_busy[k]) slot = k;
                if (slot < 0) { /* a third table within one batch: wait for the launches that read the other two */
                    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                    ctx->b->slot_busy[0] = ctx->b->slot_busy[1] = false;
                    slot = 0;
                }
                memcpy(ctx->b->h_ranges + (size_t)slot * kRangeWords, r.data(), r.size() * sizeof(uint32_t));
                ctx->b->range_slot[slot] = r;
            }
            ctx->b->slot_busy[slot] = true;
            d_slice_ranges = ctx->b->hd_ranges + (size_t)slot * kRangeWords;
        }
        const bool clocks = xcd_layout && time_it;
        if (clocks) {
            const size_t need = (size_t)2 * grid * sizeof(unsigned long long);
            if (ctx->b->clock_cap < need) {
                rc = ensure(ctx, ctx->b->d_clock, ctx->b->clock_cap, need);
                if (rc) return rc;
                HIP_TRY(ctx, hipMemsetAsync(ctx->b->d_clock, 0, need, ctx->stream)); /* k_stats_final clears the finish entries from here on */
            }
        }
        fused_args a;
        a.xcd_layout = xcd_layout ? 1u : 0u;
        a.block_clock = clocks ? ctx->b->d_clock : nullptr;
        a.slice_ranges = d_slice_ranges;
        a.range_stride = range_stride;
        a.tail_ctr = nullptr;
        a.tail_slices = 0;
        if (tail_on && d_slice_ranges) {
            if (!ctx->b->d_tail) {
                HIP_TRY(ctx, hipMalloc((void **)&ctx->b->d_tail, 16 * H2Y_TAIL_WORDS * sizeof(uint32_t)));
                HIP_TRY(ctx, hipMemsetAsync(ctx->b->d_tail, 0, 16 * H2Y_TAIL_WORDS * sizeof(uint32_t), ctx->stream)); /* k_stats_final clears it from here on */
            }
            a.tail_ctr = ctx->b->d_tail;
            a.tail_slices = tail_slices;
        }
        a.redo_count = t1 ? ctx->b->d_redo : nullptr;
        a.low_flag = approx ? ctx->b->d_low : nullptr;
        a.frames = ctx->b->d_frames + ctx->slot_base + f0;
        a.n_frames = nf;
        a.width = d->width;
        a.height = d->height;
        a.wq = g.wq;
        a.wq_magic = g.wq_magic;
        a.tiles_per_frame = g.tiles;
        a.chunks_per_frame = g.chunks;
        a.groups = (uint32_t)groups;
        a.table = ctx->d_table;
        a.table_src = a.table_dst = nullptr;
        a.lut16 = ctx->d_lut16;
        a.table1 = ctx->d_table1;
        a.sn = sn;
        a.partial = ctx->b->d_partial;
        a.assumed = d_assumed;
        a.pp = pp;
        if (pp.convert_transfer == 2 && !var.narrow && (var.pipe == 0 || var.pipe == 7)) {
            /* generic transfer pair through the table tier: source function, then destination function */
            static const int kSrcFn[4] = {H2Y_TFN_NONE, H2Y_TFN_PQ_F, H2Y_TFN_RHO_H, H2Y_TFN_G24};    /* by H2Y_TF_* class */
            static const int kDstFn[4] = {H2Y_TFN_NONE, H2Y_TFN_PQ_R, H2Y_TFN_RHO_R, H2Y_TFN_G24INV};
            const int sf = kSrcFn[pp.src_tf], df = kDstFn[pp.dst_tf];
            int rc2 = ensure_tfn(ctx, sf);
            if (!rc2) rc2 = ensure_tfn(ctx, df);
            if (rc2) return rc2;
            a.pp.src_fn = sf;
            a.pp.dst_fn = df;
            a.table_src = sf ? ctx->d_tfn[sf] : nullptr;
            a.table_dst = df ? ctx->d_tfn[df] : nullptr;
            a.pp.tf_ext[0] = sf ? ctx->d_tfn_ext[sf] : nullptr;
            a.pp.tf_ext[1] = df ? ctx->d_tfn_ext[df] : nullptr;
        }
        a.tiles_magic = g.tiles > 1 ? (uint32_t)(0x100000000ull / g.tiles) : 0xFFFFFFFFu;
        const bool ev = time_it && ctx->b->n_ev < kMaxEvents;
        if (ev) {
            HIP_TRY(ctx, hipEventRecord(ctx->b->ev[ctx->b->n_ev][0], ctx->stream));
            ctx->last_name = h2y_fused_name(var);
            static const char *const kIn[] = {"F32", "F16", "U16"}, *const kOut[] = {"420BOX", "444", "444TMP"};
            static const char *const kPipe[] = {"RUNTIME", "PQ_IDENT", "PQ_NORM", "LUT16", "PQ_IDENT", "PQ_NORM", "NONE", "TFN"};
            const char *mode = var.mode == H2Y_MODE_YCBCR ? "YCBCR" : var.mode == H2Y_MODE_YDZDX ? "YDZDX" : var.mode == H2Y_MODE_IDENTITY ? "IDENTITY" : "YPQRS";
            char buf[192];
            snprintf(buf, sizeof buf, "%s<%s,%s,%s,%s%s>%s groups=%d xcd=%d%s", ctx->last_name, kIn[var.in_kind], kOut[var.out_kind], mode,
                     kPipe[var.pipe], var.cols8 ? ",COLS8" : "", out_kind == H2Y_OUT_444TMP ? "+k_fir420" : "", groups, xcd_layout ? 1 : 0,
                     a.tail_ctr ? " tail=1" : "");
            ctx->last_variant = buf;
        }
        HIP_TRY(ctx, h2y_launch_fused(var, grid, ctx->stream, a));
        if (ev) {
            HIP_TRY(ctx, hipEventRecord(ctx->b->ev[ctx->b->n_ev][1], ctx->stream));
            ctx->b->n_ev++;
        }
        final_args fa;
        fa.partial = ctx->b->d_partial;
        fa.nblk = grid / groups * waves;
        fa.redo_count = t1 ? ctx->b->d_redo : nullptr;
        fa.low_flag = approx ? ctx->b->d_low : nullptr;
        fa.out = ctx->b->fs_out + fstats_offset + f0;
        fa.is_u16 = d->in_sample_type == H2Y_SAMPLE_U16;
        fa.src_bit_depth = d->src_bit_depth;
        fa.check = check ? 1 : 0;
        fa.assumed = d_assumed;
        fa.publish = nullptr;
        fa.block_clock = clocks ? ctx->b->d_clock : nullptr;
        fa.grid = grid;
        static_assert(sizeof(frame_stats) >= 8 * sizeof(float), "the XCD run times ride in one frame_stats entry");
        fa.xcd_time = reinterpret_cast<float *>(ctx->b->fs_out + fstats_offset + n); /* the caller's copy of the statistics takes one entry more */
        fa.tail_ctr = a.tail_ctr;
        fa.tail_n = groups * (int)H2Y_TAIL_WORDS;
        fa.block_time = clocks && d_slice_ranges && ctx->b->fs_out == ctx->b->m_fstats ? ctx->b->hd_btime : nullptr; /* (enqueued batches: what the host reads in h2y_batch_finish) */
        HIP_TRY(ctx, h2y_launch_stats_final(nf, ctx->stream, fa));
        if (clocks) {
            ctx->b->bal_slot = fstats_offset + n;
            ctx->b->bal_pending = true;
            for (int x = 0; x < 8; x++) ctx->b->bal_work[x] = work[x];
            ctx->b->bal_grid = fa.block_time ? grid : 0;
            ctx->b->bal_groups = groups;
            ctx->b->bal_bwork = bwork;
        }
        if (out_kind == H2Y_OUT_444TMP) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_fused[half], ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->fir_stream, ctx->ev_fused[half], 0));
            fir_args fr;
            fr.frames = ctx->b->d_frames + ctx->slot_base + f0;
            fr.n_frames = nf;
            fr.src_cb = fr.src_cr = nullptr;
            fr.dst_cb = fr.dst_cr = nullptr;
            fr.width = d->width;
            fr.height = d->height;
            fr.fir_max = pp.fir_max;
            fr.apply_yuv_clamp = 1;
            fr.pp = pp;
            HIP_TRY(ctx, h2y_launch_fir420(ctx->fir_stream, fr));
            HIP_TRY(ctx, hipEventRecord(ctx->ev_fir[half], ctx->fir_stream));
            ctx->fir_used[half] = true;
        }
    }
    if (out_kind == H2Y_OUT_444TMP) /* everything queued after this call on the main stream sees finished chroma */
        for (int hlf = 0; hlf < 2; hlf++)
            if (ctx->fir_used[hlf]) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_fir[hlf], 0));
    return 0;
}

int reserve_batch(h2y_ctx *ctx, int n)
{
    if ((size_t)n > ctx->b->frames_cap) {
        if (ctx->b->d_frames) HIP_TRY(ctx, hipFree(ctx->b->d_frames));
        if (ctx->b->h_frames) HIP_TRY(ctx, hipHostFree(ctx->b->h_frames));
        if (ctx->b->d_fstats) HIP_TRY(ctx, hipFree(ctx->b->d_fstats));
        if (ctx->b->h_fstats) HIP_TRY(ctx, hipHostFree(ctx->b->h_fstats));
        ctx->b->d_frames = nullptr; ctx->b->h_frames = nullptr; ctx->b->d_fstats = nullptr; ctx->b->h_fstats = nullptr;
        ctx->b->frames_cap = 0;
        ctx->b->dev_frames.clear();
        size_t cap = (size_t)n < 64 ? 64 : (size_t)n;
        HIP_TRY(ctx, hipMalloc((void **)&ctx->b->d_frames, cap * sizeof(frame_io)));
        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->b->h_frames, cap * sizeof(frame_io), hipHostMallocDefault));
        /* +1: slot for the stats pre-pass / redo */
        HIP_TRY(ctx, hipMalloc((void **)&ctx->b->d_fstats, (cap + 1) * sizeof(frame_stats)));
        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->b->h_fstats, (cap + 1) * sizeof(frame_stats), hipHostMallocDefault));
        HIP_TRY(ctx, hipHostGetDevicePointer((void **)&ctx->b->m_fstats, ctx->b->h_fstats, 0));
        ctx->b->fs_out = ctx->b->d_fstats;
        ctx->b->frames_cap = cap;
    }
    return 0;
}

/* pic_stats() of one frame on the device; result lands in d_fstats[slot] and,
 * when publish != NULL, as the assumption for later kernels -- no host sync. */
int run_stats(h2y_ctx *ctx, const h2y_desc *d, const void *const in[3], int slot, assumed_stats *publish)
{
    const size_t npix = (size_t)d->width * d->height;
    int grid = ctx->n_cu * 4;
    size_t need_blocks = (npix / 4 + H2Y_FUSED_THREADS - 1) / H2Y_FUSED_THREADS;
    if ((size_t)grid > need_blocks) grid = need_blocks ? (int)need_blocks : 1;
    int rc = ensure(ctx, ctx->b->d_partial, ctx->b->partial_cap, (size_t)grid * 6 * sizeof(float));
    if (rc) return rc;
    stats_args sa;
    bool aligned = true;
    for (int c = 0; c < 3; c++) {
        sa.in[c] = in[c];
        if (((uintptr_t)in[c]) & 15) aligned = false;
    }
    sa.npix = npix;
    sa.vec_ok = aligned ? 1 : 0;
    sa.partial = ctx->b->d_partial;
    HIP_TRY(ctx, h2y_launch_stats(in_kind_of(d), grid, ctx->stream, sa));
    final_args fa;
    fa.partial = ctx->b->d_partial;
    fa.nblk = grid;
    fa.out = ctx->b->d_fstats + slot;
    fa.is_u16 = d->in_sample_type == H2Y_SAMPLE_U16;
    fa.src_bit_depth = d->src_bit_depth;
    fa.redo_count = nullptr;
    fa.low_flag = nullptr;
    fa.check = 0;
    fa.assumed = nullptr;
    fa.publish = publish;
    fa.block_clock = nullptr;
    fa.grid = 0;
    fa.xcd_time = nullptr;
    HIP_TRY(ctx, h2y_launch_stats_final(1, ctx->stream, fa));
    return 0;
}

} // namespace

extern "C" {

int h2y_abi_version(void)
{
#ifdef H2Y_EXPERIMENT
    return H2Y_ABI_VERSION | H2Y_ABI_EXPERIMENT;
#else
    return H2Y_ABI_VERSION;
#endif
}

int h2y_desc_check(const h2y_desc *d, const char **why)
{
    const char *w = nullptr;
    int rc = H2Y_OK;
#define BAD(code, msg) do { rc = (code); w = (msg); goto done; } while (0)
    if (!d) BAD(H2Y_EINVAL, "null descriptor");
    if (d->width < 1 || d->width > 16384 || d->height < 1 || d->height > 16384) BAD(H2Y_EINVAL, "picture dimensions out of bounds");
    if (d->in_sample_type != H2Y_SAMPLE_F32 && d->in_sample_type != H2Y_SAMPLE_F16 && d->in_sample_type != H2Y_SAMPLE_U16)
        BAD(H2Y_EINVAL, "in_sample_type not recognized"); /* common.cpp:229-233 */
    if (d->dst_bit_depth < 8 || d->dst_bit_depth > 16) BAD(H2Y_EINVAL, "dst_bit_depth must be 8..16");
    if (d->in_sample_type == H2Y_SAMPLE_U16) {
        if (d->src_bit_depth < 8 || d->src_bit_depth > 16) BAD(H2Y_EINVAL, "src_bit_depth must be 8..16 for U16 input");
        if (d->dst_bit_depth > d->src_bit_depth) BAD(H2Y_EINVAL, "dst bitdepth > src bitdepth"); /* tiff.cpp:396-401 */
    }
    if (d->dst_chroma_format_idc != H2Y_CHROMA_420 && d->dst_chroma_format_idc != H2Y_CHROMA_444)
        BAD(H2Y_EUNSUPPORTED, "dst_chroma_format_idc must be 1 (4:2:0) or 3 (4:4:4)");
    if (d->dst_chroma_format_idc == H2Y_CHROMA_420) {
        if ((d->width & 1) || (d->height & 1)) BAD(H2Y_EINVAL, "4:2:0 needs even width and height");
        if (d->chroma_resampler_type == 0 && ((d->width & 3) || (d->height & 3)))
            BAD(H2Y_EINVAL, "box resampler reads 4x4 tiles: width and height must be multiples of 4"); /* convert.cpp:100-140 */
    }
    if (d->src_transfer != d->dst_transfer) {
        if (tf_class(d->src_transfer) < 0) BAD(H2Y_EUNSUPPORTED, "src_transfer_characteristics not supported (yet)");  /* convert.cpp:1061 */
        if (tf_class(d->dst_transfer) < 0) BAD(H2Y_EUNSUPPORTED, "dst_transfer_characteristics not supported (yet)");  /* convert.cpp:1107 */
    }
    if (!(d->dst_matrix == d->src_matrix && d->dst_primaries == d->src_primaries)) {
        switch (d->dst_matrix) {
        case H2Y_MATRIX_YDZDX: case H2Y_MATRIX_BT2020NC: case H2Y_MATRIX_BT709:
        case H2Y_MATRIX_YDZDX_Y100: case H2Y_MATRIX_YDZDX_Y500: break;
        default: BAD(H2Y_EUNSUPPORTED, "can't determine color difference to use"); /* convert.cpp:1195-1197 */
        }
    }
    if (d->stats_override)
        for (int c = 0; c < 3; c++)
            if (d->src_transfer != d->dst_transfer && d->ceiling[c] == d->floor[c])
                BAD(H2Y_EINVAL, "stats override with ceiling == floor (division by zero range)");
done:
#undef BAD
    if (why) *why = w ? w : "ok";
    return rc;
}

size_t h2y_frame_bytes(const h2y_desc *d)
{
    if (!d || d->width < 1 || d->height < 1) return 0;
    size_t n = (size_t)d->width * d->height;
    size_t nc = d->dst_chroma_format_idc == H2Y_CHROMA_420 ? (size_t)(d->width >> 1) * (d->height >> 1) : n;
    return (n + 2 * nc) * sizeof(uint16_t);
}

size_t h2y_plane_bytes(const h2y_desc *d)
{
    if (!d || d->width < 1 || d->height < 1) return 0;
    return (size_t)d->width * d->height * sample_bytes(d);
}

const char *h2y_last_error(const h2y_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

/* everything h2y_ctx_create() allocates; on a failure the caller destroys the half-built context */
static int ctx_init(h2y_ctx *ctx, int device)
{
    ctx->device = device;
    for (batch_state &b : ctx->bs)
        for (int i = 0; i < kMaxEvents; i++) b.ev[i][0] = b.ev[i][1] = nullptr;
    HIP_TRY(ctx, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(ctx, hipGetDeviceProperties(&prop, device));
    ctx->n_cu = prop.multiProcessorCount;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->fir_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fused[i], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fir[i], hipEventDisableTiming));
    }
    for (batch_state &b : ctx->bs) {
        for (int i = 0; i < kMaxEvents; i++) {
            HIP_TRY(ctx, hipEventCreate(&b.ev[i][0]));
            HIP_TRY(ctx, hipEventCreate(&b.ev[i][1]));
        }
        HIP_TRY(ctx, hipEventCreateWithFlags(&b.ev_done, hipEventDisableTiming));
    }
    /* PQ fast-tier table: built on the host once, lives in HBM, staged to LDS per block */
    {
        std::vector<pq_recA> A(H2Y_PQ_NREC);
        std::vector<pq_recB> B(H2Y_PQ_NREC);
        pq_build_table(A.data(), B.data());
        HIP_TRY(ctx, hipMalloc(&ctx->d_table, H2Y_PQ_TABLE_BYTES));
        char *t = static_cast<char *>(ctx->d_table);
        HIP_TRY(ctx, hipMemcpy(t, A.data(), H2Y_PQ_NREC * 16, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(t + H2Y_PQ_NREC * 16, B.data(), H2Y_PQ_NREC * 16, hipMemcpyHostToDevice));
        {
            std::vector<pq_ext_rec> X(H2Y_PQX_NSEG);
            pq_build_table_ext(X.data());
            HIP_TRY(ctx, hipMalloc(&ctx->d_table_ext, H2Y_PQX_TABLE_BYTES));
            HIP_TRY(ctx, hipMemcpy(ctx->d_table_ext, X.data(), H2Y_PQX_TABLE_BYTES, hipMemcpyHostToDevice));
        }
        std::vector<pq_rec1> T1(H2Y_T1_NREC);
        pq_build_table1(T1.data());
        HIP_TRY(ctx, hipMalloc(&ctx->d_table1, H2Y_T1_NREC * sizeof(pq_rec1)));
        HIP_TRY(ctx, hipMemcpy(ctx->d_table1, T1.data(), H2Y_T1_NREC * sizeof(pq_rec1), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_lut16, H2Y_LUT16_N * sizeof(float)));
        HIP_TRY(ctx, h2y_launch_build_lut16(ctx->stream, ctx->d_table, ctx->d_lut16));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    for (batch_state &b : ctx->bs) {
        HIP_TRY(ctx, hipMalloc((void **)&b.d_assumed, 2 * sizeof(assumed_stats)));
        HIP_TRY(ctx, hipHostMalloc((void **)&b.h_assumed, 2 * sizeof(assumed_stats), hipHostMallocDefault));
        ctx->b = &b;
        const int rc = reserve_batch(ctx, 64);
        if (rc) return rc;
    }
    ctx->b = &ctx->bs[0];
    return H2Y_OK;
}

int h2y_ctx_create(int device, h2y_ctx **out)
{
    if (!out) return fail(nullptr, H2Y_EINVAL, "null out pointer");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(nullptr, H2Y_EHIP, "no HIP device (%s): this library has no CPU path", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, H2Y_EINVAL, "device %d out of range (have %d)", device, ndev);
    h2y_ctx *ctx = new (std::nothrow) h2y_ctx();
    if (!ctx) return fail(nullptr, H2Y_ENOMEM, "out of host memory");
    const int rc = ctx_init(ctx, device);
    if (rc) { /* g_err holds the reason (h2y_last_error(NULL)); nothing of the half-built context survives */
        h2y_ctx_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return H2Y_OK;
}

/* Tuning and test knobs, per context.  Nothing in this library reads the environment.
 *   "t1"      "0" | "1"                 binary32 first tier off / on (default on)
 *   "groups"  "0" | "1" .. "64"         at most this many frame groups (rounded down to a power of two; 1 = off); "0": by the frame's size (default)
 *   "cols8"   "0" | "1"                 8-column thread tiles for half input (default on)
 *   "balance" "adaptive" | "xcd" | "off" | "<xcd mask>,<ratio>"   slices by measured block speed / XCD speed only / even / fixed XCD weights (default adaptive)
 *   "tail"    "off" | "auto" | "on"     k_fused_t1: the last frame of every frame group drawn dynamically by the blocks that finish first
 *                                       (auto: groups of eight frames or more; on: two suffice; default off: measured neutral)
 *   "fir"     "auto" | "twopass" | "fused"   how the FIR resampler runs (default auto)
 *   "firsync" "0" | "1" .. "1024"       k_fir_fused: the waves of a block meet at a barrier every so many steps (power of two; 0 = never; default "auto": 1, or 0 while many pixels go to the exact tiers) */
int h2y_ctx_set_option(h2y_ctx *ctx, const char *name, const char *value)
{
    if (!ctx) return fail(nullptr, H2Y_EINVAL, "null ctx");
    if (!name || !value) return fail(ctx, H2Y_EINVAL, "null option name or value");
    if ((ctx->q_count > 0) || ctx->streaming) return fail(ctx, H2Y_EINVAL, "a batch is pending or a stream is open");
    if (!strcmp(name, "t1")) {
        ctx->opt_t1 = value[0] != '0';
        ctx->opt_t1_steer = strcmp(value, "always") != 0; /* "always": stay on the first tier however many pixels it passes on (timing) */
    }
    else if (!strcmp(name, "cols8")) ctx->opt_cols8 = value[0] != '0';
    else if (!strcmp(name, "firsync")) {
        int v = atoi(value), p = 1;
        if (!strcmp(value, "auto")) v = -1;
        else if (v < 0) return fail(ctx, H2Y_EINVAL, "firsync must be >= 0 or \"auto\"");
        while (2 * p <= v && p < 1024) p *= 2;
        ctx->opt_fir_sync = v > 0 ? p : v;
    } else if (!strcmp(name, "groups")) {
        int v = atoi(value), p = 1;
        if (v < 0) return fail(ctx, H2Y_EINVAL, "groups must be >= 0");
        while (2 * p <= v && p < 64) p *= 2;
        ctx->opt_groups = v ? p : 0;
    } else if (!strcmp(name, "balance")) {
        if (!strcmp(value, "adaptive")) { ctx->opt_bal_mode = 0; ctx->opt_bal_blocks = true; }
        else if (!strcmp(value, "xcd")) { ctx->opt_bal_mode = 0; ctx->opt_bal_blocks = false; } /* round 2's form: A/B timing */
        else if (!strcmp(value, "off")) ctx->opt_bal_mode = 1;
        else {
            char *end = nullptr;
            const unsigned long mm = strtoul(value, &end, 0);
            const double r = (end && *end == ',') ? atof(end + 1) : 0.0;
            if (!(mm & 0xFFu) || (mm & 0xFFu) == 0xFFu || !(r > 1.0)) return fail(ctx, H2Y_EINVAL, "balance: want adaptive, xcd, off or <mask>,<ratio > 1>");
            ctx->opt_bal_mode = 2;
            ctx->opt_bal_mask = (uint32_t)(mm & 0xFFu);
            ctx->opt_bal_rho = r;
        }
    } else if (!strcmp(name, "tail")) {
        if (!strcmp(value, "auto")) ctx->opt_tail = 0;
        else if (!strcmp(value, "on")) ctx->opt_tail = 1;
        else if (!strcmp(value, "off")) ctx->opt_tail = 2;
        else return fail(ctx, H2Y_EINVAL, "tail: want auto, on or off");
    } else if (!strcmp(name, "fir")) {
        if (!strcmp(value, "auto")) ctx->opt_fir = 0;
        else if (!strcmp(value, "twopass")) ctx->opt_fir = 1;
        else if (!strcmp(value, "fused")) ctx->opt_fir = 2;
        else return fail(ctx, H2Y_EINVAL, "fir: want auto, twopass or fused");
    } else return fail(ctx, H2Y_EINVAL, "unknown option '%s'", name);
    return H2Y_OK;
}

void h2y_ctx_destroy(h2y_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->streaming) (void)h2y_stream_close(ctx);
    for (batch_state &b : ctx->bs) {
        for (int i = 0; i < kMaxEvents; i++) {
            if (b.ev[i][0]) (void)hipEventDestroy(b.ev[i][0]);
            if (b.ev[i][1]) (void)hipEventDestroy(b.ev[i][1]);
        }
        if (b.ev_done) (void)hipEventDestroy(b.ev_done);
        (void)hipFree(b.d_frames);
        (void)hipHostFree(b.h_frames);
        (void)hipFree(b.d_partial);
        (void)hipFree(b.d_redo);
        (void)hipFree(b.d_low);
        (void)hipFree(b.d_clock);
        (void)hipHostFree(b.h_ranges);
        (void)hipHostFree(b.h_btime);
        (void)hipFree(b.d_tail);
        (void)hipFree(b.d_unit_rows);
        (void)hipHostFree(b.h_unit_rows);
        (void)hipFree(b.d_fstats);
        (void)hipHostFree(b.h_fstats);
        (void)hipFree(b.d_assumed);
        (void)hipHostFree(b.h_assumed);
    }
    (void)hipFree(ctx->d_table);
    for (void *t : ctx->d_tfn) (void)hipFree(t);
    for (int fn = 0; fn < H2Y_TFN_COUNT; fn++)
        if (ctx->d_tfn_ext[fn] && ctx->d_tfn_ext[fn] != ctx->d_table_ext) (void)hipFree(ctx->d_tfn_ext[fn]);
    (void)hipFree(ctx->d_lut16);
    (void)hipFree(ctx->d_table1);
    (void)hipFree(ctx->d_table_ext);
    (void)hipFree(ctx->d_tmp);
    (void)hipFree(ctx->d_in);
    (void)hipFree(ctx->d_out);
    if (ctx->fir_stream) {
        (void)hipStreamSynchronize(ctx->fir_stream);
        (void)hipStreamDestroy(ctx->fir_stream);
    }
    for (int i = 0; i < 2; i++) {
        if (ctx->ev_fused[i]) (void)hipEventDestroy(ctx->ev_fused[i]);
        if (ctx->ev_fir[i]) (void)hipEventDestroy(ctx->ev_fir[i]);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int h2y_ctx_set_stream(h2y_ctx *ctx, void *hip_stream)
{
    if (!ctx) return fail(nullptr, H2Y_EINVAL, "null ctx");
    if ((ctx->q_count > 0)) return fail(ctx, H2Y_EINVAL, "a batch is pending: call h2y_batch_finish first");
    if (ctx->streaming) return fail(ctx, H2Y_EINVAL, "a stream is open: close it first");
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return H2Y_OK;
}

int h2y_convert_batch_enqueue(h2y_ctx *ctx, const h2y_desc *d, int n_frames, const void *const *d_in, uint16_t *const *d_out)
{
    if (!ctx) return fail(nullptr, H2Y_EINVAL, "null ctx");
    if (ctx->q_count >= 2) return fail(ctx, H2Y_EINVAL, "two batches are already in flight: call h2y_batch_finish first");
    if (ctx->streaming) return fail(ctx, H2Y_EINVAL, "a stream is open: close it first");
    const char *why;
    int rc = h2y_desc_check(d, &why);
    if (rc) return fail(ctx, rc, "descriptor: %s", why);
    if (n_frames < 1 || !d_in || !d_out) return fail(ctx, H2Y_EINVAL, "n_frames < 1 or null pointer arrays");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->b = &ctx->bs[(ctx->q_head + ctx->q_count) & 1]; /* the free slot: its last batch was finished */
    rc = reserve_batch(ctx, n_frames);
    if (rc) return rc;
    ctx->b->p_frames.resize(n_frames);
    for (int f = 0; f < n_frames; f++) {
        frame_io io;
        for (int c = 0; c < 3; c++) {
            io.in[c] = d_in[f * 3 + c];
            if (!io.in[c] || ((uintptr_t)io.in[c] & 15)) return fail(ctx, H2Y_EINVAL, "input plane %d of frame %d is null or not 16-byte aligned", c, f);
        }
        io.out = d_out[f];
        if (!io.out || ((uintptr_t)io.out & 15)) return fail(ctx, H2Y_EINVAL, "output of frame %d is null or not 16-byte aligned", f);
        io.tmp_cb = io.tmp_cr = nullptr;
        ctx->b->p_frames[f] = io;
    }
    ctx->b->n_ev = 0;
    const bool needs_stats = d->src_transfer != d->dst_transfer; /* convert.cpp:930-940: stats are only read then */
    bool check = false;
    bool host_knows = true;
    assumed_stats *as = ctx->b->h_assumed;
    if (!needs_stats || d->stats_override) {
        assumed_stats want;
        for (int c = 0; c < 3; c++) {
            want.floor_[c] = d->stats_override ? d->floor[c] : 0;
            want.ceil_[c] = d->stats_override ? d->ceiling[c] : 1;
        }
        if (!ctx->b->dev_assumed_ok || memcmp(&want, &ctx->b->dev_assumed, sizeof want) != 0) {
            *as = want;
            HIP_TRY(ctx, hipMemcpyAsync(ctx->b->d_assumed, as, sizeof *as, hipMemcpyHostToDevice, ctx->stream));
            ctx->b->dev_assumed = want;
            ctx->b->dev_assumed_ok = true;
        } else *as = want;
    } else if (ctx->have_hint && ctx->hint_kind == d->in_sample_type) {
        /* assume this batch looks like the last frame we saw; verified below */
        assumed_stats want;
        for (int c = 0; c < 3; c++) {
            want.floor_[c] = ctx->hint_floor[c];
            want.ceil_[c] = ctx->hint_ceil[c];
        }
        if (!ctx->b->dev_assumed_ok || memcmp(&want, &ctx->b->dev_assumed, sizeof want) != 0) {
            *as = want;
            HIP_TRY(ctx, hipMemcpyAsync(ctx->b->d_assumed, as, sizeof *as, hipMemcpyHostToDevice, ctx->stream));
            ctx->b->dev_assumed = want;
            ctx->b->dev_assumed_ok = true;
        } else *as = want; /* (the host copy is what pick_variant() reads) */
        check = true;
    } else {
        /* no history: measure frame 0 (pic_stats pre-pass) and assume the rest match it */
        ctx->b->dev_assumed_ok = false;
        rc = run_stats(ctx, d, ctx->b->p_frames[0].in, (int)ctx->b->frames_cap, ctx->b->d_assumed);
        if (rc) return rc;
        check = true;
        host_knows = false; /* the values exist only in device memory */
    }
    t1_begin_batch(ctx);
    ctx->b->fs_out = ctx->b->m_fstats; /* the statistics (+ the XCD run times) go straight to pinned host memory */
    rc = run_frames(ctx, d, ctx->b->p_frames.data(), n_frames, ctx->b->d_assumed, host_knows ? ctx->b->h_assumed : nullptr, check, 0, true);
    ctx->b->fs_out = ctx->b->d_fstats;
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->b->ev_done, ctx->stream));
    ctx->b->p_desc = *d;
    ctx->b->p_n = n_frames;
    ctx->b->p_check = check;
    ctx->q_count++;
    return H2Y_OK;
}

int h2y_batch_finish(h2y_ctx *ctx, int *n_redone)
{
    if (n_redone) *n_redone = 0;
    if (!ctx) return fail(nullptr, H2Y_EINVAL, "null ctx");
    if (ctx->q_count == 0) return H2Y_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->b = &ctx->bs[ctx->q_head]; /* the oldest batch in flight; a younger one may still be running behind it */
    ctx->q_head ^= 1;
    ctx->q_count--;
    HIP_TRY(ctx, hipEventSynchronize(ctx->b->ev_done));
    float ms = 0.f;
    for (int i = 0; i < ctx->b->n_ev; i++) {
        float t = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&t, ctx->b->ev[i][0], ctx->b->ev[i][1]));
        ms += t;
    }
    ctx->last_ms = ms;
    ctx->last_launches = ctx->b->n_ev;
#ifdef H2Y_BLOCK_TIMES
    if (const char *e = getenv("H2Y_BLOCK_TIMES_FILE")) { /* one file per finished batch: <name>.<n> */
        static int n_dump = 0;
        char fn[512];
        snprintf(fn, sizeof fn, "%s.%d", e, n_dump++);
        if (!strcmp(ctx->last_name, "k_fir_fused")) h2y_dump_ff_block_times(fn);
        else h2y_dump_block_times(fn);
    }
#endif
    int redone = 0;
    const h2y_desc *d = &ctx->b->p_desc;
    t1_end_batch(ctx, d, ctx->b->h_fstats, ctx->b->p_n);
    balance_update(ctx);
    ffb_update(ctx);
    if (ctx->b->p_check) {
        for (int f = 0; f < ctx->b->p_n; f++) {
            if (!ctx->b->h_fstats[f].mismatch) continue;
            if (ctx->b->approx_min) {
                /* the kernel kept only a subsample of the minimum: what it measured is exact where it matched the
                 * assumption, not here -- take pic_stats() of this frame first, then the pixels, as
                 * h2y_convert_frame() does */
                int rc = run_stats(ctx, d, ctx->b->p_frames[f].in, (int)ctx->b->frames_cap, ctx->b->d_assumed + 1);
                if (rc) return rc;
                HIP_TRY(ctx, hipMemcpyAsync(ctx->b->h_fstats + f, ctx->b->d_fstats + ctx->b->frames_cap, sizeof(frame_stats), hipMemcpyDeviceToHost, ctx->stream));
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); /* before run_frames() reuses the slot */
                rc = run_frames(ctx, d, &ctx->b->p_frames[f], 1, ctx->b->d_assumed + 1, nullptr, false, (int)ctx->b->frames_cap, false);
                if (rc) return rc;
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                redone++;
                continue;
            }
            /* the assumption was wrong for this frame: its true floor/ceiling are now
             * known (the fused kernel measured them), so run it again with those */
            assumed_stats *as = ctx->b->h_assumed + 1;
            for (int c = 0; c < 3; c++) {
                as->floor_[c] = ctx->b->h_fstats[f].floor_[c];
                as->ceil_[c] = ctx->b->h_fstats[f].ceil_[c];
            }
            HIP_TRY(ctx, hipMemcpyAsync(ctx->b->d_assumed + 1, as, sizeof *as, hipMemcpyHostToDevice, ctx->stream));
            int rc = run_frames(ctx, d, &ctx->b->p_frames[f], 1, ctx->b->d_assumed + 1, as, false, (int)ctx->b->frames_cap, false);
            if (rc) return rc;
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            redone++;
        }
    }
    if (d->src_transfer != d->dst_transfer && !d->stats_override) {
        const frame_stats &last = ctx->b->h_fstats[ctx->b->p_n - 1];
        for (int c = 0; c < 3; c++) {
            ctx->hint_floor[c] = last.floor_[c];
            ctx->hint_ceil[c] = last.ceil_[c];
        }
        ctx->have_hint = true;
        ctx->hint_kind = d->in_sample_type;
    }
    if (n_redone) *n_redone = redone;
    return H2Y_OK;
}

int h2y_convert_batch(h2y_ctx *ctx, const h2y_desc *d, int n_frames, const void *const *d_in, uint16_t *const *d_out)
{
    int rc = h2y_convert_batch_enqueue(ctx, d, n_frames, d_in, d_out);
    if (rc) return rc;
    while (ctx->q_count > 0) { /* this batch and any enqueued before it */
        rc = h2y_batch_finish(ctx, nullptr);
        if (rc) return rc;
    }
    return H2Y_OK;
}

int h2y_convert_frame(h2y_ctx *ctx, const h2y_desc *d, const void *const in_planes[3], uint16_t *out_yuv)
{
    if (!ctx) return fail(nullptr, H2Y_EINVAL, "null ctx");
    if ((ctx->q_count > 0)) return fail(ctx, H2Y_EINVAL, "a batch is pending: call h2y_batch_finish first");
    if (ctx->streaming) return fail(ctx, H2Y_EINVAL, "a stream is open: close it first");
    const char *why;
    int rc = h2y_desc_check(d, &why);
    if (rc) return fail(ctx, rc, "descriptor: %s", why);
    if (!in_planes || !in_planes[0] || !in_planes[1] || !in_planes[2] || !out_yuv) return fail(ctx, H2Y_EINVAL, "null buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t pb = h2y_plane_bytes(d), pb_al = (pb + 255) & ~(size_t)255, ob = h2y_frame_bytes(d);
    rc = ensure(ctx, ctx->d_in, ctx->in_cap, 3 * pb_al);
    if (rc) return rc;
    rc = ensure(ctx, ctx->d_out, ctx->out_cap, ob);
    if (rc) return rc;
    frame_io io;
    for (int c = 0; c < 3; c++) {
        io.in[c] = (char *)ctx->d_in + c * pb_al;
        HIP_TRY(ctx, hipMemcpyAsync((void *)io.in[c], in_planes[c], pb, hipMemcpyHostToDevice, ctx->stream));
    }
    io.out = ctx->d_out;
    io.tmp_cb = io.tmp_cr = nullptr;
    /* the reference's order: pic_stats first, then the pixel loops with its result */
    const bool needs_stats = d->src_transfer != d->dst_transfer;
    bool host_knows = true;
    if (needs_stats && !d->stats_override) {
        rc = run_stats(ctx, d, io.in, (int)ctx->b->frames_cap, ctx->b->d_assumed);
        ctx->b->dev_assumed_ok = false; /* d_assumed[0] no longer holds what the last enqueued batch left there */
        if (rc) return rc;
        host_knows = false;
    } else {
        assumed_stats *as = ctx->b->h_assumed;
        for (int c = 0; c < 3; c++) {
            as->floor_[c] = d->stats_override ? d->floor[c] : 0;
            as->ceil_[c] = d->stats_override ? d->ceiling[c] : 1;
        }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->b->d_assumed, as, sizeof *as, hipMemcpyHostToDevice, ctx->stream));
        ctx->b->dev_assumed_ok = false; /* d_assumed[0] no longer holds what the last enqueued batch left there */
    }
    ctx->b->n_ev = 0;
    t1_begin_batch(ctx);
    rc = run_frames(ctx, d, &io, 1, ctx->b->d_assumed, host_knows ? ctx->b->h_assumed : nullptr, false, 0, true);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->b->h_fstats, ctx->b->d_fstats, 2 * sizeof(frame_stats), hipMemcpyDeviceToHost, ctx->stream)); /* + the XCD run times */
    HIP_TRY(ctx, hipMemcpyAsync(out_yuv, ctx->d_out, ob, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    t1_end_batch(ctx, d, ctx->b->h_fstats, 1);
    balance_update(ctx);
    float ms = 0.f;
    for (int i = 0; i < ctx->b->n_ev; i++) {
        float t = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&t, ctx->b->ev[i][0], ctx->b->ev[i][1]));
        ms += t;
    }
    ctx->last_ms = ms;
    ctx->last_launches = ctx->b->n_ev;
    return H2Y_OK;
}

int h2y_matrix_inverse(h2y_ctx *ctx, int width, int height, int in_bit_depth, int in_full_range, int in_matrix_coeffs,
                       int out_bit_depth, const uint16_t *const d_in[3], uint16_t *const d_out[3])
{
    if (!ctx) return fail(nullptr, H2Y_EINVAL, "null ctx");
    if ((ctx->q_count > 0) || ctx->streaming) return fail(ctx, H2Y_EINVAL, "a batch is pending or a stream is open");
    if (width < 1 || height < 1 || (uint64_t)width * height >= (1ull << 28)) return fail(ctx, H2Y_EINVAL, "bad picture size");
    if (in_bit_depth < 8 || in_bit_depth > 16 || out_bit_depth < 8 || out_bit_depth > 16) return fail(ctx, H2Y_EINVAL, "bit depths must be 8..16");
    if (in_matrix_coeffs == H2Y_MATRIX_GBR) /* convert.cpp:1733-1736: "Can't determine color difference to use?" and exit(0) */
        return fail(ctx, H2Y_EUNSUPPORTED, "matrix_coeffs 0 (GBR) has no inverse in the reference (it exits)");
    if (!d_in || !d_out) return fail(ctx, H2Y_EINVAL, "null pointer arrays");
    inverse_args a;
    for (int c = 0; c < 3; c++) {
        if (!d_in[c] || !d_out[c] || ((uintptr_t)d_in[c] & 7) || ((uintptr_t)d_out[c] & 7))
            return fail(ctx, H2Y_EINVAL, "plane %d is null or not 8-byte aligned", c);
        a.in[c] = d_in[c];
        a.out[c] = d_out[c];
    }
    const clip_limits ic = make_clip(in_bit_depth, in_full_range);
    a.npix = (uint32_t)width * (uint32_t)height;
    a.d709 = in_matrix_coeffs == H2Y_MATRIX_BT709;
    a.minVR = ic.minVR;
    a.maxVR = ic.maxVR;
    a.shift_right = in_bit_depth > out_bit_depth;
    a.shift = a.shift_right ? in_bit_depth - out_bit_depth : out_bit_depth - in_bit_depth;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t blocks = (a.npix / 4 + 255) / 256;
    if (blocks > (uint32_t)ctx->n_cu * 16u) blocks = (uint32_t)ctx->n_cu * 16u;
    if (blocks < 1) blocks = 1;
    ctx->b->n_ev = 0;
    HIP_TRY(ctx, hipEventRecord(ctx->b->ev[0][0], ctx->stream));
    HIP_TRY(ctx, h2y_launch_inverse((int)blocks, ctx->stream, a));
    HIP_TRY(ctx, hipEventRecord(ctx->b->ev[0][1], ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->b->ev[0][0], ctx->b->ev[0][1]));
    ctx->last_ms = ms;
    ctx->last_launches = 1;
    ctx->last_name = "k_inverse";
    ctx->last_variant = "k_inverse";
    return H2Y_OK;
}

/* Subsample420to444(), convert.cpp:1869-1986 */
static int upsample_launch(h2y_ctx *ctx, int width, int height, int algorithm, unsigned min_cv, unsigned max_cv, const uint16_t *s0,
                           const uint16_t *s1, uint16_t *d0, uint16_t *d1)
{
    up_args a;
    a.src0 = s0; a.src1 = s1; a.dst0 = d0; a.dst1 = d1;
    a.width = width; a.height = height;
    a.algorithm = algorithm;
    a.fmin = (float)min_cv; a.fmax = (float)max_cv;
    HIP_TRY(ctx, h2y_launch_up444(ctx->stream, a));
    return H2Y_OK;
}

int h2y_upsample_444(h2y_ctx *ctx, int width, int height, int algorithm, unsigned min_cv, unsigned max_cv, const uint16_t *d_src,
                     uint16_t *d_dst)
{
    if (!ctx) return fail(nullptr, H2Y_EINVAL, "null ctx");
    if ((ctx->q_count > 0) || ctx->streaming) return fail(ctx, H2Y_EINVAL, "a batch is pending or a stream is open");
    /* odd sizes: the reference's FIR branch reads rows of its intermediate it never wrote (convert.cpp:1949 walks
     * j < height over 2 * (height / 2) written rows): no defined bytes */
    if (width < 2 || height < 2 || (width & 1) || (height & 1) || width > 32766 || height > 32766)
        return fail(ctx, H2Y_EINVAL, "upsample: width and height must be even, 2..32766 (the reference takes them as short)");
    if (min_cv > max_cv || max_cv > 65535u) return fail(ctx, H2Y_EINVAL, "upsample: need minCV <= maxCV <= 65535");
    if (!d_src || !d_dst || ((uintptr_t)d_src & 1) || ((uintptr_t)d_dst & 3)) return fail(ctx, H2Y_EINVAL, "upsample: null pointer, or the result plane is not 4-byte aligned");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = upsample_launch(ctx, width, height, algorithm, min_cv, max_cv, d_src, nullptr, d_dst, nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return H2Y_OK;
}

int h2y_inverse_420(h2y_ctx *ctx, int width, int height, int in_bit_depth, int in_full_range, int in_matrix_coeffs, int out_bit_depth,
                    int algorithm, const uint16_t *const d_in[3], uint16_t *const d_out[3])
{
    if (!ctx) return fail(nullptr, H2Y_EINVAL, "null ctx");
    if ((ctx->q_count > 0) || ctx->streaming) return fail(ctx, H2Y_EINVAL, "a batch is pending or a stream is open");
    if (width < 2 || height < 2 || (width & 3) || (height & 1) || width > 32766 || height > 32766)
        return fail(ctx, H2Y_EINVAL, "4:2:0 inverse: width a multiple of 4 and height even, up to 32766"); /* the upsampled planes feed 8-byte loads */
    if (in_bit_depth < 8 || in_bit_depth > 16) return fail(ctx, H2Y_EINVAL, "bit depths must be 8..16");
    if (!d_in || !d_in[0] || !d_in[1] || !d_in[2]) return fail(ctx, H2Y_EINVAL, "null pointer arrays");
    if (out_bit_depth < 8 || out_bit_depth > 16) return fail(ctx, H2Y_EINVAL, "bit depths must be 8..16");
    if (in_matrix_coeffs == H2Y_MATRIX_GBR) return fail(ctx, H2Y_EUNSUPPORTED, "matrix_coeffs 0 (GBR) has no inverse in the reference (it exits)");
    if (!d_out || !d_out[0] || !d_out[1] || !d_out[2]) return fail(ctx, H2Y_EINVAL, "null pointer arrays");
    for (int c = 0; c < 3; c++)
        if (((uintptr_t)d_in[c] & 3) || ((uintptr_t)d_out[c] & 3)) return fail(ctx, H2Y_EINVAL, "plane %d is not 4-byte aligned", c);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    /* one pass: both chroma planes upsampled inside the blocks (yuv2tiff.cpp:92-93,142-154: minCV 0, maxCV 2^depth - 1), then
     * matrix_inverse's pixel; the 4:4:4 chroma never reaches memory (k_inverse420, h2y_resample.hip) */
    inv420_args a;
    a.up.src0 = d_in[1]; a.up.src1 = d_in[2]; a.up.dst0 = a.up.dst1 = nullptr;
    a.up.width = width; a.up.height = height;
    a.up.algorithm = algorithm;
    a.up.fmin = 0.0f; a.up.fmax = (float)((1u << in_bit_depth) - 1u);
    const clip_limits ic = make_clip(in_bit_depth, in_full_range);
    a.inv.in[0] = d_in[0]; a.inv.in[1] = a.inv.in[2] = nullptr;
    for (int c = 0; c < 3; c++) a.inv.out[c] = d_out[c];
    a.inv.npix = (uint32_t)width * (uint32_t)height;
    a.inv.d709 = in_matrix_coeffs == H2Y_MATRIX_BT709;
    a.inv.minVR = ic.minVR;
    a.inv.maxVR = ic.maxVR;
    a.inv.shift_right = in_bit_depth > out_bit_depth;
    a.inv.shift = a.inv.shift_right ? in_bit_depth - out_bit_depth : out_bit_depth - in_bit_depth;
    ctx->b->n_ev = 0;
    HIP_TRY(ctx, hipEventRecord(ctx->b->ev[0][0], ctx->stream));
    HIP_TRY(ctx, h2y_launch_inverse420(ctx->stream, a));
    HIP_TRY(ctx, hipEventRecord(ctx->b->ev[0][1], ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->b->ev[0][0], ctx->b->ev[0][1]));
    ctx->last_ms = ms;
    ctx->last_launches = 1;
    ctx->last_name = "k_inverse420";
    ctx->last_variant = algorithm ? "k_inverse420<FIR>" : "k_inverse420<REPLICATE>";
    return H2Y_OK;
}

int h2y_inverse_frame(h2y_ctx *ctx, int width, int height, int in_chroma_format_idc, int in_bit_depth, int in_full_range,
                      int in_matrix_coeffs, int out_bit_depth, int algorithm, const uint16_t *const in_planes[3], uint16_t *const out_planes[3])
{
    if (!ctx) return fail(nullptr, H2Y_EINVAL, "null ctx");
    if ((ctx->q_count > 0) || ctx->streaming) return fail(ctx, H2Y_EINVAL, "a batch is pending or a stream is open");
    if (in_chroma_format_idc != H2Y_CHROMA_444 && in_chroma_format_idc != H2Y_CHROMA_420)
        return fail(ctx, H2Y_EUNSUPPORTED, "inverse flow: input chroma_format_idc must be 3 (4:4:4) or 1 (4:2:0)");
    if (width < 1 || height < 1 || (uint64_t)width * height >= (1ull << 28)) return fail(ctx, H2Y_EINVAL, "bad picture size");
    if (!in_planes || !out_planes) return fail(ctx, H2Y_EINVAL, "null pointer arrays");
    for (int c = 0; c < 3; c++)
        if (!in_planes[c] || !out_planes[c]) return fail(ctx, H2Y_EINVAL, "plane %d is null", c);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool sub = in_chroma_format_idc == H2Y_CHROMA_420;
    const size_t pb = (size_t)width * height * sizeof(uint16_t), pb_al = (pb + 255) & ~(size_t)255;
    const size_t cb = sub ? (size_t)(width >> 1) * (height >> 1) * sizeof(uint16_t) : pb;
    int rc = ensure(ctx, ctx->d_in, ctx->in_cap, 3 * pb_al);
    if (rc) return rc;
    rc = ensure(ctx, ctx->d_out, ctx->out_cap, 3 * pb_al);
    if (rc) return rc;
    const uint16_t *din[3];
    uint16_t *dout[3];
    for (int c = 0; c < 3; c++) {
        din[c] = reinterpret_cast<const uint16_t *>((char *)ctx->d_in + c * pb_al);
        dout[c] = reinterpret_cast<uint16_t *>((char *)ctx->d_out + c * pb_al);
        HIP_TRY(ctx, hipMemcpyAsync((void *)din[c], in_planes[c], c ? cb : pb, hipMemcpyHostToDevice, ctx->stream));
    }
    rc = sub ? h2y_inverse_420(ctx, width, height, in_bit_depth, in_full_range, in_matrix_coeffs, out_bit_depth, algorithm, din, dout)
             : h2y_matrix_inverse(ctx, width, height, in_bit_depth, in_full_range, in_matrix_coeffs, out_bit_depth, din, dout);
    if (rc) return rc;
    for (int c = 0; c < 3; c++) HIP_TRY(ctx, hipMemcpyAsync(out_planes[c], dout[c], pb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return H2Y_OK;
}

/* ---- streaming pipeline ------------------------------------------------------
 * H2D of frame k+1, conversion of frame k and D2H of frame k-1 overlap: three streams, a ring of
 * pinned host slots the caller fills and drains in place.  Every frame is converted in the
 * reference's order (pic_stats pre-pass on the device, then the pixel kernel with its result in
 * device memory): no speculation, nothing to redo, no host round trip between the stages. */
static void stream_free(h2y_ctx *ctx)
{
    for (auto &s : ctx->ss) {
        if (s.h_in) (void)hipHostFree(s.h_in);
        if (s.h_out) (void)hipHostFree(s.h_out);
        if (s.d_in) (void)hipFree(s.d_in);
        if (s.d_out) (void)hipFree(s.d_out);
        if (s.ev_h2d) (void)hipEventDestroy(s.ev_h2d);
        if (s.ev_conv) (void)hipEventDestroy(s.ev_conv);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
    }
    ctx->ss.clear();
    if (ctx->s_h2d) (void)hipStreamDestroy(ctx->s_h2d);
    if (ctx->s_d2h) (void)hipStreamDestroy(ctx->s_d2h);
    ctx->s_h2d = ctx->s_d2h = nullptr;
    ctx->streaming = false;
    ctx->s_head = ctx->s_tail = 0;
    ctx->s_lent = -1;
}

int h2y_stream_open(h2y_ctx *ctx, const h2y_desc *d, int depth)
{
    if (!ctx) return fail(nullptr, H2Y_EINVAL, "null ctx");
    if ((ctx->q_count > 0) || ctx->streaming) return fail(ctx, H2Y_EINVAL, "a batch is pending or a stream is already open");
    const char *why;
    int rc = h2y_desc_check(d, &why);
    if (rc) return fail(ctx, rc, "descriptor: %s", why);
    if (depth < 2 || depth > 16) return fail(ctx, H2Y_EINVAL, "depth must be 2..16");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    rc = reserve_batch(ctx, 64);
    if (rc) return rc;
    const size_t pb = h2y_plane_bytes(d), ob = h2y_frame_bytes(d);
    ctx->s_plane_al = (pb + 255) & ~(size_t)255;
    ctx->s_desc = *d;
    ctx->ss.assign(depth, h2y_ctx::stream_slot());
    ctx->streaming = true;
    hipError_t e = hipStreamCreateWithFlags(&ctx->s_h2d, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->s_d2h, hipStreamNonBlocking);
    for (auto &s : ctx->ss) {
        if (e == hipSuccess) e = hipHostMalloc((void **)&s.h_in, 3 * ctx->s_plane_al, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void **)&s.h_out, ob, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&s.d_in, 3 * ctx->s_plane_al);
        if (e == hipSuccess) e = hipMalloc((void **)&s.d_out, ob);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_h2d, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_conv, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        stream_free(ctx);
        return fail(ctx, H2Y_ENOMEM, "stream buffers: %s", hipGetErrorString(e));
    }
    return H2Y_OK;
}

int h2y_stream_input(h2y_ctx *ctx, void *planes[3])
{
    if (!ctx || !planes) return fail(ctx, H2Y_EINVAL, "null argument");
    if (!ctx->streaming) return fail(ctx, H2Y_EINVAL, "no stream open");
    h2y_ctx::stream_slot &s = ctx->ss[ctx->s_tail];
    if (s.state == 1) { /* asked twice without a submit: same buffers again */
    } else if (s.state != 0) return fail(ctx, H2Y_EINVAL, "all %d slots are in flight: take an output first", (int)ctx->ss.size());
    s.state = 1;
    for (int c = 0; c < 3; c++) planes[c] = s.h_in + c * ctx->s_plane_al;
    return H2Y_OK;
}

int h2y_stream_submit(h2y_ctx *ctx)
{
    if (!ctx) return fail(nullptr, H2Y_EINVAL, "null ctx");
    if (!ctx->streaming) return fail(ctx, H2Y_EINVAL, "no stream open");
    const int slot = ctx->s_tail;
    h2y_ctx::stream_slot &s = ctx->ss[slot];
    if (s.state != 1) return fail(ctx, H2Y_EINVAL, "nothing to submit: call h2y_stream_input first");
    const h2y_desc *d = &ctx->s_desc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t pb = h2y_plane_bytes(d), ob = h2y_frame_bytes(d);
    frame_io io;
    for (int c = 0; c < 3; c++) io.in[c] = s.d_in + c * ctx->s_plane_al;
    /* the slot's three planes lie one after the other (each padded to 256 bytes): one copy command, not three */
    HIP_TRY(ctx, hipMemcpyAsync(s.d_in, s.h_in, 2 * ctx->s_plane_al + pb, hipMemcpyHostToDevice, ctx->s_h2d));
    io.out = s.d_out;
    io.tmp_cb = io.tmp_cr = nullptr;
    HIP_TRY(ctx, hipEventRecord(s.ev_h2d, ctx->s_h2d));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, s.ev_h2d, 0));
    const bool needs_stats = d->src_transfer != d->dst_transfer;
    int rc;
    if (needs_stats && !d->stats_override) {
        rc = run_stats(ctx, d, io.in, (int)ctx->b->frames_cap, ctx->b->d_assumed); /* published in device memory, read by the next kernel */
        ctx->b->dev_assumed_ok = false; /* d_assumed[0] no longer holds what the last enqueued batch left there */
        if (rc) return rc;
    } else {
        /* the same six integers for every frame of the stream: staged once per slot, so an earlier copy still in flight reads its own */
        assumed_stats *as = reinterpret_cast<assumed_stats *>(s.h_out); /* the slot's pinned output is idle until its D2H */
        for (int c = 0; c < 3; c++) {
            as->floor_[c] = d->stats_override ? d->floor[c] : 0;
            as->ceil_[c] = d->stats_override ? d->ceiling[c] : 1;
        }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->b->d_assumed, as, sizeof *as, hipMemcpyHostToDevice, ctx->stream));
        ctx->b->dev_assumed_ok = false; /* d_assumed[0] no longer holds what the last enqueued batch left there */
    }
    ctx->slot_base = slot;
    ctx->b->n_ev = 0;
    ctx->cur_skip_t1 = false; /* PCIe-bound here: no steering between the tiers */
    rc = run_frames(ctx, d, &io, 1, ctx->b->d_assumed, nullptr, false, slot, false);
    ctx->slot_base = 0;
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(s.ev_conv, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_d2h, s.ev_conv, 0));
    HIP_TRY(ctx, hipMemcpyAsync(s.h_out, s.d_out, ob, hipMemcpyDeviceToHost, ctx->s_d2h));
    HIP_TRY(ctx, hipEventRecord(s.ev_done, ctx->s_d2h));
    s.state = 2;
    ctx->s_tail = (slot + 1) % (int)ctx->ss.size();
    return H2Y_OK;
}

int h2y_stream_output(h2y_ctx *ctx, const uint16_t **yuv)
{
    if (!ctx || !yuv) return fail(ctx, H2Y_EINVAL, "null argument");
    if (!ctx->streaming) return fail(ctx, H2Y_EINVAL, "no stream open");
    if (ctx->s_lent >= 0) { /* the frame handed out last time goes back into the ring */
        ctx->ss[ctx->s_lent].state = 0;
        ctx->s_lent = -1;
    }
    h2y_ctx::stream_slot &s = ctx->ss[ctx->s_head];
    if (s.state != 2) return fail(ctx, H2Y_EINVAL, "no submitted frame is waiting");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(s.ev_done));
    *yuv = s.h_out;
    s.state = 3;
    ctx->s_lent = ctx->s_head;
    ctx->s_head = (ctx->s_head + 1) % (int)ctx->ss.size();
    return H2Y_OK;
}

int h2y_stream_close(h2y_ctx *ctx)
{
    if (!ctx) return fail(nullptr, H2Y_EINVAL, "null ctx");
    if (!ctx->streaming) return H2Y_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    (void)hipStreamSynchronize(ctx->s_h2d);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->s_d2h);
    stream_free(ctx);
    return H2Y_OK;
}


int h2y_pic_stats(h2y_ctx *ctx, const h2y_desc *d, const void *const d_in[3], float fminmax[6], int32_t floor_ceiling[6])
{
    if (!ctx) return fail(nullptr, H2Y_EINVAL, "null ctx");
    if ((ctx->q_count > 0)) return fail(ctx, H2Y_EINVAL, "a batch is pending");
    if (ctx->streaming) return fail(ctx, H2Y_EINVAL, "a stream is open: close it first");
    const char *why;
    int rc = h2y_desc_check(d, &why);
    if (rc) return fail(ctx, rc, "descriptor: %s", why);
    if (!d_in || !fminmax || !floor_ceiling) return fail(ctx, H2Y_EINVAL, "null argument");
    for (int c = 0; c < 3; c++) /* run_stats() takes the scalar-load path for planes that are not 16-byte aligned */
        if (!d_in[c] || ((uintptr_t)d_in[c] & (sample_bytes(d) - 1))) return fail(ctx, H2Y_EINVAL, "input plane %d is null or not aligned to its sample size", c);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    rc = run_stats(ctx, d, d_in, (int)ctx->b->frames_cap, nullptr);
    if (rc) return rc;
    frame_stats *hs = ctx->b->h_fstats + ctx->b->frames_cap;
    HIP_TRY(ctx, hipMemcpyAsync(hs, ctx->b->d_fstats + ctx->b->frames_cap, sizeof(frame_stats), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 6; i++) fminmax[i] = hs->mm[i];
    for (int c = 0; c < 3; c++) {
        floor_ceiling[2 * c] = hs->floor_[c];
        floor_ceiling[2 * c + 1] = hs->ceil_[c];
    }
    return H2Y_OK;
}

int h2y_matrix_convert(h2y_ctx *ctx, const h2y_desc *d, const void *const d_in[3], uint16_t *const d_out444[3])
{
    if (!ctx) return fail(nullptr, H2Y_EINVAL, "null ctx");
    if ((ctx->q_count > 0)) return fail(ctx, H2Y_EINVAL, "a batch is pending");
    if (ctx->streaming) return fail(ctx, H2Y_EINVAL, "a stream is open: close it first");
    const char *why;
    int rc = h2y_desc_check(d, &why);
    if (rc) return fail(ctx, rc, "descriptor: %s", why);
    if (!d_in || !d_out444) return fail(ctx, H2Y_EINVAL, "null pointer arrays");
    for (int c = 0; c < 3; c++) { /* the kernels issue 16-byte loads and 8-byte stores */
        if (!d_in[c] || ((uintptr_t)d_in[c] & 15)) return fail(ctx, H2Y_EINVAL, "input plane %d is null or not 16-byte aligned", c);
        if (!d_out444[c] || ((uintptr_t)d_out444[c] & 15)) return fail(ctx, H2Y_EINVAL, "output plane %d is null or not 16-byte aligned", c);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    assumed_stats *as = ctx->b->h_assumed;
    for (int c = 0; c < 3; c++) {
        as->floor_[c] = d->floor[c];
        as->ceil_[c] = d->ceiling[c];
    }
    if (d->src_transfer != d->dst_transfer)
        for (int c = 0; c < 3; c++)
            if (d->floor[c] == d->ceiling[c]) return fail(ctx, H2Y_EINVAL, "floor == ceiling for plane %d", c);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->b->d_assumed, as, sizeof *as, hipMemcpyHostToDevice, ctx->stream));
    ctx->b->dev_assumed_ok = false; /* d_assumed[0] no longer holds what the last enqueued batch left there */
    pix_params pp;
    derive_params(d, &pp, true);
    pp.pq_ext = ctx->d_table_ext;
    fused_variant var;
    var.in_kind = in_kind_of(d);
    var.out_kind = H2Y_OUT_444TMP;
    var.mode = pp.mode;
    var.narrow = (d->width % 4) != 0;
    var.even_h = (d->height & 1) == 0;
    var.pipe = (pp.convert_transfer == 1 && !var.narrow) ? 2 : 0;
    const geom g = make_geom(d, h2y_fused_threads(var));
    frame_io io;
    for (int c = 0; c < 3; c++) io.in[c] = d_in[c];
    io.out = d_out444[0];
    io.tmp_cb = d_out444[1];
    io.tmp_cr = d_out444[2];
    ctx->b->h_frames[0] = io;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->b->d_frames, ctx->b->h_frames, sizeof(frame_io), hipMemcpyHostToDevice, ctx->stream));
    ctx->b->dev_frames.clear(); /* run_frames()'s record of what d_frames holds */
    const int grid = grid_for(ctx, var, g.chunks);
    rc = ensure(ctx, ctx->b->d_partial, ctx->b->partial_cap, (size_t)grid * (h2y_fused_threads(var) / 64) * 6 * sizeof(float));
    if (rc) return rc;
    fused_args a;
    a.frames = ctx->b->d_frames;
    a.n_frames = 1;
    a.width = d->width;
    a.height = d->height;
    a.wq = g.wq;
    a.wq_magic = g.wq_magic;
    a.tiles_per_frame = g.tiles;
    a.chunks_per_frame = g.chunks;
    a.groups = 1;
    a.xcd_layout = 0;
    a.block_clock = nullptr;
    a.slice_ranges = nullptr;
    a.table = ctx->d_table;
    a.table_src = a.table_dst = nullptr; /* (a generic transfer pair takes the careful tier in this stage entry) */
    a.lut16 = ctx->d_lut16;
    a.table1 = ctx->d_table1;
    memset(&a.sn, 0, sizeof a.sn);
    a.tiles_magic = 0;
    a.redo_count = nullptr;
    a.low_flag = nullptr;
    a.partial = ctx->b->d_partial;
    a.assumed = ctx->b->d_assumed;
    a.pp = pp;
    HIP_TRY(ctx, h2y_launch_fused(var, grid, ctx->stream, a));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return H2Y_OK;
}

int h2y_subsample_420(h2y_ctx *ctx, int width, int height, int bit_depth, int chroma_resampler_type, const uint16_t *d_src,
                      uint16_t *d_dst)
{
    if (!ctx) return fail(nullptr, H2Y_EINVAL, "null ctx");
    if ((ctx->q_count > 0)) return fail(ctx, H2Y_EINVAL, "a batch is pending");
    if (ctx->streaming) return fail(ctx, H2Y_EINVAL, "a stream is open: close it first");
    if (width < 2 || height < 2 || (width & 1) || (height & 1) || bit_depth < 8 || bit_depth > 16 || !d_src || !d_dst)
        return fail(ctx, H2Y_EINVAL, "bad subsample arguments");
    if (chroma_resampler_type == 0 && ((width & 3) || (height & 3))) return fail(ctx, H2Y_EINVAL, "box needs multiples of 4");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (chroma_resampler_type == 0) HIP_TRY(ctx, h2y_launch_box420(ctx->stream, d_src, d_dst, width, height));
    else {
        fir_args fr;
        memset(&fr, 0, sizeof fr);
        fr.frames = nullptr;
        fr.src_cb = d_src;
        fr.src_cr = nullptr;
        fr.dst_cb = d_dst;
        fr.dst_cr = nullptr;
        fr.width = width;
        fr.height = height;
        fr.fir_max = (float)((1u << bit_depth) - 1);
        fr.apply_yuv_clamp = 0;
        HIP_TRY(ctx, h2y_launch_fir420(ctx->stream, fr));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return H2Y_OK;
}

const char *h2y_last_kernel_name(const h2y_ctx *ctx) { return ctx ? ctx->last_name : ""; }
const char *h2y_last_kernel_variant(const h2y_ctx *ctx) { return ctx ? ctx->last_variant.c_str() : ""; }

int h2y_last_kernel_ms(const h2y_ctx *ctx, float *ms, int *launches)
{
    if (!ctx) return H2Y_EINVAL;
    if (ms) *ms = ctx->last_ms;
    if (launches) *launches = ctx->last_launches;
    return H2Y_OK;
}

} // extern "C"
