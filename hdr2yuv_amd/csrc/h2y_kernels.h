/* h2y_kernels.h -- argument blocks and launch entry points shared by
 * h2y_kernels.hip (device code) and h2y_api.hip (the C-ABI shim). */
#ifndef H2Y_KERNELS_H
#define H2Y_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "h2y_math.h"

#ifndef H2Y_FUSED_THREADS
#define H2Y_FUSED_THREADS 512
#endif
#ifndef H2Y_FUSED_MINWAVES
#define H2Y_FUSED_MINWAVES 4 /* waves per SIMD the fused kernel is register-budgeted for: 2 blocks of 512 per CU */
#endif

/* k_fused2 / k_fused_lut16: one block of 1024 per CU -- all sixteen waves of a CU draw their tiles from one
 * counter (wave_deal); with two blocks of 512 the older block's waves run ahead of the younger's */
#ifndef H2Y_LOOP_THREADS
#define H2Y_LOOP_THREADS 1024
#endif

#define H2Y_LUT16_N 16384 /* halves 0x0000..0x3FFF = [0, 2) */

enum { H2Y_IN_F32 = 0, H2Y_IN_F16 = 1, H2Y_IN_U16 = 2 };
enum { H2Y_OUT_420BOX = 0, H2Y_OUT_444 = 1, H2Y_OUT_444TMP = 2 };

/* one frame's buffers (device pointers) */
struct frame_io {
    const void *in[3]; /* planes G,B,R (convert.cpp:980-982) */
    uint16_t *out;     /* .yuv frame: Y | Cb | Cr */
    uint16_t *tmp_cb;  /* H2Y_OUT_444TMP only: 4:4:4 matrix_convert output */
    uint16_t *tmp_cr;
};

/* what k_stats_final leaves per frame */
struct frame_stats {
    float mm[6];       /* min0,max0,min1,max1,min2,max2 */
    int32_t floor_[3]; /* pic_stats estimated_floor   */
    int32_t ceil_[3];  /* pic_stats estimated_ceiling */
    int32_t mismatch;  /* assumed != measured         */
    uint32_t redone;   /* k_fused_t1: tiles of this frame the first tier could not settle (redone by the exact tiers) */
};

/* estimated_floor/ceiling the pixel kernels normalise with (convert.cpp:939-940).
 * Lives in device memory so that a stats pre-pass can hand it to the next
 * kernel without a host round trip. */
struct assumed_stats {
    int32_t floor_[3];
    int32_t ceil_[3];
};

struct fused_args {
    const frame_io *frames; /* device array, n_frames entries */
    int n_frames;
    uint32_t width, height;
    uint32_t wq;              /* width/4 (k_fused) or width (k_fused_narrow) */
    uint32_t wq_magic;        /* floor(2^32 / wq) */
    uint32_t tiles_per_frame; /* thread-tiles per frame */
    uint32_t chunks_per_frame;
    uint32_t groups;          /* k_fused2 / k_fused_t1 / k_fused_lut16: the grid works as this many groups of gridDim.x / groups
                                 blocks, group g on frames g, g + groups, ... (1: every block on every frame); divides gridDim.x */
    uint32_t xcd_layout;      /* 1: gridDim.x is a multiple of 8 * groups and groups are made of whole rounds of the eight XCDs */
    unsigned long long *block_clock; /* [gridDim.x][2]: start, finish (wall_clock64) of each block, or NULL; finish entries zero at launch */
    const uint32_t *slice_ranges; /* xcd_layout, loop-form kernels: [gridDim.x / groups + 1] first 64-tile slice of every block of a group
                                     (the last entry = slices per frame): block i of a group takes slices [r[i], r[i+1]) of each of the
                                     group's frames; NULL = the round-robin dealing of frame_walk */
    uint32_t *tail_ctr = nullptr; /* not NULL (k_fused_t1, slice ranges in force): the last frame of every group is drawn dynamically (h2y_walk.h,
                                 "The dynamic last frame"): [groups][H2Y_TAIL_WORDS] counters and bits, zero at launch (k_stats_final clears them again) */
    uint32_t tail_slices = 0; /* slices of such a frame */
    uint32_t range_stride = 0; /* 0: one table for every group; else the words from one group's table to the next's (each group cut by
                                 the speeds of its own blocks) */
    const void *table;        /* pq_recA[NREC] then pq_recB[NREC] */
    const void *table_src, *table_dst; /* k_fused, generic transfer pair (pp.convert_transfer == 2): the two stages' tables in the same
                                          format (tfn_build_table), or NULL for a stage that is the identity */
    const float *lut16;       /* k_fused_lut16: PQ of every half in [0,2) */
    const void *table1;       /* k_fused_t1: pq_rec1[H2Y_T1_NREC] */
    h2y::t1_sens sn;          /* k_fused_t1: sensitivity windows */
    uint32_t tiles_magic;     /* floor(2^32 / tiles_per_frame): k_fused_t1's redo list holds frame * tiles + tile */
    float *partial;           /* [n_frames][grid / groups][waves][6] (k_fused: [n_frames][grid][6]) */
    uint32_t *redo_count;     /* k_fused_t1: [n_frames][grid / groups * waves] tiles sent to the redo list, or NULL */
    uint32_t *low_flag;       /* k_fused_t1 with assumed floor 0 / ceiling 1: [n_frames], kept zero between launches, or NULL */
    const assumed_stats *assumed;
    h2y::pix_params pp;       /* offset/range/norm_identity are filled in-kernel from *assumed */
};

/* which instantiation of the fused kernel */
struct fused_variant {
    int in_kind, out_kind, mode;
    int pipe;      /* 0 runtime flags, 1 LINEAR->PQ with floor 0/ceiling 1, 2 LINEAR->PQ general normalisation,
                      3 as 1 for half input through the 16 384-entry table (k_fused_lut16),
                      4 / 5 as 1 / 2 with the binary32 first tier in front (k_fused_t1),
                      6 equal transfers, no PQ at all (k_fused2) */
    bool narrow;   /* width % 4 != 0: scalar-load variant */
    bool even_h;   /* height % 2 == 0: the branch-free loop forms (k_fused2, k_fused_t1) apply */
    bool t1_ok = false; /* the binary32 first tier applies to this descriptor (whatever pipe was chosen in the end): t1_sens is filled in */
    bool cols8 = false; /* k_fused_lut16 only: 8-column thread tiles (width % 8 == 0, all planes 16-byte aligned); wq and tiles count those */
};

struct stats_args {
    const void *in[3];
    size_t npix;
    int vec_ok;     /* planes 16-byte aligned and npix % 4 == 0 handled by tail loop */
    float *partial; /* [grid][6] */
};

struct final_args {
    const float *partial; /* [n_frames][nblk][6] */
    const uint32_t *redo_count; /* [n_frames][nblk] or NULL */
    uint32_t *low_flag;         /* [n_frames] or NULL: see fused_args; read and cleared */
    int nblk;
    frame_stats *out; /* [n_frames] */
    int is_u16, src_bit_depth;
    int check;                     /* compare against *assumed, set frame_stats.mismatch */
    const assumed_stats *assumed;  /* may be NULL when check == 0 */
    assumed_stats *publish;        /* not NULL: frame 0's floor/ceiling are written here */
    unsigned long long *block_clock; /* not NULL: fused_args.block_clock of the launch just finished ... */
    int grid;                        /* ... its grid ... */
    float *xcd_time;                 /* ... and where block 0 leaves the mean run time (us) of the blocks of each XCD [8]; clears the finish entries */
    float *block_time = nullptr;     /* not NULL: [min(grid, 1024)] every block's own run time (us), for the per-block balancing */
    uint32_t *tail_ctr = nullptr;    /* not NULL: fused_args.tail_ctr, cleared for the next launch ... */
    int tail_n = 0;                  /* ... its words */
};

struct fir_args {
    const frame_io *frames;          /* batch form: n_frames entries (tmp_cb/tmp_cr -> out's chroma planes); else NULL */
    int n_frames;
    const uint16_t *src_cb, *src_cr; /* single form: 4:4:4 planes; src_cr may be NULL (one plane) */
    uint16_t *dst_cb, *dst_cr;
    int width, height;
    float fir_max;       /* (float)clip->maxCV of the tmp picture, convert.cpp:314 */
    int apply_yuv_clamp; /* 1: follow with write_yuv's shift + range clamp */
    h2y::pix_params pp;
};

struct inverse_args {
    const void *in[3]; /* Y, Cb/Dz, Cr/Dx: U16 4:4:4 planes, 8-byte aligned */
    void *out[3];      /* G, B, R */
    uint32_t npix;
    int d709;          /* matrix_coeffs == 1 */
    uint32_t minVR, maxVR;
    int shift, shift_right;
};

/* k_fir_fused (h2y_fir_fused.hip): a wave's unit of work is (frame, segment of chroma rows, strip of 240 columns) */
struct firf_args {
    const frame_io *frames;
    int n_frames;
    uint32_t width, height;
    uint32_t wq;              /* width / 4 */
    uint32_t n_strips;        /* ceil(wq / 60) */
    uint32_t n_seg, seg_rows; /* segments per frame, chroma rows per segment (the last may be shorter): the even cut */
    const uint32_t *unit_rows; /* [total_units]: first chroma row | one past the last << 16 of every unit, or NULL for the even cut.
                                  Strips are independent of each other, so every (frame, strip) column may be cut its own way:
                                  the host gives the units that run on slower XCDs fewer rows */
    unsigned long long *block_clock; /* as fused_args.block_clock */
    uint32_t mix_xcds;        /* 1 (grid a multiple of 8): block b works as block h2y_firf_vblock(b), so that the segments of one
                                 (frame, strip) column land on XCDs of both halves of the card and of both parities */
    uint32_t units_per_frame; /* n_seg * n_strips */
    uint32_t total_units;     /* n_frames * units_per_frame */
    uint32_t sync_mask;       /* the block's waves meet at a barrier every sync_mask + 1 steps (a power of two); ~0u: never */
    const void *table, *table1;
    const float *lut16;       /* FF_TIER_LUT16: PQ of every half in [0, 2) */
    h2y::t1_sens sn;
    float *partial;           /* [n_frames][units_per_frame][6] */
    uint32_t *redo_count;     /* [n_frames][units_per_frame] */
    uint32_t *low_flag;       /* [n_frames] or NULL (see fused_args) */
    const assumed_stats *assumed;
    h2y::pix_params pp;
};

struct up_args { /* k_up444: one or two chroma planes, (width/2 x height/2) -> (width x height) */
    const uint16_t *src0, *src1; /* src1 may be NULL (one plane) */
    uint16_t *dst0, *dst1;
    int width, height;           /* of the 4:4:4 result; both even */
    int algorithm;               /* 0 replication, else the FIR pair */
    float fmin, fmax;            /* (float) of minCV / maxCV, convert.cpp:1932-1934 */
};

struct inv420_args { /* k_inverse420: Subsample420to444 of both chroma planes and matrix_inverse in one pass */
    up_args up;       /* src0/src1 = the 4:2:0 Cb/Dz and Cr/Dx planes; dst0/dst1 unused; width, height, algorithm, fmin, fmax */
    inverse_args inv; /* in[0] = the luma plane (in[1], in[2] unused), out[3] = G, B, R */
};

/* k_fir_fused: lanes of a wave that own chroma columns (the others, half on either side, only feed the horizontal taps):
 * a strip is 4 x this many picture columns */
#ifndef H2Y_FF_OWN_LANES
#define H2Y_FF_OWN_LANES 60
#endif
#ifdef H2Y_BLOCK_TIMES
void h2y_dump_block_times(const char *path); /* timing experiments only */
void h2y_dump_ff_block_times(const char *path);
#endif
int h2y_fused_threads(const fused_variant &v);
const char *h2y_fused_name(const fused_variant &v);
bool h2y_fused_grouped(const fused_variant &v); /* does the kernel honour fused_args.groups? */
int h2y_fused_blocks_per_cu(const fused_variant &v);
hipError_t h2y_launch_fused(const fused_variant &v, int grid, hipStream_t st, const fused_args &a);
hipError_t h2y_launch_build_lut16(hipStream_t st, const void *table, float *lut);
hipError_t h2y_launch_stats(int in_kind, int grid, hipStream_t st, const stats_args &a);
hipError_t h2y_launch_stats_final(int n_frames, hipStream_t st, const final_args &a);
hipError_t h2y_launch_fir420(hipStream_t st, const fir_args &a);
hipError_t h2y_launch_inverse(int grid, hipStream_t st, const inverse_args &a);
/* k_fir_fused: bits 1 and 2 of the block number exchanged (an involution on [0, 8k)).  Block b runs on XCD b % 8; with four
 * segments per column, b = 4 f + segment would give a column the XCDs {0..3} or {4..7} -- after the exchange {0,1,4,5} or {2,3,6,7} */
static inline uint32_t h2y_firf_vblock(uint32_t b) { return (b & ~6u) | ((b & 2u) << 1) | ((b & 4u) >> 1); }
hipError_t h2y_launch_fir_fused(int in_kind, int mode, bool ident, bool lut16, int grid, hipStream_t st, const firf_args &a);
hipError_t h2y_launch_up444(hipStream_t st, const up_args &a);
hipError_t h2y_launch_inverse420(hipStream_t st, const inv420_args &a);
hipError_t h2y_launch_box420(hipStream_t st, const uint16_t *src, uint16_t *dst, int W, int H);

#endif
