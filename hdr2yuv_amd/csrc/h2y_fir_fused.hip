/*
 * h2y_fir_fused.hip -- k_fir_fused: the whole path with the FIR chroma resampler in ONE pass.
 *
 *   pic_stats min/max + matrix_convert (convert.cpp:879-1221) + Subsample444to420_FIR (convert.cpp:261-383)
 *   + write_yuv's clamp (tiff.cpp:457-550): planar float RGB in, .yuv 4:2:0 out, nothing in between touches HBM.
 *
 * The two-pass form (fused kernel -> 4:4:4 Cb/Cr scratch -> k_fir420) moves 23 bytes per pixel against 15
 * algorithmic.  Here the 4:4:4 chroma never leaves the registers:
 *
 *   A WAVE owns a column strip of 240 picture columns and walks it top to bottom, one row pair per step; lane L
 *   holds columns 4L-8 .. 4L-5 of the strip (lanes 0, 1 and 62, 63 are halo: they recompute the neighbour strips'
 *   pixels -- 6 % -- so that no wave ever waits for another).
 *   Horizontal 7-tap (even columns): the taps at odd columns come from lanes L-2, L-1, L+1 by three DPP wave shifts
 *   of one packed dword per row and plane.
 *   Vertical 12-tap: each lane keeps the last eleven rows of ITS two 4:2:2 columns (both planes, two rows to a
 *   register: 24 registers) and emits chroma row j = s - 3 at step s; the taps go two at a time (v_dot2_i32_i16).  Picture edges replicate as the reference's index clamps do: rows above
 *   the picture are row 0 (the history is filled with it at step 0), rows below are the last row (three virtual
 *   steps), the first and last column stand in for the columns beyond them.
 *   A frame's strip is cut into segments of rows so that a launch has enough waves' worth of work; a segment that
 *   starts inside the picture first recomputes the three row pairs above it (and the one before it three below):
 *   6 row pairs per cut.
 *
 * Arithmetic: the binary32 first tier (pq_t1 + pix_matrix_t1) for every pixel; where a lane's pixel is not settled
 * by it (unsure sample near a rounding boundary, sample outside the table: 0.05 % of pixels at 12 bits, every pixel
 * of a black bar) the wave takes the binary64 tier for that pixel position on the spot -- the FIR consumes the
 * 4:4:4 values at once, so there is no "provisional bytes now, exact bytes later" here.  Both tables sit in LDS
 * (100 KB + 50 KB), one block of 1024 threads per CU.  The FIR stages run in integers (fir_h_int / fir_v_int:
 * equal to the reference's float expressions for code values up to 14 bits, tools/fir_int_check.cpp).
 *
 * The loop body keeps k_fused_t1's discipline: global memory operations in a fixed order, none inside a branch
 * (lanes that own no chroma store lane 2's sample again, rows not yet final go to the segment's first row and are
 * overwritten in order by the same lane), so every s_waitcnt is exact and the next step's rows arrive while this
 * step computes.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdio.h>

#include "h2y_device.h"

#define FF_THREADS 1024
#ifndef FF_BUFFER_STORE
#define FF_BUFFER_STORE 1 /* 1: chroma through range-checked buffer stores (lanes without a column are dropped by the hardware) */
#endif
#define FF_OWN_LANES H2Y_FF_OWN_LANES /* lanes 2 .. 61 own chroma: 240 picture columns per strip (h2y_kernels.h) */
#define FF_HALO ((64 - FF_OWN_LANES) / 2) /* lanes on either side that only feed the horizontal taps */

/* one row pair of the lane's four columns, three planes: as floats, or -- FF_TIER_LUT16, whose table is indexed by the
 * half's bits -- as the halves they were loaded as, two to a register (widening them only to narrow them again cost 48
 * conversions a step, 7 % of that tier's instructions) */
template <bool RAW> struct ff_rows;
template <> struct ff_rows<false> { float g0[4], b0[4], r0[4], g1[4], b1[4], r1[4]; };
template <> struct ff_rows<true> { uint32_t g0[2], b0[2], r0[2], g1[2], b1[2], r1[2]; };
template <int IN_KIND> __device__ __forceinline__ void ff_load(const void *p, uint32_t q, float (&d)[4]) { in_traits<IN_KIND>::load4q(p, q, d); }
template <int IN_KIND> __device__ __forceinline__ void ff_load(const void *p, uint32_t q, uint32_t (&d)[2])
{
    const u32x2 t = gload_nt<u32x2>(p, q); /* four halves */
    d[0] = t.x; d[1] = t.y;
}
__device__ __forceinline__ uint32_t ff_pk_min_h(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_min_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t ff_pk_max_h(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float ff_half_to_float(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (uint16_t)h); }

/* lane L gets lane L-1's / L+1's value (DPP wave shifts; what arrives in lane 0 resp. 63 is never used) */
__device__ __forceinline__ uint32_t from_lane_below(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t from_lane_above(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, true); }

struct ff_edges { /* per lane, fixed for a unit */
    uint32_t any; /* the wave holds a picture edge (uniform, in a scalar register) */
    bool left0;  /* this lane's columns are the picture's first four */
    bool left1;  /* ... the next four */
    bool right0; /* ... the last four */
};

/* two signed 16-bit products and a 32-bit sum in one instruction (v_dot2_i32_i16): the FIR taps in pairs */
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int32_t dot2(uint32_t v, uint32_t k, int32_t acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, v), __builtin_bit_cast(s16x2, k), acc, false);
}
#define FF_K(lo, hi) (((uint32_t)(lo) & 0xFFFFu) | ((uint32_t)(hi) << 16))

/* Horizontal stage for one row and plane (convert.cpp:291-320): c[0..3] = this lane's four 4:4:4 values (below
 * 2^14); a, b = the 4:2:2 values at its even columns 4L and 4L+2, clamped to [0, maxCV] and truncated.  Integer
 * form of fir_h() (h2y_math.h: exact up to 14 bits), the taps taken two at a time.
 * RAWC (the first tier): c[] are pix_matrix_t1()'s raw chroma integers -- signed 16-bit quantities, the code value
 * minus (Half - 1).  The taps sum to 512, so the offset is one constant in the stage's rounding term: k0 = 512 (Half - 1)
 * + 256 (without RAWC: 256).  a and b come out as code values either way. */
template <bool RAWC>
__device__ __forceinline__ uint32_t ff_pack16(uint32_t lo, uint32_t hi)
{
    /* low halves of both: one v_perm_b32 (bytes 1, 0 of hi over bytes 1, 0 of lo); written as and / shift / or it came out as two instructions */
    return RAWC ? __builtin_amdgcn_perm(hi, lo, 0x05040100u) : (lo | (hi << 16));
}
template <bool RAWC>
__device__ __forceinline__ void ff_hstage(const uint32_t (&c)[4], const ff_edges &e, int32_t maxcv, uint32_t k0, uint32_t &a, uint32_t &b)
{
    const uint32_t p13 = ff_pack16<RAWC>(c[1], c[3]);
    uint32_t l1 = from_lane_below(p13); /* columns 4L-3, 4L-1 */
    uint32_t l2 = from_lane_below(l1);  /* columns 4L-7, 4L-5 */
    uint32_t r1 = from_lane_above(p13); /* columns 4L+5, 4L+7 */
    if (e.any) { /* the reference's "picture border logic": indices below 0 read column 0, beyond width-1 the last column */
        const uint32_t c0_below = from_lane_below(c[0]);
        const uint32_t own0 = ff_pack16<RAWC>(c[0], c[0]), own3 = ff_pack16<RAWC>(c[3], c[3]);
        l1 = e.left0 ? own0 : l1;
        l2 = e.left0 ? own0 : (e.left1 ? (c0_below << 16) : l2); /* 4L-5 = -1 -> column 0, which is the lane below's c[0] */
        r1 = e.right0 ? own3 : r1;
    }
    /* even column i = 4L: 21 (s[i-5] + s[i+5]) - 52 (s[i-3] + s[i+3]) + 159 (s[i-1] + s[i+1]) + 256 s[i] + 256, all over 512 */
    int32_t sa = (int32_t)((c[0] << 8) + k0);
    sa = dot2(l1, FF_K(-52, 159), sa);                               /* s[4L-3], s[4L-1] */
    sa = dot2(p13, FF_K(159, -52), sa);                              /* s[4L+1], s[4L+3] */
    sa = dot2(__builtin_amdgcn_alignbit(r1, l2, 16), FF_K(21, 21), sa); /* s[4L-5] (high half of l2), s[4L+5] (low half of r1) */
    /* even column i = 4L+2 */
    int32_t sb = (int32_t)((c[2] << 8) + k0);
    sb = dot2(l1, FF_K(21, -52), sb);   /* s[4L-3], s[4L-1] */
    sb = dot2(p13, FF_K(159, 159), sb); /* s[4L+1], s[4L+3] */
    sb = dot2(r1, FF_K(-52, 21), sb);   /* s[4L+5], s[4L+7] */
    a = (uint32_t)imed3_0(sa >> 9, maxcv);
    b = (uint32_t)imed3_0(sb >> 9, maxcv);
}

/* The lane's history of one 4:2:2 column and plane, rows two to a register: p[i] = row 2s-11+2i | row 2s-10+2i << 16
 * (rows 2s-11 .. 2s-2), last = row 2s-1. */
struct ff_hist {
    uint32_t p[5], last;
    __device__ __forceinline__ void fill(uint32_t v) /* every row is v (rows above the picture are row 0) */
    {
#pragma unroll
        for (int i = 0; i < 5; i++) p[i] = v | (v << 16);
        last = v;
    }
    /* Vertical stage (convert.cpp:323-376) for chroma row j = s - 3 with n0 = row 2s: rows 2j-5 .. 2j+6 against
     * (5 11 -21 -37 70 228 228 70 -37 -21 11 5)/512, then write_yuv's chroma clamp (inside [0, maxCV]: one clamp). */
    __device__ __forceinline__ uint32_t out(uint32_t n0, int32_t clo, int32_t chi) const
    {
        int32_t sv = dot2(p[0], FF_K(5, 11), 256);
        sv = dot2(p[1], FF_K(-21, -37), sv);
        sv = dot2(p[2], FF_K(70, 228), sv);
        sv = dot2(p[3], FF_K(228, 70), sv);
        sv = dot2(p[4], FF_K(-37, -21), sv);
        sv = dot2(last | (n0 << 16), FF_K(11, 5), sv);
        return (uint32_t)imed3(sv >> 9, clo, chi);
    }
    __device__ __forceinline__ void push(uint32_t n0, uint32_t n1) /* on to step s + 1 */
    {
        p[0] = p[1]; p[1] = p[2]; p[2] = p[3]; p[3] = p[4];
        p[4] = last | (n0 << 16);
        last = n1;
    }
};

/* TIER: how a sample's PQ value is found.
 *   FF_TIER_T1     float or half input: binary32 first tier, binary64 tier for the pixel positions it cannot settle
 *   FF_TIER_LUT16  half input with floor 0 / ceiling 1: the table of all 16 384 halves in [0, 2) (k_fused_lut16's, exact
 *                  by construction); a pixel with a sample outside it, or whose division guard fires, takes the careful tier */
#define FF_TIER_T1 0
#define FF_TIER_LUT16 1

#ifdef H2Y_BLOCK_TIMES /* timing experiments only: when does each block start and finish? (tools/blocktimes.py) */
__device__ unsigned long long g_ff_block_times[2 * 1024];
void h2y_dump_ff_block_times(const char *path)
{
    static unsigned long long h[2 * 1024];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ff_block_times), sizeof h) != hipSuccess) return;
    FILE *f = fopen(path, "w");
    if (!f) return;
    for (int i = 0; i < 1024; i++) fprintf(f, "%d %llu %llu\n", i, h[2 * i], h[2 * i + 1]);
    fclose(f);
}
#endif

template <int IN_KIND, int MODE, int PIPE, int TIER>
__global__ __launch_bounds__(FF_THREADS) void k_fir_fused(firf_args a)
{
#ifdef H2Y_BLOCK_TIMES
    if (threadIdx.x == 0) { g_ff_block_times[2 * blockIdx.x] = wall_clock64(); g_ff_block_times[2 * blockIdx.x + 1] = 0ull; }
#endif
    if (a.block_clock && threadIdx.x == 0) a.block_clock[2 * blockIdx.x] = wall_clock64();
    __shared__ pq_rec1 s_t1[TIER == FF_TIER_T1 ? H2Y_T1_NREC : 1];
    __shared__ pq_recA s_t2[TIER == FF_TIER_T1 ? 2 * H2Y_PQ_NREC : 1]; /* A records, then B records */
    __shared__ float s_lut_y[TIER == FF_TIER_LUT16 ? H2Y_LUT16_N : 1], s_lut_c[TIER == FF_TIER_LUT16 ? H2Y_LUT16_N : 1]; /* stage_lut16_scaled() */
    __shared__ pix_params s_pp;
    const pq_recA *sA = s_t2;
    const pq_recB *sB = reinterpret_cast<const pq_recB *>(s_t2 + (TIER == FF_TIER_T1 ? H2Y_PQ_NREC : 0));
    const pix_params pp = with_assumed(a.pp, a.assumed);
    if (TIER == FF_TIER_T1) {
        stage16<FF_THREADS, H2Y_T1_NREC>(a.table1, s_t1);
        stage_table<FF_THREADS>(a.table, s_t2);
    } else stage_lut16_scaled<FF_THREADS>(a.lut16, pp, s_lut_y, s_lut_c);
    t1_sens sn = a.sn;
    asm volatile("" : "+v"(sn.a_lo), "+v"(sn.a_hi));
    if (threadIdx.x == 0) s_pp = pp;
    __syncthreads();
    /* a pixel of three +0.0 samples (black bars): the first tier cannot answer it (zero is outside its table), so rows of
     * them are recognised before the arithmetic (below) and given this pixel's code values, worked out once by the exact tiers */
    __shared__ uint32_t s_black[3];
    if (TIER == FF_TIER_T1) {
        if (threadIdx.x == 0) {
            const float G0 = norm1<PIPE>(pp, 0, 0.0f), B0 = norm1<PIPE>(pp, 1, 0.0f), R0 = norm1<PIPE>(pp, 2, 0.0f);
            uint32_t y, cb, cr;
            if (pixel_fast<MODE, PIPE>(pp, sA, sB, G0, B0, R0, y, cb, cr)) {
                const ycc k = pixel_careful<MODE>(&s_pp, G0, B0, R0);
                y = k.y; cb = k.cb; cr = k.cr;
            }
            s_black[0] = y; s_black[1] = cb - pp.half_m1; s_black[2] = cr - pp.half_m1; /* chroma raw, as pix_matrix_t1() gives it */
        }
        __syncthreads();
    }

    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t GW = gridDim.x * (FF_THREADS / WAVE);
    const uint32_t vblock = a.mix_xcds ? ((blockIdx.x & ~6u) | ((blockIdx.x & 2u) << 1) | ((blockIdx.x & 4u) >> 1)) : blockIdx.x; /* h2y_firf_vblock() */
    const uint32_t gw = __builtin_amdgcn_readfirstlane(vblock * (FF_THREADS / WAVE) + threadIdx.x / WAVE); /* uniform, and known to be */
    const uint32_t W = a.width, H = a.height, WQ = a.wq, H2 = H >> 1;
    const uint32_t npix = W * H, ncb = (W >> 1) * (H >> 1);
    const int32_t maxcv = (int32_t)pp.maxCV, clo = (int32_t)pp.clo_s, chi = (int32_t)pp.chi_s; /* float input: down_shift == 0 */
    const uint32_t hk0 = TIER == FF_TIER_T1 ? 512u * pp.half_m1 + 256u : 256u; /* ff_hstage()'s rounding term: with the raw chroma's offset */

    /* a wave's units: u = gw, gw + GW, ...; unit = (frame, segment, strip), strips of one band next to each other
     * so that the sixteen waves of a block fill whole lines of the output between them */
    /* The launch's arguments are read where a unit begins and where it ends, through a pointer the compiler cannot see through
     * (the kernel-argument segment, laundered): read once at the top, a dozen of them stay in scalar registers across the step
     * loop, which does not use them, while the loop's own plane pointers were spilled (twelve lane reads a step). */
    typedef const __attribute__((address_space(4))) firf_args *args_ptr;
    args_ptr ap = (args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ap));
    const uint32_t total_units = ap->total_units, sync_mask = ap->sync_mask;
    for (uint32_t u = gw; u < total_units; u += GW) {
        asm volatile("" : "+s"(ap));
        /* (uniform values; the divisions run on the vector unit, so say so) */
        const uint32_t upf = ap->units_per_frame, nst = ap->n_strips, seg_rows = ap->seg_rows;
        const uint32_t f = __builtin_amdgcn_readfirstlane(u / upf), r = u - f * upf;
        const uint32_t seg = __builtin_amdgcn_readfirstlane(r / nst), strip = r - seg * nst;
        const frame_io io = uniform_io(ap->frames + f);
        uint32_t j0 = seg * seg_rows, j1 = j0 + seg_rows < H2 ? j0 + seg_rows : H2;   /* chroma rows [j0, j1) are this unit's */
        if (ap->unit_rows) { /* the host's own cut of this (frame, strip) column (weights by XCD speed) */
            const uint32_t rw = __builtin_amdgcn_readfirstlane(ap->unit_rows[u]);
            j0 = rw & 0xFFFFu;
            j1 = rw >> 16;
        }
        const uint32_t s_begin = j0 >= 3u ? j0 - 3u : 0u, s_end = j1 + 2u;          /* steps: row pairs s_begin .. s_end (those >= H2 are virtual) */
        const int32_t qxu = (int32_t)(FF_OWN_LANES * strip + lane) - FF_HALO;       /* this lane's quad column, before clamping */
        const uint32_t qx = (uint32_t)min(max(qxu, 0), (int32_t)WQ - 1);
        const bool own = lane >= (uint32_t)FF_HALO && lane < (uint32_t)FF_HALO + FF_OWN_LANES && qxu < (int32_t)WQ;
#if FF_BUFFER_STORE
        const uint32_t qx_store = own ? qx : 0x20000000u; /* times four: past any frame, short of wrapping */
        /* the frame's output as a raw buffer: stores beyond its last byte are dropped by the hardware's range check */
        const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(io.out, 0, (int)((npix + 2u * ncb) * 2u), 0x00020000);
#endif
        ff_edges e;
        e.any = __builtin_amdgcn_readfirstlane((uint32_t)(strip == 0u) | (uint32_t)(strip + 1u == nst));
        e.left0 = qxu == 0;
        e.left1 = qxu == 1;
        e.right0 = qxu == (int32_t)WQ - 1;

        mm6 mm;
        mm.reset();
        uint32_t flagged = 0; /* pixels the first tier could not settle (uniform) */
        uint64_t low_m = 0;   /* lanes that met a sample <= -1 (see k_fused_t1: the minimum is only sampled) */
        ff_hist hb[2], hr[2]; /* rows 2s-11 .. 2s-1 of the 4:2:2 intermediate: Cb and Cr, this lane's columns 4L and 4L+2 */
        hb[0].fill(0u); hb[1].fill(0u); hr[0].fill(0u); hr[1].fill(0u);

        constexpr bool RAW = TIER == FF_TIER_LUT16;
        constexpr int NREG = RAW ? 2 : 4;
        uint32_t hmn[3] = {0x7C007C00u, 0x7C007C00u, 0x7C007C00u}, hmx[3] = {0u, 0u, 0u}; /* RAW: packed-half statistics, as k_fused_lut16's */
        ff_rows<RAW> v; /* the row pair in hand; refilled row by row with the next one */
        uint32_t q0 = 2u * (s_begin < H2 - 1u ? s_begin : H2 - 1u) * WQ + qx; /* (scalar min: the loop's bounds stay in scalar registers) */
        ff_load<IN_KIND>(io.in[0], q0, v.g0); ff_load<IN_KIND>(io.in[1], q0, v.b0); ff_load<IN_KIND>(io.in[2], q0, v.r0);
        ff_load<IN_KIND>(io.in[0], q0 + WQ, v.g1); ff_load<IN_KIND>(io.in[1], q0 + WQ, v.b1); ff_load<IN_KIND>(io.in[2], q0 + WQ, v.r1);
#pragma unroll
        for (int j = 0; j < NREG; j++) asm volatile("" ::"v"(v.g0[j]), "v"(v.b0[j]), "v"(v.r0[j]), "v"(v.g1[j]), "v"(v.b1[j]), "v"(v.r1[j]));

        for (uint32_t s = s_begin; s <= s_end; s++) {
            /* The block's sixteen waves are the sixteen strips of one band of rows: kept in step, they read a row of the
             * picture at about the same time -- one DRAM page after the other instead of sixteen places at once (4K x 64
             * frames: 1.90-1.95 ms per launch without, 1.74 every second step -- tools/firsyncbench.sh).  Waves whose units differ in length, or that
             * have none left, just meet less often: a barrier only waits for the waves still running. */
            if (((s - s_begin) & sync_mask) == 0u) __builtin_amdgcn_s_barrier();
            const uint32_t q0n = 2u * (s + 1u < H2 - 1u ? s + 1u : H2 - 1u) * WQ + qx; /* the next step's rows (the last rows again once the picture ends) */
            uint32_t yp[2][2], n_cb[2][2], n_cr[2][2]; /* new 4:2:2 values: [row][column 4L / 4L+2] */
#pragma unroll
            for (int row = 0; row < 2; row++) {
                const auto &gv = row ? v.g1 : v.g0; /* float[4], or (RAW) two registers of two halves */
                const auto &bv = row ? v.b1 : v.b0;
                const auto &rv = row ? v.r1 : v.r0;
                uint32_t Y[4], Cb[4], Cr[4];
                bool zrow = false;
                uint64_t row_out_m = 0; /* RAW: lanes with an out-of-table sample in this row (a compare's own lane mask) */
                if constexpr (RAW) { /* (always floor 0 / ceiling 1 here) every sample's maximum, a subsample of the minimum */
                    if (row == 0) { hmn[0] = ff_pk_min_h(hmn[0], gv[0]); hmn[1] = ff_pk_min_h(hmn[1], bv[0]); hmn[2] = ff_pk_min_h(hmn[2], rv[0]); }
                    hmx[0] = ff_pk_max_h(ff_pk_max_h(hmx[0], gv[0]), gv[1]);
                    hmx[1] = ff_pk_max_h(ff_pk_max_h(hmx[1], bv[0]), bv[1]);
                    hmx[2] = ff_pk_max_h(ff_pk_max_h(hmx[2], rv[0]), rv[1]);
                    /* a sample outside the table (sign or bit 14 set: negative, >= 2.0, inf, NaN) in this lane's row?  Looked for
                     * once per row on the packed halves, not per pixel: such a lane's four pixels all take the careful tier */
                    row_out_m = __builtin_amdgcn_ballot_w64((((gv[0] | gv[1] | bv[0]) | (bv[1] | rv[0] | rv[1])) & 0xC000C000u) != 0u);
                } else {
                if (PIPE == H2Y_PIPE_PQ_IDENT) { /* as k_fused_t1: every sample's maximum, a subsample of the minimum */
                    if (row == 0) { mm.add2(0, gv[0], gv[1]); mm.add2(1, bv[0], bv[1]); mm.add2(2, rv[0], rv[1]); }
                    else { mm.add2_max(0, gv[0], gv[1]); mm.add2_max(1, bv[0], bv[1]); mm.add2_max(2, rv[0], rv[1]); }
                    mm.add2_max(0, gv[2], gv[3]); mm.add2_max(1, bv[2], bv[3]); mm.add2_max(2, rv[2], rv[3]);
                } else {
                    mm.add2(0, gv[0], gv[1]); mm.add2(0, gv[2], gv[3]);
                    mm.add2(1, bv[0], bv[1]); mm.add2(1, bv[2], bv[3]);
                    mm.add2(2, rv[0], rv[1]); mm.add2(2, rv[2], rv[3]);
                }
                /* a row of zeros in every lane (letterbox bars)?  One compare per row for ordinary pictures: lane 0's first sample */
                if (TIER == FF_TIER_T1 && __builtin_expect(__builtin_amdgcn_ballot_w64(f2bits(gv[0]) != 0u) == 0, 0)) {
                    const uint32_t z = (f2bits(gv[1]) | f2bits(gv[2]) | f2bits(gv[3])) | (f2bits(bv[0]) | f2bits(bv[1]) | f2bits(bv[2])) |
                                       (f2bits(bv[3]) | f2bits(rv[0]) | f2bits(rv[1])) | (f2bits(rv[2]) | f2bits(rv[3]));
                    zrow = __builtin_amdgcn_ballot_w64(z != 0u) == 0;
                }
                }
                if (zrow) {
                    const uint32_t y0 = s_black[0], cb0 = s_black[1], cr0 = s_black[2];
#pragma unroll
                    for (int col = 0; col < 4; col++) { Y[col] = y0; Cb[col] = cb0; Cr[col] = cr0; }
                } else
#pragma unroll
                for (int col = 0; col < 4; col++) {
                    if constexpr (RAW) {
                        const uint32_t hg = (gv[col >> 1] >> (16 * (col & 1))) & 0xFFFFu, hb = (bv[col >> 1] >> (16 * (col & 1))) & 0xFFFFu,
                                       hr = (rv[col >> 1] >> (16 * (col & 1))) & 0xFFFFu;
#ifdef H2Y_EXP_NOCOMPUTE /* timing experiment only (wrong bytes) */
                        Y[col] = hg >> 5; Cb[col] = hb >> 5; Cr[col] = hr >> 5;
                        continue;
#endif
                        const float g = s_lut_y[hg & (H2Y_LUT16_N - 1)], b = s_lut_c[hb & (H2Y_LUT16_N - 1)], rr = s_lut_c[hr & (H2Y_LUT16_N - 1)];
                        __builtin_amdgcn_sched_barrier(0);
                        bool um;
                        pix_matrix<MODE, false>(pp, g, b, rr, Y[col], Cb[col], Cr[col], &um);
                        /* a sample of the row outside the table, or the division guard (lane masks ORed in scalar registers) */
                        const uint64_t fm = __builtin_amdgcn_ballot_w64(um) | row_out_m;
                        if (__builtin_expect(fm != 0, 0)) {
                            const bool fl = ((fm >> lane) & 1u) != 0;
                            const float fg = ff_half_to_float(hg), fb = ff_half_to_float(hb), fr = ff_half_to_float(hr);
                            if (fl) {
                                const ycc k = pixel_careful<MODE>(&s_pp, fg, fb, fr);
                                Y[col] = k.y; Cb[col] = k.cb; Cr[col] = k.cr;
                            }
                            flagged += (uint32_t)__popcll(fm);
                            low_m |= __builtin_amdgcn_ballot_w64(fl && min3f(fg, fb, fr) <= -1.0f);
                        }
                    } else {
#ifdef H2Y_EXP_NOCOMPUTE /* timing experiment only (wrong bytes): the step's loads, FIR stages and stores without the pixel arithmetic */
                    Y[col] = f2bits(gv[col]) >> 21; Cb[col] = f2bits(bv[col]) >> 21; Cr[col] = f2bits(rv[col]) >> 21;
                    continue;
#endif
                    const float Gn = norm1<PIPE>(pp, 0, gv[col]), Bn = norm1<PIPE>(pp, 1, bv[col]), Rn = norm1<PIPE>(pp, 2, rv[col]);
                    const pq_rec1 cg = pq_t1_fetch(Gn, s_t1), cb = pq_t1_fetch(Bn, s_t1), cr = pq_t1_fetch(Rn, s_t1);
                    __builtin_amdgcn_sched_barrier(0);
                    float mg, mb, mr;
                    const float g = pix_scale(pq_t1_eval_m(Gn, cg, &mg), pp.mulY, pp.addY);
                    const float b = pix_scale(pq_t1_eval_m(Bn, cb, &mb), pp.mulC, pp.addC);
                    const float rr = pix_scale(pq_t1_eval_m(Rn, cr, &mr), pp.mulC, pp.addC);
                    bool ra, rb;
                    pix_matrix_t1<MODE>(pp, sn, g, b, rr, pq_t1_unsure3(mg, mb, mr), Y[col], Cb[col], Cr[col], &ra, &rb);
                    const uint64_t fm = __builtin_amdgcn_ballot_w64(ra) | __builtin_amdgcn_ballot_w64(rb);
                    if (__builtin_expect(fm != 0, 0)) {
                        const bool ug = pq_t1_unsure(mg), ub = pq_t1_unsure(mb), ur = pq_t1_unsure(mr);
                        /* some lane's pixel here is not settled: the binary64 tier for this pixel position, all lanes
                         * together (no memory operation in this branch but LDS reads and -- samples below the tables -- scalar
                         * loads).  Only the planes that hold an unsure sample of a flagged lane go through it: a sample the first
                         * tier was sure of IS the reference's float, and most unsettled pixels have one unsure sample, not three */
                        uint32_t Y2, Cb2, Cr2;
                        const bool fl = ((fm >> lane) & 1u) != 0;
                        float g2 = g, b2 = b, r2 = rr;
                        bool un = false;
                        if (__builtin_amdgcn_ballot_w64(fl & ug) != 0) {
                            bool sl;
                            const float t = pq_ext_inline(Gn, pq_fast(Gn, sA, sB, &sl), sl, pp.pq_ext);
                            g2 = (fl & ug) ? pix_scale(t, pp.mulY, pp.addY) : g2;
                            un |= fl & ug & sl;
                        }
                        if (__builtin_amdgcn_ballot_w64(fl & ub) != 0) {
                            bool sl;
                            const float t = pq_ext_inline(Bn, pq_fast(Bn, sA, sB, &sl), sl, pp.pq_ext);
                            b2 = (fl & ub) ? pix_scale(t, pp.mulC, pp.addC) : b2;
                            un |= fl & ub & sl;
                        }
                        if (__builtin_amdgcn_ballot_w64(fl & ur) != 0) {
                            bool sl;
                            const float t = pq_ext_inline(Rn, pq_fast(Rn, sA, sB, &sl), sl, pp.pq_ext);
                            r2 = (fl & ur) ? pix_scale(t, pp.mulC, pp.addC) : r2;
                            un |= fl & ur & sl;
                        }
                        bool um;
                        pix_matrix<MODE, false>(pp, g2, b2, r2, Y2, Cb2, Cr2, &um);
                        un |= um;
                        if (__builtin_expect(fl & un, 0)) {
                            const ycc k = pixel_careful<MODE>(&s_pp, Gn, Bn, Rn);
                            Y2 = k.y; Cb2 = k.cb; Cr2 = k.cr;
                        }
                        Y[col] = fl ? Y2 : Y[col];
                        Cb[col] = fl ? Cb2 - pp.half_m1 : Cb[col]; /* raw, as the first tier's */
                        Cr[col] = fl ? Cr2 - pp.half_m1 : Cr[col];
                        flagged += (uint32_t)__popcll(fm);
                        if (PIPE == H2Y_PIPE_PQ_IDENT) low_m |= __builtin_amdgcn_ballot_w64(fl && min3f(Gn, Bn, Rn) <= -1.0f);
                    }
                    }
                }
                yp[row][0] = pix_yuv_clamp<true>(pp, Y[0], false) | (pix_yuv_clamp<true>(pp, Y[1], false) << 16);
                yp[row][1] = pix_yuv_clamp<true>(pp, Y[2], false) | (pix_yuv_clamp<true>(pp, Y[3], false) << 16);
#if defined(H2Y_EXP_NOCOMPUTE) && H2Y_EXP_NOCOMPUTE >= 2 /* ... nor the FIR stages: the access pattern alone */
                n_cb[row][0] = Cb[0]; n_cb[row][1] = Cb[2]; n_cr[row][0] = Cr[0]; n_cr[row][1] = Cr[2];
#else
                ff_hstage<TIER == FF_TIER_T1>(Cb, e, maxcv, hk0, n_cb[row][0], n_cb[row][1]);
                ff_hstage<TIER == FF_TIER_T1>(Cr, e, maxcv, hk0, n_cr[row][0], n_cr[row][1]);
#endif
                if (row == 0) { /* row 0 of the next step, into the registers just read for the last time */
                    ff_load<IN_KIND>(io.in[0], q0n, v.g0);
                    ff_load<IN_KIND>(io.in[1], q0n, v.b0);
                    ff_load<IN_KIND>(io.in[2], q0n, v.r0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            gstore_nt<u32x2>(io.out, q0, u32x2{yp[0][0], yp[0][1]});
            gstore_nt<u32x2>(io.out, q0 + WQ, u32x2{yp[1][0], yp[1][1]});
            ff_load<IN_KIND>(io.in[0], q0n + WQ, v.g1);
            ff_load<IN_KIND>(io.in[1], q0n + WQ, v.b1);
            ff_load<IN_KIND>(io.in[2], q0n + WQ, v.r1);

            if (s >= H2) { /* below the picture: both new rows are its last row (convert.cpp:337-347) */
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    n_cb[0][k] = n_cb[1][k] = hb[k].last;
                    n_cr[0][k] = n_cr[1][k] = hr[k].last;
                }
            }
            if (s == 0u) { /* above the picture: row 0 (the same clamps) */
#pragma unroll
                for (int k = 0; k < 2; k++) { hb[k].fill(n_cb[0][k]); hr[k].fill(n_cr[0][k]); }
            }
            /* chroma row j = s - 3 from rows 2j-5 .. 2j+6 = history + the new row 2s.  Before the history is whole
             * (j < j0) the value is not yet that row's: it goes to row j0, which this lane overwrites in order */
#if defined(H2Y_EXP_NOCOMPUTE) && H2Y_EXP_NOCOMPUTE >= 2
            const uint32_t ocb = n_cb[0][0] | (n_cb[1][1] << 16), ocr = n_cr[0][0] | (n_cr[1][1] << 16);
#else
            const uint32_t ocb = hb[0].out(n_cb[0][0], clo, chi) | (hb[1].out(n_cb[0][1], clo, chi) << 16);
            const uint32_t ocr = hr[0].out(n_cr[0][0], clo, chi) | (hr[1].out(n_cr[0][1], clo, chi) << 16);
#endif
            const uint32_t jj = (s >= j0 + 3u) ? s - 3u : j0;
            /* dword index in a chroma plane: two samples per quad column.  Lanes that own no chroma column of this
             * strip carry an index far beyond the frame: the buffer store drops them (no branch, no select) */
#if FF_BUFFER_STORE
            const uint32_t cidx = jj * WQ + qx_store;
            __builtin_amdgcn_raw_buffer_store_b32(ocb, out_rsrc, (int)(((npix >> 1) + cidx) << 2), 0, 2 /* nt */);
            __builtin_amdgcn_raw_buffer_store_b32(ocr, out_rsrc, (int)((((npix + ncb) >> 1) + cidx) << 2), 0, 2 /* nt */);
#else
            uint32_t cidx = jj * WQ + qx;
            /* lanes that own no chroma column of this strip store lane 2's dword once more */
            const uint32_t ocb2 = __builtin_amdgcn_readlane(ocb, 2), ocr2 = __builtin_amdgcn_readlane(ocr, 2), cidx2 = __builtin_amdgcn_readlane(cidx, 2);
            cidx = own ? cidx : cidx2;
            gstore_nt<uint32_t>(io.out, (npix >> 1) + cidx, own ? ocb : ocb2);
            gstore_nt<uint32_t>(io.out, ((npix + ncb) >> 1) + cidx, own ? ocr : ocr2);
#endif
#pragma unroll
            for (int k = 0; k < 2; k++) {
                hb[k].push(n_cb[0][k], n_cb[1][k]);
                hr[k].push(n_cr[0][k], n_cr[1][k]);
            }
            q0 = q0n;
        }
        /* drain the loads the last step asked for (their registers are reused by the next unit's first loads) */
#pragma unroll
        for (int j = 0; j < NREG; j++) asm volatile("" ::"v"(v.g0[j]), "v"(v.b0[j]), "v"(v.r0[j]), "v"(v.g1[j]), "v"(v.b1[j]), "v"(v.r1[j]));
        if constexpr (RAW) {
#pragma unroll
            for (int c = 0; c < 3; c++) { /* fold the two packed halves, widen; pic_stats' initial values when nothing beat them (k_fused_lut16) */
                mm.lo[c] = fminf(ff_half_to_float(hmn[c] & 0xFFFFu), ff_half_to_float(hmn[c] >> 16));
                mm.hi[c] = fmaxf(ff_half_to_float(hmx[c] & 0xFFFFu), ff_half_to_float(hmx[c] >> 16));
                mm.lo[c] = mm.lo[c] > 65504.0f ? 3.402823466e+38f : mm.lo[c];
                mm.hi[c] = mm.hi[c] <= 0.0f ? 1.175494351e-38f : mm.hi[c];
            }
        }
        asm volatile("" : "+s"(ap));
        const uint32_t f_end = __builtin_amdgcn_readfirstlane(u / ap->units_per_frame); /* (again, rather than carried through the loop) */
        const size_t slot = (size_t)u; /* = f * units_per_frame + r */
        wave_store_mm(mm, ap->partial + slot * 6);
        if (lane == 0) {
            ap->redo_count[slot] = flagged >> 3; /* in tiles of eight pixels, the unit the host steers by */
            if (PIPE == H2Y_PIPE_PQ_IDENT && ap->low_flag && low_m != 0) ap->low_flag[f_end] = 1u;
        }
    }
    /* NO BARRIER MAY FOLLOW THE UNIT LOOP.  The pacing s_barrier inside the step loop is executed a different number of times by
     * each wave (units of different length; waves without a unit leave at once): that is outside the HIP programming model and
     * works because gfx950's barrier counts only the waves of the block that have not ended, and because nothing after the loop
     * waits on one.  A __syncthreads() here would pair with another wave's pacing barrier (wrong hand-over) or never complete.
     * tests/test_gpu_parity.py::test_fir_fused_kernel_geometries (blocks whose units differ in length) is the guard. */
    if (a.block_clock && (threadIdx.x & (WAVE - 1)) == 0) atomicMax(&a.block_clock[2 * blockIdx.x + 1], (unsigned long long)wall_clock64());
#ifdef H2Y_BLOCK_TIMES
    if ((threadIdx.x & (WAVE - 1)) == 0) atomicMax(&g_ff_block_times[2 * blockIdx.x + 1], (unsigned long long)wall_clock64());
#endif
}

/* ---- launch ------------------------------------------------------------- */
typedef void (*firf_fn)(firf_args);
template <int IN_KIND> static firf_fn pick_firf(int mode, int pipe)
{
    if (mode == H2Y_MODE_YCBCR)
        return pipe == H2Y_PIPE_PQ_IDENT ? k_fir_fused<IN_KIND, H2Y_MODE_YCBCR, H2Y_PIPE_PQ_IDENT, FF_TIER_T1>
                                         : k_fir_fused<IN_KIND, H2Y_MODE_YCBCR, H2Y_PIPE_PQ_NORM, FF_TIER_T1>;
    return pipe == H2Y_PIPE_PQ_IDENT ? k_fir_fused<IN_KIND, H2Y_MODE_YDZDX, H2Y_PIPE_PQ_IDENT, FF_TIER_T1>
                                     : k_fir_fused<IN_KIND, H2Y_MODE_YDZDX, H2Y_PIPE_PQ_NORM, FF_TIER_T1>;
}

hipError_t h2y_launch_fir_fused(int in_kind, int mode, bool ident, bool lut16, int grid, hipStream_t st, const firf_args &a)
{
    const int pipe = ident ? H2Y_PIPE_PQ_IDENT : H2Y_PIPE_PQ_NORM;
    firf_fn fn;
    if (lut16) /* half input, floor 0 / ceiling 1 */
        fn = mode == H2Y_MODE_YCBCR ? k_fir_fused<H2Y_IN_F16, H2Y_MODE_YCBCR, H2Y_PIPE_PQ_IDENT, FF_TIER_LUT16>
                                    : k_fir_fused<H2Y_IN_F16, H2Y_MODE_YDZDX, H2Y_PIPE_PQ_IDENT, FF_TIER_LUT16>;
    else fn = in_kind == H2Y_IN_F16 ? pick_firf<H2Y_IN_F16>(mode, pipe) : pick_firf<H2Y_IN_F32>(mode, pipe);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(FF_THREADS), 0, st, a);
    return hipGetLastError();
}
