/*
 * h2y_kernels.hip -- gfx950 kernels of the convert path.
 *
 *   k_stats        pic_stats()            common.cpp:66-139   (pre-pass form)
 *   the fused kernels: matrix_convert()   convert.cpp:879-1221
 *                  + convert() box / 4:4:4 convert.cpp:802-859 (+ :91-172)
 *                  + write_yuv() clamp     tiff.cpp:457-550
 *                  + pic_stats() min/max of the same samples, as a by-product
 *     k_fused_t1     LINEAR -> PQ, float input, <= 12 bits: binary32 first tier, the rest redone in place
 *     k_fused2       the binary64 tier for every sample (16 bits, u16 input), or no PQ at all (equal transfers)
 *     k_fused_lut16  half-float input: PQ of all 16 384 halves in LDS
 *     k_fused        generic form (runtime matrix / transfer flags, odd heights), k_fused_narrow (width % 4 != 0)
 *   k_fir420       Subsample444to420_FIR  convert.cpp:261-383 + write_yuv clamp
 *   k_box420       Subsample444to420_box  convert.cpp:91-172 (stage entry only)
 *   k_inverse      matrix_inverse()       convert.cpp:1320-1867
 *   k_stats_final  (int) floor/ceiling    common.cpp:135-136, and the check of
 *                  the values the fused kernel assumed against the ones it measured
 *
 * The path is HBM-bound elementwise work: no MFMA.  Layout in HBM is the
 * reference's: three planar row-major planes in (G,B,R) order, stride = width;
 * output is one .yuv frame (Y plane, Cb plane, Cr plane, little-endian u16).
 *
 * Built with -ffp-contract=off: the reference's bytes depend on products and
 * sums being rounded separately; fma() appears only where
 * h2y_math.h asks for it by name.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>

#include "h2y_math.h"
#include "h2y_kernels.h"
#include "h2y_walk.h"
#include "h2y_device.h"

/* ---- the thread tile: 4 columns x 2 rows ------------------------------- */
/* Positions are kept in QUADS (four samples: one 16-byte load of floats, one 8-byte store of code values):
 * tile tt = rp * WQ + cg (row pair, column group; WQ = W / 4) starts at quad q0 = 2 rp WQ + cg = 2 tt - cg,
 * and its two 4:2:0 chroma samples are dword tt of the chroma plane -- only cg = tt mod WQ costs a division. */
struct tile_pos {
    uint32_t tt;     /* tile number within the frame */
    uint32_t q0, q1; /* quad index of row 0 / row 1 (row 1 == row 0 when the picture ends); pictures hold < 2^28 samples */
    bool row1;
};
__device__ __forceinline__ tile_pos tile_locate(uint32_t tt, uint32_t W, uint32_t H, uint32_t WQ, uint32_t magic)
{
    tile_pos t;
    uint32_t cg;
    (void)udiv_magic(tt, WQ, magic, cg);
    t.tt = tt;
    t.q0 = 2u * tt - cg;
    t.row1 = tt < WQ * (H >> 1); /* odd height: the last row of tiles has one picture row */
    t.q1 = t.row1 ? t.q0 + WQ : t.q0;
    return t;
}
struct tile_in {
    float g0[4], b0[4], r0[4], g1[4], b1[4], r1[4];
};
template <int IN_KIND> __device__ __forceinline__ void tile_load(const frame_io &io, const tile_pos &t, tile_in &v)
{
    typedef in_traits<IN_KIND> IN;
    IN::load4q(io.in[0], t.q0, v.g0);
    IN::load4q(io.in[1], t.q0, v.b0);
    IN::load4q(io.in[2], t.q0, v.r0);
    IN::load4q(io.in[0], t.q1, v.g1);
    IN::load4q(io.in[1], t.q1, v.b1);
    IN::load4q(io.in[2], t.q1, v.r1);
}
/* packed results of a tile, ready to store */
struct tile_out {
    uint32_t yp0[2], yp1[2];                     /* Y rows, two u16 per dword */
    uint32_t cbp0[2], cbp1[2], crp0[2], crp1[2]; /* 4:4:4 chroma rows */
    uint32_t cb_box, cr_box;                     /* 4:2:0 box chroma, two samples each */
};
template <int OUT_KIND>
__device__ __forceinline__ void tile_pack(const pix_params &pp, int jb, const uint32_t (&Y)[4], uint32_t (&Cb)[4],
                                          uint32_t (&Cr)[4], tile_out &o)
{
    o.yp0[jb] = pix_yuv_clamp(pp, Y[0], false) | (pix_yuv_clamp(pp, Y[1], false) << 16);
    o.yp1[jb] = pix_yuv_clamp(pp, Y[2], false) | (pix_yuv_clamp(pp, Y[3], false) << 16);
    if (OUT_KIND == H2Y_OUT_420BOX) {
        /* convert.cpp:157-160: (a+b+c+d)/4, unsigned truncation; then write_yuv's clamp */
        uint32_t cb = pix_box_clamp(pp, Cb[0] + Cb[1] + Cb[2] + Cb[3]);
        uint32_t cr = pix_box_clamp(pp, Cr[0] + Cr[1] + Cr[2] + Cr[3]);
        if (jb == 0) { o.cb_box = cb; o.cr_box = cr; }
        else { o.cb_box |= cb << 16; o.cr_box |= cr << 16; }
    } else {
        if (OUT_KIND == H2Y_OUT_444) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                Cb[q] = pix_yuv_clamp(pp, Cb[q], true);
                Cr[q] = pix_yuv_clamp(pp, Cr[q], true);
            }
        }
        o.cbp0[jb] = Cb[0] | (Cb[1] << 16); o.cbp1[jb] = Cb[2] | (Cb[3] << 16);
        o.crp0[jb] = Cr[0] | (Cr[1] << 16); o.crp1[jb] = Cr[2] | (Cr[3] << 16);
    }
}
/* the same packing one picture row (four pixels) at a time; sb/sr carry the 2x2 box sums from row 0 to row 1 */
/* NOSHIFT: write_yuv's down shift is known to be zero (float input: the temporary picture has the output's depth)
 * RAWC: the chroma comes raw from pix_matrix_t1() (signed, without "+ Half - 1"; no clamp needed): the offset goes in
 * once per 2x2 box, as the third operand of the first row's sum */
template <int OUT_KIND, bool NOSHIFT = false, bool RAWC = false>
__device__ __forceinline__ void row_pack(const pix_params &pp, int row, const uint32_t (&Y)[4], uint32_t (&Cb)[4], uint32_t (&Cr)[4],
                                         tile_out &o, uint32_t (&sb)[2], uint32_t (&sr)[2])
{
    uint32_t(&yp)[2] = row ? o.yp1 : o.yp0;
    yp[0] = pix_yuv_clamp<NOSHIFT>(pp, Y[0], false) | (pix_yuv_clamp<NOSHIFT>(pp, Y[1], false) << 16);
    yp[1] = pix_yuv_clamp<NOSHIFT>(pp, Y[2], false) | (pix_yuv_clamp<NOSHIFT>(pp, Y[3], false) << 16);
    if (OUT_KIND == H2Y_OUT_420BOX) {
        if (row == 0) {
            const uint32_t k4 = RAWC ? pp.half_m1 << 2 : 0u;
            sb[0] = Cb[0] + Cb[1] + k4; sb[1] = Cb[2] + Cb[3] + k4;
            sr[0] = Cr[0] + Cr[1] + k4; sr[1] = Cr[2] + Cr[3] + k4;
        } else {
            /* convert.cpp:157-160: (a+b+c+d)/4, unsigned truncation; then write_yuv's clamp */
            o.cb_box = pix_box_clamp<NOSHIFT>(pp, sb[0] + Cb[0] + Cb[1]) | (pix_box_clamp<NOSHIFT>(pp, sb[1] + Cb[2] + Cb[3]) << 16);
            o.cr_box = pix_box_clamp<NOSHIFT>(pp, sr[0] + Cr[0] + Cr[1]) | (pix_box_clamp<NOSHIFT>(pp, sr[1] + Cr[2] + Cr[3]) << 16);
        }
    } else {
        if (RAWC) {
#pragma unroll
            for (int q = 0; q < 4; q++) { Cb[q] += pp.half_m1; Cr[q] += pp.half_m1; }
        }
        if (OUT_KIND == H2Y_OUT_444) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                Cb[q] = pix_yuv_clamp<NOSHIFT>(pp, Cb[q], true);
                Cr[q] = pix_yuv_clamp<NOSHIFT>(pp, Cr[q], true);
            }
        }
        uint32_t(&cbp)[2] = row ? o.cbp1 : o.cbp0;
        uint32_t(&crp)[2] = row ? o.crp1 : o.crp0;
        cbp[0] = Cb[0] | (Cb[1] << 16); cbp[1] = Cb[2] | (Cb[3] << 16);
        crp[0] = Cr[0] | (Cr[1] << 16); crp[1] = Cr[2] | (Cr[3] << 16);
    }
}
/* offsets in units of the store (u32x2: a quad of code values; dword: two 4:2:0 chroma samples) from
 * the start of the frame's output: < 2^32 bytes.  W % 4 == 0, so W * H and the chroma plane's size are even. */
template <int OUT_KIND>
__device__ __forceinline__ void tile_store(const frame_io &io, const tile_pos &t, uint32_t W, uint32_t H, const tile_out &o)
{
    const uint32_t npix = W * H;
    gstore_nt<u32x2>(io.out, t.q0, u32x2{o.yp0[0], o.yp0[1]});
    if (t.row1) gstore_nt<u32x2>(io.out, t.q1, u32x2{o.yp1[0], o.yp1[1]});
    if (OUT_KIND == H2Y_OUT_420BOX) {
        /* the tile's two chroma samples: index rp * (W / 2) + x / 2 = 2 tt, i.e. dword tt of each plane */
        const uint32_t ncb = (W >> 1) * (H >> 1);
        gstore_nt<uint32_t>(io.out, (npix >> 1) + t.tt, o.cb_box);
        gstore_nt<uint32_t>(io.out, ((npix + ncb) >> 1) + t.tt, o.cr_box);
    } else {
        uint16_t *Cbp = OUT_KIND == H2Y_OUT_444 ? io.out + npix : io.tmp_cb;
        uint16_t *Crp = OUT_KIND == H2Y_OUT_444 ? io.out + 2 * (size_t)npix : io.tmp_cr;
        if (OUT_KIND == H2Y_OUT_444) {
            gstore_nt<u32x2>(Cbp, t.q0, u32x2{o.cbp0[0], o.cbp0[1]});
            gstore_nt<u32x2>(Crp, t.q0, u32x2{o.crp0[0], o.crp0[1]});
            if (t.row1) {
                gstore_nt<u32x2>(Cbp, t.q1, u32x2{o.cbp1[0], o.cbp1[1]});
                gstore_nt<u32x2>(Crp, t.q1, u32x2{o.crp1[0], o.crp1[1]});
            }
        } else { /* scratch planes: k_fir420 reads them next */
            gstore<u32x2>(Cbp, t.q0, u32x2{o.cbp0[0], o.cbp0[1]});
            gstore<u32x2>(Crp, t.q0, u32x2{o.crp0[0], o.crp0[1]});
            if (t.row1) {
                gstore<u32x2>(Cbp, t.q1, u32x2{o.cbp1[0], o.cbp1[1]});
                gstore<u32x2>(Crp, t.q1, u32x2{o.crp1[0], o.crp1[1]});
            }
        }
    }
}

/* exact tier for one tile.  Pixels are taken two at a time (the two rows of
 * one column): the fast tier of both is one straight line of code, so six
 * samples' table loads are in flight together, and one rarely-taken branch
 * covers the careful tier of the pair.  (One block for all eight pixels makes
 * the register allocator spill; one pixel at a time exposes the LDS latency.) */
#ifndef H2Y_PAIR
#define H2Y_PAIR 1
#endif
template <int OUT_KIND, int MODE, int PIPE>
__device__ __forceinline__ void tile_exact(const pix_params &pp, const pix_params *spp, const pq_recA *sA, const pq_recB *sB,
                                           tile_in &v, tile_out &o)
{
#pragma unroll
    for (int jb = 0; jb < 2; jb++) {
        uint32_t Y[4], Cb[4], Cr[4];
#if H2Y_PAIR
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int col = 2 * jb + c;
            const float G0 = norm1<PIPE>(pp, 0, v.g0[col]), B0 = norm1<PIPE>(pp, 1, v.b0[col]), R0 = norm1<PIPE>(pp, 2, v.r0[col]);
            const float G1 = norm1<PIPE>(pp, 0, v.g1[col]), B1 = norm1<PIPE>(pp, 1, v.b1[col]), R1 = norm1<PIPE>(pp, 2, v.r1[col]);
            const bool u0 = pixel_fast<MODE, PIPE>(pp, sA, sB, G0, B0, R0, Y[c], Cb[c], Cr[c]);
            const bool u1 = pixel_fast<MODE, PIPE>(pp, sA, sB, G1, B1, R1, Y[2 + c], Cb[2 + c], Cr[2 + c]);
            if (__builtin_expect(u0 | u1, 0)) {
                if (u0) {
                    const ycc k = pixel_careful<MODE>(spp, G0, B0, R0);
                    Y[c] = k.y; Cb[c] = k.cb; Cr[c] = k.cr;
                }
                if (u1) {
                    const ycc k = pixel_careful<MODE>(spp, G1, B1, R1);
                    Y[2 + c] = k.y; Cb[2 + c] = k.cb; Cr[2 + c] = k.cr;
                }
            }
        }
#else
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int col = 2 * jb + (q & 1);
            const float G = norm1<PIPE>(pp, 0, q < 2 ? v.g0[col] : v.g1[col]);
            const float B = norm1<PIPE>(pp, 1, q < 2 ? v.b0[col] : v.b1[col]);
            const float R = norm1<PIPE>(pp, 2, q < 2 ? v.r0[col] : v.r1[col]);
            if (__builtin_expect(pixel_fast<MODE, PIPE>(pp, sA, sB, G, B, R, Y[q], Cb[q], Cr[q]), 0)) {
                const ycc k = pixel_careful<MODE>(spp, G, B, R);
                Y[q] = k.y; Cb[q] = k.cb; Cr[q] = k.cr;
            }
        }
#endif
        tile_pack<OUT_KIND>(pp, jb, Y, Cb, Cr, o);
    }
}

/*
 * k_fused: one thread = 4 columns x 2 rows of one frame, exact tier only.
 *
 * A chunk is blockDim.x consecutive thread-tiles of ONE frame (chunks never
 * straddle frames); global chunk g belongs to block g % gridDim.x, so a block
 * walks the frames in order and all its threads are always in the same frame.
 * After its last chunk of a frame the block reduces the min/max its threads
 * saw and stores six floats to partial[frame][block] -- every (frame, block)
 * slot is written exactly once per launch, no atomics, no initialisation.
 *
 * OUT_KIND:
 *   H2Y_OUT_420BOX  Y final; Cb/Cr = truncating mean of the thread's own two
 *                   2x2 blocks (convert.cpp:157-160), final
 *   H2Y_OUT_444     Y, Cb, Cr final, full resolution
 *   H2Y_OUT_444TMP  Y final; Cb/Cr = matrix_convert() output (NOT yet
 *                   range-clamped) into scratch planes for k_fir420
 * MODE: H2Y_MODE_YCBCR / H2Y_MODE_YDZDX compiled in, or H2Y_MODE_RUNTIME.
 */
template <int IN_KIND, int OUT_KIND, int MODE, int PIPE>
__global__ __launch_bounds__(H2Y_FUSED_THREADS, PIPE == H2Y_PIPE_RUNTIME ? 2 : H2Y_FUSED_MINWAVES) void k_fused(fused_args a) /* (the runtime form holds two tables: one block of 512 per CU) */
{
    /* A records, then B records; the runtime form has room for a second table (the two stages of a generic transfer pair) */
    __shared__ pq_recA s_tab[(PIPE == H2Y_PIPE_RUNTIME ? 4 : 2) * H2Y_PQ_NREC];
    const pq_recA *sA = s_tab;
    const pq_recB *sB = reinterpret_cast<const pq_recB *>(s_tab + H2Y_PQ_NREC);
    __shared__ pix_params s_pp;
    if (PIPE != H2Y_PIPE_RUNTIME || a.pp.convert_transfer == 1) stage_table<H2Y_FUSED_THREADS>(a.table, s_tab);
    if (PIPE == H2Y_PIPE_RUNTIME && a.pp.convert_transfer == 2) {
        if (a.table_src) stage_table<H2Y_FUSED_THREADS>(a.table_src, s_tab);
        if (a.table_dst) stage_table<H2Y_FUSED_THREADS>(a.table_dst, s_tab + 2 * H2Y_PQ_NREC);
    }
    const pix_params pp = with_assumed(a.pp, a.assumed);
    if (threadIdx.x == 0) s_pp = pp;
    __syncthreads();

    const uint32_t W = a.width, H = a.height, G = gridDim.x;
    for (int f = 0; f < a.n_frames; f++) {
        const frame_io io = uniform_io(a.frames + f);
        mm6 mm;
        mm.reset();
        /* first chunk of frame f owned by this block */
        const uint32_t gbase = (uint32_t)(((uint64_t)f * a.chunks_per_frame) % G);
        uint32_t k = (blockIdx.x + G - gbase) % G;
        for (; k < a.chunks_per_frame; k += G) {
            const uint32_t tt = k * H2Y_FUSED_THREADS + threadIdx.x;
            if (tt >= a.tiles_per_frame) continue;
            const tile_pos t = tile_locate(tt, W, H, a.wq, a.wq_magic);
            tile_in v;
            tile_load<IN_KIND>(io, t, v);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                mm.add2(0, v.g0[j], v.g1[j]);
                mm.add2(1, v.b0[j], v.b1[j]);
                mm.add2(2, v.r0[j], v.r1[j]);
            }
            tile_out o;
            tile_exact<OUT_KIND, MODE, PIPE>(pp, &s_pp, sA, sB, v, o);
            tile_store<OUT_KIND>(io, t, W, H, o);
        }
        wave_store_mm(mm, a.partial + (((size_t)f * G + blockIdx.x) * (H2Y_FUSED_THREADS / WAVE) + threadIdx.x / WAVE) * 6);
    }
}

/* frame_walk (h2y_walk.h): which frames a block works on and which chunks of them -- frame groups, XCD-aware
 * layout.  Glue to the launch arguments: */
__device__ __forceinline__ void walk_init(frame_walk &fw, const fused_args &a)
{
    walk_args wa;
    wa.groups = a.groups; wa.xcd_layout = a.xcd_layout;
    wa.chunks_per_frame = a.chunks_per_frame;
    wa.n_frames = a.n_frames;
    fw.init(wa, blockIdx.x, gridDim.x);
}
/* this wave's slot in the per-frame arrays: [frame][block of the group][wave] */
__device__ __forceinline__ size_t walk_slot(const frame_walk &fw, uint32_t waves) { return fw.slot() * waves + threadIdx.x / WAVE; }

/*
 * Waves take their tiles by ticket.
 *
 * The SIMD arbiter serves its oldest wave first.  With a fixed share per wave -- every wave one 64-tile
 * slice of every chunk -- the four waves of a SIMD do not finish together: measured on C2
 * (tools/blocktimes.py), the oldest wave of each SIMD left the frame loop at 57 % of the launch, the next
 * at 70 %, the third at 87 %, and the last ran alone, at a third of the four-wave rate.  So the slices of
 * a block's chunks are not tied to waves: a frame's slices are numbered (slice i = chunk i / WPB of the
 * block's chunks of that frame, 64-tile part i % WPB) and a wave draws the next number from a counter in
 * LDS (one ds_add_rtn per tile, claimed one tile ahead so that the prefetch knows its target).  Old
 * waves simply draw more often; all leave a frame within one tile of each other.  One counter per frame
 * of the group (waves drift across frame boundaries), zeroed at kernel start: H2Y_CLAIM_FRAMES bounds the
 * frames of a group per launch (the host splits longer batches).
 */
#define H2Y_CLAIM_FRAMES 128
struct wave_deal { /* a block's share of one frame, in slices */
    uint32_t k0, G, total; /* round robin: first chunk and stride; slices in all */
    uint32_t s0;           /* ranged form: the block's first slice */
    bool ranged;
    __device__ __forceinline__ void set(const frame_walk &fw, uint32_t k, uint32_t wpb)
    {
        k0 = k; G = fw.G;
        total = fw.count(k) * wpb;
        ranged = false;
        s0 = 0;
    }
    /* Ranged form (fused_args.slice_ranges): the block owns the slices [first, first + count) of every frame of its
     * group -- one contiguous run of 64-tile slices, as long as its XCD is fast (h2y_walk.h: slice_ranges()). */
    __device__ __forceinline__ void set_range(uint32_t first, uint32_t count)
    {
        k0 = 0; G = 1;
        s0 = first;
        total = count;
        ranged = true;
    }
    /* first tile of slice i (WPB slices of 64 tiles per chunk of THREADS tiles) */
    template <int THREADS> __device__ __forceinline__ uint32_t tile0(uint32_t i) const
    {
        constexpr uint32_t WPB = THREADS / WAVE;
        if (ranged) return (s0 + i) * WAVE;
        const uint32_t j = i / WPB, sub = i % WPB;
        return (k0 + j * G) * THREADS + sub * WAVE;
    }
};
/* this block's run of slices, if the launch deals by ranges: [group-relative block number] and the next entry */
__device__ __forceinline__ bool block_range(const fused_args &a, const frame_walk &fw, uint32_t &first, uint32_t &count)
{
    first = count = 0;
    if (!a.slice_ranges) return false;
    const uint32_t *r = a.slice_ranges + (size_t)a.range_stride * (uint32_t)fw.f; /* (before the frame loop fw.f is the group's number) */
    first = __builtin_amdgcn_readfirstlane(r[fw.bi]);
    count = __builtin_amdgcn_readfirstlane(r[fw.bi + 1u]) - first;
    return true;
}
/* draw a number: lane 0 adds to the counter, the other lanes to scratch words of their own (no branch) */
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t wave_claim(uint32_t *ctr, uint32_t *scratch /* [WAVE] */, bool real)
{
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    uint32_t *p = (real && lane == 0) ? ctr : scratch + lane;
    const uint32_t v = __hip_atomic_fetch_add((lds_u32 *)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_readfirstlane(v);
}

/* the two halves of wave_claim(): the add goes out early, its result is read when the prefetch needs it */
__device__ __forceinline__ uint32_t wave_claim_issue(uint32_t *ctr, uint32_t *scratch)
{
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    uint32_t *p = lane == 0 ? ctr : scratch + lane;
    return __hip_atomic_fetch_add((lds_u32 *)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
/* The slice after the one in hand, from the number asked for with wave_claim_issue() at the top of the tile: of
 * this frame (kind 1), else -- this frame is dealt out -- of the group's next frame if there is one (kind 2), else
 * none (kind 0: the request repeats the current tile).  Returns the first tile of that slice. */
template <int THREADS>
__device__ __forceinline__ uint32_t ticket_resolve(uint32_t n1v, const wave_deal &deal, const wave_deal &deal_n, uint32_t *claim, uint32_t *scratch,
                                                   uint32_t fo, bool has_next, uint32_t tick, int *kind, uint32_t *tick2)
{
    const uint32_t n1 = __builtin_amdgcn_readfirstlane(n1v);
    if (__builtin_expect(n1 < deal.total, 1)) {
        *kind = 1;
        *tick2 = n1;
        return deal.tile0<THREADS>(n1);
    }
    const uint32_t n2 = wave_claim(&claim[(fo + 1u) & (H2Y_CLAIM_FRAMES - 1u)], scratch, has_next);
    const bool nextf = has_next && n2 < deal_n.total;
    *kind = nextf ? 2 : 0;
    *tick2 = nextf ? n2 : tick;
    return nextf ? deal_n.tile0<THREADS>(n2) : deal.tile0<THREADS>(tick);
}

/* finish time of the block, for the host's balancing: the latest of its waves */
__device__ __forceinline__ void block_clock_start(const fused_args &a)
{
    if (a.block_clock && threadIdx.x == 0) a.block_clock[2 * blockIdx.x] = wall_clock64();
}
__device__ __forceinline__ void block_clock_end(const fused_args &a)
{
    if (a.block_clock && (threadIdx.x & (WAVE - 1)) == 0) atomicMax(&a.block_clock[2 * blockIdx.x + 1], (unsigned long long)wall_clock64());
}

/*
 * k_fused2: k_fused's work (binary64 tier for every sample) in the loop form of k_fused_t1 below --
 * row-wise tiles with rolling prefetch, one basic block of memory operations, no divergence around
 * them (see the comment in k_fused_t1).  LINEAR -> PQ with compile-time MODE and PIPE, even height.
 * The rare cases stay inside the loop, as branches without memory operations: a sample whose
 * binary64 value is too close to a rounding tie (1 in 63 763, or outside the table) goes through
 * pq_slow() alone -- not its whole pixel --, a pixel whose reciprocal-division guard fires gets the
 * IEEE divisions.
 */
template <int PIPE>
__device__ __forceinline__ float pq_sample(const pix_params &pp, int c, float raw, const pq_recA *sA, const pq_recB *sB, bool &any_slow,
                                           const void *const *s_ext /* LDS: pp.pq_ext, read only here */)
{
    if (PIPE == H2Y_PIPE_NONE) return raw; /* convert.cpp:930: no transfer conversion, no normalisation, no scale step */
    const float x = norm1<PIPE>(pp, c, raw);
    bool slow;
    float v = pq_fast(x, sA, sB, &slow);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(slow) != 0, 0)) {
        v = pq_ext_inline(x, v, slow, *s_ext); /* below the LDS table: the full-range table by scalar loads, no wait for the prefetch */
        if (slow) v = pq_slow(x, *s_ext);
    }
    any_slow |= slow; /* through pq_slow(): the value may be NaN */
    return pix_scale(v, c == 0 ? pp.mulY : pp.mulC, c == 0 ? pp.addY : pp.addC);
}
/* the same from a record already fetched (x normalised) */
__device__ __forceinline__ float pq_sample_rec(const pix_params &pp, int c, float x, const pq_rec &rec, bool &any_slow, const void *const *s_ext)
{
    bool sl;
    float v = pq_eval(x, rec, &sl);
    const bool zero = f2bits(x) == 0u; /* as pq_fast(): +0.0 is outside the table but its value is a constant */
    const bool slow = sl & !zero;
    bool slow2 = slow;
    v = zero ? bits2f(H2Y_PQ_AT_ZERO_BITS) : v;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(slow) != 0, 0)) {
        v = pq_ext_inline(x, v, slow2, *s_ext);
        if (slow2) v = pq_slow(x, *s_ext);
    }
    any_slow |= slow2; /* through pq_slow(): the value may be NaN */
    return pix_scale(v, c == 0 ? pp.mulY : pp.mulC, c == 0 ? pp.addY : pp.addC);
}
/* H2Y_PIPE_TFN: the three samples of a pixel through ONE stage of a generic transfer pair (convert.cpp:1024-1109: source
 * function -> linear light, a float as in the reference -> destination function), the stage's table in LDS (tfn_fast's
 * arithmetic, with the six records of the pixel on their way before the first is used); what the table does not reach through
 * the stage's full-range table in global memory by scalar loads (pq_ext_inline: no wait for the prefetch); what is left
 * (ambiguous roundings, subnormals, NaNs) through the careful tier, sample by sample. */
/* what tfn_index() / tfn_fast() derive from the function's number, worked out once per kernel (scalar registers) */
struct tfn_consts {
    uint32_t lo_base, hi_base;   /* first float (bits) of the table's low part and of its finer high part (0x80000000: none) */
    uint32_t lo_shift, hi_shift; /* 23 - seg_bits of either part */
    uint32_t nseg_lo;            /* segments of the low part */
    uint32_t zero_bits, one_bits, lo_bits;
    uint32_t rho_h;              /* the source stage starts with RHO_GAMMA_f's inner powf */
    __device__ __forceinline__ void set(int fn)
    {
        const tfn_cut c = tfn_cut_of(fn);
        lo_base = (uint32_t)(127 + c.emin) << 23;
        hi_base = c.hi_emin < 1 ? (uint32_t)(127 + c.hi_emin) << 23 : 0x80000000u; /* (none: no non-negative float reaches it) */
        lo_shift = 23u - (uint32_t)c.seg_bits;
        hi_shift = 23u - (uint32_t)c.hi_seg_bits;
        nseg_lo = (uint32_t)tfn_nseg_lo(c);
        zero_bits = tfn_zero_bits(fn);
        one_bits = tfn_one_bits(fn);
        lo_bits = tfn_lo_bits(fn);
        rho_h = fn == H2Y_TFN_RHO_H;
    }
};
template <bool SRC>
__device__ __forceinline__ void tfn_stage3(const pix_params &pp, const tfn_consts &k, const pq_recA *tab, const void *ext, float (&x)[3])
{
    float x0[3];
    uint32_t bits[3], idx[3], low_bits[3];
    pq_recA ra[3];
    pq_recB rb[3];
    if (SRC && k.rho_h) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            x0[c] = x[c];
            x[c] = (powf25(x[c]) - 1.0f) * 0.0625f; /* RHO_GAMMA_f's inner powf, then (P - 1) / 16: both exact */
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; c++) x0[c] = x[c];
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        bits[c] = f2bits(x[c]);
        /* tfn_index(): everything outside the table (0, tiny, >= 2, negative, NaN) wraps above it and lands on the sentinel */
        const bool hi = bits[c] >= k.hi_base && (int32_t)bits[c] >= 0;
        const uint32_t i_lo = (bits[c] - k.lo_base) >> k.lo_shift, i_hi = k.nseg_lo + ((bits[c] - k.hi_base) >> k.hi_shift);
        idx[c] = umin32(hi ? i_hi : i_lo, (uint32_t)H2Y_PQ_NSEG);
        low_bits[c] = hi ? k.hi_shift : k.lo_shift;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        ra[c] = tab[idx[c]];
        rb[c] = reinterpret_cast<const pq_recB *>(tab + H2Y_PQ_NREC)[idx[c]];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double v = tfn_poly(bits[c], 23 - (int)low_bits[c], ra[c], rb[c]);
        const bool zero = bits[c] == 0u, one = k.one_bits != 0u && bits[c] == 0x3F800000u;
        bool slow = pq_ambiguous(v) & !(zero | one);
        float r = zero ? bits2f(k.zero_bits) : one ? bits2f(k.one_bits) : (float)v;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(slow) != 0, 0)) {
            r = pq_ext_inline(x[c], r, slow, ext, k.lo_bits);
            if (slow) r = SRC ? tf_to_linear_careful(pp.src_tf, x0[c]) : tf_from_linear_careful(pp.dst_tf, x0[c]);
        }
        x[c] = r;
    }
}
template <int IN_KIND, int OUT_KIND, int MODE, int PIPE>
__global__ __launch_bounds__(H2Y_LOOP_THREADS) void k_fused2(fused_args a)
{
    __shared__ pq_recA s_tab[PIPE == H2Y_PIPE_NONE ? 1 : (PIPE == H2Y_PIPE_TFN ? 4 : 2) * H2Y_PQ_NREC]; /* A records, then B records (TFN: of the source stage, then of the destination stage) */
    const pq_recA *sA = s_tab;
    const pq_recB *sB = reinterpret_cast<const pq_recB *>(s_tab + (PIPE == H2Y_PIPE_NONE ? 0 : H2Y_PQ_NREC));
    __shared__ uint32_t s_claim[H2Y_CLAIM_FRAMES], s_scratch[WAVE];
    __shared__ const void *s_ext; /* a.pp.pq_ext, for the rare pq_slow() */
    if (PIPE == H2Y_PIPE_TFN) {
        if (a.table_src) stage_table<H2Y_LOOP_THREADS>(a.table_src, s_tab);
        if (a.table_dst) stage_table<H2Y_LOOP_THREADS>(a.table_dst, s_tab + (PIPE == H2Y_PIPE_TFN ? 2 * H2Y_PQ_NREC : 0));
    } else if (PIPE != H2Y_PIPE_NONE) stage_table<H2Y_LOOP_THREADS>(a.table, s_tab);
    if (threadIdx.x < H2Y_CLAIM_FRAMES) s_claim[threadIdx.x] = 0u;
    if (threadIdx.x == 0) s_ext = a.pp.pq_ext;
    const pix_params pp = with_assumed(a.pp, a.assumed);
    tfn_consts k_src, k_dst; /* H2Y_PIPE_TFN: the two stages' table geometry */
    if (PIPE == H2Y_PIPE_TFN) {
        k_src.set(pp.src_fn > 0 ? pp.src_fn : H2Y_TFN_PQ_R);
        k_dst.set(pp.dst_fn > 0 ? pp.dst_fn : H2Y_TFN_PQ_R);
    }
    __syncthreads();

    block_clock_start(a);
    const uint32_t W = a.width, H = a.height;
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    tile_in v;         /* the tile being worked on */
    tile_pos t_cur;    /* and where it is */
    bool have = false; /* v holds the tile this wave meets next (uniform) */
    typedef in_traits<IN_KIND> IN;
    frame_walk fw;
    uint32_t fo = 0, tick = 0; /* waves take their tiles by ticket: see wave_deal */
    walk_init(fw, a);
    uint32_t r_first, r_count;
    const bool ranged = block_range(a, fw, r_first, r_count);
    for (; fw.f < a.n_frames; fw.advance(), fo++) {
        const int f = fw.f;
        const frame_io io = uniform_io(a.frames + f);
        const frame_io io_next = uniform_io(a.frames + (fw.has_next() ? f + (int)fw.NG : f));
        mm6 mm;
        mm.reset();
        wave_deal deal, deal_n;
        if (ranged) {
            deal.set_range(r_first, r_count);
            deal_n.set_range(r_first, r_count);
        } else {
            deal.set(fw, fw.k0, H2Y_LOOP_THREADS / WAVE);
            deal_n.set(fw, fw.k0_n, H2Y_LOOP_THREADS / WAVE);
        }
        if (!have) tick = wave_claim(&s_claim[fo], s_scratch, true);
        bool more = tick < deal.total; /* (a slice carried over from the previous frame is always below the total) */
        if (!have && more) {
            t_cur = tile_locate(umin32(deal.tile0<H2Y_LOOP_THREADS>(tick) + lane, a.tiles_per_frame - 1u), W, H, a.wq, a.wq_magic);
            tile_load<IN_KIND>(io, t_cur, v);
#pragma unroll
            for (int j = 0; j < 4; j++)
                asm volatile("" ::"v"(v.g0[j]), "v"(v.b0[j]), "v"(v.r0[j]), "v"(v.g1[j]), "v"(v.b1[j]), "v"(v.r1[j]));
        }
        if (!more) have = false;
        while (more) {
            tile_pos t = t_cur;
            t.row1 = true;
            const void *src[3];
            int kind;
            uint32_t tick2;
            tile_pos t2;
            const uint32_t n1v = wave_claim_issue(&s_claim[fo], s_scratch); /* read after row 0, when the prefetch needs it */

            tile_out o;
            uint32_t sb[2], sr[2];
#pragma unroll
            for (int row = 0; row < 2; row++) {
                const float(&gv)[4] = row ? v.g1 : v.g0;
                const float(&bv)[4] = row ? v.b1 : v.b0;
                const float(&rv)[4] = row ? v.r1 : v.r0;
                mm.add2(0, gv[0], gv[1]); mm.add2(0, gv[2], gv[3]);
                mm.add2(1, bv[0], bv[1]); mm.add2(1, bv[2], bv[3]);
                mm.add2(2, rv[0], rv[1]); mm.add2(2, rv[2], rv[3]);
                uint32_t Y[4], Cb[4], Cr[4];
#pragma unroll
                for (int col = 0; col < 4; col++) {
                    bool odd = false; /* a sample went through pq_slow(): it may be NaN (negative input, 0/0 normalisation) */
                    float g, b, r;
                    if (PIPE == H2Y_PIPE_NONE) { g = gv[col]; b = bv[col]; r = rv[col]; }
                    else if (PIPE == H2Y_PIPE_TFN) {
                        float x[3] = {norm1<H2Y_PIPE_RUNTIME>(pp, 0, gv[col]), norm1<H2Y_PIPE_RUNTIME>(pp, 1, bv[col]), norm1<H2Y_PIPE_RUNTIME>(pp, 2, rv[col])};
                        if (pp.src_fn > 0) tfn_stage3<true>(pp, k_src, s_tab, pp.tf_ext[0], x);
                        if (pp.dst_fn > 0) tfn_stage3<false>(pp, k_dst, s_tab + 2 * H2Y_PQ_NREC, pp.tf_ext[1], x);
                        odd = !(x[0] == x[0]) | !(x[1] == x[1]) | !(x[2] == x[2]); /* a NaN (negative sample through a power, ...): the careful matrix */
                        g = pix_scale(x[0], pp.mulY, pp.addY);
                        b = pix_scale(x[1], pp.mulC, pp.addC);
                        r = pix_scale(x[2], pp.mulC, pp.addC);
                    } else {
                        /* the six table records of a pixel on their way before the first is used */
                        const float xg = norm1<PIPE>(pp, 0, gv[col]), xb = norm1<PIPE>(pp, 1, bv[col]), xr = norm1<PIPE>(pp, 2, rv[col]);
                        const pq_rec cg = pq_fetch(xg, sA, sB), cb = pq_fetch(xb, sA, sB), cr = pq_fetch(xr, sA, sB);
                        __builtin_amdgcn_sched_barrier(0);
                        g = pq_sample_rec(pp, 0, xg, cg, odd, &s_ext);
                        b = pq_sample_rec(pp, 1, xb, cb, odd, &s_ext);
                        r = pq_sample_rec(pp, 2, xr, cr, odd, &s_ext);
                    }
                    bool um;
                    pix_matrix<MODE, false>(pp, g, b, r, Y[col], Cb[col], Cr[col], &um);
                    /* the careful form of the matrix: IEEE divisions, the reference's NaN conversions */
                    if (__builtin_expect(um | odd, 0)) pix_matrix<MODE, true>(pp, g, b, r, Y[col], Cb[col], Cr[col], &um);
                }
                row_pack<OUT_KIND, IN_KIND != H2Y_IN_U16>(pp, row, Y, Cb, Cr, o, sb, sr);
                if (row == 0) {
                    const uint32_t tt2 = ticket_resolve<H2Y_LOOP_THREADS>(n1v, deal, deal_n, s_claim, s_scratch, fo, fw.has_next(), tick, &kind, &tick2);
                    have = kind != 0;
#pragma unroll
                    for (int c = 0; c < 3; c++) src[c] = kind == 2 ? io_next.in[c] : io.in[c];
                    t2 = tile_locate(umin32(tt2 + lane, a.tiles_per_frame - 1u), W, H, a.wq, a.wq_magic);
                    IN::load4q(src[0], t2.q0, v.g0);
                    IN::load4q(src[1], t2.q0, v.b0);
                    IN::load4q(src[2], t2.q0, v.r0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            tile_store<OUT_KIND>(io, t, W, H, o);
            IN::load4q(src[0], t2.q1, v.g1);
            IN::load4q(src[1], t2.q1, v.b1);
            IN::load4q(src[2], t2.q1, v.r1);
            t_cur = t2;
            more = kind == 1;
            tick = tick2;
        }
        wave_store_mm(mm, a.partial + walk_slot(fw, H2Y_LOOP_THREADS / WAVE) * 6);
    }
    block_clock_end(a);
}

/*
 * k_fused_t1: the same work with a binary32 FIRST TIER in front (LINEAR -> PQ, YCbCr or YDzDx,
 * float input, even height).
 *
 *   tier 1  pq_t1(): one 16-byte LDS record and ~15 binary32 instructions per sample give the
 *           reference's float for 99 % of samples and say so; the matrix step is the exact one
 *           (pix_matrix_t1).  A pixel is final unless one of its samples was "unsure" AND one of
 *           its three integers could change with a one-ulp change of that sample (0.05 % of pixels
 *           at 12 bits, 0.8 % at 16), or the reciprocal-division guard fires, or a sample lies
 *           outside the table (+0.0 above all: black bars).
 *   The tile loop has no branch for such pixels.  The lanes whose tile holds one append the tile's
 *   number to their wave's list in LDS; when a wave has 64 of them it leaves the loop, redoes those
 *   64 tiles -- one per lane, all lanes busy -- with
 *   tier 2  pixel_fast<>(), the binary64 polynomial (its table is in LDS too), and
 *   tier 3  pixel_careful(), exactly as k_fused does (redo_pass), stores them over T1's
 *           provisional bytes and re-enters the loop.  No second kernel, no atomics.
 *
 * 1024 threads per block, one block per CU: 100 KB (T1 table) + 50 KB (binary64 table) + 8 KB
 * (sixteen wave lists) of the CU's 160 KB of LDS.
 */
#define H2Y_T1_THREADS 1024
#define H2Y_TAIL_QLEN 64 /* chunks of the dynamic last frame a block can draw (h2y_walk.h): 64 x 16 slices = a sixteenth of a 4K frame */
#define H2Y_REDO_CAP 128 /* list entries per wave: up to 63 left over plus the 64 one tile can add */
struct redo_ctx {        /* what redo_pass needs, handed over in LDS so that the call carries two pointers */
    const frame_io *frames;
    uint32_t *low_flag;  /* [n_frames]: set when a frame holds a sample <= -1 (subsampled minimum, see the tile loop) */
    uint32_t width, height, wq, wq_magic, tiles_per_frame, tiles_magic;
};
/* one tile per lane out of the wave's list: entries [first, first + cnt).  (Until round 3, when every tile the first tier could
 * not settle came here -- one in 250 --, the pass recomputed the loop's bytes and stored only the parts that differed: scattered
 * partial lines were what it cost.  Now one tile in tens of thousands comes, and is stored whole.) */
template <int IN_KIND, int OUT_KIND, int MODE, int PIPE>
__device__ __forceinline__ void redo_pass(const redo_ctx *rc, const pix_params *spp, const pq_recA *sA, const uint32_t *list,
                                                    uint32_t first, uint32_t cnt)
{
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    if (lane >= cnt) return;
    const pq_recB *sB = reinterpret_cast<const pq_recB *>(sA + H2Y_PQ_NREC);
    uint32_t tt;
    const uint32_t f = udiv_magic(list[first + lane], rc->tiles_per_frame, rc->tiles_magic, tt); /* entry = frame * tiles + tile */
    const frame_io io = rc->frames[f]; /* differs from lane to lane */
    const tile_pos t = tile_locate(tt, rc->width, rc->height, rc->wq, rc->wq_magic);
    tile_in v;
    tile_load<IN_KIND>(io, t, v);
    if (PIPE == H2Y_PIPE_PQ_IDENT && rc->low_flag) {
        /* the loop keeps only a subsample of the minimum (see there): a sample <= -1 changes pic_stats' floor */
        float lo = min3f(min3f(v.g0[0], v.g0[1], v.g0[2]), min3f(v.g0[3], v.g1[0], v.g1[1]), min3f(v.g1[2], v.g1[3], v.b0[0]));
        lo = min3f(lo, min3f(v.b0[1], v.b0[2], v.b0[3]), min3f(v.b1[0], v.b1[1], v.b1[2]));
        lo = min3f(lo, min3f(v.b1[3], v.r0[0], v.r0[1]), min3f(v.r0[2], v.r0[3], v.r1[0]));
        lo = min3f(lo, min3f(v.r1[1], v.r1[2], v.r1[3]), lo);
        if (lo <= -1.0f) rc->low_flag[f] = 1u;
    }
    tile_out o;
    tile_exact<OUT_KIND, MODE, PIPE>(*spp, spp, sA, sB, v, o);
    tile_store<OUT_KIND>(io, t, rc->width, rc->height, o);
}

#ifdef H2Y_BLOCK_TIMES /* timing experiments only: when does each block start its first tile and finish its last? */
__device__ unsigned long long g_block_times[2 * 1024];
void h2y_dump_block_times(const char *path)
{
    static unsigned long long h[2 * 1024];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_block_times), sizeof h) != hipSuccess) return;
    FILE *f = fopen(path, "w");
    if (!f) return;
    for (int i = 0; i < 1024; i++) fprintf(f, "%d %llu %llu\n", i, h[2 * i], h[2 * i + 1]);
    fclose(f);
}
#endif
template <int IN_KIND, int OUT_KIND, int MODE, int PIPE>
__global__ __launch_bounds__(H2Y_T1_THREADS) void k_fused_t1(fused_args a)
{
#ifdef H2Y_BLOCK_TIMES
    if (threadIdx.x == 0) g_block_times[2 * blockIdx.x] = wall_clock64();
#endif
    __shared__ pq_rec1 s_t1[H2Y_T1_NREC];
    __shared__ pq_recA s_t2[2 * H2Y_PQ_NREC]; /* A records, then B records */
    __shared__ uint32_t s_redo[H2Y_T1_THREADS / WAVE][H2Y_REDO_CAP];
    __shared__ pix_params s_pp;
    __shared__ redo_ctx s_rc;
    __shared__ uint32_t s_claim[H2Y_CLAIM_FRAMES], s_scratch[WAVE];
    __shared__ uint32_t s_tq[H2Y_TAIL_QLEN + 3], s_tbounds[1]; /* the dynamic last frame: the block's chunks, its drawing state (part, parts found empty), its ticket; the XCDs' runs */
    {
        stage16<H2Y_T1_THREADS, H2Y_T1_NREC>(a.table1, s_t1);
        stage_table<H2Y_T1_THREADS>(a.table, s_t2);
        if (threadIdx.x < H2Y_CLAIM_FRAMES) s_claim[threadIdx.x] = 0u;
        if (threadIdx.x < H2Y_TAIL_QLEN) s_tq[threadIdx.x] = 0xFFFFFFFFu; /* not drawn yet */
        if (threadIdx.x == 0) {
            s_tq[H2Y_TAIL_QLEN] = (blockIdx.x * 37u) & (H2Y_TAIL_PARTS - 1u); /* the part of the dynamic frame this block starts at (37 is odd: the blocks spread over all of them) */
            s_tq[H2Y_TAIL_QLEN + 1] = 0u;
            s_tq[H2Y_TAIL_QLEN + 2] = 0u;
        }
        if (a.tail_ctr && threadIdx.x == 0) s_tbounds[0] = a.tail_slices;
    }
    const pix_params pp = with_assumed(a.pp, a.assumed);
    t1_sens sn = a.sn;
    /* the two window constants selected per pixel (v_cndmask_b32 wants one operand in a vector register):
     * kept in registers for the whole kernel instead of being moved there for every pixel */
    asm volatile("" : "+v"(sn.a_lo), "+v"(sn.a_hi));
    if (threadIdx.x == 0) {
        s_pp = pp;
        s_rc.frames = a.frames;
        s_rc.low_flag = a.low_flag;
        s_rc.width = a.width; s_rc.height = a.height; s_rc.wq = a.wq; s_rc.wq_magic = a.wq_magic;
        s_rc.tiles_per_frame = a.tiles_per_frame; s_rc.tiles_magic = a.tiles_magic;
    }
    __syncthreads();
    /* a pixel of three +0.0 samples (black bars): the first tier cannot answer it (zero is outside its table), so rows of
     * them are recognised before the arithmetic (in the tile loop) and given this pixel's code values, worked out once by
     * the exact tiers */
    __shared__ uint32_t s_black[3];
    if (threadIdx.x == 0) {
        const pq_recB *sB2 = reinterpret_cast<const pq_recB *>(s_t2 + H2Y_PQ_NREC);
        const float G0 = norm1<PIPE>(pp, 0, 0.0f), B0 = norm1<PIPE>(pp, 1, 0.0f), R0 = norm1<PIPE>(pp, 2, 0.0f);
        uint32_t y, cb, cr;
        if (pixel_fast<MODE, PIPE>(pp, s_t2, sB2, G0, B0, R0, y, cb, cr)) {
            const ycc k = pixel_careful<MODE>(&s_pp, G0, B0, R0);
            y = k.y; cb = k.cb; cr = k.cr;
        }
        s_black[0] = y; s_black[1] = cb - pp.half_m1; s_black[2] = cr - pp.half_m1; /* chroma raw, as pix_matrix_t1() gives it */
    }
    __syncthreads();

    block_clock_start(a);
    const uint32_t W = a.width, H = a.height;
    const uint32_t lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const pq_recB *const sB2 = reinterpret_cast<const pq_recB *>(s_t2 + H2Y_PQ_NREC);
    uint32_t *const my_list = s_redo[wave];
    uint32_t n_redo = 0; /* entries in my_list (uniform over the wave) */
    /*
     * Prefetch.  The loop is bound by how many bytes it has on their way from memory, not by its arithmetic (a launch with
     * the arithmetic left out is 3-8 % shorter; a tenth fewer vector instructions changed nothing): sixteen waves per CU are
     * all the 100 KB table allows, so each must keep a whole tile -- six 16-byte loads per lane, 6 KB per wave -- in flight
     * while it works on the one before: the next tile's requests go out at the top of the loop body into a second set of
     * registers, which is copied over the first at the bottom (24 moves, 128 registers in all).  Until round 3 the loop
     * refilled the tile's own registers row by row as they fell free ("rolling": no second set, no copies, half as many
     * bytes in flight): 1.474 -> 1.446 ms per 64 x 4K launch on the same box.
     * The loop body is one basic block with a fixed number of memory operations in a fixed order, so
     * every wait is for exactly the loads it needs (memory operations complete in issue order; a
     * store issued conditionally would have to be assumed absent, and its wait would swallow the
     * loads behind it).  To that end:
     *  - lanes past the end of a frame work on its last tile again and store the same bytes again;
     *  - the redo list is appended to with LDS writes only, and working it off happens outside the loop;
     *  - the picture height is even (the host sends odd heights to k_fused).
     */
    tile_in v;         /* the tile being worked on */
    tile_pos t_cur;    /* and where it is */
    bool have = false; /* v holds the tile this block meets next (uniform) */
    typedef in_traits<IN_KIND> IN;
    frame_walk fw;
    uint32_t fo = 0;   /* ordinal of the frame among the group's: its counter is s_claim[fo] */
    uint32_t tick = 0; /* the slice in hand, if hold: of the frame the loop is at (or about to enter) */
    bool hold = false; /* (have implies hold: v is tick's tile; a redo pass drops the data, not the slice) */
    uint32_t dyn_carry = 0, dyn_t0_next = 0; /* the dynamic frame: first tile of a slice drawn from the frame before it / of the slice asked for */
    walk_init(fw, a);
    typedef const __attribute__((address_space(4))) fused_args *args_ptr;
    args_ptr ap = (args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    const uint32_t grp = (uint32_t)fw.f; /* the block's group: its first frame */
    uint32_t r_first, r_count;
    const bool ranged = block_range(a, fw, r_first, r_count);
    /*
     * The dynamic last frame of the group (h2y_walk.h): same loop, other source of slices.  The BLOCK draws chunks of sixteen
     * slices from the counters in global memory; ticket t of the block's counter s_tq[QLEN + 2] stands for slice t % 16 of its
     * chunk t / 16, and s_tq[c] holds chunk c once it is drawn: first slice | count << 24, kTqEmpty when nothing was left.
     * The wave that holds ticket 16 c reads chunk c (it is there, or about to be) and then draws chunk c + 1: one chunk ahead,
     * one drawer at a time, the atomic's latency on one wave in sixteen.  A ticket beyond a short last chunk stands for that
     * chunk's last slice once more (the same bytes again, as for lanes beyond a frame's end).
     */
    constexpr uint32_t kTqNone = 0xFFFFFFFFu, kTqEmpty = 0xFFFFFFFEu;
    volatile uint32_t *const tq = s_tq;
    uint32_t *const tail_ctr = a.tail_ctr ? a.tail_ctr + grp * H2Y_TAIL_WORDS : nullptr;
    auto draw_chunk = [&](uint32_t c) { /* one chunk from the counters: by ONE wave at a time (the block's drawing state is in LDS) */
        tail_state ts;
        ts.part = tq[H2Y_TAIL_QLEN];
        ts.tried = tq[H2Y_TAIL_QLEN + 1];
        const uint32_t n_slices = s_tbounds[0];
        uint32_t got = kTqEmpty;
        uint32_t known[2] = {0u, 0u}; /* parts known to be exhausted (bits set by whoever found them so): read once one is met */
        while (!ts.done()) {
            if ((known[ts.part >> 5] >> (ts.part & 31u)) & 1u) { ts.skip(); continue; }
            uint32_t cv = 0u;
            if (lane == 0) cv = __hip_atomic_fetch_add(tail_ctr + ts.part, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t first, count;
            if (ts.take(n_slices, __builtin_amdgcn_readfirstlane(cv), &first, &count)) {
                got = first | (count << 24);
                break;
            }
            uint32_t k0 = 0u, k1 = 0u;
            if (lane == 0) {
                k0 = __hip_atomic_fetch_or(tail_ctr + H2Y_TAIL_PARTS + (ts.part >> 5), 1u << (ts.part & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                k1 = __hip_atomic_load(tail_ctr + H2Y_TAIL_PARTS + ((ts.part >> 5) ^ 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            known[ts.part >> 5] = __builtin_amdgcn_readfirstlane(k0) | (1u << (ts.part & 31u));
            known[(ts.part >> 5) ^ 1u] = __builtin_amdgcn_readfirstlane(k1);
            ts.skip();
        }
        if (lane == 0) {
            /* (LDS operations of one wave are carried out in the order it issued them: the state is there before the chunk that
             * tells the next drawer to read it -- no fence: one would also wait for the wave's picture loads and stores) */
            tq[H2Y_TAIL_QLEN] = ts.part;
            tq[H2Y_TAIL_QLEN + 1] = ts.tried;
            asm volatile("" ::: "memory");
            tq[c] = got;
        }
    };
    /* ticket tk of the dynamic frame -> its slice's first tile; false: the frame is dealt out */
    auto dyn_tile0 = [&](uint32_t tk, uint32_t &tile0) -> bool {
        const uint32_t c = tk / H2Y_TAIL_CHUNK, idx = tk % H2Y_TAIL_CHUNK;
        if (c >= H2Y_TAIL_QLEN) return false; /* (the host sees to it that a block never gets this far: run_frames()) */
        if (tk == 0u) draw_chunk(0u);          /* the first wave of the block to get here */
        uint32_t e;
        while ((e = tq[c]) == kTqNone) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        if (idx == 0u && c + 1u < H2Y_TAIL_QLEN) {
            if (e == kTqEmpty) { if (lane == 0) tq[c + 1u] = kTqEmpty; }
            else draw_chunk(c + 1u);
        }
        if (e == kTqEmpty) return false;
        const uint32_t cnt = e >> 24;
        tile0 = ((e & 0xFFFFFFu) + (idx < cnt ? idx : cnt - 1u)) * WAVE;
        return true;
    };
    for (; fw.f < fw.n_frames; fw.advance(), fo++) {
        const int f = fw.f;
        const bool dyn = tail_ctr != nullptr && !fw.has_next();                         /* this frame is the group's dynamic one */
        const bool dyn_n = tail_ctr != nullptr && fw.has_next() && f + 2 * (int)fw.NG >= fw.n_frames; /* the next one is */
        uint32_t dyn_t0 = 0; /* dyn: first tile of the slice in hand / asked for */
        asm volatile("" : "+s"(ap)); /* (what only a frame's beginning and end need is read there, through a pointer the compiler cannot see through: k_fir_fused, "scalar registers") */
        const frame_io *const frames = ap->frames;
        const frame_io io = uniform_io(frames + f);
        /* the frame after this one in the group, for the prefetch across the frame boundary: its plane pointers are read when a wave
         * gets there (scalar loads: the array is written before the launch), not kept in six scalar registers all frame long */
        typedef const __attribute__((address_space(4))) frame_io *frame_cptr;
        const frame_cptr io_next_p = (frame_cptr)(uintptr_t)(frames + (fw.has_next() ? f + (int)fw.NG : f));
        mm6 mm;
        mm.reset();
        const uint32_t id_base = (uint32_t)f * a.tiles_per_frame;
        uint32_t flagged_f = 0;  /* tiles of this frame this wave sent to the list */
        uint32_t flagged_px = 0; /* pixels of this frame the wave's lanes sent through the binary64 tier in the loop (the host steers by their share, in tiles of eight) */
        wave_deal deal, deal_n;
        if (ranged) {
            deal.set_range(r_first, r_count);
            deal_n.set_range(r_first, r_count);
        } else {
            deal.set(fw, fw.k0, H2Y_T1_THREADS / WAVE);
            deal_n.set(fw, fw.k0_n, H2Y_T1_THREADS / WAVE);
        }
        if (dyn) {
            if (!hold) {
                tick = wave_claim(&s_tq[H2Y_TAIL_QLEN + 2], s_scratch, true);
                hold = dyn_tile0(tick, dyn_t0);
            } else dyn_t0 = dyn_carry; /* the slice was drawn while the previous frame was finishing */
        } else if (!hold) {
            tick = wave_claim(&s_claim[fo], s_scratch, true);
            hold = tick < deal.total;
        }
        bool more = hold;
        while (more) {
            if (!have) { /* nothing on its way (first tile of the launch, after a redo pass, or a wave that found a frame dealt out) */
                t_cur = tile_locate(umin32((dyn ? dyn_t0 : deal.tile0<H2Y_T1_THREADS>(tick)) + lane, a.tiles_per_frame - 1u), W, H, a.wq, a.wq_magic);
                tile_load<IN_KIND>(io, t_cur, v);
            }
            /* have the data arrive here: entering the loop with loads outstanding would make the loop's
             * own waits (computed over all ways into it) wait for everything */
#pragma unroll
            for (int j = 0; j < 4; j++)
                asm volatile("" ::"v"(v.g0[j]), "v"(v.b0[j]), "v"(v.r0[j]), "v"(v.g1[j]), "v"(v.b1[j]), "v"(v.r1[j]));
            do {
                const uint32_t tt = (dyn ? dyn_t0 : deal.tile0<H2Y_T1_THREADS>(tick)) + lane;
                tile_pos t = t_cur; /* located one iteration ago, as the prefetch target */
                t.row1 = true;
                const void *src[3];
                int kind;
                uint32_t tick2;
                tile_pos t2;
                /* the next slice's number */
                const uint32_t n1v = wave_claim_issue(dyn ? &s_tq[H2Y_TAIL_QLEN + 2] : &s_claim[fo], s_scratch);

                tile_in nx; /* the next tile */
                {
                        { /* (spelled out: through ticket_resolve() the same code came out 3 % slower here) */
                            const uint32_t n1 = __builtin_amdgcn_readfirstlane(n1v);
                            uint32_t tt2;
                            if (__builtin_expect(dyn, 0)) { /* the dynamic frame: the ticket's slice out of the block's chunks; no frame behind it */
                                uint32_t t0n = 0;
                                const bool same = dyn_tile0(n1, t0n);
                                kind = same ? 1 : 0;
                                tick2 = same ? n1 : tick;
                                tt2 = same ? t0n : dyn_t0;
                                dyn_t0_next = tt2;
                            } else {
                            const bool same = n1 < deal.total;
                            if (__builtin_expect(same, 1)) {
                                kind = 1;
                                tick2 = n1;
                                tt2 = deal.tile0<H2Y_T1_THREADS>(n1);
                            } else { /* this frame is dealt out: a slice of the group's next frame, if there is one */
                                const bool try_next = fw.has_next();
                                bool nextf;
                                uint32_t n2, t0n = 0;
                                if (dyn_n) { /* ... which is the dynamic one */
                                    n2 = wave_claim(&s_tq[H2Y_TAIL_QLEN + 2], s_scratch, true);
                                    nextf = dyn_tile0(n2, t0n);
                                    dyn_carry = t0n;
                                } else {
                                    n2 = wave_claim(&s_claim[(fo + 1u) & (H2Y_CLAIM_FRAMES - 1u)], s_scratch, try_next);
                                    nextf = try_next && n2 < deal_n.total;
                                    if (nextf) t0n = deal_n.tile0<H2Y_T1_THREADS>(n2);
                                }
                                kind = nextf ? 2 : 0;
                                tick2 = nextf ? n2 : tick;
                                tt2 = nextf ? t0n : deal.tile0<H2Y_T1_THREADS>(tick);
                            }
                            }
                            /* uniform values, and known to be: the dynamic frame's branch reads them out of LDS, after which the compiler
                             * kept all three in vector registers and chose the plane pointers with vector selects (thirty
                             * instructions a tile) */
                            kind = __builtin_amdgcn_readfirstlane(kind);
                            tick2 = __builtin_amdgcn_readfirstlane(tick2);
                            tt2 = __builtin_amdgcn_readfirstlane(tt2);
                            hold = have = kind != 0;
#pragma unroll
                            for (int c = 0; c < 3; c++) src[c] = io.in[c];
                            if (__builtin_expect(kind == 2, 0)) {
#pragma unroll
                                for (int c = 0; c < 3; c++) src[c] = io_next_p->in[c];
                            }
                            t2 = tile_locate(umin32(tt2 + lane, a.tiles_per_frame - 1u), W, H, a.wq, a.wq_magic);
                        }
                    IN::load4q(src[0], t2.q0, nx.g0);
                    IN::load4q(src[1], t2.q0, nx.b0);
                    IN::load4q(src[2], t2.q0, nx.r0);
                    IN::load4q(src[0], t2.q1, nx.g1);
                    IN::load4q(src[1], t2.q1, nx.b1);
                    IN::load4q(src[2], t2.q1, nx.r1);
                    __builtin_amdgcn_sched_barrier(0); /* none of the tile's arithmetic before the requests are out */
                }
                tile_out o;
                uint32_t sb[2], sr[2]; /* 2x2 box: chroma sums of the two blocks */
                uint64_t redo_m = 0; /* lanes whose tile holds a pixel to redo: the guards' own lane masks, ORed in scalar registers */
#pragma unroll
                for (int row = 0; row < 2; row++) {
                    const float(&gv)[4] = row ? v.g1 : v.g0;
                    const float(&bv)[4] = row ? v.b1 : v.b0;
                    const float(&rv)[4] = row ? v.r1 : v.r0;
#ifdef H2Y_EXP_NOSTATS /* timing experiment only: what do twelve vector instructions per tile cost? */
                    if (false) {
#else
                    if (PIPE == H2Y_PIPE_PQ_IDENT) {
#endif
                        /* Assumed floor 0 / ceiling 1: the maximum of every sample is needed (is there one >= 1?),
                         * of the minimum only that it lies in (-1, 1).  In-table samples are positive; a sample
                         * <= -1 is out of the table, so its tile meets redo_pass(), which reports it.  That leaves
                         * "some sample is below 1": the minimum over a subsample (one pair per tile and plane)
                         * shows it for every ordinary picture; if it does not, the frame counts as a mismatch and
                         * is redone with exact statistics.  Nine instructions less per tile. */
                        if (row == 0) { mm.add2(0, gv[0], gv[1]); mm.add2(1, bv[0], bv[1]); mm.add2(2, rv[0], rv[1]); }
                        else { mm.add2_max(0, gv[0], gv[1]); mm.add2_max(1, bv[0], bv[1]); mm.add2_max(2, rv[0], rv[1]); }
                        mm.add2_max(0, gv[2], gv[3]); mm.add2_max(1, bv[2], bv[3]); mm.add2_max(2, rv[2], rv[3]);
                    } else {
#ifndef H2Y_EXP_NOSTATS
                        mm.add2(0, gv[0], gv[1]); mm.add2(0, gv[2], gv[3]);
                        mm.add2(1, bv[0], bv[1]); mm.add2(1, bv[2], bv[3]);
                        mm.add2(2, rv[0], rv[1]); mm.add2(2, rv[2], rv[3]);
#endif
                    }
                    uint32_t Y[4], Cb[4], Cr[4];
                    /* a row of zeros in every lane (letterbox bars)?  One compare per row for ordinary pictures: each lane's
                     * first sample.  Such rows get the black pixel's code values and are not flagged. */
                    bool zrow = false;
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(f2bits(gv[0]) != 0u) == 0, 0)) {
                        const uint32_t z = (f2bits(gv[1]) | f2bits(gv[2]) | f2bits(gv[3])) | (f2bits(bv[0]) | f2bits(bv[1]) | f2bits(bv[2])) |
                                           (f2bits(bv[3]) | f2bits(rv[0]) | f2bits(rv[1])) | (f2bits(rv[2]) | f2bits(rv[3]));
                        zrow = __builtin_amdgcn_ballot_w64(z != 0u) == 0;
                    }
                    if (zrow) {
                        const uint32_t y0 = s_black[0], cb0 = s_black[1], cr0 = s_black[2];
#pragma unroll
                        for (int col = 0; col < 4; col++) { Y[col] = y0; Cb[col] = cb0; Cr[col] = cr0; }
                    } else
#pragma unroll
                    for (int col = 0; col < 4; col++) {
#ifdef H2Y_EXP_NOCOMPUTE /* timing experiment only (wrong bytes): the loop's loads, stores and tickets alone */
                        Y[col] = f2bits(gv[col]) >> 20; Cb[col] = f2bits(bv[col]) >> 20; Cr[col] = f2bits(rv[col]) >> 20;
                        continue;
#endif
#ifdef H2Y_HALF_COMPUTE /* timing experiments only: how much of the time is arithmetic? */
                        if (row == 1) { Y[col] = o.yp0[col >> 1] >> (16 * (col & 1)) & 0xFFFFu; Cb[col] = sb[col >> 1] >> 1; Cr[col] = sr[col >> 1] >> 1; continue; }
#endif
                        const float Gn = norm1<PIPE>(pp, 0, gv[col]);
                        const float Bn = norm1<PIPE>(pp, 1, bv[col]);
                        const float Rn = norm1<PIPE>(pp, 2, rv[col]);
                        /* the three records of a pixel on their way before the first is used: -1.7 % (one pixel further ahead: no better) */
                        const pq_rec1 cg = pq_t1_fetch(Gn, s_t1), cb = pq_t1_fetch(Bn, s_t1), cr = pq_t1_fetch(Rn, s_t1);
                        __builtin_amdgcn_sched_barrier(0);
                        float mg, mb, mr;
                        const float g = pix_scale(pq_t1_eval_m(Gn, cg, &mg), pp.mulY, pp.addY);
                        const float b = pix_scale(pq_t1_eval_m(Bn, cb, &mb), pp.mulC, pp.addC);
                        const float r = pix_scale(pq_t1_eval_m(Rn, cr, &mr), pp.mulC, pp.addC);
                        bool ra, rb;
                        pix_matrix_t1<MODE>(pp, sn, g, b, r, pq_t1_unsure3(mg, mb, mr), Y[col], Cb[col], Cr[col], &ra, &rb);
                        const uint64_t fm = __builtin_amdgcn_ballot_w64(ra) | __builtin_amdgcn_ballot_w64(rb); /* a ballot of a compare is the compare's own result */
                        if (__builtin_expect(fm != 0, 0)) {
                            /* Some lane's pixel here is not settled (one wave-pixel in 25 on the headline's picture): the binary64
                             * tier for this pixel position on the spot, as in k_fir_fused -- no global memory operation in the
                             * branch, only LDS reads and, for samples below the tables, scalar loads; only the planes that hold an
                             * unsure sample of a flagged lane go through it.  What it cannot settle either (a value too close to a
                             * rounding tie for binary64, a NaN, the division guard: one pixel in tens of thousands) sends the
                             * tile to the list.  Rounds 1-2 listed every such tile: on dark pictures one tile in 70. */
                            const bool fl = ((fm >> lane) & 1u) != 0;
                            const bool ug = pq_t1_unsure(mg), ub = pq_t1_unsure(mb), ur = pq_t1_unsure(mr);
                            float g2 = g, b2 = b, r2 = r;
                            bool un = false;
                            if (__builtin_amdgcn_ballot_w64(fl & ug) != 0) {
                                bool sl;
                                const float t = pq_ext_inline(Gn, pq_fast(Gn, s_t2, sB2, &sl), sl, pp.pq_ext);
                                g2 = (fl & ug) ? pix_scale(t, pp.mulY, pp.addY) : g2;
                                un |= fl & ug & sl;
                            }
                            if (__builtin_amdgcn_ballot_w64(fl & ub) != 0) {
                                bool sl;
                                const float t = pq_ext_inline(Bn, pq_fast(Bn, s_t2, sB2, &sl), sl, pp.pq_ext);
                                b2 = (fl & ub) ? pix_scale(t, pp.mulC, pp.addC) : b2;
                                un |= fl & ub & sl;
                            }
                            if (__builtin_amdgcn_ballot_w64(fl & ur) != 0) {
                                bool sl;
                                const float t = pq_ext_inline(Rn, pq_fast(Rn, s_t2, sB2, &sl), sl, pp.pq_ext);
                                r2 = (fl & ur) ? pix_scale(t, pp.mulC, pp.addC) : r2;
                                un |= fl & ur & sl;
                            }
                            uint32_t Y2, Cb2, Cr2;
                            bool um;
                            pix_matrix<MODE, false>(pp, g2, b2, r2, Y2, Cb2, Cr2, &um);
                            Y[col] = fl ? Y2 : Y[col];
                            Cb[col] = fl ? Cb2 - pp.half_m1 : Cb[col]; /* raw, as the first tier's */
                            Cr[col] = fl ? Cr2 - pp.half_m1 : Cr[col];
                            redo_m |= __builtin_amdgcn_ballot_w64(fl & (un | um));
                            flagged_px += (uint32_t)__popcll(fm);
                        }
                        /* pin the mask here: left alone, the compiler postpones every pixel's guard arithmetic
                         * to the end of the tile and keeps its operands alive until then (register spills) */
                        asm volatile("" : "+s"(redo_m));
                    }
                    row_pack<OUT_KIND, true, true>(pp, row, Y, Cb, Cr, o, sb, sr);
                }
                tile_store<OUT_KIND>(io, t, W, H, o);
                v = nx; /* (24 register moves; waits for the loads asked for at the top) */
                t_cur = t2;
                /* tiles to redo (a few lanes of about one wave in four at 12 bits; whole waves on black
                 * bars) go to the wave's list: position = entries so far + flagged lanes below this one */
                {
                    /* Lanes past the frame's end have redone its last tile and stored its provisional bytes
                     * again -- possibly after the tile's owner already replaced them with the exact ones.  So
                     * they list the tile too: the last store to a flagged tile is then always a redo_pass()
                     * store (a wave's own redo follows its own provisional store). */
                    const uint64_t m = redo_m;
                    const bool flagged = ((m >> lane) & 1u) != 0;
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    /* unflagged lanes write to the spare last slot: no branch in the loop body */
                    my_list[flagged ? n_redo + below : (uint32_t)(H2Y_REDO_CAP - 1)] = id_base + umin32(tt, a.tiles_per_frame - 1u);
                    n_redo = __builtin_amdgcn_readfirstlane(n_redo + (uint32_t)__popcll(m)); /* (uniform, and said to be: the loop's exit test stays a scalar one) */
                    flagged_f = __builtin_amdgcn_readfirstlane(flagged_f + (uint32_t)__popcll(m));
                }
                more = kind == 1;
                tick = tick2;
                if (dyn) dyn_t0 = dyn_t0_next;
            } while (more && n_redo < WAVE);
            if (n_redo >= WAVE) { /* 64 tiles to redo: one per lane */
                n_redo -= WAVE;
#ifndef H2Y_SKIP_REDO /* timing experiments only: what do the passes cost? */
                redo_pass<IN_KIND, OUT_KIND, MODE, PIPE>(&s_rc, &s_pp, s_t2, my_list, n_redo, WAVE);
                have = false; /* the prefetched tile is not carried through the pass (registers): it is asked for again */
#endif
            }
        }
        asm volatile("" : "+s"(ap));
        wave_store_mm(mm, ap->partial + walk_slot(fw, H2Y_T1_THREADS / WAVE) * 6);
        if (ap->redo_count && lane == 0) ap->redo_count[walk_slot(fw, H2Y_T1_THREADS / WAVE)] = flagged_f + (flagged_px >> 3);
    }
#ifdef H2Y_BLOCK_TIMES /* blocks 0..3: when each wave left the frame loop */
    if (blockIdx.x < 4 && lane == 0) g_block_times[2 * (800 + blockIdx.x * 16 + wave)] = wall_clock64();
#endif
    if (n_redo) redo_pass<IN_KIND, OUT_KIND, MODE, PIPE>(&s_rc, &s_pp, s_t2, my_list, 0u, n_redo);
    block_clock_end(a);
#ifdef H2Y_BLOCK_TIMES
    __syncthreads();
    if (threadIdx.x == 0) g_block_times[2 * blockIdx.x + 1] = wall_clock64();
#endif
    (void)lane;
}

/*
 * k_fused_lut16: half-float input (EXR, exr.cpp:233), LINEAR -> PQ, floor 0 /
 * ceiling 1.  A half in [0, 2) has only 16 384 bit patterns, so PQ10000_r() of
 * every one of them fits in LDS (64 KB of binary32, built once per context by
 * k_build_lut16 with the exact tiers): the per-sample transfer is one 4-byte LDS
 * read, exact by construction, zero included.  Samples outside [0, 2) (sign or
 * bit 14 set) send their pixel to the careful tier.  Everything after the
 * transfer is the same arithmetic as k_fused.  min/max for pic_stats run on
 * the packed halves (v_pk_min_f16 / v_pk_max_f16 order halves exactly as their
 * widened floats).
 */
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_h(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_min_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_max_h(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float half_bits_to_float(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (uint16_t)h); }

/* COLS = 8: the thread tile is 8 columns x 2 rows (width % 8 == 0, planes 16-byte aligned): halves are two
 * bytes, so only then are the accesses the 16 bytes per lane the memory pipeline likes best (loads and luma
 * stores; 8 bytes for the 4:2:0 chroma) -- the same registers per thread as a 4-column tile of floats.
 * a.wq = width / COLS, a.tiles_per_frame accordingly; COLS = 4 is the general form. */
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int COLS> struct lut16_raw;
template <> struct lut16_raw<4> { typedef u32x2 type; };
template <> struct lut16_raw<8> { typedef u32x4 type; };
/* two 4-column halves of an 8-column tile, stored together */
template <int OUT_KIND>
__device__ __forceinline__ void tile_store8(const frame_io &io, const tile_pos &t, uint32_t W, uint32_t H, const tile_out &a, const tile_out &b)
{
    const uint32_t npix = W * H;
    gstore_nt<u32x4>(io.out, t.q0, u32x4{a.yp0[0], a.yp0[1], b.yp0[0], b.yp0[1]});
    gstore_nt<u32x4>(io.out, t.q1, u32x4{a.yp1[0], a.yp1[1], b.yp1[0], b.yp1[1]});
    if (OUT_KIND == H2Y_OUT_420BOX) {
        /* four chroma samples per tile: sample index 4 tt in each plane, i.e. 8-byte unit tt */
        const uint32_t ncb = (W >> 1) * (H >> 1);
        gstore_nt<u32x2>(io.out, (npix >> 2) + t.tt, u32x2{a.cb_box, b.cb_box});
        gstore_nt<u32x2>(io.out, ((npix + ncb) >> 2) + t.tt, u32x2{a.cr_box, b.cr_box});
    } else if (OUT_KIND == H2Y_OUT_444) {
        uint16_t *Cbp = io.out + npix, *Crp = io.out + 2 * (size_t)npix;
        gstore_nt<u32x4>(Cbp, t.q0, u32x4{a.cbp0[0], a.cbp0[1], b.cbp0[0], b.cbp0[1]});
        gstore_nt<u32x4>(Crp, t.q0, u32x4{a.crp0[0], a.crp0[1], b.crp0[0], b.crp0[1]});
        gstore_nt<u32x4>(Cbp, t.q1, u32x4{a.cbp1[0], a.cbp1[1], b.cbp1[0], b.cbp1[1]});
        gstore_nt<u32x4>(Crp, t.q1, u32x4{a.crp1[0], a.crp1[1], b.crp1[0], b.crp1[1]});
    } else { /* scratch planes: k_fir420 reads them next */
        gstore<u32x4>(io.tmp_cb, t.q0, u32x4{a.cbp0[0], a.cbp0[1], b.cbp0[0], b.cbp0[1]});
        gstore<u32x4>(io.tmp_cr, t.q0, u32x4{a.crp0[0], a.crp0[1], b.crp0[0], b.crp0[1]});
        gstore<u32x4>(io.tmp_cb, t.q1, u32x4{a.cbp1[0], a.cbp1[1], b.cbp1[0], b.cbp1[1]});
        gstore<u32x4>(io.tmp_cr, t.q1, u32x4{a.crp1[0], a.crp1[1], b.crp1[0], b.crp1[1]});
    }
}

template <int OUT_KIND, int MODE, int COLS>
__global__ __launch_bounds__(H2Y_LOOP_THREADS) void k_fused_lut16(fused_args a)
{
    typedef typename lut16_raw<COLS>::type RV;
    constexpr int NH = COLS / 4; /* 4-column halves per tile */
    __shared__ float s_lut_y[H2Y_LUT16_N], s_lut_c[H2Y_LUT16_N]; /* PQ and the scale step of every half in [0, 2): luma / chroma constants */
    __shared__ pix_params s_pp;
    __shared__ uint32_t s_claim[H2Y_CLAIM_FRAMES], s_scratch[WAVE];
    const pix_params pp = with_assumed(a.pp, a.assumed);
    {
        stage_lut16_scaled<H2Y_LOOP_THREADS>(a.lut16, pp, s_lut_y, s_lut_c);
        if (threadIdx.x < H2Y_CLAIM_FRAMES) s_claim[threadIdx.x] = 0u;
    }
    if (threadIdx.x == 0) s_pp = pp;
    __syncthreads();

    block_clock_start(a);
    const uint32_t W = a.width, H = a.height;
    /* the loop form of k_fused_t1: row-wise tiles, rolling prefetch, one basic block of memory
     * operations; even height (the host sends odd heights to k_fused) */
    RV raw[3][2];   /* the tile being worked on, raw halves: [plane][row], COLS samples each */
    tile_pos t_cur;
    bool have = false;
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    frame_walk fw;
    uint32_t fo = 0, tick = 0; /* waves take their tiles by ticket: see wave_deal */
    walk_init(fw, a);
    uint32_t r_first, r_count;
    const bool ranged = block_range(a, fw, r_first, r_count);
    for (; fw.f < a.n_frames; fw.advance(), fo++) {
        const int f = fw.f;
        const frame_io io = uniform_io(a.frames + f);
        const frame_io io_next = uniform_io(a.frames + (fw.has_next() ? f + (int)fw.NG : f));
        /* packed-half accumulators: {min, max} x plane, two halves per dword */
        uint32_t mn[3], mx[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            mn[c] = 0x7C007C00u; /* +inf stands in for FLT_MAX (common.cpp:118): no finite or infinite sample is "< FLT_MAX" unless it is finite */
            mx[c] = 0x00000000u; /* +0: FLT_MIN (1.2e-38) is below the smallest half; (int) of either is 0 */
        }
        wave_deal deal, deal_n;
        if (ranged) {
            deal.set_range(r_first, r_count);
            deal_n.set_range(r_first, r_count);
        } else {
            deal.set(fw, fw.k0, H2Y_LOOP_THREADS / WAVE);
            deal_n.set(fw, fw.k0_n, H2Y_LOOP_THREADS / WAVE);
        }
        if (!have) tick = wave_claim(&s_claim[fo], s_scratch, true);
        bool more = tick < deal.total;
        if (!more) have = false;
        if (!have && more) {
            t_cur = tile_locate(umin32(deal.tile0<H2Y_LOOP_THREADS>(tick) + lane, a.tiles_per_frame - 1u), W, H, a.wq, a.wq_magic);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                raw[c][0] = gload_nt<RV>(io.in[c], t_cur.q0);
                raw[c][1] = gload_nt<RV>(io.in[c], t_cur.q1);
            }
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int w = 0; w < COLS / 2; w++) asm volatile("" ::"v"(raw[c][0][w]), "v"(raw[c][1][w]));
        }
        while (more) {
            tile_pos t = t_cur;
            t.row1 = true;
            const void *src[3];
            int kind;
            uint32_t tick2;
            tile_pos t2;
            const uint32_t n1v = wave_claim_issue(&s_claim[fo], s_scratch); /* read after row 0, when the prefetch needs it */
            tile_out o[NH];
            uint32_t sb[NH][2], sr[NH][2];
#pragma unroll
            for (int row = 0; row < 2; row++) {
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int w = 0; w < COLS / 2; w++) {
                        mn[c] = pk_min_h(mn[c], raw[c][row][w]);
                        mx[c] = pk_max_h(mx[c], raw[c][row][w]);
                    }
#pragma unroll
                for (int hf = 0; hf < NH; hf++) {
                    uint32_t Y[4], Cb[4], Cr[4];
                    /* a sample outside the table (sign or bit 14 set: negative, >= 2.0, inf, NaN) among this lane's four
                     * columns of the row?  Looked for once on the packed halves, not per pixel: the four pixels then all
                     * take the careful tier */
                    const bool outside = (((raw[0][row][2 * hf] | raw[0][row][2 * hf + 1] | raw[1][row][2 * hf]) |
                                           (raw[1][row][2 * hf + 1] | raw[2][row][2 * hf] | raw[2][row][2 * hf + 1])) & 0xC000C000u) != 0u;
#pragma unroll
                    for (int col = 0; col < 4; col++) {
                        const uint32_t wg = raw[0][row][2 * hf + (col >> 1)];
                        const uint32_t wb = raw[1][row][2 * hf + (col >> 1)];
                        const uint32_t wr = raw[2][row][2 * hf + (col >> 1)];
                        const uint32_t hg = col & 1 ? wg >> 16 : wg & 0xFFFFu, hb = col & 1 ? wb >> 16 : wb & 0xFFFFu, hr = col & 1 ? wr >> 16 : wr & 0xFFFFu;
                        /* the three table reads of a pixel on their way before the first is used (as in k_fused_t1) */
                        const float g = s_lut_y[hg & (H2Y_LUT16_N - 1)], b = s_lut_c[hb & (H2Y_LUT16_N - 1)], r = s_lut_c[hr & (H2Y_LUT16_N - 1)];
#ifndef H2Y_LUT_NOBARRIER
                        __builtin_amdgcn_sched_barrier(0);
#endif
                        bool um;
                        pix_matrix<MODE, false>(pp, g, b, r, Y[col], Cb[col], Cr[col], &um);
                        if (__builtin_expect(outside | um, 0)) {
                            const ycc c = pixel_careful<MODE>(&s_pp, half_bits_to_float(hg), half_bits_to_float(hb), half_bits_to_float(hr));
                            Y[col] = c.y; Cb[col] = c.cb; Cr[col] = c.cr;
                        }
                    }
                    row_pack<OUT_KIND, true>(pp, row, Y, Cb, Cr, o[hf], sb[hf], sr[hf]);
                }
                if (row == 0) {
                    const uint32_t tt2 = ticket_resolve<H2Y_LOOP_THREADS>(n1v, deal, deal_n, s_claim, s_scratch, fo, fw.has_next(), tick, &kind, &tick2);
                    have = kind != 0;
#pragma unroll
                    for (int c = 0; c < 3; c++) src[c] = kind == 2 ? io_next.in[c] : io.in[c];
                    t2 = tile_locate(umin32(tt2 + lane, a.tiles_per_frame - 1u), W, H, a.wq, a.wq_magic);
#pragma unroll
                    for (int c = 0; c < 3; c++) raw[c][0] = gload_nt<RV>(src[c], t2.q0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (COLS == 8) tile_store8<OUT_KIND>(io, t, W, H, o[0], o[NH - 1]);
            else tile_store<OUT_KIND>(io, t, W, H, o[0]);
#pragma unroll
            for (int c = 0; c < 3; c++) raw[c][1] = gload_nt<RV>(src[c], t2.q1);
            t_cur = t2;
            more = kind == 1;
            tick = tick2;
        }
        mm6 mm;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            /* fold the two packed halves, widen; floats from here on (block_store_mm) */
            mm.lo[c] = fminf(half_bits_to_float(mn[c] & 0xFFFFu), half_bits_to_float(mn[c] >> 16));
            mm.hi[c] = fmaxf(half_bits_to_float(mx[c] & 0xFFFFu), half_bits_to_float(mx[c] >> 16));
            /* restore pic_stats' initial values when nothing beat them */
            mm.lo[c] = mm.lo[c] > 65504.0f ? 3.402823466e+38f : mm.lo[c]; /* still +inf: no sample was below FLT_MAX */
            mm.hi[c] = mm.hi[c] <= 0.0f ? 1.175494351e-38f : mm.hi[c];
        }
        wave_store_mm(mm, a.partial + walk_slot(fw, H2Y_LOOP_THREADS / WAVE) * 6);
    }
    block_clock_end(a);
}

/* PQ10000_r() of every half in [0, 2) through the exact tiers; table records read from HBM */
__global__ __launch_bounds__(256) void k_build_lut16(const void *table, float *lut)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H2Y_LUT16_N) return;
    const pq_recA *A = static_cast<const pq_recA *>(table);
    const pq_recB *B = reinterpret_cast<const pq_recB *>(A + H2Y_PQ_NREC);
    const float x = half_bits_to_float(h);
    bool slow;
    float v = pq_fast(x, A, B, &slow);
    if (slow) v = pq_slow(x);
    lut[h] = v;
}

/*
 * Generic-width variant (width % 4 != 0, so rows are not 16-byte aligned):
 * one thread = 1 column x 2 rows, 4:4:4 or 4:4:4-to-scratch only (box needs
 * width % 4 == 0).  Correctness path for odd sizes, not tuned.
 */
template <int IN_KIND, int OUT_KIND>
__global__ __launch_bounds__(H2Y_FUSED_THREADS) void k_fused_narrow(fused_args a)
{
    __shared__ pq_recA s_tab[2 * H2Y_PQ_NREC]; /* A records, then B records */
    const pq_recA *sA = s_tab;
    const pq_recB *sB = reinterpret_cast<const pq_recB *>(s_tab + H2Y_PQ_NREC);
    __shared__ pix_params s_pp;
    if (a.pp.convert_transfer == 1) stage_table<H2Y_FUSED_THREADS>(a.table, s_tab);
    typedef in_traits<IN_KIND> IN;
    const pix_params pp = with_assumed(a.pp, a.assumed);
    if (threadIdx.x == 0) s_pp = pp;
    __syncthreads();
    const uint32_t W = a.width, H = a.height;
    const size_t npix = (size_t)W * H;
    const uint32_t G = gridDim.x;
    for (int f = 0; f < a.n_frames; f++) {
        const frame_io io = uniform_io(a.frames + f);
        mm6 mm;
        mm.reset();
        const uint32_t gbase = (uint32_t)(((uint64_t)f * a.chunks_per_frame) % G);
        uint32_t k = (blockIdx.x + G - gbase) % G;
        for (; k < a.chunks_per_frame; k += G) {
            const uint32_t tt = k * H2Y_FUSED_THREADS + threadIdx.x;
            if (tt >= a.tiles_per_frame) continue;
            uint32_t x;
            const uint32_t rp = udiv_magic(tt, W, a.wq_magic, x);
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const uint32_t y = rp * 2 + rr;
                if (y >= H) break;
                const size_t i = (size_t)y * W + x;
                float g = IN::load1(io.in[0], i), b = IN::load1(io.in[1], i), r = IN::load1(io.in[2], i);
                mm.add(0, g); mm.add(1, b); mm.add(2, r);
                uint32_t Y, Cb, Cr;
                pixel<H2Y_MODE_RUNTIME>(pp, &s_pp, sA, sB, g, b, r, Y, Cb, Cr);
                io.out[i] = (uint16_t)pix_yuv_clamp(pp, Y, false);
                if (OUT_KIND == H2Y_OUT_444) {
                    io.out[npix + i] = (uint16_t)pix_yuv_clamp(pp, Cb, true);
                    io.out[2 * npix + i] = (uint16_t)pix_yuv_clamp(pp, Cr, true);
                } else {
                    io.tmp_cb[i] = (uint16_t)Cb;
                    io.tmp_cr[i] = (uint16_t)Cr;
                }
            }
        }
        wave_store_mm(mm, a.partial + (((size_t)f * G + blockIdx.x) * (H2Y_FUSED_THREADS / WAVE) + threadIdx.x / WAVE) * 6);
    }
}

/* ---- pic_stats() as a pre-pass, common.cpp:116-139 ---------------------- */
template <int IN_KIND>
__global__ __launch_bounds__(H2Y_FUSED_THREADS) void k_stats(stats_args a)
{
    __shared__ float s_red[(H2Y_FUSED_THREADS / WAVE) * 6];
    typedef in_traits<IN_KIND> IN;
    mm6 mm;
    mm.reset();
    const size_t n = a.npix, n4 = a.vec_ok ? n / 4 : 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float v[4];
            IN::load4(a.in[c], i * 4, v);
            mm.add2(c, v[0], v[1]);
            mm.add2(c, v[2], v[3]);
        }
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
#pragma unroll
        for (int c = 0; c < 3; c++) mm.add(c, IN::load1(a.in[c], i));
    block_store_mm<H2Y_FUSED_THREADS / WAVE>(mm, s_red, a.partial + (size_t)blockIdx.x * 6);
}

/*
 * k_stats_final: one block per frame.  Reduces partial[frame][0..nblk) to the
 * six floats, derives estimated_floor/ceiling and compares with the values
 * the fused kernel assumed.
 *   F32/F16: (int)min, (int)max                     common.cpp:135-136
 *   U16    : min, max with the ceiling snap         common.cpp:91-106
 */
#define H2Y_FINAL_THREADS 1024 /* one block per frame over up to 4096 per-wave records: 15 us with 256 threads, the gap between two launches */
__global__ __launch_bounds__(H2Y_FINAL_THREADS) void k_stats_final(final_args a)
{
    __shared__ float s_red[(H2Y_FINAL_THREADS / WAVE) * 6];
    __shared__ uint32_t s_cnt;
    __shared__ int s_start[1024], s_finish[1024], s_min[8]; /* block clocks (grids hold at most 1024 blocks: four per CU) */
    const int f = blockIdx.x;
    const float *p = a.partial + (size_t)f * a.nblk * 6;
    frame_stats *out = a.out + f;
    /* Everything this block reads from memory is asked for here, at once: the kernel is a chain of short steps, and every
     * step that began with a load of its own cost another round trip -- 15 us in all, which is time between two launches. */
    mm6 mm;
    mm.reset();
    uint32_t cnt = 0;
    for (int i = threadIdx.x; i < a.nblk; i += blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            /* plain min/max: partials are never NaN (fminf/fmaxf dropped them) */
            mm.lo[c] = fminf(mm.lo[c], p[i * 6 + 2 * c]);
            mm.hi[c] = fmaxf(mm.hi[c], p[i * 6 + 2 * c + 1]);
        }
        if (a.redo_count) cnt += a.redo_count[(size_t)f * a.nblk + i];
    }
    int as_floor[3] = {0, 0, 0}, as_ceil[3] = {0, 0, 0};
    uint32_t low = 0;
    if (threadIdx.x == 0) {
        if (a.check)
            for (int c = 0; c < 3; c++) { as_floor[c] = a.assumed->floor_[c]; as_ceil[c] = a.assumed->ceil_[c]; }
        if (a.low_flag) low = a.low_flag[f];
    }
    const bool clocks = a.block_clock && f == 0;
    const int nb = min(a.grid, 1024);
    if (clocks) {
        const unsigned long long ref = a.block_clock[0];
        for (int b = (int)threadIdx.x; b < a.grid; b += (int)blockDim.x) {
            if (b < nb) {
                s_start[b] = (int)(long long)(a.block_clock[2 * b] - ref);
                s_finish[b] = (int)(long long)(a.block_clock[2 * b + 1] - ref);
            }
            a.block_clock[2 * b + 1] = 0ull; /* ready for the next launch's atomicMax */
        }
    }
    if (a.tail_ctr && f == 0)
        for (int i = (int)threadIdx.x; i < a.tail_n; i += (int)blockDim.x) a.tail_ctr[i] = 0u; /* ready for the next launch */
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    if (a.redo_count) { /* one add per wave */
#pragma unroll
        for (int o = WAVE / 2; o > 0; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o, WAVE);
        if ((threadIdx.x & (WAVE - 1)) == 0) atomicAdd(&s_cnt, cnt);
    }
    block_store_mm<H2Y_FINAL_THREADS / WAVE>(mm, s_red, out->mm); /* (two barriers inside: s_cnt is complete after it) */
    if (threadIdx.x == 0) {
        out->redone = a.redo_count ? s_cnt : 0u;
        int bad = 0;
        for (int c = 0; c < 3; c++) {
            float lo = s_red[2 * c], hi = s_red[2 * c + 1]; /* (out is pinned host memory: not to be read back) */
            int fl, ce;
            if (a.is_u16) {
                fl = (int)lo;
                ce = (int)hi;
                const int D = 1 << (a.src_bit_depth - 8), SMin = D * 16;
                const int YMax = 219 * D + SMin, CMax = 224 * D + SMin;
                if (ce < YMax && ce > (YMax * 3) / 4) ce = YMax;
                if (ce < CMax && ce > (CMax * 3) / 4) ce = CMax;
            } else {
                /* C truncation; the reference leaves |v| >= 2^31 undefined,
                 * we saturate (unpinned) */
                lo = fminf(fmaxf(lo, -2147483648.0f), 2147483520.0f);
                hi = fminf(fmaxf(hi, -2147483648.0f), 2147483520.0f);
                fl = (int)lo;
                ce = (int)hi;
            }
            out->floor_[c] = fl;
            out->ceil_[c] = ce;
            if (a.check && (fl != as_floor[c] || ce != as_ceil[c])) bad = 1;
            if (a.publish && f == 0) {
                a.publish->floor_[c] = fl;
                a.publish->ceil_[c] = ce;
            }
        }
        if (a.low_flag) { /* k_fused_t1 with the subsampled minimum: a sample <= -1 somewhere makes floor_ unknown */
            if (low) bad = 1;
            a.low_flag[f] = 0u; /* ready for the next launch */
        }
        out->mismatch = bad;
    }
    /* the launch's block clocks, per XCD (block b ran on XCD b % 8): the host balances the next launch by them.
     * 32-bit ticks relative to the earliest start (the starts lie within microseconds, a launch lasts milliseconds: 100 MHz
     * ticks fit easily); eight threads, one per XCD, add up their own blocks' from LDS. */
    if (clocks) {
        const int x = (int)threadIdx.x;
        if (x < 8) {
            int mn = 0; /* block 0 itself */
            for (int b = x; b < nb; b += 8) mn = min(mn, s_start[b]);
            s_min[x] = mn;
        }
        __syncthreads();
        if (x < 8) {
            int t0 = s_min[0];
#pragma unroll
            for (int k = 1; k < 8; k++) t0 = min(t0, s_min[k]);
            uint32_t sum = 0u, n = 0u;
            for (int b = x; b < nb; b += 8) {
                sum += (uint32_t)(s_finish[b] - t0);
                n++;
            }
            a.xcd_time[x] = n ? (float)sum / (float)n * 0.01f : 0.f; /* 100 MHz ticks -> us */
        }
        if (a.block_time) {
            int t0 = s_min[0];
#pragma unroll
            for (int k = 1; k < 8; k++) t0 = min(t0, s_min[k]);
            for (int b = x; b < nb; b += (int)blockDim.x) a.block_time[b] = (float)(s_finish[b] - t0) * 0.01f;
        }
    }
}

/* ---- Subsample444to420_box as a stand-alone stage, convert.cpp:91-172 --- */
__global__ __launch_bounds__(256) void k_box420(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, int W, int H)
{
    const int wc = W >> 1, hc = H >> 1;
    const int xc = blockIdx.x * blockDim.x + threadIdx.x, yc = blockIdx.y;
    if (xc >= wc || yc >= hc) return;
    const uint32_t *r0 = reinterpret_cast<const uint32_t *>(src + (size_t)(2 * yc) * W);
    const uint32_t *r1 = reinterpret_cast<const uint32_t *>(src + (size_t)(2 * yc + 1) * W);
    uint32_t a = r0[xc], b = r1[xc];
    dst[(size_t)yc * wc + xc] = (uint16_t)(((a & 0xFFFF) + (a >> 16) + (b & 0xFFFF) + (b >> 16)) >> 2);
}

/*
 * k_fir420: Subsample444to420_FIR (convert.cpp:261-383) for one plane, then
 * the write_yuv clamp.  Block = output tile FIR_TW x FIR_TH chroma samples.
 *   1. stage the 4:4:4 source rows 2*r0-5 .. 2*r0+2*FIR_TH+4, columns
 *      2*c0-5 .. 2*c0+2*FIR_TW+4 into LDS with the reference's edge
 *      replication (indices clamped, convert.cpp:295-300, :337-347)
 *   2. horizontal 7-tap -> u16 4:2:2 tile in LDS (clamped + truncated exactly
 *      as the reference stores dst422, convert.cpp:314-317)
 *   3. vertical 12-tap -> 4:2:0 sample, clamp, truncate, then shift + range
 *      clamp of write_yuv
 */
#ifndef FIR_TW
#define FIR_TW 64
#endif
#ifndef FIR_TH
#define FIR_TH 32
#endif
#define FIR_ROWS (2 * FIR_TH + 10)
#define FIR_COLS (2 * FIR_TW + 10)
#define FIR_LCOLS (2 * FIR_TW + 16) /* staged columns: 8 left of the tile (16-byte aligned), 8 right */

__global__ __launch_bounds__(256) void k_fir420(fir_args a)
{
    /* 4:4:4 source rows 2*r0-5 .. 2*r0+36, columns 2*c0-8 .. 2*c0+135 (16-byte aligned start) */
    __shared__ __attribute__((aligned(16))) uint16_t s444[FIR_ROWS][FIR_LCOLS];
    __shared__ __attribute__((aligned(16))) uint16_t s422[FIR_ROWS][FIR_TW];
    const int W = a.width, H = a.height, wc = W >> 1, hc = H >> 1;
    const uint16_t *src;
    uint16_t *dst;
    if (a.frames) { /* batch form: blockIdx.z = 2*frame + plane */
        const frame_io io = a.frames[blockIdx.z >> 1];
        uint16_t *cb = io.out + (size_t)W * H;
        src = (blockIdx.z & 1) ? io.tmp_cr : io.tmp_cb;
        dst = (blockIdx.z & 1) ? cb + (size_t)wc * hc : cb;
    } else {
        src = blockIdx.z == 0 ? a.src_cb : a.src_cr;
        dst = blockIdx.z == 0 ? a.dst_cb : a.dst_cr;
    }
    const int c0 = blockIdx.x * FIR_TW, r0 = blockIdx.y * FIR_TH;
    const int ys = 2 * r0 - 5, xs = 2 * c0 - 8;

    /* 1. stage the source tile; rows and columns outside the picture replicate the edge
     *    (the reference's clamped indices, convert.cpp:295-300 and :337-347) */
    const bool interior = xs >= 0 && xs + FIR_LCOLS <= W && (W & 7) == 0 && ((uintptr_t)src & 15) == 0;
    if (interior) {
        /* all of a thread's loads are issued before the first is stored to LDS (one exposed memory
         * latency per block instead of one per round) */
        constexpr int N16 = FIR_ROWS * (FIR_LCOLS / 8), ROUNDS = (N16 + 255) / 256;
        typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
        u32x4v q[ROUNDS];
#pragma unroll
        for (int k = 0; k < ROUNDS; k++) {
            const int i = min((int)threadIdx.x + 256 * k, N16 - 1);
            const int r = i / (FIR_LCOLS / 8), c8 = i - r * (FIR_LCOLS / 8);
            const int y = min(max(ys + r, 0), H - 1);
            q[k] = gload<u32x4v>(src, (size_t)(((size_t)y * W + xs + c8 * 8) >> 3));
        }
#pragma unroll
        for (int k = 0; k < ROUNDS; k++) {
            const int i = (int)threadIdx.x + 256 * k;
            if (i < N16) {
                const int r = i / (FIR_LCOLS / 8), c8 = i - r * (FIR_LCOLS / 8);
                *reinterpret_cast<u32x4v *>(&s444[r][c8 * 8]) = q[k];
            }
        }
    } else {
        for (int i = threadIdx.x; i < FIR_ROWS * FIR_LCOLS; i += 256) {
            const int r = i / FIR_LCOLS, c = i - r * FIR_LCOLS;
            const int y = min(max(ys + r, 0), H - 1), x = min(max(xs + c, 0), W - 1);
            s444[r][c] = src[(size_t)y * W + x];
        }
    }
    __syncthreads();

    /* 2. horizontal 7-tap at the even columns -> u16 4:2:2 tile (clamped and truncated exactly as
     *    the reference stores dst422, convert.cpp:314-317).  One item = one row x 8 outputs:
     *    32 source samples from four 16-byte LDS reads. */
    for (int i = threadIdx.x; i < FIR_ROWS * (FIR_TW / 8); i += 256) {
        const int r = i / (FIR_TW / 8), g = i - r * (FIR_TW / 8);
        uint32_t w[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 t = *reinterpret_cast<const uint4 *>(&s444[r][16 * g + 8 * q]);
            w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
        }
        float s[32];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            s[2 * q] = (float)(w[q] & 0xFFFFu);
            s[2 * q + 1] = (float)(w[q] >> 16);
        }
        uint32_t o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int ctr = 8 + 2 * j; /* local index of the even column 2*(c0 + 8g + j) */
            o[j] = fir_h(s[ctr - 5], s[ctr - 3], s[ctr - 1], s[ctr], s[ctr + 1], s[ctr + 3], s[ctr + 5], a.fir_max);
        }
        *reinterpret_cast<uint4 *>(&s422[r][8 * g]) =
            make_uint4(o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16));
    }
    __syncthreads();

    /* 3. vertical 12-tap (convert.cpp:365-374), then write_yuv's shift + range clamp.
     *    One item = one output row x 4 outputs. */
    for (int i = threadIdx.x; i < FIR_TH * (FIR_TW / 4); i += 256) {
        const int r = i / (FIR_TW / 4), g = i - r * (FIR_TW / 4);
        const int yo = r0 + r, xo = c0 + 4 * g;
        if (yo >= hc || xo >= wc) continue;
        float t[12][4];
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const uint2 q = *reinterpret_cast<const uint2 *>(&s422[2 * r + k][4 * g]);
            t[k][0] = (float)(q.x & 0xFFFFu); t[k][1] = (float)(q.x >> 16);
            t[k][2] = (float)(q.y & 0xFFFFu); t[k][3] = (float)(q.y >> 16);
        }
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t v = fir_v(t[0][j], t[1][j], t[2][j], t[3][j], t[4][j], t[5][j], t[6][j], t[7][j], t[8][j], t[9][j], t[10][j],
                               t[11][j], a.fir_max);
            if (a.apply_yuv_clamp) v = pix_yuv_clamp(a.pp, v, true);
            o[j] = v;
        }
        uint16_t *d = dst + (size_t)yo * wc + xo;
        if (xo + 3 < wc && (((uintptr_t)d) & 7) == 0) *reinterpret_cast<uint2 *>(d) = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
        else
            for (int j = 0; j < 4 && xo + j < wc; j++) d[j] = (uint16_t)o[j];
    }
}

/*
 * k_inverse: matrix_inverse(), convert.cpp:1320-1867, U16 4:4:4 Y'CbCr / Y'DzDx -> U16 G,B,R planes
 * (the .yuv -> .tiff flow, hdr2yuv.cpp:818-819).  Elementwise, 12 B/px, HBM-bound.  One thread = four
 * consecutive samples of each plane (8-byte loads and stores), grid-stride.  The arithmetic is the
 * reference's, operation by operation and type by type, with the behaviour of the compiled function:
 * Half = 2048 and Full = 4096 at every bit depth; only matrix_coeffs 1 takes the BT.709 equations
 * (the test at convert.cpp:1391 compares matrix_coeffs with booleans), everything else Y'DzDx; the
 * video-range clamp always runs, with the input picture's limits.
 */
__global__ __launch_bounds__(256) void k_inverse(inverse_args a)
{
    const uint32_t n4 = a.npix >> 2;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const u32x2 y = gload<u32x2>(a.in[0], i), cb = gload<u32x2>(a.in[1], i), cr = gload<u32x2>(a.in[2], i);
        uint32_t G[4], B[4], R[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t wy = j & 2 ? y.y : y.x, wb = j & 2 ? cb.y : cb.x, wr = j & 2 ? cr.y : cr.x;
            inverse_pixel(a, j & 1 ? wy >> 16 : wy & 0xFFFFu, j & 1 ? wb >> 16 : wb & 0xFFFFu, j & 1 ? wr >> 16 : wr & 0xFFFFu, G[j], B[j], R[j]);
        }
        gstore<u32x2>(a.out[0], i, u32x2{G[0] | (G[1] << 16), G[2] | (G[3] << 16)});
        gstore<u32x2>(a.out[1], i, u32x2{B[0] | (B[1] << 16), B[2] | (B[3] << 16)});
        gstore<u32x2>(a.out[2], i, u32x2{R[0] | (R[1] << 16), R[2] | (R[3] << 16)});
    }
    /* the last npix % 4 samples */
    const uint32_t tail = a.npix & 3u;
    if (blockIdx.x == 0 && threadIdx.x < tail) {
        const uint32_t i = (n4 << 2) + threadIdx.x;
        uint32_t G, B, R;
        inverse_pixel(a, gload<uint16_t>(a.in[0], i), gload<uint16_t>(a.in[1], i), gload<uint16_t>(a.in[2], i), G, B, R);
        gstore<uint16_t>(a.out[0], i, (uint16_t)G);
        gstore<uint16_t>(a.out[1], i, (uint16_t)B);
        gstore<uint16_t>(a.out[2], i, (uint16_t)R);
    }
}

/* ---- launch helpers (called from h2y_api.hip) --------------------------- */
typedef void (*fused_fn)(fused_args);

template <int IN_KIND, int OUT_KIND, int MODE> static fused_fn pick_pipe(int pipe, bool even_h)
{
    switch (pipe) {
    case H2Y_PIPE_PQ_IDENT: return even_h ? k_fused2<IN_KIND, OUT_KIND, MODE, H2Y_PIPE_PQ_IDENT> : k_fused<IN_KIND, OUT_KIND, MODE, H2Y_PIPE_PQ_IDENT>;
    case H2Y_PIPE_PQ_NORM: return even_h ? k_fused2<IN_KIND, OUT_KIND, MODE, H2Y_PIPE_PQ_NORM> : k_fused<IN_KIND, OUT_KIND, MODE, H2Y_PIPE_PQ_NORM>;
    case H2Y_PIPE_NONE: return k_fused2<IN_KIND, OUT_KIND, MODE, H2Y_PIPE_NONE>;
    case H2Y_PIPE_TFN: return k_fused2<IN_KIND, OUT_KIND, MODE, H2Y_PIPE_TFN>;
    default: return k_fused<IN_KIND, OUT_KIND, MODE, H2Y_PIPE_RUNTIME>;
    }
}
template <int IN_KIND, int OUT_KIND> static fused_fn pick_mode(int mode, int pipe, bool even_h)
{
    switch (mode) {
    case H2Y_MODE_YCBCR: return pick_pipe<IN_KIND, OUT_KIND, H2Y_MODE_YCBCR>(pipe, even_h);
    case H2Y_MODE_YDZDX: return pick_pipe<IN_KIND, OUT_KIND, H2Y_MODE_YDZDX>(pipe, even_h);
    default: return k_fused<IN_KIND, OUT_KIND, H2Y_MODE_RUNTIME, H2Y_PIPE_RUNTIME>; /* identity / Y100 / Y500: generic */
    }
}
template <int IN_KIND> static fused_fn pick_out(const fused_variant &v)
{
    if (v.narrow) return v.out_kind == H2Y_OUT_444 ? k_fused_narrow<IN_KIND, H2Y_OUT_444> : k_fused_narrow<IN_KIND, H2Y_OUT_444TMP>;
    switch (v.out_kind) {
    case H2Y_OUT_420BOX: return pick_mode<IN_KIND, H2Y_OUT_420BOX>(v.mode, v.pipe, v.even_h);
    case H2Y_OUT_444: return pick_mode<IN_KIND, H2Y_OUT_444>(v.mode, v.pipe, v.even_h);
    default: return pick_mode<IN_KIND, H2Y_OUT_444TMP>(v.mode, v.pipe, v.even_h);
    }
}
template <int OUT_KIND> static fused_fn pick_lut_mode(int mode, bool cols8)
{
    if (cols8) return mode == H2Y_MODE_YCBCR ? k_fused_lut16<OUT_KIND, H2Y_MODE_YCBCR, 8> : k_fused_lut16<OUT_KIND, H2Y_MODE_YDZDX, 8>;
    return mode == H2Y_MODE_YCBCR ? k_fused_lut16<OUT_KIND, H2Y_MODE_YCBCR, 4> : k_fused_lut16<OUT_KIND, H2Y_MODE_YDZDX, 4>;
}
template <int IN_KIND, int OUT_KIND> static fused_fn pick_t1_mode(int mode, int pipe)
{
    if (mode == H2Y_MODE_YCBCR)
        return pipe == 4 ? k_fused_t1<IN_KIND, OUT_KIND, H2Y_MODE_YCBCR, H2Y_PIPE_PQ_IDENT> : k_fused_t1<IN_KIND, OUT_KIND, H2Y_MODE_YCBCR, H2Y_PIPE_PQ_NORM>;
    return pipe == 4 ? k_fused_t1<IN_KIND, OUT_KIND, H2Y_MODE_YDZDX, H2Y_PIPE_PQ_IDENT> : k_fused_t1<IN_KIND, OUT_KIND, H2Y_MODE_YDZDX, H2Y_PIPE_PQ_NORM>;
}
template <int IN_KIND> static fused_fn pick_t1_out(int out_kind, int mode, int pipe)
{
    switch (out_kind) {
    case H2Y_OUT_420BOX: return pick_t1_mode<IN_KIND, H2Y_OUT_420BOX>(mode, pipe);
    case H2Y_OUT_444: return pick_t1_mode<IN_KIND, H2Y_OUT_444>(mode, pipe);
    default: return pick_t1_mode<IN_KIND, H2Y_OUT_444TMP>(mode, pipe);
    }
}
const char *h2y_fused_name(const fused_variant &v)
{
    if (v.narrow) return "k_fused_narrow";
    if (v.pipe == 3) return "k_fused_lut16";
    if (v.pipe == 4 || v.pipe == 5) return "k_fused_t1";
    if (v.pipe == H2Y_PIPE_TFN) return "k_fused2";
    return ((v.pipe == 1 || v.pipe == 2 || v.pipe == H2Y_PIPE_NONE) && v.even_h && (v.mode == H2Y_MODE_YCBCR || v.mode == H2Y_MODE_YDZDX)) ? "k_fused2" : "k_fused";
}
bool h2y_fused_grouped(const fused_variant &v)
{
    const char *n = h2y_fused_name(v); /* the loop-form kernels: k_fused2, k_fused_t1, k_fused_lut16 */
    return !strcmp(n, "k_fused2") || !strcmp(n, "k_fused_t1") || !strcmp(n, "k_fused_lut16");
}
int h2y_fused_threads(const fused_variant &v)
{
    if (v.pipe == 4 || v.pipe == 5) return H2Y_T1_THREADS;
    return h2y_fused_grouped(v) ? H2Y_LOOP_THREADS : H2Y_FUSED_THREADS; /* k_fused2 / k_fused_lut16 : k_fused / k_fused_narrow */
}

static fused_fn pick_fused(const fused_variant &v)
{
    if (v.pipe == 4 || v.pipe == 5) /* first-tier kernels: float inputs only */
        return v.in_kind == H2Y_IN_F16 ? pick_t1_out<H2Y_IN_F16>(v.out_kind, v.mode, v.pipe) : pick_t1_out<H2Y_IN_F32>(v.out_kind, v.mode, v.pipe);
    if (v.pipe == 3) {
        switch (v.out_kind) {
        case H2Y_OUT_420BOX: return pick_lut_mode<H2Y_OUT_420BOX>(v.mode, v.cols8);
        case H2Y_OUT_444: return pick_lut_mode<H2Y_OUT_444>(v.mode, v.cols8);
        default: return pick_lut_mode<H2Y_OUT_444TMP>(v.mode, v.cols8);
        }
    }
    switch (v.in_kind) {
    case H2Y_IN_F32: return pick_out<H2Y_IN_F32>(v);
    case H2Y_IN_F16: return pick_out<H2Y_IN_F16>(v);
    default: return pick_out<H2Y_IN_U16>(v);
    }
}

/* resident blocks per CU for this variant (the grid is sized to exactly fill the chip) */
int h2y_fused_blocks_per_cu(const fused_variant &v)
{
    int nb = 0;
    fused_fn fn = pick_fused(v);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(fn), h2y_fused_threads(v), 0) != hipSuccess || nb < 1)
        nb = 1;
    return nb;
}

hipError_t h2y_launch_fused(const fused_variant &v, int grid, hipStream_t st, const fused_args &a)
{
    fused_fn fn = pick_fused(v);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(h2y_fused_threads(v)), 0, st, a);
    return hipGetLastError();
}

hipError_t h2y_launch_build_lut16(hipStream_t st, const void *table, float *lut)
{
    hipLaunchKernelGGL(k_build_lut16, dim3(H2Y_LUT16_N / 256), dim3(256), 0, st, table, lut);
    return hipGetLastError();
}

hipError_t h2y_launch_stats(int in_kind, int grid, hipStream_t st, const stats_args &a)
{
    dim3 blk(H2Y_FUSED_THREADS);
    switch (in_kind) {
    case H2Y_IN_F32: hipLaunchKernelGGL((k_stats<H2Y_IN_F32>), dim3(grid), blk, 0, st, a); break;
    case H2Y_IN_F16: hipLaunchKernelGGL((k_stats<H2Y_IN_F16>), dim3(grid), blk, 0, st, a); break;
    default: hipLaunchKernelGGL((k_stats<H2Y_IN_U16>), dim3(grid), blk, 0, st, a); break;
    }
    return hipGetLastError();
}

hipError_t h2y_launch_stats_final(int n_frames, hipStream_t st, const final_args &a)
{
    hipLaunchKernelGGL(k_stats_final, dim3(n_frames), dim3(H2Y_FINAL_THREADS), 0, st, a);
    return hipGetLastError();
}

hipError_t h2y_launch_inverse(int grid, hipStream_t st, const inverse_args &a)
{
    hipLaunchKernelGGL(k_inverse, dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t h2y_launch_fir420(hipStream_t st, const fir_args &a)
{
    const int wc = a.width >> 1, hc = a.height >> 1;
    dim3 grid((wc + FIR_TW - 1) / FIR_TW, (hc + FIR_TH - 1) / FIR_TH, a.frames ? 2 * a.n_frames : (a.src_cr ? 2 : 1));
    hipLaunchKernelGGL(k_fir420, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t h2y_launch_box420(hipStream_t st, const uint16_t *src, uint16_t *dst, int W, int H)
{
    const int wc = W >> 1, hc = H >> 1;
    dim3 grid((wc + 255) / 256, hc);
    hipLaunchKernelGGL(k_box420, grid, dim3(256), 0, st, src, dst, W, H);
    return hipGetLastError();
}
