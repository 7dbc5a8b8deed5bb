/* h2y_walk.h -- which frames and which parts of them a block of the loop-form kernels (k_fused2, k_fused_t1,
 * k_fused_lut16) works on.  Host and device: tools/walk_check.cpp (run by tests/test_walk.py) checks on the CPU that
 * every chunk / slice of every frame is dealt to exactly one block, for the same code the kernels and the shim run. */
#ifndef H2Y_WALK_H
#define H2Y_WALK_H
#include "h2y_math.h" /* H2Y_FN */

namespace h2y {

struct walk_args {
    uint32_t groups, xcd_layout, chunks_per_frame;
    int n_frames;
};

/*
 * Frame groups.  A wave pays a fixed price per frame (min/max reduction over the wave, frame descriptors, the first
 * tile's latency: ~230 vector instructions, a third of a tile's), and with every block on every frame a 4K frame is only
 * four tiles per lane.  So the grid works as a.groups groups of G = gridDim.x / groups blocks; group g takes frames
 * g, g + groups, ...: the same price, groups times as many tiles per lane and frame.  With a.xcd_layout (gridDim.x a
 * multiple of 8 * groups) a group is made of whole rounds of the eight XCDs (block b runs on XCD b % 8): group
 * (b / 8) % groups, number ((b / 8) / groups) * 8 + b % 8 in it -- so block i of a group runs on XCD i % 8.
 *
 * Within a frame, a group's blocks share the work in one of two ways:
 *   round robin   chunk c (blockDim.x tiles) of the group's n-th frame goes to block (n * cpf + c) % G: first(), succ()
 *                 and the closed form k0 + j G, j < count() that the kernels' ticket dealing uses (the rotation from
 *                 frame to frame keeps the blocks level when G does not divide cpf);
 *   slice ranges  block i takes the 64-tile slices [r[i], r[i+1]) of every frame of the group, r cut in proportion to
 *                 the speeds of the blocks' XCDs (slice_ranges() below; the XCDs of a card are not equally fast on these
 *                 kernels, and a launch ends when its slowest block ends).  Used whenever the XCD layout applies.
 */
struct frame_walk {
    uint32_t G, NG, bi; /* blocks per group, groups, this block's number in its group */
    uint32_t cpf, mod, gb;
    int f, n_frames;
    uint32_t k0, k0_n;  /* this block's first chunk in this frame and in the group's next one; >= cpf: none */

    H2Y_FN uint32_t wrap(uint32_t x) const { return x >= G ? x - G : x; } /* x < 2 G */
    H2Y_FN void set_firsts()
    {
        k0 = wrap(bi + G - gb);
        k0_n = wrap(bi + G - wrap(gb + mod));
    }
    H2Y_FN void init(const walk_args &a, uint32_t b /* block */, uint32_t grid /* blocks */)
    {
        NG = a.groups;
        G = grid / NG;
        cpf = a.chunks_per_frame;
        n_frames = a.n_frames;
        if (a.xcd_layout) {
            f = (int)((b >> 3) % NG);
            bi = ((b >> 3) / NG) * 8u + (b & 7u);
        } else {
            f = (int)(b % NG);
            bi = b / NG;
        }
        mod = cpf % G;
        gb = 0;
        set_firsts();
    }
    H2Y_FN bool has_next() const { return f + (int)NG < n_frames; }
    H2Y_FN void advance()
    {
        gb = wrap(gb + mod);
        f += (int)NG;
        set_firsts();
    }
    /* first() / succ(): a block's chunks of a frame in order, and the step across the frame boundary -- the definition
     * tools/walk_check.cpp checks the closed form (k0, count()) against */
    H2Y_FN bool first(uint32_t &k) const
    {
        k = k0;
        return k0 < cpf;
    }
    /* the chunk after k: 1 = in the same frame, 2 = in the group's next frame, 0 = none (k2 = k) */
    H2Y_FN int succ(uint32_t k, uint32_t &k2) const
    {
        const uint32_t ks = k + G;
        const bool same = ks < cpf;
        const bool next = !same && has_next() && k0_n < cpf;
        k2 = same ? ks : next ? k0_n : k;
        return same ? 1 : next ? 2 : 0;
    }
    /* how many chunks this block owns in a frame whose first one is k (k, k + G, ... below cpf) */
    H2Y_FN uint32_t count(uint32_t k) const { return k < cpf ? (cpf - 1u - k) / G + 1u : 0u; }
    /* a block's slot in the per-frame arrays: [frame][block of the group] */
    H2Y_FN size_t slot() const { return (size_t)f * G + bi; }
};

/* Slice ranges: r[0 .. G] with r[0] = 0, r[G] = n_slices, block i of a group taking [r[i], r[i+1]); the share of block i
 * is proportional to w[i]: the measured speed of that block (round 3), or of the XCD it runs on (speed[i % 8] under the XCD
 * layout). */
inline void slice_ranges_w(const double *w /* [G] */, uint32_t G, uint32_t n_slices, uint32_t *r)
{
    double tot = 0.0;
    for (uint32_t i = 0; i < G; i++) tot += w[i];
    double cum = 0.0;
    for (uint32_t i = 0; i <= G; i++) {
        r[i] = i == G ? n_slices : (uint32_t)(cum / tot * (double)n_slices + 0.5);
        if (i < G) cum += w[i];
    }
}
inline void slice_ranges(const double speed[8], uint32_t G, uint32_t n_slices, uint32_t *r)
{
    double w[1024];
    for (uint32_t i = 0; i < G && i < 1024u; i++) w[i] = speed[i % 8u];
    slice_ranges_w(w, G, n_slices, r);
}
/* the block of the grid that is block i of group g under the XCD layout (the inverse of frame_walk::init) */
H2Y_FN uint32_t walk_block_of(uint32_t g, uint32_t i, uint32_t NG) { return (((i >> 3) * NG + g) << 3) | (i & 7u); }

/*
 * The dynamic last frame (round 3).  With every frame dealt in fixed shares a launch ends when its slowest block ends, and
 * the blocks' finish times spread over 2-4 % of a launch however the shares are cut (the spread is not the same from one
 * launch to the next: shares cut by every block's own measured speed ran no faster than shares cut by its XCD's).  So the
 * LAST frame of every group is not dealt at all: a block that has finished its shares of the other frames draws slices of
 * that frame from counters in global memory until none is left -- blocks that finish early do more of it.  Draws on one
 * word go one after the other, 0.4 us each under load (measured: 2 025 draws per counter made the frame take 330 us instead
 * of 45), so a BLOCK draws sixteen slices at a time -- its waves share them through a ticket in LDS, the next chunk is asked
 * for while this one is worked on -- and the frame is cut in sixty-four parts with a counter each (sixteen draws per counter
 * and 4K frame, at device scope): a block starts at a part of its own and goes round; a part found exhausted is marked in
 * two words of bits next to the counters, read before every draw, so that at the end of the frame a block does not ask
 * sixty-four empty counters in turn.  tail_state is the whole rule; the kernels add the atomics, tools/walk_check.cpp plays
 * it with random interleavings.
 */
#define H2Y_TAIL_CHUNK 32u /* slices a BLOCK draws at a time: two per wave of a 1024-thread block (a draw takes microseconds under load:
                              with sixteen the drawing wave could not stay a chunk ahead of the other fifteen) */
#define H2Y_TAIL_PARTS 64u /* counters per group: the frame's slices cut in as many equal parts */
#define H2Y_TAIL_WORDS (H2Y_TAIL_PARTS + 2u) /* per group: the counters, then two words of "part is exhausted" bits */
struct tail_state {
    uint32_t part, tried; /* the part being drawn from; parts found (or known to be) exhausted so far */
    H2Y_FN void init(uint32_t start) { part = start & (H2Y_TAIL_PARTS - 1u); tried = 0; }
    H2Y_FN bool done() const { return tried >= H2Y_TAIL_PARTS; }
    H2Y_FN void range(uint32_t n_slices, uint32_t *lo, uint32_t *hi) const
    {
        *lo = (uint32_t)(((uint64_t)n_slices * part) / H2Y_TAIL_PARTS);
        *hi = (uint32_t)(((uint64_t)n_slices * (part + 1u)) / H2Y_TAIL_PARTS);
    }
    H2Y_FN void skip() /* the part is exhausted: on to the next one */
    {
        tried++;
        part = (part + 1u) & (H2Y_TAIL_PARTS - 1u);
    }
    /* `c` = what the part's counter held before this block's increment: the c-th chunk of H2Y_TAIL_CHUNK slices of the part
     * (the last one may be shorter); false: the part is exhausted (the caller marks it so and skip()s) */
    H2Y_FN bool take(uint32_t n_slices, uint32_t c, uint32_t *first, uint32_t *count) const
    {
        uint32_t lo, hi;
        range(n_slices, &lo, &hi);
        const uint64_t start = (uint64_t)lo + (uint64_t)c * H2Y_TAIL_CHUNK;
        if (start >= hi) return false;
        *first = (uint32_t)start;
        *count = hi - (uint32_t)start < H2Y_TAIL_CHUNK ? hi - (uint32_t)start : H2Y_TAIL_CHUNK;
        return true;
    }
};

} // namespace h2y
#endif
