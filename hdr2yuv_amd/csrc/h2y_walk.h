/* h2y_walk.h -- which frames and chunks a block of the loop-form kernels (k_fused2, k_fused_t1,
 * k_fused_lut16) works on.  Host and device: tools/walk_check.cpp (run by tests/test_walk.py) checks on
 * the CPU that every chunk of every frame is dealt to exactly one block, for the same code the kernels run. */
#ifndef H2Y_WALK_H
#define H2Y_WALK_H
#include "h2y_math.h" /* H2Y_FN */

namespace h2y {

struct walk_args {
    uint32_t groups, xcd_layout, fast_mask, chunks_a, chunks_per_frame;
    int n_frames;
};

/*
 * Which frames a block of the loop-form kernels works on, and which chunks of them.
 *
 * Frame groups.  A wave pays a fixed price per frame (min/max reduction over the wave, frame
 * descriptors, the first tile's latency: ~230 vector instructions, a third of a tile's), and with every
 * block on every frame a 4K frame is only four tiles per lane.  So the grid works as a.groups groups of
 * G = gridDim.x / groups blocks; group g takes frames g, g + groups, ...: the same price, groups times
 * as many tiles per lane and frame.  With a.xcd_layout (gridDim.x a multiple of 8 * groups) a group is
 * made of whole rounds of the eight XCDs (block b runs on XCD b % 8): group (b / 8) % groups, number
 * ((b / 8) / groups) * 8 + b % 8 in it.
 *
 * Weighted rounds.  The XCDs of a card are not equally fast on this kernel (measured: the odd ones
 * 7 % slower; tools/blocktimes.py), and a static round-robin ends when the slowest block ends.  A
 * frame's chunks are therefore dealt in two parts: chunks [0, chunks_a) go round ALL blocks of the
 * group, chunks [chunks_a, chunks_per_frame) round the blocks on the FAST XCDs (a.fast_mask) only.  The
 * host sets the split from the finish times of the previous launch (balance_update() in h2y_api.hip).
 * Both rounds rotate from frame to frame (chunk c of the group's n-th frame: block (n chunks_a + c) % G
 * in part A, fast block (n chunks_b + c - chunks_a) % Gf in part B), so blocks stay level across frames.
 */
struct frame_walk {
    uint32_t G, NG, bi;           /* blocks per group, groups, this block's number in its group */
    uint32_t Gf, fi;              /* fast blocks per group, this block's number among them */
    bool fast;
    uint32_t cpf, cpfA;           /* chunks per frame, chunks of part A */
    uint32_t modA, modB, gbA, gbB;
    int f, n_frames;
    uint32_t kA, kB, kA_n, kB_n;  /* this block's first chunk of part A / B in this frame and in the group's next one; >= the part's end: none */

    H2Y_FN uint32_t wrapA(uint32_t x) const { return x >= G ? x - G : x; }   /* x < 2 G */
    H2Y_FN uint32_t wrapB(uint32_t x) const { return x >= Gf ? x - Gf : x; } /* x < 2 Gf */
    H2Y_FN void set_firsts()
    {
        kA = wrapA(bi + G - gbA);
        kB = fast ? cpfA + wrapB(fi + Gf - gbB) : cpf;
        kA_n = wrapA(bi + G - wrapA(gbA + modA));
        kB_n = fast ? cpfA + wrapB(fi + Gf - wrapB(gbB + modB)) : cpf;
    }
    H2Y_FN void init(const walk_args &a, uint32_t b /* block */, uint32_t grid /* blocks */)
    {
        NG = a.groups;
        G = grid / NG;
        cpf = a.chunks_per_frame;
        n_frames = a.n_frames;
        if (a.xcd_layout) {
            const uint32_t xcd = b & 7u, nf8 = (uint32_t)__builtin_popcount(a.fast_mask & 0xFFu);
            f = (int)((b >> 3) % NG);
            bi = ((b >> 3) / NG) * 8u + xcd;
            Gf = (G >> 3) * nf8;
            fast = ((a.fast_mask >> xcd) & 1u) != 0;
            fi = (bi >> 3) * nf8 + (uint32_t)__builtin_popcount(a.fast_mask & ((1u << xcd) - 1u));
            cpfA = a.chunks_a;
        } else {
            f = (int)(b % NG);
            bi = b / NG;
            Gf = G;
            fast = false;
            fi = 0;
            cpfA = cpf; /* one part */
        }
        modA = cpfA % G;
        modB = (cpf - cpfA) % Gf;
        gbA = gbB = 0;
        set_firsts();
    }
    H2Y_FN bool has_next() const { return f + (int)NG < n_frames; }
    H2Y_FN void advance()
    {
        gbA = wrapA(gbA + modA);
        gbB = wrapB(gbB + modB);
        f += (int)NG;
        set_firsts();
    }
    /* first() / succ(): a block's chunks of a frame in order, and the step across the frame boundary.  The kernels
     * used to walk them wave by wave; since waves take their tiles by ticket (wave_deal in h2y_kernels.hip) they
     * use kA / kB / count_a() / count_b() directly, and these two remain as the definition tools/walk_check.cpp
     * checks the closed form against. */
    /* this block's first chunk of the current frame */
    H2Y_FN bool first(uint32_t &k, bool &inB) const
    {
        const bool hasA = kA < cpfA;
        k = hasA ? kA : kB;
        inB = !hasA;
        return hasA || kB < cpf;
    }
    /* the chunk after k: 1 = in the same frame, 2 = in the group's next frame, 0 = none (k2 = k) */
    H2Y_FN int succ(uint32_t k, bool inB, uint32_t &k2, bool &inB2) const
    {
        const uint32_t ks = k + (inB ? Gf : G);
        const bool same1 = ks < (inB ? cpf : cpfA);
        const bool toB = !same1 && !inB && kB < cpf;
        const bool nA = kA_n < cpfA, nB = kB_n < cpf;
        const bool next = !same1 && !toB && has_next() && (nA || nB);
        k2 = same1 ? ks : toB ? kB : next ? (nA ? kA_n : kB_n) : k;
        inB2 = same1 ? inB : toB ? true : next ? !nA : inB;
        return (same1 || toB) ? 1 : next ? 2 : 0;
    }
    /* how many chunks of parts A and B this block owns in the current frame / the group's next one
     * (they are kA, kA + G, ... below cpfA and kB, kB + Gf, ... below cpf) */
    H2Y_FN uint32_t count_a(uint32_t k) const { return k < cpfA ? (cpfA - 1u - k) / G + 1u : 0u; }
    H2Y_FN uint32_t count_b(uint32_t k) const { return k < cpf ? (cpf - 1u - k) / Gf + 1u : 0u; }
    /* a block's slot in the per-frame arrays: [frame][block of the group] */
    H2Y_FN size_t slot() const { return (size_t)f * G + bi; }
};


} // namespace h2y
#endif
