// tools/walk_check.cpp -- CPU check of hdr2yuv_amd/csrc/h2y_walk.h, the code by which the loop-form kernels deal frames
// and their parts to blocks: frame groups, XCD-aware layout, round-robin chunks, slice ranges by XCD speed.
// For every configuration: each chunk of each frame is visited by exactly one block, exactly once; a block's prefetch
// target across a frame boundary (succ() kind 2) is the chunk it starts the next frame with (first()); a frame is only
// touched by blocks of its group; the closed form the kernels' ticket dealing uses (k0 + j G, j < count()) deals the
// same chunks.  For the slice ranges: r is a partition of the frame's slices in block order, and every block's share is
// within one slice of its XCD's proportional share.
// Build: g++ -O2 -std=c++17 -I hdr2yuv_amd/csrc tools/walk_check.cpp -o /tmp/walk_check
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "h2y_walk.h"
using namespace h2y;

static long n_cfg = 0, n_bad = 0;

static void check(uint32_t grid, uint32_t groups, uint32_t layout, uint32_t cpf, int nfr)
{
    walk_args a;
    a.groups = groups; a.xcd_layout = layout; a.chunks_per_frame = cpf; a.n_frames = nfr;
    std::vector<std::vector<int>> seen(nfr, std::vector<int>(cpf, 0)), seen2(nfr, std::vector<int>(cpf, 0));
    std::vector<std::vector<int>> group_of(nfr, std::vector<int>(cpf, -1));
    bool bad = false;
    for (uint32_t b = 0; b < grid && !bad; b++) {
        frame_walk fw;
        bool expect = false, expect_none = false; /* succ() announced this block's first chunk of the next frame / that there is none */
        uint32_t ek = 0;
        int grp = -1;
        bool have_n = false;
        uint32_t k0_n = 0;
        for (fw.init(a, b, grid); fw.f < nfr; fw.advance()) {
            if (grp < 0) grp = fw.f;
            if ((fw.f - grp) % (int)groups != 0) bad = true;
            if (layout && fw.bi % 8u != b % 8u) bad = true; /* block i of a group runs on XCD i % 8 */
            if (have_n && fw.k0 != k0_n) { bad = true; break; }
            have_n = true;
            k0_n = fw.k0_n;
            const uint32_t cnt = fw.count(fw.k0);
            for (uint32_t j = 0; j < cnt; j++) {
                const uint32_t k = fw.k0 + j * fw.G;
                if (k >= cpf) { bad = true; break; }
                seen2[fw.f][k]++;
            }
            uint32_t k;
            bool more = fw.first(k);
            if (expect && (!more || k != ek)) { bad = true; break; }
            if (expect_none && more) { bad = true; break; }
            expect = expect_none = false;
            long guard = 0;
            while (more) {
                if (k >= cpf) { bad = true; break; }
                seen[fw.f][k]++;
                group_of[fw.f][k] = grp;
                uint32_t k2;
                const int kind = fw.succ(k, k2);
                if (kind == 2) { expect = true; ek = k2; if (!fw.has_next()) bad = true; }
                if (kind == 0) { if (fw.has_next()) expect_none = true; if (k2 != k) bad = true; }
                more = kind == 1;
                k = k2;
                if (++guard > (long)cpf + 4) { bad = true; break; }
            }
            if (bad) break;
        }
    }
    for (int f = 0; f < nfr && !bad; f++)
        for (uint32_t c = 0; c < cpf; c++)
            if (seen[f][c] != 1 || seen2[f][c] != 1 || group_of[f][c] != f % (int)groups) { bad = true; break; }
    n_cfg++;
    if (bad) {
        n_bad++;
        if (n_bad < 20) printf("BAD grid %u groups %u layout %u cpf %u frames %d\n", grid, groups, layout, cpf, nfr);
    }
}

static void check_ranges(uint32_t G, uint32_t n_slices, const double sp[8])
{
    std::vector<uint32_t> r(G + 1);
    slice_ranges(sp, G, n_slices, r.data());
    bool bad = r[0] != 0 || r[G] != n_slices;
    double tot = 0;
    for (uint32_t i = 0; i < G; i++) tot += sp[i % 8];
    for (uint32_t i = 0; i < G && !bad; i++) {
        if (r[i + 1] < r[i]) bad = true;
        const double want = sp[i % 8] / tot * n_slices;
        if (std::fabs((double)(r[i + 1] - r[i]) - want) > 1.0) bad = true;
    }
    n_cfg++;
    if (bad) {
        n_bad++;
        if (n_bad < 20) printf("BAD ranges G %u slices %u\n", G, n_slices);
    }
}

/* per-block weights: every group's table is a partition with shares within one slice of their proportion, and
 * walk_block_of() is the inverse of frame_walk::init()'s numbering under the XCD layout */
static void check_block_ranges(uint32_t grid, uint32_t groups, uint32_t n_slices, unsigned seed)
{
    const uint32_t G = grid / groups;
    std::vector<double> bs(grid);
    unsigned s = seed;
    for (uint32_t b = 0; b < grid; b++) {
        s = s * 1664525u + 1013904223u;
        bs[b] = 0.75 + 0.5 * (double)(s >> 8) / 16777216.0;
    }
    bool bad = false;
    std::vector<int> hit(grid, 0);
    for (uint32_t g = 0; g < groups && !bad; g++) {
        std::vector<double> w(G);
        double tot = 0;
        for (uint32_t i = 0; i < G; i++) {
            const uint32_t b = walk_block_of(g, i, groups);
            if (b >= grid) { bad = true; break; }
            hit[b]++;
            walk_args wa;
            wa.groups = groups; wa.xcd_layout = 1; wa.chunks_per_frame = 16; wa.n_frames = (int)groups;
            frame_walk fw;
            fw.init(wa, b, grid);
            if ((uint32_t)fw.f != g || fw.bi != i) bad = true;
            w[i] = bs[b];
            tot += w[i];
        }
        std::vector<uint32_t> r(G + 1);
        slice_ranges_w(w.data(), G, n_slices, r.data());
        if (r[0] != 0 || r[G] != n_slices) bad = true;
        for (uint32_t i = 0; i < G && !bad; i++) {
            if (r[i + 1] < r[i]) bad = true;
            if (std::fabs((double)(r[i + 1] - r[i]) - w[i] / tot * n_slices) > 1.0) bad = true;
        }
    }
    for (uint32_t b = 0; b < grid && !bad; b++) bad = hit[b] != 1;
    n_cfg++;
    if (bad) {
        n_bad++;
        if (n_bad < 20) printf("BAD block ranges grid %u groups %u slices %u\n", grid, groups, n_slices);
    }
}

/* the dynamic last frame: blocks draw chunks from the sixty-four counters in a random interleaving, skipping the parts they
 * know to be exhausted (bits that lag behind the truth by a random amount), until tail_state says done -- every slice of the
 * frame must come out exactly once */
static void check_tail(uint32_t n_slices, uint32_t n_blocks, unsigned seed)
{
    bool bad = false;
    std::vector<int> seen(n_slices, 0);
    uint32_t ctr[H2Y_TAIL_PARTS] = {0};
    uint64_t bits = 0;
    std::vector<tail_state> w(n_blocks);
    unsigned s = seed;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (uint32_t i = 0; i < n_blocks; i++) w[i].init(i * 37u);
    uint32_t live = n_blocks;
    long guard = 0;
    while (live && !bad) {
        const uint32_t i = rnd() % n_blocks;
        if (w[i].done()) continue;
        const uint64_t known = (rnd() & 1u) ? bits : 0; /* a stale view now and then */
        if ((known >> w[i].part) & 1u) w[i].skip();
        else {
            uint32_t first, count;
            if (w[i].take(n_slices, ctr[w[i].part]++, &first, &count)) {
                if (count < 1 || count > H2Y_TAIL_CHUNK || first + count > n_slices) bad = true;
                else for (uint32_t k = 0; k < count; k++) seen[first + k]++;
            } else {
                bits |= 1ull << w[i].part;
                w[i].skip();
            }
        }
        if (w[i].done()) live--;
        if (++guard > 64L * (long)(n_slices + 64u * n_blocks) + 1000000L) bad = true;
    }
    for (uint32_t k = 0; k < n_slices && !bad; k++) bad = seen[k] != 1;
    n_cfg++;
    if (bad) {
        n_bad++;
        if (n_bad < 20) printf("BAD tail slices %u blocks %u\n", n_slices, n_blocks);
    }
}

int main()
{
    const uint32_t grids[] = {1, 7, 8, 16, 24, 60, 64, 256};
    const uint32_t cpfs[] = {1, 2, 3, 5, 8, 31, 32, 33, 100, 1013, 4051};
    const int frames[] = {1, 2, 3, 5, 8, 16, 17, 128};
    for (uint32_t grid : grids)
        for (uint32_t groups = 1; groups <= 32; groups *= 2) {
            if (grid % groups) continue;
            const uint32_t layout = grid % (8 * groups) == 0 ? 1u : 0u;
            for (uint32_t cpf : cpfs)
                for (int nfr : frames) check(grid, groups, layout, cpf, nfr);
        }
    const double speeds[][8] = {{1, 1, 1, 1, 1, 1, 1, 1}, {1.06, 0.94, 1.06, 0.94, 1.06, 0.94, 1.06, 0.94}, {1.25, 0.75, 1, 1, 1.1, 0.9, 1.02, 0.98},
                                {1, 1, 1, 1, 0.9, 0.9, 0.9, 0.9}, {1.5, 1, 1, 1, 1, 1, 1, 1}};
    for (const auto &sp : speeds)
        for (uint32_t G : {8u, 16u, 32u, 64u, 128u, 256u})
            for (uint32_t ns : {1u, 7u, 8u, 63u, 256u, 4050u, 16200u, 64800u}) check_ranges(G, ns, sp);
    for (uint32_t ns : {1u, 7u, 8u, 63u, 64u, 65u, 1000u, 4050u, 16200u, 64800u})
        for (uint32_t nb : {1u, 8u, 16u, 128u, 256u}) check_tail(ns, nb, ns * 131u + nb);
    for (uint32_t grid : {8u, 16u, 64u, 256u, 512u, 1024u})
        for (uint32_t groups = 1; groups <= 32; groups *= 2) {
            if (grid % (8 * groups)) continue;
            for (uint32_t ns : {1u, 63u, 4050u, 16200u, 64800u}) check_block_ranges(grid, groups, ns, grid * 31u + groups * 7u + ns);
        }
    printf("%ld configurations, %ld bad\n", n_cfg, n_bad);
    return n_bad ? 1 : 0;
}
