// tools/walk_check.cpp -- CPU check of frame_walk (hdr2yuv_amd/csrc/h2y_walk.h), the code by which the
// loop-form kernels deal frames and chunks to blocks: frame groups, XCD-aware layout, weighted rounds.
// For every configuration: each chunk of each frame is visited by exactly one block, exactly once;
// a block's prefetch target across a frame boundary (succ() kind 2) is the chunk it starts the next
// frame with (first()); a frame is only touched by blocks of its group.
// Build: g++ -O2 -std=c++17 -I hdr2yuv_amd/csrc tools/walk_check.cpp -o /tmp/walk_check
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "h2y_walk.h"
using namespace h2y;

static long n_cfg = 0, n_bad = 0;

static void check(uint32_t grid, uint32_t groups, uint32_t layout, uint32_t mask, uint32_t cpf, uint32_t ca, int nfr)
{
    walk_args a;
    a.groups = groups; a.xcd_layout = layout; a.fast_mask = mask; a.chunks_a = ca; a.chunks_per_frame = cpf; a.n_frames = nfr;
    std::vector<std::vector<int>> seen(nfr, std::vector<int>(cpf, 0));
    std::vector<std::vector<int>> group_of(nfr, std::vector<int>(cpf, -1));
    std::vector<long> work(grid, 0);
    bool bad = false;
    for (uint32_t b = 0; b < grid && !bad; b++) {
        frame_walk fw;
        bool expect = false; /* succ() announced this block's first chunk of the next frame */
        uint32_t ek = 0;
        bool eB = false;
        bool expect_none = false;
        int grp = -1;
        for (fw.init(a, b, grid); fw.f < nfr; fw.advance()) {
            if (grp < 0) grp = fw.f;
            if ((fw.f - grp) % (int)groups != 0) bad = true;
            uint32_t k;
            bool inB;
            bool more = fw.first(k, inB);
            if (expect && (!more || k != ek || inB != eB)) { bad = true; break; }
            if (expect_none && more) { bad = true; break; }
            expect = expect_none = false;
            long guard = 0;
            while (more) {
                if (k >= cpf) { bad = true; break; }
                seen[fw.f][k]++;
                group_of[fw.f][k] = grp;
                work[b]++;
                uint32_t k2;
                bool inB2;
                const int kind = fw.succ(k, inB, k2, inB2);
                if (kind == 2) { expect = true; ek = k2; eB = inB2; if (!fw.has_next()) bad = true; }
                if (kind == 0) { if (fw.has_next()) expect_none = true; if (k2 != k) bad = true; }
                more = kind == 1;
                k = k2;
                inB = inB2;
                if (++guard > (long)cpf + 4) { bad = true; break; }
            }
            if (bad) break;
        }
    }
    for (int f = 0; f < nfr && !bad; f++)
        for (uint32_t c = 0; c < cpf; c++)
            if (seen[f][c] != 1 || group_of[f][c] != f % (int)groups) { bad = true; break; }
    /* the closed form the kernels' ticket dealing uses (wave_deal in h2y_kernels.hip): a block's chunks of a frame
     * are kA + j G (j < count_a) and kB + j Gf (j < count_b); and kA_n / kB_n are the next frame's kA / kB */
    std::vector<std::vector<int>> seen2(nfr, std::vector<int>(cpf, 0));
    for (uint32_t b = 0; b < grid && !bad; b++) {
        frame_walk fw;
        bool have_n = false;
        uint32_t ka_n = 0, kb_n = 0;
        for (fw.init(a, b, grid); fw.f < nfr; fw.advance()) {
            if (have_n && (fw.kA != ka_n || fw.kB != kb_n)) { bad = true; break; }
            have_n = true; ka_n = fw.kA_n; kb_n = fw.kB_n;
            const uint32_t na = fw.count_a(fw.kA), nb = fw.count_b(fw.kB);
            for (uint32_t j = 0; j < na; j++) { const uint32_t k = fw.kA + j * fw.G; if (k >= cpf) { bad = true; break; } seen2[fw.f][k]++; }
            for (uint32_t j = 0; j < nb; j++) { const uint32_t k = fw.kB + j * fw.Gf; if (k >= cpf) { bad = true; break; } seen2[fw.f][k]++; }
        }
    }
    for (int f = 0; f < nfr && !bad; f++)
        for (uint32_t c = 0; c < cpf; c++)
            if (seen2[f][c] != 1) { bad = true; break; }
    n_cfg++;
    if (bad) {
        n_bad++;
        if (n_bad < 20) printf("BAD grid %u groups %u layout %u mask %#x cpf %u chunks_a %u frames %d\n", grid, groups, layout, mask, cpf, ca, nfr);
    }
}

int main()
{
    const uint32_t grids[] = {1, 7, 8, 16, 24, 60, 64, 256};
    const uint32_t cpfs[] = {1, 2, 3, 5, 8, 31, 32, 33, 100, 1013};
    const uint32_t masks[] = {0xFFu, 0x55u, 0xAAu, 0x01u, 0xFEu, 0x0Fu, 0x81u};
    const int frames[] = {1, 2, 3, 5, 8, 16, 17};
    for (uint32_t grid : grids)
        for (uint32_t groups = 1; groups <= 32; groups *= 2) {
            if (grid % groups) continue;
            const uint32_t layout = grid % (8 * groups) == 0 ? 1u : 0u;
            for (uint32_t cpf : cpfs)
                for (int nfr : frames)
                    for (uint32_t mask : masks) {
                        if (!layout && mask != 0xFFu) continue; /* the host weights only with the XCD layout */
                        const uint32_t cas[] = {cpf, cpf > 1 ? cpf - 1 : 1, cpf / 2 ? cpf / 2 : 1, 1u, (uint32_t)(cpf * 0.966) ? (uint32_t)(cpf * 0.966) : 1u};
                        for (uint32_t ca : cas) {
                            if (mask == 0xFFu && ca != cpf) continue; /* one part */
                            check(grid, groups, layout, mask, cpf, ca, nfr);
                        }
                    }
        }
    /* the share a fast block gets: 4K frames, the measured 7 % */
    {
        walk_args a;
        a.groups = 8; a.xcd_layout = 1; a.fast_mask = 0x55u; a.chunks_per_frame = 1013; a.chunks_a = 979; a.n_frames = 64;
        double wf = 0, ws = 0;
        for (uint32_t b = 0; b < 256; b++) {
            frame_walk fw;
            long w = 0;
            for (fw.init(a, b, 256); fw.f < a.n_frames; fw.advance()) {
                uint32_t k, k2;
                bool inB, inB2;
                bool more = fw.first(k, inB);
                while (more) { w++; more = fw.succ(k, inB, k2, inB2) == 1; k = k2; inB = inB2; }
            }
            ((b & 1) ? ws : wf) += w;
        }
        printf("share fast/slow %.4f (dealt: 1 + (1013 - 979) / 0.5 / 979 = %.4f)\n", wf / ws, 1.0 + 34.0 / 0.5 / 979.0);
        if (wf / ws < 1.06 || wf / ws > 1.08) n_bad++;
    }
    printf("%ld configurations, %ld bad\n", n_cfg, n_bad);
    return n_bad ? 1 : 0;
}
