/*
 * tools/t1_check.cpp -- TEST TOOL (host build of h2y_math.h).
 * First-tier pixel pipeline (pq_t1 + pix_matrix_t1) against the exact tier on
 * random pixels: a pixel the first tier ACCEPTS must have exactly the exact
 * tier's Y, Cb, Cr.  Also reports how many pixels it sends on to the second tier.
 * usage: t1_check NPIX [threads]
 */
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>
#include "../hdr2yuv_amd/csrc/h2y_math.h"
using namespace h2y;

struct cfg { const char *name; int depth, full, mode, m709, dist; };

static void make_params(const cfg &c, pix_params *pp)
{
    memset(pp, 0, sizeof *pp);
    unsigned maxCV = (1u << c.depth) - 1, D = 1u << (c.depth - 8);
    unsigned minVR = c.full ? 0 : 16 * D, maxVR = c.full ? maxCV : 235 * D, maxVRC = c.full ? maxCV : 240 * D;
    pp->convert_transfer = 1; pp->norm_identity = 1; pp->src_tf = H2Y_TF_LINEAR; pp->dst_tf = H2Y_TF_PQ;
    if (c.full) pp->mulY = pp->mulC = (float)maxCV;
    else { pp->mulY = (float)maxVR; pp->addY = (float)minVR; pp->mulC = (float)maxVRC; pp->addC = (float)minVR; }
    pp->mode = c.mode;
    if (c.mode == H2Y_MODE_YCBCR) {
        if (c.m709) { pp->kr = 0.2126; pp->kg = 0.7152; pp->kb = 0.0722; pp->dcb = 1.8556; pp->dcr = 1.5748; }
        else { pp->kr = 0.2627; pp->kg = 0.6780; pp->kb = 0.0593; pp->dcb = 1.8814; pp->dcr = 1.4746; }
        pp->inv_dcb = 1.0 / pp->dcb; pp->inv_dcr = 1.0 / pp->dcr;
    }
    pp->half_m1 = (1u << (c.depth - 1)) - 1; pp->maxCV = maxCV;
}
static inline float pq_exact_f(float x, const pq_recA *A, const pq_recB *B)
{
    bool slow; float v = pq_fast(x, A, B, &slow); return slow ? pq_slow(x) : v;
}
int main(int argc, char **argv)
{
    long npix = argc > 1 ? atol(argv[1]) : 10000000;
    int T = argc > 2 ? atoi(argv[2]) : 8;
    std::vector<pq_recA> A(H2Y_PQ_NREC); std::vector<pq_recB> B(H2Y_PQ_NREC); std::vector<pq_rec1> T1(H2Y_T1_NREC);
    pq_build_table(A.data(), B.data()); pq_build_table1(T1.data());
    const cfg cfgs[] = {
        {"10b video 2020nc uniform", 10, 0, H2Y_MODE_YCBCR, 0, 0}, {"12b video 2020nc uniform", 12, 0, H2Y_MODE_YCBCR, 0, 0},
        {"12b video 709 floatbits", 12, 0, H2Y_MODE_YCBCR, 1, 1},  {"12b full 2020nc dark", 12, 1, H2Y_MODE_YCBCR, 0, 2},
        {"14b video 2020nc uniform", 14, 0, H2Y_MODE_YCBCR, 0, 0}, {"16b video 2020nc uniform", 16, 0, H2Y_MODE_YCBCR, 0, 0},
        {"10b video YDzDx uniform", 10, 0, H2Y_MODE_YDZDX, 0, 0},  {"12b video YDzDx floatbits", 12, 0, H2Y_MODE_YDZDX, 0, 1},
        {"16b video YDzDx uniform", 16, 0, H2Y_MODE_YDZDX, 0, 0},
    };
    int bad = 0;
    for (const cfg &c : cfgs) {
        pix_params pp; make_params(c, &pp);
        t1_sens sn; bool worth = t1_bounds(pp, &sn);
        std::atomic<long> wrong{0}, redo{0}, vu{0}, differ_v{0};
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&, t]() {
            std::mt19937_64 rng(99 + t * 104729 + c.depth); long w = 0, r = 0, v = 0, dv = 0;
            for (long i = t; i < npix; i += T) {
                float x[3];
                for (int k = 0; k < 3; k++) {
                    uint64_t q = rng();
                    if (c.dist == 0) x[k] = (float)(q >> 40) * (1.0f / 16777216.0f);
                    else if (c.dist == 1) x[k] = bits2f(0x33800000u + (uint32_t)(q % (0x40000000u - 0x33800000u)));
                    else { float u = (float)(q >> 40) * (1.0f / 16777216.0f); x[k] = u * u * u * u; }
                }
                bool ug, ub, ur;
                float vg = pq_t1(x[0], T1.data(), &ug), vb = pq_t1(x[1], T1.data(), &ub), vr = pq_t1(x[2], T1.data(), &ur);
                float eg = pq_exact_f(x[0], A.data(), B.data()), eb = pq_exact_f(x[1], A.data(), B.data()), er = pq_exact_f(x[2], A.data(), B.data());
                const bool vunc = ug | ub | ur;
                if (vunc) v++;
                if (f2bits(vg) != f2bits(eg) || f2bits(vb) != f2bits(eb) || f2bits(vr) != f2bits(er)) dv++;
                uint32_t Ya, Cba, Cra, Ye, Cbe, Cre; bool dummy;
                bool need2 = c.mode == H2Y_MODE_YCBCR
                    ? pix_matrix_t1<H2Y_MODE_YCBCR>(pp, sn, pix_scale(vg, pp.mulY, pp.addY), pix_scale(vb, pp.mulC, pp.addC), pix_scale(vr, pp.mulC, pp.addC), vunc, Ya, Cba, Cra)
                    : pix_matrix_t1<H2Y_MODE_YDZDX>(pp, sn, pix_scale(vg, pp.mulY, pp.addY), pix_scale(vb, pp.mulC, pp.addC), pix_scale(vr, pp.mulC, pp.addC), vunc, Ya, Cba, Cra);
                if (need2) { r++; continue; }
                Ya = Ya < pp.maxCV ? Ya : pp.maxCV; /* the T1 form leaves the maxCV clamp of Y to pix_yuv_clamp() */
                Cba += pp.half_m1; Cra += pp.half_m1; /* ... and gives the chroma raw: offset by the kernels, NO clamp (t1_chroma_in_range() says none is needed: checked here) */
                float G = pix_scale(eg, pp.mulY, pp.addY), Bv = pix_scale(eb, pp.mulC, pp.addC), R = pix_scale(er, pp.mulC, pp.addC);
                if (c.mode == H2Y_MODE_YCBCR) pix_matrix<H2Y_MODE_YCBCR, true>(pp, G, Bv, R, Ye, Cbe, Cre, &dummy);
                else pix_matrix<H2Y_MODE_YDZDX, true>(pp, G, Bv, R, Ye, Cbe, Cre, &dummy);
                if (Ya != Ye || Cba != Cbe || Cra != Cre) { if (w++ < 3) fprintf(stderr, "WRONG %s x=%a %a %a t1 %u %u %u exact %u %u %u\n", c.name, x[0], x[1], x[2], Ya, Cba, Cra, Ye, Cbe, Cre); }
            }
            wrong += w; redo += r; vu += v; differ_v += dv; });
        for (auto &x : th) x.join();
        printf("%-28s worth=%d  unsure-sample pixels %.2f%%  (V really differs %.3f%%)  second tier %.4f%%  wrong %ld\n", c.name, (int)worth,
               100.0 * vu / npix, 100.0 * differ_v / npix, 100.0 * redo / npix, wrong.load());
        bad += wrong.load() != 0;
    }
    return bad ? 1 : 0;
}
