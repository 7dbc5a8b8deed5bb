/*
 * tools/approx_check.cpp -- TEST TOOL (host build of h2y_math.h).
 * Validates the binary32 screening pass against the exact tier:
 *   - every pixel the screening pass calls "certain" must have exactly the
 *     exact tier's Y, Cb, Cr                                  (must be 0 wrong)
 *   - reports the fraction of uncertain pixels and the largest observed
 *     |screening value - reference pre-truncation value| / E      (must be < 1)
 * usage: approx_check NPIX [threads]   (runs a fixed list of configurations)
 */
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "h2y_screening.h"
using namespace h2y;

struct cfg {
    const char *name;
    int depth, full, mode; /* mode: H2Y_MODE_YCBCR (2020nc or 709) / YDZDX */
    int m709;
    int dist; /* 0 uniform k/2^24, 1 uniform float bits in table range, 2 x^4 (dark-heavy) */
};

static void make_params(const cfg &c, pix_params *pp)
{
    memset(pp, 0, sizeof *pp);
    unsigned maxCV = (1u << c.depth) - 1, D = 1u << (c.depth - 8);
    unsigned minVR = c.full ? 0 : 16 * D, maxVR = c.full ? maxCV : 235 * D, minVRC = minVR, maxVRC = c.full ? maxCV : 240 * D;
    pp->convert_transfer = 1;
    pp->norm_identity = 1;
    if (c.full) { pp->mulY = pp->mulC = (float)maxCV; }
    else { pp->mulY = (float)maxVR; pp->addY = (float)minVR; pp->mulC = (float)maxVRC; pp->addC = (float)minVRC; }
    pp->mode = c.mode;
    if (c.mode == H2Y_MODE_YCBCR) {
        if (c.m709) { pp->kr = 0.2126; pp->kg = 0.7152; pp->kb = 0.0722; pp->dcb = 1.8556; pp->dcr = 1.5748; }
        else { pp->kr = 0.2627; pp->kg = 0.6780; pp->kb = 0.0593; pp->dcb = 1.8814; pp->dcr = 1.4746; }
        pp->inv_dcb = 1.0 / pp->dcb; pp->inv_dcr = 1.0 / pp->dcr;
    }
    pp->half_m1 = (1u << (c.depth - 1)) - 1;
    pp->maxCV = maxCV;
}

static inline float pq_exact_f(float x, const pq_recA *A, const pq_recB *B)
{
    bool slow;
    float v = pq_fast(x, A, B, &slow);
    return slow ? pq_slow(x) : v;
}

int main(int argc, char **argv)
{
    long npix = argc > 1 ? atol(argv[1]) : 10000000;
    int T = argc > 2 ? atoi(argv[2]) : 8;
    std::vector<pq_recA> A(H2Y_PQ_NREC);
    std::vector<pq_recB> B(H2Y_PQ_NREC);
    std::vector<pq_rec32> T32(H2Y_PQ_NREC);
    pq_build_table(A.data(), B.data());
    pq_build_table32(T32.data());
    const cfg cfgs[] = {
        {"10b video 2020nc uniform", 10, 0, H2Y_MODE_YCBCR, 0, 0}, {"12b video 2020nc uniform", 12, 0, H2Y_MODE_YCBCR, 0, 0},
        {"12b video 709 uniform", 12, 0, H2Y_MODE_YCBCR, 1, 0},    {"12b full 2020nc uniform", 12, 1, H2Y_MODE_YCBCR, 0, 0},
        {"12b video 2020nc floatbits", 12, 0, H2Y_MODE_YCBCR, 0, 1}, {"12b video 2020nc dark", 12, 0, H2Y_MODE_YCBCR, 0, 2},
        {"14b video 2020nc uniform", 14, 0, H2Y_MODE_YCBCR, 0, 0}, {"16b video 2020nc uniform", 16, 0, H2Y_MODE_YCBCR, 0, 0},
        {"10b video YDzDx uniform", 10, 0, H2Y_MODE_YDZDX, 0, 0},  {"12b video YDzDx floatbits", 12, 0, H2Y_MODE_YDZDX, 0, 1},
        {"16b video YDzDx uniform", 16, 0, H2Y_MODE_YDZDX, 0, 0},  {"16b full YDzDx dark", 16, 1, H2Y_MODE_YDZDX, 0, 2},
    };
    int bad_total = 0;
    for (const cfg &c : cfgs) {
        pix_params pp;
        make_params(c, &pp);
        approx_params ap;
        bool worth = approx_bounds(pp, &ap);
        const double Ey = 0.5 - ap.ty, Ecb = 0.5 - ap.tcb, Ecr = 0.5 - ap.tcr;
        std::atomic<long> wrong{0}, uncertain{0};
        std::vector<double> worst(T, 0.0);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t]() {
                std::mt19937_64 rng(1234 + t * 7919 + c.depth);
                long w = 0, u = 0;
                double wr = 0;
                for (long i = t; i < npix; i += T) {
                    float x[3];
                    for (int k = 0; k < 3; k++) {
                        uint64_t r = rng();
                        if (c.dist == 0) x[k] = (float)(r >> 40) * (1.0f / 16777216.0f);
                        else if (c.dist == 1) x[k] = bits2f(0x33800000u + (uint32_t)(r % (0x40000000u - 0x33800000u)));
                        else { float v = (float)(r >> 40) * (1.0f / 16777216.0f); x[k] = v * v * v * v; }
                    }
                    uint32_t Ya, Cba, Cra, Ye, Cbe, Cre;
                    bool ok = c.mode == H2Y_MODE_YCBCR ? pix_approx<H2Y_MODE_YCBCR>(pp, ap, T32.data(), x[0], x[1], x[2], Ya, Cba, Cra)
                                                       : pix_approx<H2Y_MODE_YDZDX>(pp, ap, T32.data(), x[0], x[1], x[2], Ya, Cba, Cra);
                    /* exact tier */
                    float G = pix_scale(pq_exact_f(x[0], A.data(), B.data()), pp.mulY, pp.addY);
                    float Bv = pix_scale(pq_exact_f(x[1], A.data(), B.data()), pp.mulC, pp.addC);
                    float R = pix_scale(pq_exact_f(x[2], A.data(), B.data()), pp.mulC, pp.addC);
                    bool dummy;
                    if (c.mode == H2Y_MODE_YCBCR) pix_matrix<H2Y_MODE_YCBCR, true>(pp, G, Bv, R, Ye, Cbe, Cre, &dummy);
                    else pix_matrix<H2Y_MODE_YDZDX, true>(pp, G, Bv, R, Ye, Cbe, Cre, &dummy);
                    if (!ok) { u++; continue; }
                    if (Ya != Ye || Cba != Cbe || Cra != Cre) {
                        if (++w < 3) fprintf(stderr, "WRONG %s x=%a %a %a approx %u %u %u exact %u %u %u\n", c.name, x[0], x[1], x[2], Ya, Cba, Cra, Ye, Cbe, Cre);
                    }
                    /* observed error of the screening values vs the reference pre-truncation values */
                    if (c.mode == H2Y_MODE_YCBCR) {
                        float g = __builtin_fmaf(pq_approx32(x[0], T32.data()), pp.mulY, pp.addY);
                        float b = __builtin_fmaf(pq_approx32(x[1], T32.data()), pp.mulC, pp.addC);
                        float r = __builtin_fmaf(pq_approx32(x[2], T32.data()), pp.mulC, pp.addC);
                        float y = __builtin_fmaf(ap.kr, r, __builtin_fmaf(ap.kg, g, __builtin_fmaf(ap.kb, b, 0.5f)));
                        float cb = __builtin_fmaf(b - y, ap.inv_dcb, 0.5f), cr = __builtin_fmaf(r - y, ap.inv_dcr, 0.5f);
                        float tmpF = (float)(((pp.kr * (double)R + pp.kg * (double)G) + pp.kb * (double)Bv) + 0.5);
                        double qb = (double)(Bv - tmpF) / pp.dcb + 0.5, qr = (double)(R - tmpF) / pp.dcr + 0.5;
                        double e = fmax(fabs(y - tmpF) / Ey, fmax(fabs(cb - qb) / Ecb, fabs(cr - qr) / Ecr));
                        if (e > wr) wr = e;
                    } else {
                        float g = __builtin_fmaf(pq_approx32(x[0], T32.data()), pp.mulY, pp.addY);
                        float b = __builtin_fmaf(pq_approx32(x[1], T32.data()), pp.mulC, pp.addC);
                        float r = __builtin_fmaf(pq_approx32(x[2], T32.data()), pp.mulC, pp.addC);
                        double qb = ((double)(-G) * 0.5 + (double)Bv * 0.5) + 0.5, qr = ((double)(-G) * 0.5 + (double)R * 0.5) + 0.5;
                        double e = fmax(fabs(g - G) / Ey, fmax(fabs(__builtin_fmaf(b - g, 0.5f, 0.5f) - qb) / Ecb, fabs(__builtin_fmaf(r - g, 0.5f, 0.5f) - qr) / Ecr));
                        if (e > wr) wr = e;
                    }
                }
                wrong += w;
                uncertain += u;
                worst[t] = wr;
            });
        for (auto &x : th) x.join();
        double wr = 0;
        for (double x : worst) wr = fmax(wr, x);
        printf("%-30s E(y,cb,cr)=%.5f %.5f %.5f worth=%d  uncertain %.3f%%  wrong %ld  max observed err/E %.3f\n", c.name, Ey, Ecb, Ecr,
               (int)worth, 100.0 * uncertain.load() / npix, wrong.load(), wr);
        bad_total += wrong.load() != 0 || wr >= 1.0;
    }
    return bad_total ? 1 : 0;
}
