/*
 * tools/h2y_screening.h -- EXPERIMENT, not part of the product.
 *
 * A binary32 "screening" version of the pixel pipeline with proven error
 * bounds: it decides the output integers for ~98 % of pixels at 12 bits without
 * any binary64 work; the rest would be redone by the exact tier.  Round 1
 * measured: the screening kernel alone runs at the memory
 * floor (27 us per 4K frame), but re-doing the ~17 % of 4x2 tiles that contain
 * an uncertain pixel costs as much again when it goes through HBM (scattered
 * 16-byte reads touch 77 % of all cache lines), so the two-tier form was not
 * faster than the exact-only kernel and was removed from the library.  Kept
 * here with its validation tool (tools/approx_check.cpp) for a later round: it
 * needs an on-chip (LDS) hand-off of the uncertain pixels to pay off.
 */
#ifndef H2Y_SCREENING_H
#define H2Y_SCREENING_H
#include "../hdr2yuv_amd/csrc/h2y_math.h"
namespace h2y {
struct alignas(16) pq_rec32 {
    float c0, c1, c2, c3;
};
/* ------------------------------------------------------------------------
 * Screening pass (binary32 only).
 *
 * The exact tiers above reproduce every rounding of the reference; that costs
 * ~115 instructions per pixel, a third of them binary64.  Most pixels do not
 * need it: the output is an integer, so an APPROXIMATE pre-truncation value v^
 * with a proven bound |v^ - v_ref| <= E decides trunc(v_ref) whenever v^ is
 * farther than E from every integer.  The screening pass computes Y, Cb, Cr
 * that way in binary32 (one 16-byte table record and ~10 instructions per
 * sample) and reports "certain" or not; the caller re-does uncertain tiles
 * with the exact tier.  E is derived on the host (approx_bounds()) from the
 * rounding steps of both computations and is checked against the exact tier
 * over >10^9 random pixels by tools/approx_check.cpp.
 *
 * H2Y_PQ32_EPS: |pq_approx32(x) - PQ_reference_double(x)| / value over every
 * float of the table's domain, measured exhaustively (tools/pq_check approx):
 * 1.186e-7 (2^-23.0); the constant below adds 6 %.
 * ---------------------------------------------------------------------- */
#define H2Y_PQ32_EPS 1.26e-7f
#define H2Y_PQ_VMAX 1.2f /* PQ(x) < 1.2 for x < 2 (PQ(2) = 1.0726) */

H2Y_FN float pq_approx32(float x, const pq_rec32 *__restrict__ T)
{
    const uint32_t bits = f2bits(x);
    const uint32_t off = pq_rec_offset(bits);
    const pq_rec32 r = *reinterpret_cast<const pq_rec32 *>(reinterpret_cast<const char *>(T) + off);
    float f = bits2f((bits & ((1u << H2Y_PQ_LOW_BITS) - 1u)) | 0x3F800000u);
    float u = f - (1.0f + 1.0f / (float)(2 << H2Y_PQ_SEG_BITS));
    float p = __builtin_fmaf(r.c3, u, r.c2);
    p = __builtin_fmaf(p, u, r.c1);
    return __builtin_fmaf(p, u, r.c0);
}

struct approx_params {
    float kr, kg, kb;       /* RN32 of the luma weights */
    float inv_dcb, inv_dcr; /* RN32 of 1/divisor */
    float ty, tcb, tcr;     /* 0.5 - E: "certain" iff |fract(v) - 0.5| < t */
};

/* NaN-safe: a NaN v gives false */
H2Y_FN bool far_from_integer(float v, float t) { return __builtin_fabsf(fract_f32(v) - 0.5f) < t; }

/* Screening version of matrix_convert()'s pixel body for PQ output
 * (convert_transfer == 1).  x are the normalised linear samples.  Returns true
 * when Y, Cb, Cr are certainly what the reference computes. */
template <int MODE>
H2Y_FN bool pix_approx(const pix_params &pp, const approx_params &ap, const pq_rec32 *T, float xg, float xb, float xr,
                       uint32_t &Yo, uint32_t &Cbo, uint32_t &Cro)
{
    const float g = __builtin_fmaf(pq_approx32(xg, T), pp.mulY, pp.addY);
    const float b = __builtin_fmaf(pq_approx32(xb, T), pp.mulC, pp.addC);
    const float r = __builtin_fmaf(pq_approx32(xr, T), pp.mulC, pp.addC);
    float y, cb, cr;
    if (MODE == H2Y_MODE_YCBCR) {
        y = __builtin_fmaf(ap.kr, r, __builtin_fmaf(ap.kg, g, __builtin_fmaf(ap.kb, b, 0.5f)));
        cb = __builtin_fmaf(b - y, ap.inv_dcb, 0.5f);
        cr = __builtin_fmaf(r - y, ap.inv_dcr, 0.5f);
    } else { /* H2Y_MODE_YDZDX */
        y = g;
        cb = __builtin_fmaf(b - g, 0.5f, 0.5f);
        cr = __builtin_fmaf(r - g, 0.5f, 0.5f);
    }
    const bool ok = far_from_integer(y, ap.ty) & far_from_integer(cb, ap.tcb) & far_from_integer(cr, ap.tcr);
    Yo = f2u_clamped(y, pp.maxCV);
    Cbo = chroma_clamped(sat_i32_f32(cb), pp.half_m1, pp.maxCV);
    Cro = chroma_clamped(sat_i32_f32(cr), pp.half_m1, pp.maxCV);
    return ok;
}

/* Host: the error bounds E of the screening values against the reference's
 * pre-truncation values, from the rounding steps of both computations.
 * eps24 = 2^-24 bounds one binary32 rounding relative to the magnitude.
 *   s  (scaled code value): reference rounds V, V*mul, +add; screening has the
 *      polynomial error and one fma rounding.
 *   y  (YCbCr): reference rounds the double sum once to float (double-side
 *      errors ~2^-50 are covered by the slack); screening: three fma roundings
 *      and the binary32 weights.
 *   cb/cr: both round the float difference; screening rounds 1/c and the fma.
 * A 5 % slack plus 1e-6 absolute is added on top. */
inline bool approx_bounds(const pix_params &pp, approx_params *ap)
{
    const double e24 = 0x1p-24;
    const double sY = (double)pp.mulY * H2Y_PQ_VMAX + pp.addY, sC = (double)pp.mulC * H2Y_PQ_VMAX + pp.addC;
    const double EsY = sY * (e24 * 1.001 + H2Y_PQ32_EPS + 3 * e24), EsC = sC * (e24 * 1.001 + H2Y_PQ32_EPS + 3 * e24);
    double Ey, Ecb, Ecr;
    if (pp.mode == H2Y_MODE_YCBCR) {
        const double ymax = (sY > sC ? sY : sC) + 0.5;
        Ey = pp.kr * EsC + pp.kg * EsY + pp.kb * EsC + 5 * e24 * ymax;
        Ecb = (EsC + Ey + 4 * e24 * ymax) / pp.dcb + e24 * (ymax / pp.dcb + 1.0);
        Ecr = (EsC + Ey + 4 * e24 * ymax) / pp.dcr + e24 * (ymax / pp.dcr + 1.0);
        ap->kr = (float)pp.kr; ap->kg = (float)pp.kg; ap->kb = (float)pp.kb;
        ap->inv_dcb = (float)pp.inv_dcb; ap->inv_dcr = (float)pp.inv_dcr;
    } else if (pp.mode == H2Y_MODE_YDZDX) {
        const double dmax = (sY > sC ? sY : sC);
        Ey = EsY;
        Ecb = Ecr = (EsC + EsY) * 0.5 + 2 * e24 * dmax;
        ap->kr = ap->kg = ap->kb = 0.f; ap->inv_dcb = ap->inv_dcr = 0.5f;
    } else return false;
    Ey = Ey * 1.05 + 1e-6; Ecb = Ecb * 1.05 + 1e-6; Ecr = Ecr * 1.05 + 1e-6;
    ap->ty = (float)(0.5 - Ey); ap->tcb = (float)(0.5 - Ecb); ap->tcr = (float)(0.5 - Ecr);
    /* worth screening only while most pixels pass */
    return Ey < 0.05 && Ecb < 0.05 && Ecr < 0.05;
}


/* degree-3 binary32 records for the screening pass, same segments as the exact table */
inline void pq_build_table32(pq_rec32 *T32)
{
    for (int i = 0; i < H2Y_PQ_NSEG; i++) {
        int e = H2Y_PQ_EMIN + i / H2Y_PQ_SEG_PER_BINADE;
        int s = i % H2Y_PQ_SEG_PER_BINADE;
        double scale = bits2d((uint64_t)(1023 + e) << 52);
        double mid = scale * (1.0 + (s + 0.5) / H2Y_PQ_SEG_PER_BINADE);
        double half = scale * (0.5 / H2Y_PQ_SEG_PER_BINADE);
        const double r = (double)(2 << H2Y_PQ_SEG_BITS);
        if (T32) {
            /* degree-3 fit through four near-Chebyshev nodes of the same segment */
            const double wn[4] = {-0.9238795325112867, -0.3826834323650898, 0.3826834323650898, 0.9238795325112867};
            dd ev[4];
            for (int j = 0; j < 4; j++) ev[j] = pq_exact_dd(dd_add_d(two_prod(wn[j], half), mid));
            for (int lvl = 1; lvl < 4; lvl++)
                for (int j = 3; j >= lvl; j--) {
                    dd num = dd_add(ev[j], dd{-ev[j - 1].hi, -ev[j - 1].lo});
                    ev[j] = dd_div(num, dd{wn[j] - wn[j - lvl], 0.0});
                }
            dd q[4] = {ev[3], {0, 0}, {0, 0}, {0, 0}};
            int dg = 0;
            for (int j = 2; j >= 0; j--) {
                dd nq[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
                for (int k = 0; k <= dg; k++) {
                    nq[k + 1] = dd_add(nq[k + 1], q[k]);
                    nq[k] = dd_add(nq[k], dd_mul_d(q[k], -wn[j]));
                }
                nq[0] = dd_add(nq[0], ev[j]);
                dg++;
                for (int k = 0; k <= dg; k++) q[k] = nq[k];
            }
            T32[i].c0 = (float)q[0].hi;
            T32[i].c1 = (float)(q[1].hi * r);
            T32[i].c2 = (float)(q[2].hi * r * r);
            T32[i].c3 = (float)(q[3].hi * r * r * r);
        }
    }
    T32[H2Y_PQ_NSEG].c0 = bits2f(0x7FC00000u); /* out-of-table inputs: NaN => "not certain" */
    T32[H2Y_PQ_NSEG].c1 = T32[H2Y_PQ_NSEG].c2 = T32[H2Y_PQ_NSEG].c3 = 0.0f;
}
} // namespace h2y
#endif
