/*
 * h2y_oracle.c -- TEST INFRASTRUCTURE ONLY (see h2y_oracle.h).
 *
 * CPU restatement of the hdr2yuv convert path.  Every function cites the
 * reference lines whose behaviour it restates.  Types and operation order are
 * part of the behaviour (float vs double, truncating casts, unsigned
 * compares); comments call out each place where that matters.
 */
#include "h2y_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- clip limits: set_pic_clip(), common.cpp:300-327 ------------------- */
void h2y_oracle_set_clip(int bit_depth, int full_range, h2y_oracle_clip *clip)
{
    clip->minCV = 0;
    clip->maxCV = (uint64_t)((1 << bit_depth) - 1);
    clip->Half = (uint16_t)(1 << (bit_depth - 1));
    if (full_range == 0) {
        /* video range: note maxVR = 219*D + 16*D = 235*D, maxVRC = 240*D */
        uint16_t D = (uint16_t)(1 << (bit_depth - 8));
        clip->minVR = (uint16_t)(16 * D);
        clip->maxVR = (uint16_t)(219 * D + clip->minVR);
        clip->minVRC = clip->minVR;
        clip->maxVRC = (uint16_t)(224 * D + clip->minVRC);
    } else {
        clip->minVR = 0;
        clip->maxVR = (uint16_t)clip->maxCV;
        clip->minVRC = 0;
        clip->maxVRC = (uint16_t)clip->maxCV;
    }
}

/* ---- PQ10000_r(), convert.cpp:56-63 ------------------------------------ */
float h2y_oracle_pq10000_r(float L)
{
    /* Both pow() calls are double precision (float argument promoted, double
     * literal exponents); the quotient is double; only the final assignment
     * rounds to float.  The first exponent is the literal 0.1593017578. */
    double Ln = pow((double)L, 0.1593017578);
    double num = 0.8359375 + 18.8515625 * Ln;
    double den = 1 + 18.6875 * Ln;
    float V = (float)pow(num / den, 78.84375);
    return V;
}

/* ---- the other transfer functions at the same dispatch point, convert.cpp:12-87.
 * The reference is C++: <math.h> there supplies float overloads, so pow(float,float)
 * is powf and log(float) is logf, while any double argument makes the call double. */
static float o_rho_gamma_f(float V) /* convert.cpp:12-27 */
{
    const float rho = 25.0f, gamma = 2.4f;
    float P = powf(rho, V);
    return (float)pow(((double)P - 1.0) / ((double)rho - 1.0), (double)gamma);
}
static float o_rho_gamma_r(float L) /* convert.cpp:30-38 */
{
    const float rho = 25.0f, gamma = 2.4f;
    return (float)(log(1.0 + ((double)rho - 1.0) * pow((double)L, 1.0 / (double)gamma)) / (double)logf(rho));
}
static float o_pq10000_f(float V) /* convert.cpp:43-51 */
{
    double p = pow((double)V, 1.0 / 78.84375);
    return (float)pow(fmax(p - 0.8359375, 0.0) / (18.8515625 - 18.6875 * p), 1.0 / 0.1593017578);
}
/* bt1886_f / bt1886_r, convert.cpp:67-87, as called with gamma 2.4f, Lw 1, Lb 0
 * (convert.cpp:1052-1058, :1094-1100): a folds to 1.0f and b to 0.0f */
static float o_bt1886_f(float V) { return (float)(1.0 * pow(fmax((double)(V + 0.0f), 0.), (double)2.4f)); }
static float o_bt1886_r(float L) { return (float)(pow(fmax((double)(L / 1.0f), 0.), 1. / (double)2.4f) - 0.0); }

static int tf_class(int t) /* 0 linear, 1 PQ, 2 rho-gamma, 3 BT.1886 family, -1 not handled */
{
    switch (t) {
    case 8: return 0;
    case 16: return 1;
    case 18: return 2;
    case 1: case 6: case 14: case 15: return 3; /* BT709, BT601, BT2020_10bit, BT2020_12bit: convert.cpp:1047-1050 */
    default: return -1;
    }
}
/* convert.cpp:1024-1109: source transfer -> linear -> destination transfer, one sample */
static float o_transfer_chain(int src, int dst, float x)
{
    switch (tf_class(src)) {
    case 1: x = o_pq10000_f(x); break;
    case 2: x = o_rho_gamma_f(x); break;
    case 3: x = o_bt1886_f(x); break;
    default: break;
    }
    switch (tf_class(dst)) {
    case 1: x = h2y_oracle_pq10000_r(x); break;
    case 2: x = o_rho_gamma_r(x); break;
    case 3: x = o_bt1886_r(x); break;
    default: break;
    }
    return x;
}
float h2y_oracle_transfer_chain(int src, int dst, float x) { return o_transfer_chain(src, dst, x); }

/* ---- pic_stats(), common.cpp:66-168 ------------------------------------ */
void h2y_oracle_stats_f32(const float *const planes[3], size_t n, float mm[6],
                          int32_t floor_[3], int32_t ceil_[3])
{
    for (int c = 0; c < 3; c++) {
        /* common.cpp:118-119: max starts at the smallest POSITIVE normal */
        float lo = FLT_MAX;
        float hi = FLT_MIN;
        const float *p = planes[c];
        for (size_t i = 0; i < n; i++) {
            float s = p[i];
            lo = s < lo ? s : lo;
            hi = s > hi ? s : hi;
        }
        mm[2 * c] = lo;
        mm[2 * c + 1] = hi;
        /* common.cpp:135-136: C truncation toward zero */
        floor_[c] = (int32_t)lo;
        ceil_[c] = (int32_t)hi;
    }
}

void h2y_oracle_stats_u16(const uint16_t *const planes[3], size_t n, int bit_depth,
                          uint16_t mm[6], int32_t floor_[3], int32_t ceil_[3])
{
    for (int c = 0; c < 3; c++) {
        uint16_t lo = 0xFFFF, hi = 0;
        const uint16_t *p = planes[c];
        for (size_t i = 0; i < n; i++) {
            uint16_t s = p[i];
            lo = s < lo ? s : lo;
            hi = s > hi ? s : hi;
        }
        mm[2 * c] = lo;
        mm[2 * c + 1] = hi;
        /* common.cpp:94-106: snap the ceiling up to nominal luma/chroma peak */
        int D = 1 << (bit_depth - 8);
        int SMin = D * 16;
        int YMax = 219 * D + SMin;
        int CMax = 224 * D + SMin;
        int fl = lo, ce = hi;
        if (ce < YMax && ce > (YMax * 3) / 4) ce = YMax;
        if (ce < CMax && ce > (CMax * 3) / 4) ce = CMax;
        floor_[c] = fl;
        ceil_[c] = ce;
    }
}

/* float -> unsigned int the way the reference's x86-64 build does it
 * (cvttss2si to 64 bits, low 32 bits kept).  In the pinned domain (finite,
 * 0 <= f < 2^32) this is plain truncation; outside it C leaves the cast
 * undefined and this documents what the reference binary does. */
static inline uint32_t f2u(float f)
{
    if (!(f > -9.2e18f && f < 9.2e18f)) return 0u; /* NaN/huge: 0x8000000000000000 -> low word 0 */
    return (uint32_t)(int64_t)f;
}

/* double -> int, truncation toward zero (cvttsd2si, 32 bit) */
static inline int32_t d2i(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
    return (int32_t)v;
}

/* ---- matrix_convert(), convert.cpp:879-1221 ---------------------------- */
int h2y_oracle_matrix_convert(const h2y_desc *d, const void *const in_planes[3],
                              const int32_t floor_[3], const int32_t ceil_[3],
                              int tmp_bit_depth, uint16_t *const out444[3])
{
    h2y_oracle_clip clip;
    h2y_oracle_set_clip(tmp_bit_depth, d->dst_full_range, &clip);

    const int is_u16 = d->in_sample_type == H2Y_SAMPLE_U16;
    const int convert_transfer = d->src_transfer != d->dst_transfer; /* :930 */

    /* :913-925 */
    float P = 0, Q = 0, RR = 0, S = 0;
    if (d->dst_matrix == H2Y_MATRIX_YDZDX_Y100) {
        P = -0.5f; Q = 0.491722f; RR = 0.5f; S = -0.49495f;
    } else if (d->dst_matrix == H2Y_MATRIX_YDZDX_Y500) {
        P = -0.5f; Q = 0.493393f; RR = 0.5f; S = -0.49602f;
    }

    /* :939-940 ints converted to float */
    float range[3] = {1, 1, 1}, offset[3] = {0, 0, 0};
    if (convert_transfer) {
        for (int c = 0; c < 3; c++) {
            range[c] = (float)(ceil_[c] - floor_[c]);
            offset[c] = (float)floor_[c];
        }
    }

    if (convert_transfer) {
        /* transfers the reference has code for (convert.cpp:1024-1109); for the rest it only prints a warning per pixel */
        if (tf_class(d->src_transfer) < 0 || tf_class(d->dst_transfer) < 0) return H2Y_EUNSUPPORTED;
    }

    const int identity = d->dst_matrix == d->src_matrix && d->dst_primaries == d->src_primaries; /* :1159 */
    switch (d->dst_matrix) {
    case H2Y_MATRIX_YDZDX: case H2Y_MATRIX_BT2020NC: case H2Y_MATRIX_BT709:
    case H2Y_MATRIX_YDZDX_Y100: case H2Y_MATRIX_YDZDX_Y500:
        break;
    default:
        if (!identity) return H2Y_EUNSUPPORTED; /* reference exit(0)s, :1196 */
    }

    /* unsigned short members promoted to int, then to float in the products */
    const float maxCVf = (float)clip.maxCV;
    const float maxVR = (float)(int)clip.maxVR, minVR = (float)(int)clip.minVR;
    const float maxVRC = (float)(int)clip.maxVRC, minVRC = (float)(int)clip.minVRC;

    const size_t n = (size_t)d->width * (size_t)d->height;
    for (size_t i = 0; i < n; i++) {
        float G, B, R;
        if (is_u16) {
            G = (float)((const uint16_t *)in_planes[0])[i];
            B = (float)((const uint16_t *)in_planes[1])[i];
            R = (float)((const uint16_t *)in_planes[2])[i];
        } else {
            G = ((const float *)in_planes[0])[i];
            B = ((const float *)in_planes[1])[i];
            R = ((const float *)in_planes[2])[i];
        }

        if (convert_transfer) {
            /* :1017-1019 float subtract, float divide */
            G = (G - offset[0]) / range[0];
            B = (B - offset[1]) / range[1];
            R = (R - offset[2]) / range[2];
            /* :1024-1109 (LINEAR -> PQ is :1072-1079) */
            G = o_transfer_chain(d->src_transfer, d->dst_transfer, G);
            B = o_transfer_chain(d->src_transfer, d->dst_transfer, B);
            R = o_transfer_chain(d->src_transfer, d->dst_transfer, R);
            /* :1123-1145 separate float multiply and float add */
            if (d->dst_full_range) {
                G = G * maxCVf; B = B * maxCVf; R = R * maxCVf;
            } else if (d->dst_matrix == H2Y_MATRIX_GBR) {
                G = G * maxVR + minVR; B = B * maxVR + minVR; R = R * maxVR + minVR;
            } else {
                G = G * maxVR + minVR; B = B * maxVRC + minVRC; R = R * maxVRC + minVRC;
            }
        }

        uint32_t Y;
        int64_t Cb, Cr;
        if (identity) {
            Y = f2u(G); Cb = (int64_t)f2u(B); Cr = (int64_t)f2u(R); /* :1163-1165 */
        } else {
            float tmpF;
            if (d->dst_matrix == H2Y_MATRIX_YDZDX) { /* :1169-1173 */
                Y = f2u(G);
                Cb = d2i(-G / 2.0 + B / 2.0 + 0.5);
                Cr = d2i(-G / 2.0 + R / 2.0 + 0.5);
            } else if (d->dst_matrix == H2Y_MATRIX_BT2020NC) { /* :1176-1180 */
                /* double products and sums, rounded to float once; the 0.5
                 * stays inside tmpF when the colour differences are formed */
                tmpF = (float)((0.2627 * R + 0.6780 * G + 0.0593 * B) + 0.5);
                Y = f2u(tmpF);
                Cb = d2i((B - tmpF) / 1.8814 + 0.5);
                Cr = d2i((R - tmpF) / 1.4746 + 0.5);
            } else if (d->dst_matrix == H2Y_MATRIX_BT709) { /* :1181-1185 */
                tmpF = (float)((0.2126 * R + 0.7152 * G + 0.0722 * B) + 0.5);
                Y = f2u(tmpF);
                Cb = d2i((B - tmpF) / 1.8556 + 0.5);
                Cr = d2i((R - tmpF) / 1.5748 + 0.5);
            } else { /* Y100 / Y500, :1186-1190: float products and float sum, double +0.5 */
                Y = f2u(G);
                Cb = d2i((double)(P * G + Q * B) + 0.5);
                Cr = d2i((double)(RR * R + S * G) + 0.5);
            }
            Cb = Cb + clip.Half - 1; /* :1200-1201 */
            Cr = Cr + clip.Half - 1;
        }
        /* :1207-1213 all compares are unsigned 64-bit (long vs unsigned long):
         * a negative Cb/Cr wraps high and clamps to maxCV; "< minCV(0)" never fires */
        uint64_t y64 = Y, cb64 = (uint64_t)Cb, cr64 = (uint64_t)Cr;
        if (y64 > clip.maxCV) y64 = clip.maxCV;
        if (y64 < clip.minCV) y64 = clip.minCV;
        if (cb64 > clip.maxCV) cb64 = clip.maxCV;
        if (cb64 < clip.minCV) cb64 = clip.minCV;
        if (cr64 > clip.maxCV) cr64 = clip.maxCV;
        if (cr64 < clip.minCV) cr64 = clip.minCV;
        out444[0][i] = (uint16_t)y64;
        out444[1][i] = (uint16_t)cb64;
        out444[2][i] = (uint16_t)cr64;
    }
    return 0;
}

/* ---- Subsample444to420_box(), convert.cpp:91-172 ----------------------- */
void h2y_oracle_sub420_box(uint16_t *dst, const uint16_t *src, int width, int height)
{
    /* the reference walks 4x4 tiles (so needs width,height % 4 == 0) but each
     * output is simply the truncating mean of its own 2x2 block, :157-160 */
    const int wc = width / 2;
    for (int y = 0; y < height; y += 2)
        for (int x = 0; x < width; x += 2) {
            uint64_t s = (uint64_t)src[(size_t)y * width + x] + src[(size_t)y * width + x + 1] +
                         src[(size_t)(y + 1) * width + x] + src[(size_t)(y + 1) * width + x + 1];
            dst[(size_t)(y / 2) * wc + x / 2] = (uint16_t)(s / 4);
        }
}

/* ---- Subsample444to420_FIR(), convert.cpp:261-383 ---------------------- */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void h2y_oracle_sub420_fir(uint16_t *dst, const uint16_t *src, int width, int height,
                           uint64_t minCV, uint64_t maxCV)
{
    const int w2 = width >> 1;
    uint16_t *mid = (uint16_t *)malloc((size_t)height * w2 * sizeof(uint16_t));
    const float scale = 512.0f;
    const float c21 = 21.0f / scale, c52 = 52.0f / scale, c159 = 159.0f / scale, c256 = 256.0f / scale;
    const float fmaxCV = (float)maxCV, fminCV = (float)minCV;

    /* stage 1, :291-320: 7-tap horizontal, even columns, edges replicate */
    for (int j = 0; j < height; j++) {
        const uint16_t *s = src + (size_t)j * width;
        for (int i = 0; i < width; i += 2) {
            int im5 = clampi(i - 5, 0, width - 1), im3 = clampi(i - 3, 0, width - 1);
            int im1 = clampi(i - 1, 0, width - 1), ip1 = clampi(i + 1, 0, width - 1);
            int ip3 = clampi(i + 3, 0, width - 1), ip5 = clampi(i + 5, 0, width - 1);
            /* float products and sums left to right; the trailing +0.5 is a
             * double add whose result is rounded back to float */
            float acc = c21 * ((float)s[im5] + (float)s[ip5]) - c52 * ((float)s[im3] + (float)s[ip3]) +
                        c159 * ((float)s[im1] + (float)s[ip1]) + c256 * (float)s[i];
            float t = (float)((double)acc + 0.5);
            if (t > fmaxCV) t = fmaxCV;
            if (t < fminCV) t = fminCV;
            mid[(size_t)j * w2 + (i >> 1)] = (uint16_t)t;
        }
    }

    /* stage 2, :323-376: 12-tap vertical at half-sample phase, even rows */
    const float c228 = 228.0f / scale, c70 = 70.0f / scale, c37 = 37.0f / scale;
    const float c11 = 11.0f / scale, c5 = 5.0f / scale;
    for (int i = 0; i < w2; i++) {
        for (int j = 0; j < height; j += 2) {
            const int h1 = height - 1;
#define M(r) ((float)mid[(size_t)clampi((r), 0, h1) * w2 + i])
            float acc = c228 * (M(j) + M(j + 1)) + c70 * (M(j - 1) + M(j + 2)) -
                        c37 * (M(j - 2) + M(j + 3)) - c21 * (M(j - 3) + M(j + 4)) +
                        c11 * (M(j - 4) + M(j + 5)) + c5 * (M(j - 5) + M(j + 6));
#undef M
            float t = (float)((double)acc + 0.5);
            if (t > fmaxCV) t = fmaxCV;
            if (t < fminCV) t = fminCV;
            dst[(size_t)(j >> 1) * w2 + i] = (uint16_t)t;
        }
    }
    free(mid);
}

/* ---- Subsample420to444(), convert.cpp:1869-1986 (also yuv2tiff.cpp:575-692) ----
 * src: (width/2) x (height/2) plane, dst: width x height plane, both row-major here (the reference
 * indexes arrays of column pointers [col][row]; the arithmetic does not depend on that).
 * algorithm 0: sample replication (:1871-1881).  Otherwise (:1882-1983): vertical 6-tap pair
 * (3 -16 67 227 -32 7)/256 per output-row parity into a (width/2) x height intermediate that is
 * clamped and truncated to unsigned short, then horizontally even samples copied and odd samples
 * from (21 -52 159 159 -52 21)/256, edges replicated by index clamping. */
void h2y_oracle_up444(uint16_t *dst, const uint16_t *src, int width, int height, int algorithm,
                      unsigned minCV, unsigned maxCV)
{
    const int w2 = width >> 1, h2 = height >> 1; /* short w422 = width>>1, h420 = height>>1, :1892-1893 */
    if (algorithm == 0) {
        for (int l = 0; l < height / 2; l++)
            for (int p = 0; p < width / 2; p++) {
                const uint16_t v = src[(size_t)l * w2 + p];
                dst[(size_t)(2 * l) * width + 2 * p] = v;
                dst[(size_t)(2 * l) * width + 2 * p + 1] = v;
                dst[(size_t)(2 * l + 1) * width + 2 * p] = v;
                dst[(size_t)(2 * l + 1) * width + 2 * p + 1] = v;
            }
        return;
    }
    /* rows 2*h2 .. height-1 and columns 2*w2 .. width-1 (odd sizes) are never written by the reference */
    uint16_t *mid = (uint16_t *)malloc((size_t)(height > 0 ? height : 1) * (w2 > 0 ? w2 : 1) * sizeof(uint16_t));
    const float scale = 256.0f;
    const float c3 = 3.0f / scale, c16 = 16.0f / scale, c67 = 67.0f / scale, c227 = 227.0f / scale, c32 = 32.0f / scale, c7 = 7.0f / scale;
    const float fmaxCV = (float)maxCV, fminCV = (float)minCV;
    for (int i = 0; i < w2; i++)
        for (int j = 0; j < h2; j++) {
#define S(r) ((float)src[(size_t)clampi((r), 0, h2 - 1) * w2 + i])
            /* :1925-1931: products and sums left to right in float, the trailing +0.5 a double add rounded back */
            float acc = c3 * S(j - 3) - c16 * S(j - 2) + c67 * S(j - 1) + c227 * S(j) - c32 * S(j + 1) + c7 * S(j + 2);
            float t = (float)((double)acc + 0.5);
            if (t > fmaxCV) t = fmaxCV;
            if (t < fminCV) t = fminCV;
            mid[(size_t)(2 * j) * w2 + i] = (uint16_t)t;
            /* :1936-1944 */
            acc = c3 * S(j + 3) - c16 * S(j + 2) + c67 * S(j + 1) + c227 * S(j) - c32 * S(j - 1) + c7 * S(j - 2);
            t = (float)((double)acc + 0.5);
            if (t > fmaxCV) t = fmaxCV;
            if (t < fminCV) t = fminCV;
            mid[(size_t)(2 * j + 1) * w2 + i] = (uint16_t)t;
#undef S
        }
    /* :1949-1979; the reference walks j < height over the intermediate, whose rows >= 2*h2 (odd height) it never
     * wrote (malloc'ed garbage): here they are not produced at all */
    const float d21 = 21.0f / scale, d52 = 52.0f / scale, d159 = 159.0f / scale;
    for (int j = 0; j < 2 * h2; j++)
        for (int i = 0; i < w2; i++) {
#define D(c) ((float)mid[(size_t)j * w2 + clampi((c), 0, w2 - 1)])
            dst[(size_t)j * width + 2 * i] = mid[(size_t)j * w2 + i];
            float acc = d21 * (D(i - 2) + D(i + 3)) - d52 * (D(i - 1) + D(i + 2)) + d159 * (D(i) + D(i + 1));
            float t = (float)((double)acc + 0.5);
            if (t > fmaxCV) t = fmaxCV;
            if (t < fminCV) t = fminCV;
            dst[(size_t)j * width + 2 * i + 1] = (uint16_t)t;
#undef D
        }
    free(mid);
}

/* ---- write_yuv() per-sample arithmetic, tiff.cpp:457-550 --------------- */
void h2y_oracle_yuv_clamp(uint16_t *plane, size_t n, int down_shift, int full_range,
                          unsigned lo, unsigned hi, uint64_t maxCV)
{
    for (size_t i = 0; i < n; i++) {
        uint16_t v = (uint16_t)(plane[i] >> down_shift);
        if (full_range == 0) {
            v = (v < lo) ? (uint16_t)lo : v;
            v = (v > hi) ? (uint16_t)hi : v;
        } else {
            v = (v > maxCV) ? (uint16_t)maxCV : v;
        }
        plane[i] = v;
    }
}

size_t h2y_oracle_frame_bytes(const h2y_desc *d)
{
    size_t n = (size_t)d->width * d->height;
    size_t nc = d->dst_chroma_format_idc == H2Y_CHROMA_420 ? (size_t)(d->width >> 1) * (d->height >> 1) : n;
    return (n + 2 * nc) * sizeof(uint16_t);
}

/* ---- half <-> float (exr.cpp:233 widens half to float exactly) --------- */
float h2y_oracle_f16_to_f32(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu, bits;
    if (e == 0) {
        if (m == 0) bits = sign;
        else { /* subnormal half -> normal float */
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; sh++; }
            m &= 0x3FFu;
            bits = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31) bits = sign | 0x7F800000u | (m << 13);
    else bits = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

uint16_t h2y_oracle_f32_to_f16(float f)
{
    /* round to nearest even, as SURVEY 8c's fp16 frames are made */
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    int32_t e = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
    uint32_t m = x & 0x7FFFFFu;
    if (((x >> 23) & 0xFF) == 0xFF) return (uint16_t)(sign | 0x7C00u | (m ? 0x200u : 0));
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        int shift = 14 - e;
        uint32_t r = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)e << 10) | (m >> 13), rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) r++;
    return (uint16_t)(sign | r);
}

void h2y_oracle_synth_plane_f32(float *plane, size_t n, uint32_t *lcg_state)
{
    uint32_t s = *lcg_state;
    for (size_t i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        plane[i] = (float)(s >> 8) * (1.0f / 16777216.0f);
    }
    *lcg_state = s;
}

/* ---- main() sequence, hdr2yuv.cpp:797-928 ------------------------------ */
int h2y_oracle_convert_frame(const h2y_desc *d, const void *const in_planes[3], uint16_t *out_yuv)
{
    if (d->width < 1 || d->height < 1) return H2Y_EINVAL;
    const size_t n = (size_t)d->width * d->height;
    const int is_u16 = d->in_sample_type == H2Y_SAMPLE_U16;
    const int is_f16 = d->in_sample_type == H2Y_SAMPLE_F16;
    int rc = 0;

    /* F16: widen first (exr.cpp:233-235) */
    float *wide[3] = {0, 0, 0};
    const void *planes[3] = {in_planes[0], in_planes[1], in_planes[2]};
    h2y_desc dd = *d;
    if (is_f16) {
        for (int c = 0; c < 3; c++) {
            wide[c] = (float *)malloc(n * sizeof(float));
            const uint16_t *h = (const uint16_t *)in_planes[c];
            for (size_t i = 0; i < n; i++) wide[c][i] = h2y_oracle_f16_to_f32(h[i]);
            planes[c] = wide[c];
        }
        dd.in_sample_type = H2Y_SAMPLE_F32;
    }

    /* pic_stats(in_pic), hdr2yuv.cpp:797 */
    int32_t fl[3], ce[3];
    if (d->stats_override) {
        for (int c = 0; c < 3; c++) { fl[c] = d->floor[c]; ce[c] = d->ceiling[c]; }
    } else if (is_u16) {
        uint16_t mm[6];
        h2y_oracle_stats_u16((const uint16_t *const *)planes, n, d->src_bit_depth, mm, fl, ce);
    } else {
        float mm[6];
        h2y_oracle_stats_f32((const float *const *)planes, n, mm, fl, ce);
    }

    /* hdr2yuv.cpp:803-812: tmp_pic is U16 (the .yuv output type); it takes the
     * input's depth when the input is U16 too, else the output's */
    const int tmp_bit_depth = is_u16 ? d->src_bit_depth : d->dst_bit_depth;
    uint16_t *tmp[3];
    for (int c = 0; c < 3; c++) tmp[c] = (uint16_t *)malloc(n * sizeof(uint16_t));
    rc = h2y_oracle_matrix_convert(&dd, planes, fl, ce, tmp_bit_depth, tmp);

    if (rc == 0) {
        h2y_oracle_clip tclip, oclip;
        h2y_oracle_set_clip(tmp_bit_depth, d->dst_full_range, &tclip);
        h2y_oracle_set_clip(d->dst_bit_depth, d->dst_full_range, &oclip);
        uint16_t *Y = out_yuv;
        size_t nc;
        /* convert(), convert.cpp:802-859, or the 4:4:4 memcpy at hdr2yuv.cpp:912-921 */
        memcpy(Y, tmp[0], n * sizeof(uint16_t));
        if (d->dst_chroma_format_idc == H2Y_CHROMA_420) {
            nc = (size_t)(d->width >> 1) * (d->height >> 1);
            for (int c = 1; c < 3; c++) {
                uint16_t *dst = out_yuv + n + (size_t)(c - 1) * nc;
                if (d->chroma_resampler_type == 0) h2y_oracle_sub420_box(dst, tmp[c], d->width, d->height);
                else h2y_oracle_sub420_fir(dst, tmp[c], d->width, d->height, tclip.minCV, tclip.maxCV);
            }
        } else {
            nc = n;
            memcpy(out_yuv + n, tmp[1], n * sizeof(uint16_t));
            memcpy(out_yuv + 2 * n, tmp[2], n * sizeof(uint16_t));
        }
        /* write_yuv(dst, &h, out_pic, tmp_pic->bit_depth), hdr2yuv.cpp:928: the
         * clamp limits are the OUTPUT picture's */
        const int down_shift = tmp_bit_depth - d->dst_bit_depth;
        if (down_shift < 0) rc = H2Y_EINVAL; /* reference exit(0)s, tiff.cpp:396-401 */
        else {
            h2y_oracle_yuv_clamp(Y, n, down_shift, d->dst_full_range, oclip.minVR, oclip.maxVR, oclip.maxCV);
            h2y_oracle_yuv_clamp(out_yuv + n, nc, down_shift, d->dst_full_range, oclip.minVRC, oclip.maxVRC, oclip.maxCV);
            h2y_oracle_yuv_clamp(out_yuv + n + nc, nc, down_shift, d->dst_full_range, oclip.minVRC, oclip.maxVRC, oclip.maxCV);
        }
    }
    for (int c = 0; c < 3; c++) { free(tmp[c]); free(wide[c]); }
    return rc;
}

/* ---- matrix_inverse, convert.cpp:1320-1867 -------------------------------------------------
 * What the compiled function does, oddities included:
 *  - Half = 2048 and Full = 4096 whatever the bit depth (convert.cpp:1337-1338);
 *  - "DXYZ = 0" is decided by comparing matrix_coeffs with the BOOLEANS D709, D2020, Y100, Y500
 *    (convert.cpp:1391): true for matrix 1 (BT.709, D709 == 1) and matrix 0 (== Y100 == 0) only, so
 *    BT.2020 pictures take the Y'DzDx formula like everything else; matrix 0 ends in exit(0)
 *    (convert.cpp:1733-1736);
 *  - 0.07222 (sic) in the BT.709 green, convert.cpp:1676;
 *  - the video-range clamp always runs, with the INPUT picture's limits (FULLRANGE is a constant 0,
 *    convert.cpp:1343,1783-1793): for a full-range input those limits are 0 and maxCV. */
int h2y_oracle_matrix_inverse(int width, int height, int in_bit_depth, int in_full_range, int in_matrix, int out_bit_depth,
                              const uint16_t *const in_planes[3], uint16_t *const out_planes[3])
{
    const unsigned short Half = 2048, Full = 4096;
    h2y_oracle_clip clip;
    h2y_oracle_set_clip(in_bit_depth, in_full_range, &clip);
    const int d709 = in_matrix == 1;
    if (in_matrix == 0) return 1;
    const size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; i++) {
        float Yav = (float)in_planes[0][i], Cb = (float)in_planes[1][i], Cr = (float)in_planes[2][i];
        float Rp, Bp, tmpF;
        if (!d709) {
            Rp = (float)(2.0 * Cr - (Full - 1.0) + Yav);
            Bp = (float)(2.0 * Cb - (Full - 1.0) + Yav);
        } else {
            tmpF = (float)(((float)(Cb) - (Half - 0.5)) * 1.8556 + Yav);
            if (tmpF > (Full - 1.0)) tmpF = (float)(Full - 1.0);
            Bp = tmpF;
            tmpF = (float)(((float)(Cr) - (Half - 0.5)) * 1.5748 + Yav);
            if (tmpF > (Full - 1.0)) tmpF = (float)(Full - 1.0);
            Rp = tmpF;
            tmpF = (float)(((float)Yav - 0.07222 * (float)Bp - 0.2126 * (float)Rp) / 0.7152 + 0.5);
            if (tmpF > (Full - 1.0)) tmpF = (float)(Full - 1.0);
            Yav = tmpF;
        }
        int G = (int)Yav, B = (int)Bp, R = (int)Rp;
        if (G < 0) G = 0;
        if (R < 0) R = 0;
        if (B < 0) B = 0;
        R = (R < clip.minVR) ? clip.minVR : R;
        G = (G < clip.minVR) ? clip.minVR : G;
        B = (B < clip.minVR) ? clip.minVR : B;
        R = (R > clip.maxVR) ? clip.maxVR : R;
        G = (G > clip.maxVR) ? clip.maxVR : G;
        B = (B > clip.maxVR) ? clip.maxVR : B;
        if (in_bit_depth > out_bit_depth) {
            int shift = in_bit_depth - out_bit_depth;
            R = R >> shift; G = G >> shift; B = B >> shift;
        } else {
            int shift = out_bit_depth - in_bit_depth;
            R = R << shift; G = G << shift; B = B << shift;
        }
        out_planes[0][i] = (uint16_t)G;
        out_planes[1][i] = (uint16_t)B;
        out_planes[2][i] = (uint16_t)R;
    }
    return 0;
}
