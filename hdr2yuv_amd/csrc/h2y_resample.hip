/*
 * h2y_resample.hip -- chroma resampling kernels that stand alone (their own launches):
 *
 *   k_up444   Subsample420to444   convert.cpp:1869-1986 (also yuv2tiff.cpp:575-692): 4:2:0 -> 4:4:4 of one
 *             chroma plane, by replication (algorithm 0) or by the reference's FIR pair -- vertical
 *             (3 -16 67 227 -32 7)/256 per output-row parity into a U16 4:2:2 intermediate, then
 *             horizontal: even samples copied, odd samples (21 -52 159 159 -52 21)/256; edges replicated
 *             by index clamping; every stage clamps to [minCV, maxCV] and truncates to unsigned short.
 *
 *   k_inverse420  the .yuv 4:2:0 -> RGB flow in ONE pass (yuv2tiff.cpp:341-342 then its pixel loop = matrix_inverse,
 *             convert.cpp:1320-1867): both chroma planes upsampled as above inside the block (source tiles and the U16
 *             intermediate in LDS), the 4:4:4 chroma never written: 3 B/px read, 6 B/px written, against 17 B/px for
 *             k_up444 followed by k_inverse.
 *
 * HBM-bound stencil work (0.5 B/px read, 2 B/px written per plane): no MFMA.  Built -ffp-contract=off: the
 * products and sums round one by one, as the reference's do.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "h2y_device.h"

#define UP_TW 64 /* source (4:2:0) columns per block -> 128 output columns */
#define UP_TH 16 /* source rows per block -> 32 output rows */
#define UP_SW (UP_TW + 5) /* staged source / intermediate columns: 2 left, 3 right of the tile */
#define UP_SH (UP_TH + 6) /* staged source rows: 3 above, 3 below */

template <bool FIR>
__global__ __launch_bounds__(256) void k_up444(up_args a)
{
    const int W = a.width, w2 = W >> 1, h2 = a.height >> 1;
    const uint16_t *src = blockIdx.z ? a.src1 : a.src0;
    uint16_t *dst = blockIdx.z ? a.dst1 : a.dst0;
    const int c0 = blockIdx.x * UP_TW, r0 = blockIdx.y * UP_TH;
    if (!FIR) {
        /* :1871-1881: dst[2p..2p+1][2l..2l+1] = src[p][l]; one thread = one source sample, two 4-byte stores */
        for (int i = threadIdx.x; i < UP_TW * UP_TH; i += 256) {
            const int r = i / UP_TW, c = i - r * UP_TW;
            const int y = r0 + r, x = c0 + c;
            if (y >= h2 || x >= w2) continue;
            const uint32_t v = src[(size_t)y * w2 + x];
            const uint32_t vv = v | (v << 16);
            uint32_t *d0 = reinterpret_cast<uint32_t *>(dst + (size_t)(2 * y) * W) + x; /* W even, planes 4-byte aligned (host checks) */
            d0[0] = vv;
            d0[W >> 1] = vv;
        }
        return;
    }
    __shared__ uint16_t s_src[UP_SH][UP_SW + 1];
    __shared__ uint16_t s_mid[2 * UP_TH][UP_SW + 1];
    /* source rows r0-3 .. r0+TH+2, columns c0-2 .. c0+TW+2, indices clamped into the plane: the reference's edge
     * ternaries (:1918-1923 rows, :1960-1964 columns) -- a clamped column of the intermediate is the intermediate
     * of the clamped column */
    for (int i = threadIdx.x; i < UP_SH * UP_SW; i += 256) {
        const int r = i / UP_SW, c = i - r * UP_SW;
        const int y = min(max(r0 - 3 + r, 0), h2 - 1), x = min(max(c0 - 2 + c, 0), w2 - 1);
        s_src[r][c] = src[(size_t)y * w2 + x];
    }
    __syncthreads();
    /* vertical stage, :1911-1946: intermediate rows 2j (taps j-3..j+2) and 2j+1 (taps j+3..j-2, mirrored) */
    for (int i = threadIdx.x; i < UP_TH * UP_SW; i += 256) {
        const int r = i / UP_SW, c = i - r * UP_SW;
        float s[7];
#pragma unroll
        for (int k = 0; k < 7; k++) s[k] = (float)s_src[r + k][c]; /* source rows j-3 .. j+3 */
        s_mid[2 * r][c] = (uint16_t)up_fir6(s[0], s[1], s[2], s[3], s[4], s[5], a.fmin, a.fmax);
        s_mid[2 * r + 1][c] = (uint16_t)up_fir6(s[6], s[5], s[4], s[3], s[2], s[1], a.fmin, a.fmax);
    }
    __syncthreads();
    /* horizontal stage, :1956-1979: even output = intermediate, odd output from columns i-2 .. i+3 */
    for (int i = threadIdx.x; i < 2 * UP_TH * UP_TW; i += 256) {
        const int r = i / UP_TW, c = i - r * UP_TW;
        const int y = 2 * r0 + r, x = c0 + c;
        if (y >= 2 * h2 || x >= w2) continue;
        float m[6];
#pragma unroll
        for (int k = 0; k < 6; k++) m[k] = (float)s_mid[r][c + k]; /* intermediate columns x-2 .. x+3 */
        const uint32_t even = s_mid[r][c + 2];
        const uint32_t odd = up_fir_odd(m[0], m[1], m[2], m[3], m[4], m[5], a.fmin, a.fmax);
        reinterpret_cast<uint32_t *>(dst + (size_t)y * W)[x] = even | (odd << 16);
    }
}

/* Same tiles as k_up444 (64 x 16 source samples -> 128 x 32 pixels), both chroma planes.  LDS rows are 72 samples (144 bytes) so
 * that four neighbouring samples are one 8-byte read: the vertical stage works on four columns at a time (seven reads, eight
 * results), the last stage on eight pixels of a row (its twelve intermediate samples per plane: three reads; its luma: one
 * 16-byte load; G, B, R: three 16-byte stores). */
#define IV_LW 72 /* staged columns: c0-2 .. c0+69 (the last three only pad the row) */
#define IV_TH 8  /* source rows per tile (16 output rows): smaller than k_up444's, and a block takes several tiles one after the
                    other -- with one tile per block the whole grid loads, computes and stores in step and the three never overlap */
#define IV_SH (IV_TH + 6)
typedef uint32_t iv_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t iv_u32x4 __attribute__((ext_vector_type(4)));
template <bool FIR>
__global__ __launch_bounds__(256) void k_inverse420(inv420_args a)
{
    const up_args &u = a.up;
    const int W = u.width, w2 = W >> 1, h2 = u.height >> 1;
    const int tiles_x = (w2 + UP_TW - 1) / UP_TW, tiles = tiles_x * ((h2 + IV_TH - 1) / IV_TH);
    /* both stages' samples are kept as floats in LDS (every one of them is an integer below 2^16: exact), converted once where
     * they are produced instead of at each of their seven (vertical) or six (horizontal) uses */
    __shared__ __attribute__((aligned(16))) float s_src[2][FIR ? IV_SH : 1][IV_LW];
    __shared__ __attribute__((aligned(16))) float s_mid[2][FIR ? 2 * IV_TH : 1][IV_LW];
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int ty = tile / tiles_x, c0 = (tile - ty * tiles_x) * UP_TW, r0 = ty * IV_TH;
    if (FIR) {
        __syncthreads(); /* the previous tile's last stage has read s_mid */
        /* source rows r0-3 .. r0+TH+2, columns c0-2 .. c0+69 in PAIRS (c0 - 2 and w2 are even: a pair lies inside the plane, where
         * it is one 4-byte load, or wholly beyond an edge, where both are the edge sample -- the reference's index ternaries) */
        /* (item = (plane, row, column pair); 256 = 7 x 36 + 4: the indices advance without divisions) */
        int sp = 0, sr = (int)threadIdx.x / (IV_LW / 2), sc = (int)threadIdx.x - sr * (IV_LW / 2);
        for (int i = threadIdx.x; i < 2 * IV_SH * (IV_LW / 2); i += 256, sc += 4, sr += 7) {
            if (sc >= IV_LW / 2) { sc -= IV_LW / 2; sr++; }
            if (sr >= IV_SH) { sr -= IV_SH; sp++; }
            const int p = sp, r = sr, c = 2 * sc;
            const int y = min(max(r0 - 3 + r, 0), h2 - 1), x = c0 - 2 + c;
            const uint16_t *row = (p ? u.src1 : u.src0) + (size_t)y * w2;
            uint32_t v0, v1;
            if (x >= 0 && x + 1 < w2) {
                const uint32_t v = *reinterpret_cast<const uint32_t *>(row + x);
                v0 = v & 0xFFFFu;
                v1 = v >> 16;
            } else v0 = v1 = row[x < 0 ? 0 : w2 - 1];
            *reinterpret_cast<float2 *>(&s_src[p][r][c]) = float2{(float)v0, (float)v1};
        }
        __syncthreads();
        /* vertical stage: item = (plane, source row r, four columns) -> intermediate rows 2r and 2r+1 of those columns */
        for (int i = threadIdx.x; i < 2 * IV_TH * (IV_LW / 4); i += 256) { /* 288 items: one or two per thread */
            const int p = i >= IV_TH * (IV_LW / 4), j = i - p * (IV_TH * (IV_LW / 4));
            const int r = j / (IV_LW / 4), c = (j - r * (IV_LW / 4)) * 4;
            float4 s[7];
#pragma unroll
            for (int k = 0; k < 7; k++) s[k] = *reinterpret_cast<const float4 *>(&s_src[p][r + k][c]);
            float4 e, o;
            e.x = (float)up_fir6(s[0].x, s[1].x, s[2].x, s[3].x, s[4].x, s[5].x, u.fmin, u.fmax);
            e.y = (float)up_fir6(s[0].y, s[1].y, s[2].y, s[3].y, s[4].y, s[5].y, u.fmin, u.fmax);
            e.z = (float)up_fir6(s[0].z, s[1].z, s[2].z, s[3].z, s[4].z, s[5].z, u.fmin, u.fmax);
            e.w = (float)up_fir6(s[0].w, s[1].w, s[2].w, s[3].w, s[4].w, s[5].w, u.fmin, u.fmax);
            o.x = (float)up_fir6(s[6].x, s[5].x, s[4].x, s[3].x, s[2].x, s[1].x, u.fmin, u.fmax);
            o.y = (float)up_fir6(s[6].y, s[5].y, s[4].y, s[3].y, s[2].y, s[1].y, u.fmin, u.fmax);
            o.z = (float)up_fir6(s[6].z, s[5].z, s[4].z, s[3].z, s[2].z, s[1].z, u.fmin, u.fmax);
            o.w = (float)up_fir6(s[6].w, s[5].w, s[4].w, s[3].w, s[2].w, s[1].w, u.fmin, u.fmax);
            *reinterpret_cast<float4 *>(&s_mid[p][2 * r][c]) = e;
            *reinterpret_cast<float4 *>(&s_mid[p][2 * r + 1][c]) = o;
        }
        __syncthreads();
    }
    /* last stage: item = (row of the tile, eight pixels = four chroma columns c .. c+3) */
    for (int i = threadIdx.x; i < 2 * IV_TH * (UP_TW / 4); i += 256) {
        const int r = i / (UP_TW / 4), c = (i - r * (UP_TW / 4)) * 4;
        const int y = 2 * r0 + r, x = c0 + c;
        if (y >= 2 * h2 || x >= w2) continue;
        uint32_t ce[2][4], co[2][4]; /* [plane][pair]: chroma at the even and the odd column */
        if (FIR) {
#pragma unroll
            for (int p = 0; p < 2; p++) {
                float m[12]; /* intermediate columns x-2 .. x+9 */
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float4 v = *reinterpret_cast<const float4 *>(&s_mid[p][r][c + 4 * k]);
                    m[4 * k] = v.x; m[4 * k + 1] = v.y; m[4 * k + 2] = v.z; m[4 * k + 3] = v.w;
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    ce[p][q] = (uint32_t)m[q + 2]; /* even output = the intermediate sample (convert.cpp:1966) */
                    co[p][q] = up_fir_odd(m[q], m[q + 1], m[q + 2], m[q + 3], m[q + 4], m[q + 5], u.fmin, u.fmax);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int xs = min(x + q, w2 - 1);
                ce[0][q] = co[0][q] = u.src0[(size_t)(y >> 1) * w2 + xs];
                ce[1][q] = co[1][q] = u.src1[(size_t)(y >> 1) * w2 + xs];
            }
        }
        const size_t p0 = (size_t)y * w2 + x; /* index of the first pixel pair; W a multiple of 4: x + 3 < w2 or x + 1 < w2 */
        const bool whole = x + 3 < w2;
        uint32_t yy[4];
        if (whole && ((p0 & 3u) == 0)) {
            const iv_u32x4 v = reinterpret_cast<const iv_u32x4 *>(a.inv.in[0])[p0 >> 2];
            yy[0] = v.x; yy[1] = v.y; yy[2] = v.z; yy[3] = v.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) yy[q] = x + q < w2 ? reinterpret_cast<const uint32_t *>(a.inv.in[0])[p0 + q] : 0u;
        }
        uint32_t G[4], B[4], R[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t g0, b0, r0_, g1, b1, r1;
            inverse_pixel(a.inv, yy[q] & 0xFFFFu, ce[0][q], ce[1][q], g0, b0, r0_);
            inverse_pixel(a.inv, yy[q] >> 16, co[0][q], co[1][q], g1, b1, r1);
            G[q] = g0 | (g1 << 16); B[q] = b0 | (b1 << 16); R[q] = r0_ | (r1 << 16);
        }
        if (whole && ((p0 & 3u) == 0)) {
            reinterpret_cast<iv_u32x4 *>(a.inv.out[0])[p0 >> 2] = iv_u32x4{G[0], G[1], G[2], G[3]};
            reinterpret_cast<iv_u32x4 *>(a.inv.out[1])[p0 >> 2] = iv_u32x4{B[0], B[1], B[2], B[3]};
            reinterpret_cast<iv_u32x4 *>(a.inv.out[2])[p0 >> 2] = iv_u32x4{R[0], R[1], R[2], R[3]};
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (x + q < w2) {
                    reinterpret_cast<uint32_t *>(a.inv.out[0])[p0 + q] = G[q];
                    reinterpret_cast<uint32_t *>(a.inv.out[1])[p0 + q] = B[q];
                    reinterpret_cast<uint32_t *>(a.inv.out[2])[p0 + q] = R[q];
                }
        }
    }
    }
}

hipError_t h2y_launch_inverse420(hipStream_t st, const inv420_args &a)
{
    const int w2 = a.up.width >> 1, h2 = a.up.height >> 1;
    const int tiles = ((w2 + UP_TW - 1) / UP_TW) * ((h2 + IV_TH - 1) / IV_TH);
    dim3 grid(tiles < 2048 ? tiles : 2048, 1, 1); /* eight blocks of 256 per CU at most: the rest of the tiles by the same blocks */
    if (a.up.algorithm == 0) hipLaunchKernelGGL(k_inverse420<false>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_inverse420<true>, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t h2y_launch_up444(hipStream_t st, const up_args &a)
{
    const int w2 = a.width >> 1, h2 = a.height >> 1;
    dim3 grid((w2 + UP_TW - 1) / UP_TW, (h2 + UP_TH - 1) / UP_TH, a.src1 ? 2 : 1);
    if (a.algorithm == 0) hipLaunchKernelGGL(k_up444<false>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_up444<true>, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}
