/*
 * h2y_resample.hip -- chroma resampling kernels that stand alone (their own launches):
 *
 *   k_up444   Subsample420to444   convert.cpp:1869-1986 (also yuv2tiff.cpp:575-692): 4:2:0 -> 4:4:4 of one
 *             chroma plane, by replication (algorithm 0) or by the reference's FIR pair -- vertical
 *             (3 -16 67 227 -32 7)/256 per output-row parity into a U16 4:2:2 intermediate, then
 *             horizontal: even samples copied, odd samples (21 -52 159 159 -52 21)/256; edges replicated
 *             by index clamping; every stage clamps to [minCV, maxCV] and truncates to unsigned short.
 *
 * HBM-bound stencil work (0.5 B/px read, 2 B/px written per plane): no MFMA.  Built -ffp-contract=off: the
 * products and sums round one by one, as the reference's do.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "h2y_kernels.h"
#include "h2y_math.h"

using namespace h2y;

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

hipError_t h2y_launch_up444(hipStream_t st, const up_args &a)
{
    const int w2 = a.width >> 1, h2 = a.height >> 1;
    dim3 grid((w2 + UP_TW - 1) / UP_TW, (h2 + UP_TH - 1) / UP_TH, a.src1 ? 2 : 1);
    if (a.algorithm == 0) hipLaunchKernelGGL(k_up444<false>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_up444<true>, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}
