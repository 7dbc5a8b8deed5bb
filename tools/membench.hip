// tools/membench.hip -- memory-pattern experiments for the fused kernel's access shape (GPU box).
// Frame = 3 fp32 planes in (W*H each), out = Y (u16 W*H) + Cb,Cr (u16 W/2*H/2): 15 B/px.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define W 3840
#define H 2160
#define NF 8

struct frame { const float* in[3]; uint16_t* out; };

// pattern A: thread = 4 cols x 2 rows (current k_fused), trivial math
__global__ __launch_bounds__(512) void kA(const frame* fr, int nf, int grid_chunks)
{
    const uint32_t WQ = W / 4, tiles = WQ * (H / 2), chunks = (tiles + 511) / 512;
    const size_t npix = (size_t)W * H;
    for (int f = 0; f < nf; f++) {
        frame io = fr[f];
        uint32_t gbase = ((uint64_t)f * chunks) % gridDim.x;
        for (uint32_t k = (blockIdx.x + gridDim.x - gbase) % gridDim.x; k < chunks; k += gridDim.x) {
            uint32_t tt = k * 512 + threadIdx.x;
            if (tt >= tiles) continue;
            uint32_t rp = tt / WQ, cg = tt - rp * WQ, x = cg * 4, y = rp * 2;
            size_t i0 = (size_t)y * W + x, i1 = i0 + W;
            float4 g0 = *(const float4*)(io.in[0] + i0), b0 = *(const float4*)(io.in[1] + i0), r0 = *(const float4*)(io.in[2] + i0);
            float4 g1 = *(const float4*)(io.in[0] + i1), b1 = *(const float4*)(io.in[1] + i1), r1 = *(const float4*)(io.in[2] + i1);
            auto q = [](float a, float b, float c) { return (uint32_t)((a + b + c) * 1000.0f) & 0xFFFFu; };
            uint32_t y00 = q(g0.x, b0.x, r0.x), y01 = q(g0.y, b0.y, r0.y), y02 = q(g0.z, b0.z, r0.z), y03 = q(g0.w, b0.w, r0.w);
            uint32_t y10 = q(g1.x, b1.x, r1.x), y11 = q(g1.y, b1.y, r1.y), y12 = q(g1.z, b1.z, r1.z), y13 = q(g1.w, b1.w, r1.w);
            *(uint2*)(io.out + i0) = make_uint2(y00 | (y01 << 16), y02 | (y03 << 16));
            *(uint2*)(io.out + i1) = make_uint2(y10 | (y11 << 16), y12 | (y13 << 16));
            size_t ic = (size_t)rp * (W / 2) + (x >> 1);
            uint16_t* cb = io.out + npix; uint16_t* cr = cb + npix / 4;
            *(uint32_t*)(cb + ic) = ((y00 + y01 + y10 + y11) >> 2) | (((y02 + y03 + y12 + y13) >> 2) << 16);
            *(uint32_t*)(cr + ic) = ((y00 + y10) >> 1) | (((y03 + y13) >> 1) << 16);
        }
    }
}

// pattern B: thread = 8 cols x 2 rows via two float4 loads per row per plane, 16B Y stores, 8B chroma stores
__global__ __launch_bounds__(512) void kB(const frame* fr, int nf, int)
{
    const uint32_t WQ = W / 8, tiles = WQ * (H / 2), chunks = (tiles + 511) / 512;
    const size_t npix = (size_t)W * H;
    for (int f = 0; f < nf; f++) {
        frame io = fr[f];
        uint32_t gbase = ((uint64_t)f * chunks) % gridDim.x;
        for (uint32_t k = (blockIdx.x + gridDim.x - gbase) % gridDim.x; k < chunks; k += gridDim.x) {
            uint32_t tt = k * 512 + threadIdx.x;
            if (tt >= tiles) continue;
            uint32_t rp = tt / WQ, cg = tt - rp * WQ, x = cg * 8, y = rp * 2;
            size_t i0 = (size_t)y * W + x, i1 = i0 + W;
            uint32_t yy[2][8];
            for (int r = 0; r < 2; r++) {
                size_t i = r ? i1 : i0;
                float4 ga = *(const float4*)(io.in[0] + i), gb = *(const float4*)(io.in[0] + i + 4);
                float4 ba = *(const float4*)(io.in[1] + i), bb = *(const float4*)(io.in[1] + i + 4);
                float4 ra = *(const float4*)(io.in[2] + i), rb = *(const float4*)(io.in[2] + i + 4);
                auto q = [](float a, float b, float c) { return (uint32_t)((a + b + c) * 1000.0f) & 0xFFFFu; };
                yy[r][0] = q(ga.x, ba.x, ra.x); yy[r][1] = q(ga.y, ba.y, ra.y); yy[r][2] = q(ga.z, ba.z, ra.z); yy[r][3] = q(ga.w, ba.w, ra.w);
                yy[r][4] = q(gb.x, bb.x, rb.x); yy[r][5] = q(gb.y, bb.y, rb.y); yy[r][6] = q(gb.z, bb.z, rb.z); yy[r][7] = q(gb.w, bb.w, rb.w);
                *(uint4*)(io.out + i) = make_uint4(yy[r][0] | (yy[r][1] << 16), yy[r][2] | (yy[r][3] << 16), yy[r][4] | (yy[r][5] << 16), yy[r][6] | (yy[r][7] << 16));
            }
            size_t ic = (size_t)rp * (W / 2) + (x >> 1);
            uint16_t* cb = io.out + npix; uint16_t* cr = cb + npix / 4;
            uint32_t c0 = (yy[0][0] + yy[0][1] + yy[1][0] + yy[1][1]) >> 2, c1 = (yy[0][2] + yy[0][3] + yy[1][2] + yy[1][3]) >> 2;
            uint32_t c2 = (yy[0][4] + yy[0][5] + yy[1][4] + yy[1][5]) >> 2, c3 = (yy[0][6] + yy[0][7] + yy[1][6] + yy[1][7]) >> 2;
            *(uint2*)(cb + ic) = make_uint2(c0 | (c1 << 16), c2 | (c3 << 16));
            *(uint2*)(cr + ic) = make_uint2(c1 | (c0 << 16), c3 | (c2 << 16));
        }
    }
}

// plain float4 copy of the same byte volume (read 12 B/px, write 3 B/px equivalent)
__global__ __launch_bounds__(512) void kcopy(const float4* in, float4* out, size_t n_in4, size_t n_out4)
{
    size_t stride = (size_t)gridDim.x * blockDim.x;
    float4 acc = make_float4(0, 0, 0, 0);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t j = i; j < n_in4; j += stride) { float4 v = in[j]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        if ((j & 3) == 0 && (j >> 2) < n_out4) out[j >> 2] = v; }
    if (acc.x == 123.456f) out[0] = acc;
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int ncu = p.multiProcessorCount;
    size_t npix = (size_t)W * H, out_elems = npix * 3 / 2;
    std::vector<frame> fr(NF);
    for (int f = 0; f < NF; f++) {
        for (int c = 0; c < 3; c++) { float* d; hipMalloc(&d, npix * 4); hipMemset(d, 0x3c, npix * 4); fr[f].in[c] = d; }
        hipMalloc(&fr[f].out, out_elems * 2);
    }
    frame* dfr; hipMalloc(&dfr, sizeof(frame) * NF); hipMemcpy(dfr, fr.data(), sizeof(frame) * NF, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 5; i++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double us = ms * 1000 / 5 / NF;
        printf("%-34s %7.1f us/frame  %6.0f GB/s\n", name, us, npix * 15.0 / us / 1e3);
    };
    for (int bpc : {1, 2, 3, 4, 6}) {
        char nm[64]; snprintf(nm, 64, "A: 4x2 tile, %d blocks/CU", bpc);
        timeit(nm, [&]() { hipLaunchKernelGGL(kA, dim3(ncu * bpc), dim3(512), 0, 0, dfr, NF, 0); });
    }
    for (int bpc : {1, 2, 3, 4}) {
        char nm[64]; snprintf(nm, 64, "B: 8x2 tile 16B stores, %d blocks/CU", bpc);
        timeit(nm, [&]() { hipLaunchKernelGGL(kB, dim3(ncu * bpc), dim3(512), 0, 0, dfr, NF, 0); });
    }
    // copy: same volume per frame: in 3*npix floats, out npix*3/2 u16 = npix*3 bytes
    float4* cin; float4* cout; size_t n_in4 = npix * 3 / 4 * NF, n_out4 = npix * 3 / 16 * NF;
    hipMalloc(&cin, n_in4 * 16); hipMalloc(&cout, n_out4 * 16);
    for (int bpc : {2, 4, 8}) {
        char nm[64]; snprintf(nm, 64, "copy 12B in + 3B out, %d blocks/CU", bpc);
        timeit(nm, [&]() { hipLaunchKernelGGL(kcopy, dim3(ncu * bpc), dim3(512), 0, 0, cin, cout, n_in4, n_out4); });
    }
    return 0;
}
