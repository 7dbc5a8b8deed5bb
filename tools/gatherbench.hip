// tools/gatherbench.hip -- random 16-byte gathers from a small table: LDS vs global(L1) (GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define ITER 256
__global__ __launch_bounds__(512) void kg(const uint4* __restrict__ tab, const int* rnd, long long* out, int nrec, int mode)
{
    __shared__ uint4 stab[1601];
    for (int j = threadIdx.x; j < 1601; j += blockDim.x) stab[j] = tab[j % nrec];
    __syncthreads();
    int idx[8];
    for (int c = 0; c < 8; c++) idx[c] = rnd[(threadIdx.x * 8 + c + blockIdx.x * 4096) & 65535] % nrec;
    unsigned acc = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < 8; c++) {
            uint4 v;
            if (mode == 0) v = stab[idx[c]];
            else v = tab[idx[c]];
            acc += v.x;
            idx[c] = (idx[c] * 5 + 1 + (v.y & 1)) % nrec; // data-dependent next index: keeps loads in the loop, random walk
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 12345) out[100000] = 1;
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int ncu = p.multiProcessorCount;
    uint4* tab; hipMalloc(&tab, 1601 * 16 * 4); hipMemset(tab, 1, 1601 * 16 * 4);
    std::vector<int> r(65536); srand(3); for (auto& x : r) x = rand();
    int* drnd; hipMalloc(&drnd, 65536 * 4); hipMemcpy(drnd, r.data(), 65536 * 4, hipMemcpyHostToDevice);
    long long* dout; hipMalloc(&dout, (ncu * 4 * 8 + 200000) * 8);
    for (int mode = 0; mode < 2; mode++)
        for (int nrec : {64, 256, 1600})
            for (int bpc : {1, 2}) {
                int grid = ncu * bpc;
                hipLaunchKernelGGL(kg, dim3(grid), dim3(512), 0, 0, tab, drnd, dout, nrec, mode); hipDeviceSynchronize();
                hipLaunchKernelGGL(kg, dim3(grid), dim3(512), 0, 0, tab, drnd, dout, nrec, mode); hipDeviceSynchronize();
                std::vector<long long> h(grid * 8); hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
                double mx = 0; for (auto v : h) mx = v > mx ? v : mx;
                double per_cu_instrs = (double)ITER * 8 * 8 * bpc; // wave-instrs per CU
                printf("%s nrec %4d blocks/CU %d : %8.0f cycles -> %.1f cycles per wave-gather per CU\n", mode ? "global" : "LDS   ", nrec, bpc, mx, mx / per_cu_instrs);
            }
    return 0;
}
