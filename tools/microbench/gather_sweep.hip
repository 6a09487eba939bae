// Micro-benchmark behind DESIGN.md section 5.5 (round 6): what bounds a lane-per-document dictionary walk -- the rate of dependent 8-byte
// gathers per CU as a function of (a) the table's footprint (0.5 .. 16 MB against 32 KB of L1 per CU and 4 MB of L2 per XCD), (b) the
// resident waves per CU (set by a dynamic LDS reservation, like the Unigram lane program whose rings cap it), (c) independent chains per lane.
// Each lane follows idx -> T[idx] -> idx' chains (uniformly random: the worst case; a dictionary's top levels are hotter).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MLP>
__global__ __launch_bounds__(64) void k_chain(const uint64_t *T, uint32_t mask /* entries */, int iters, uint32_t *out)
{
    extern __shared__ uint32_t pad[];
    uint32_t idx[MLP]; uint32_t acc = 0;
#pragma unroll
    for (int m = 0; m < MLP; ++m) idx[m] = (blockIdx.x * 64u + threadIdx.x + 0x9E3779B9u * (uint32_t)m) * 2654435761u;
    for (int i = 0; i < iters; ++i) {
        uint64_t e[MLP];
#pragma unroll
        for (int m = 0; m < MLP; ++m) e[m] = T[__umulhi(idx[m], mask)];
#pragma unroll
        for (int m = 0; m < MLP; ++m) { idx[m] = ((uint32_t)e[m] + (uint32_t)i * 40503u) * 2654435761u; acc += idx[m]; }
    }
    if (acc == 0x12345678u) pad[threadIdx.x] = acc;
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main()
{
    const uint32_t nmax = 1u << 21;                   // 2 M entries x 8 B = 16 MB
    std::vector<uint64_t> h(nmax);
    uint32_t s = 12345; for (uint32_t i = 0; i < nmax; ++i) { s = s * 1664525u + 1013904223u; h[i] = s >> 3; }
    uint64_t *d; uint32_t *out;
    CK(hipMalloc(&d, (size_t)nmax * 8)); CK(hipMalloc(&out, 256 * 40 * 64 * 4));
    CK(hipMemcpy(d, h.data(), (size_t)nmax * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount; const double clk = prop.clockRate * 1e3;
    printf("device %s, %d CUs, %.2f GHz\n", prop.name, cus, clk * 1e-9);
    CK(hipFuncSetAttribute((const void *)k_chain<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void *)k_chain<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void *)k_chain<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    const int iters = 1500;
    printf("%-10s %-9s %-4s %9s %22s %14s\n", "table", "waves/CU", "mlp", "ms", "lane-gathers/clk/CU", "G gathers/s");
    for (uint32_t n : {1u << 16, 1u << 17, 1u << 18, 3u << 17, 1u << 19, 3u << 18, 1u << 20, 1u << 21}) {
        for (int wpc : {8, 13, 20, 32}) {
            const size_t lds = (size_t)(160 * 1024 / wpc) & ~(size_t)1023;      // caps the resident workgroups (of one wave) per CU at wpc
            for (int mlp : {1, 2, 4}) {
                float ms = 0;
                const int blocks = cus * wpc;
                auto run = [&](int it) {
                    if (mlp == 1) k_chain<1><<<blocks, 64, lds>>>(d, n, it, out);
                    else if (mlp == 2) k_chain<2><<<blocks, 64, lds>>>(d, n, it, out);
                    else k_chain<4><<<blocks, 64, lds>>>(d, n, it, out);
                };
                run(20);
                CK(hipEventRecord(e0)); run(iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double gathers = (double)blocks * 64.0 * iters * mlp;
                printf("%6.1f MB  %-9d %-4d %9.3f %22.3f %14.1f\n", n * 8.0 / (1 << 20), wpc, mlp, ms, gathers / (ms * 1e-3 * clk * cus), gathers / (ms * 1e-3) * 1e-9);
            }
        }
    }
    return 0;
}
