// Micro-benchmark behind DESIGN.md section 5: cost of a wave-level divergent table gather on gfx950 as a function of
// the number of active lanes, entry width and where the table lives (global through the vector L1 / L2, or LDS).
// Each lane follows a dependent chain idx -> T[idx] -> idx' (like a DFA walk); occupancy is set by the launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <typename E>
__global__ __launch_bounds__(256) void k_global(const E *T, uint32_t mask, int iters, int active, uint32_t *out)
{
    const int lane = threadIdx.x & 63;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
            const E e = T[idx & mask];
            idx = (uint32_t)e + (uint32_t)i * 40503u;
            acc += idx;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_lds(const uint32_t *T, uint32_t mask, int iters, int active, uint32_t *out)
{
    extern __shared__ uint32_t lt[];
    for (uint32_t i = threadIdx.x; i <= mask; i += THREADS) lt[i] = T[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t idx = (blockIdx.x * (uint32_t)THREADS + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
            const uint32_t e = lt[idx & mask];
            idx = e + (uint32_t)i * 40503u;
            acc += idx;
        }
    }
    out[blockIdx.x * THREADS + threadIdx.x] = acc;
}

// mixed: a fraction of the lanes (idx below the LDS prefix) reads LDS, the rest reads global -- the lexer's planned step
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_mixed(const uint32_t *T, uint32_t mask, uint32_t lds_n, int iters, uint32_t *out)
{
    extern __shared__ uint32_t lt[];
    for (uint32_t i = threadIdx.x; i < lds_n; i += THREADS) lt[i] = T[i];
    __syncthreads();
    uint32_t idx = (blockIdx.x * (uint32_t)THREADS + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint32_t j = idx & mask;
        uint32_t e;
        if (j < lds_n) e = lt[j]; else e = T[j];
        idx = e + (uint32_t)i * 40503u;
        acc += idx;
    }
    out[blockIdx.x * THREADS + threadIdx.x] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main()
{
    const uint32_t n = 1u << 17;                      // 128 K entries: 512 KB at 4 B, 1 MB at 8 B
    std::vector<uint32_t> h32(n); std::vector<uint64_t> h64(n);
    uint32_t s = 12345; for (uint32_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h32[i] = s >> 3; h64[i] = s >> 3; }
    uint32_t *d32; uint64_t *d64; uint32_t *out;
    CK(hipMalloc(&d32, n * 4)); CK(hipMalloc(&d64, n * 8)); CK(hipMalloc(&out, 256 * 32 * 1024 * 4));
    CK(hipMemcpy(d32, h32.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d64, h64.data(), n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    const double clk = 2.4e9;
    auto report = [&](const char *name, int waves_per_cu, int active, float ms) {
        const double wave_instr = 256.0 * waves_per_cu * iters;
        const double cyc = ms * 1e-3 * clk * 256.0 / wave_instr;          // CU-cycles per wave-level gather
        printf("%-28s waves/CU %2d active lanes %2d: %7.3f ms  %6.1f CU-cycles per wave-gather  %.2f lane-gathers/clk/CU\n", name, waves_per_cu, active, ms, cyc, active / cyc);
    };
    for (int wpc : {28, 16}) {
        const int blocks = 256 * wpc / 4;
        for (int active : {64, 32, 16, 8, 4, 1}) {
            float ms;
            k_global<uint32_t><<<blocks, 256>>>(d32, n - 1, 10, active, out);
            CK(hipEventRecord(e0)); k_global<uint32_t><<<blocks, 256>>>(d32, n - 1, iters, active, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); report("global 4 B, 512 KB table", wpc, active, ms);
            CK(hipEventRecord(e0)); k_global<uint64_t><<<blocks, 256>>>(d64, n - 1, iters, active, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); report("global 8 B, 1 MB table", wpc, active, ms);
        }
    }
    for (uint32_t small : {1u << 12, 1u << 15}) {      // 16 KB (L1-resident) and 128 KB (L2) 4-byte tables
        float ms; const int wpc = 28, blocks = 256 * wpc / 4;
        for (int active : {64, 16}) {
            CK(hipEventRecord(e0)); k_global<uint32_t><<<blocks, 256>>>(d32, small - 1, iters, active, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); char nm[64]; snprintf(nm, 64, "global 4 B, %u KB table", small * 4 / 1024); report(nm, wpc, active, ms);
        }
    }
    {
        const uint32_t lds_n = 24576;                 // 96 KB
        CK(hipFuncSetAttribute((const void *)k_lds<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipFuncSetAttribute((const void *)k_mixed<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int active : {64, 32, 16}) {
            float ms;
            k_lds<1024><<<256, 1024, 16384 * 4>>>(d32, 16383, 10, active, out);
            CK(hipEventRecord(e0)); k_lds<1024><<<256, 1024, 16384 * 4>>>(d32, 16383, iters, active, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); report("LDS 4 B, 64 KB table", 16, active, ms);
        }
        for (uint32_t mask : {(1u << 15) - 1, (1u << 17) - 1}) {     // 75 % / 19 % of the lanes in the LDS prefix
            float ms;
            CK(hipEventRecord(e0)); k_mixed<1024><<<256, 1024, lds_n * 4>>>(d32, mask, lds_n, iters, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            char nm[64]; snprintf(nm, 64, "mixed, %.0f%% of lanes in LDS", 100.0 * lds_n / (mask + 1.0)); report(nm, 16, 64, ms);
        }
    }
    return 0;
}
