// VALU / SALU issue rate of one SIMD of gfx950, measured: how many wave64 instructions per cycle a SIMD issues as a function of the
// resident waves per SIMD and of the instruction-level parallelism inside a wave.  Settles the constant DESIGN.md section 5 rests on
// (one wave64 VALU instruction per 4 cycles per SIMD -- or per 2, MI355X_MICROARCH.md "Wave scheduling").
//
// A workgroup is four waves (one per SIMD); W workgroups per CU give W waves per SIMD.  Every wave runs ITER iterations of a loop body
// of 64 instructions of one kind:
//   add_dep    v_add_u32 chain, each instruction depends on the one before it (issue interval of ONE wave)
//   add_ilp8   eight independent v_add_u32 chains, round robin
//   and_ilp8   the same with v_and_b32
//   mix        v_add_u32 and s_add_u32 alternating (does the scalar unit issue next to the vector unit?)
//   salu_dep   s_add_u32 chain
//   mix_3v1s / mix_1v3s   three vector instructions per scalar one, and the reverse: is the limit per kind, or one limit for all instructions?
// Cycles are s_memtime differences (shader clock) taken by every wave around its loop; the kernel's duration comes from HIP events.
// Output: per kind and W, wave-instructions per cycle per SIMD by the slowest wave's own cycle count, and the clock = cycles / time.
// usage: valu_issue [iter]      (run alone; under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES` for the counters)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(256) void k_issue(int iter, unsigned long long *cyc, unsigned *sink)
{
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned s0 = blockIdx.x, s1 = blockIdx.x + 1;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iter; ++it) {
        if (KIND == 0) {
            REP8(REP8(asm volatile("v_add_u32 %0, %0, 1" : "+v"(a0));))
        } else if (KIND == 1) {
            REP8(asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 2) {
            REP8(asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %4\n v_and_b32 %4, %4, %5\n v_and_b32 %5, %5, %6\n v_and_b32 %6, %6, %7\n v_and_b32 %7, %7, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 3) {
            REP8(asm volatile("v_add_u32 %0, %0, 1\n s_add_u32 %4, %4, 1\n v_add_u32 %1, %1, 1\n s_add_u32 %4, %4, 1\n v_add_u32 %2, %2, 1\n s_add_u32 %4, %4, 1\n v_add_u32 %3, %3, 1\n s_add_u32 %4, %4, 1"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0) : : "scc");)
        } else if (KIND == 4) {
            REP8(REP8(asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) : : "scc");))
        } else if (KIND == 5) {        // three vector : one scalar
            REP8(asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n s_add_u32 %4, %4, 1\n v_add_u32 %3, %3, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n s_add_u32 %4, %4, 1"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0) : : "scc");)
        } else {                       // one vector : three scalar (two scalar chains)
            REP8(asm volatile("v_add_u32 %0, %0, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %2, %2, 1\n v_add_u32 %1, %1, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1"
                              : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1) : : "scc");)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1) == 0x12345u) sink[0] = a0;
}

template <int KIND>
static void run(const char *name, int vper, int sper, int iter, int cus, unsigned long long *d_cyc, unsigned *d_sink)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int W = 1; W <= 8; ++W) {
        const int blocks = cus * W;
        hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), 0, 0, iter / 8, d_cyc, d_sink);      // warm-up
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), 0, 0, iter, d_cyc, d_sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c((size_t)blocks * 4);
        hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost);
        std::sort(c.begin(), c.end());
        const double med = (double)c[c.size() / 2], mx = (double)c.back();
        const double per_wave = (double)iter * 64.0;               // instructions of the loop body per wave
        // W waves share a SIMD: instructions issued by the SIMD while its slowest wave ran its loop = W * per_wave
        const double v_rate = (double)vper / 64.0 * W * per_wave / mx, s_rate = (double)sper / 64.0 * W * per_wave / mx;
        // the same by the wall clock (if s_memtime should not tick with the shader clock): wave-instructions per ns per SIMD
        const double v_ns = (double)vper / 64.0 * W * per_wave / (ms * 1e6), s_ns = (double)sper / 64.0 * W * per_wave / (ms * 1e6);
        printf("%-9s waves/SIMD %d  cycles/wave median %.0f max %.0f  kernel %.3f ms  counter %.2f GHz  VALU %.3f  SALU %.3f wave-instr/cycle/SIMD  (cycles per VALU instr of one wave: %.2f)  per ns: VALU %.3f SALU %.3f\n",
               name, W, med, mx, ms, mx / (ms * 1e6), v_rate, s_rate, vper ? med / (per_wave * vper / 64.0) : 0.0, v_ns, s_ns);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char **argv)
{
    const int iter = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned long long *d_cyc = nullptr; unsigned *d_sink = nullptr;
    if (hipMalloc(&d_cyc, (size_t)cus * 8 * 4 * 8) != hipSuccess || hipMalloc(&d_sink, 64) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    printf("%s, %d CUs, clockRate %.2f GHz, %d loop iterations of 64 instructions per wave\n", prop.gcnArchName, cus, prop.clockRate / 1e6, iter);
    run<0>("add_dep", 64, 0, iter, cus, d_cyc, d_sink);
    run<1>("add_ilp8", 64, 0, iter, cus, d_cyc, d_sink);
    run<2>("and_ilp8", 64, 0, iter, cus, d_cyc, d_sink);
    run<3>("mix", 32, 32, iter, cus, d_cyc, d_sink);
    run<4>("salu_dep", 0, 64, iter, cus, d_cyc, d_sink);
    run<5>("mix_3v1s", 48, 16, iter, cus, d_cyc, d_sink);
    run<6>("mix_1v3s", 16, 48, iter, cus, d_cyc, d_sink);
    hipFree(d_cyc); hipFree(d_sink);
    return 0;
}
