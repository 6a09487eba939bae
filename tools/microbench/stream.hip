// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of the tokenising kernels
// (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports 1/2 of a wide coalesced 16 B/lane read; other widths are uncalibrated -- calibrate on
// a known byte count in your own access pattern").  Each kernel moves a KNOWN number of bytes once:
//   k_read<8>   8 bytes per lane, consecutive lanes consecutive (the text read of k_wp_wave)
//   k_read<16>  16 bytes per lane (the guide's reference shape)
//   k_write4    4 bytes per lane, consecutive (id stores)
// usage: stream <MiB>; run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` and divide.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int B> struct Vec;
template <> struct Vec<8> { typedef uint2 T; };
template <> struct Vec<16> { typedef uint4 T; };

template <int B>
__global__ __launch_bounds__(256) void k_read(const typename Vec<B>::T *p, size_t n, unsigned *out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const typename Vec<B>::T v = p[i]; acc += v.x ^ v.y; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write4(unsigned *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (unsigned)i;
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 2048;
    const size_t bytes = mib << 20;
    void *buf = nullptr; unsigned *out = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_read<8>, dim3(256 * 32), dim3(256), 0, 0, (const uint2 *)buf, bytes / 8, out);
    hipLaunchKernelGGL(k_read<16>, dim3(256 * 32), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, out);
    hipLaunchKernelGGL(k_write4, dim3(256 * 32), dim3(256), 0, 0, (unsigned *)buf, bytes / 4);
    hipDeviceSynchronize();
    printf("bytes per kernel %zu\n", bytes);
    return 0;
}
