// bf_kernels_common.h -- what the kernel files share: the launch helpers, the lane / wave intrinsics of gfx950 under the names the wave
// programs use (namespace wv; the test build supplies a simulator behind the same names).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "bf_kernels.h"

namespace bfa {

#define BF_WAVE 64

// compute units of the current device (persistent kernels launch resident-waves-per-CU x CUs workgroups)
static int device_cus()
{
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] <= 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }

__device__ __forceinline__ uint32_t cpmap_get(const DevCpMap &m, int cp)
{
    return m.pages[(uint32_t)m.l1[cp >> 8] * 256u + (uint32_t)(cp & 255)];
}

// Hand-off between the lanes of ONE wave through LDS (or global memory): the producer's stores are released and the consumer's
// loads acquired at wavefront scope, and the compiler may not move either across this point.  (The lanes of a wave run in
// lockstep and DS operations of a wave complete in order, so this costs nothing at run time; it pins what the code relies on.)
// the number of this wave inside its workgroup AS A SCALAR: threadIdx.x / 64 is the same in all lanes of a wave, but only readfirstlane
// tells the compiler so -- what a wave-per-document kernel derives from it (document number, lengths, loop bounds, LDS block) then
// stays in scalar registers and its loops are scalar branches instead of execution-mask loops (k_wp_wave: 96 -> 78 VGPRs, 39 -> 33 ms)
__device__ __forceinline__ int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

__device__ __forceinline__ void wave_handoff()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// inclusive wave scan by shuffles (6 steps)
__device__ __forceinline__ int wave_incl_scan(int v)
{
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(v, o, 64); if (l >= o) v += t; }
    return v;
}


// ------------------------------------------------------------------------------------------
// k_lex_wp: one document per lane (bf_lex.h).  Staging slot of document d in ids_tmp (32-byte aligned so that
// 8-id chunks are whole 32-byte sectors): base = align8(doc_off[d]) + 8*d.
// ------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int64_t ids_slot(int64_t doc_off_d, int64_t d) { return ((doc_off_d + 7) & ~(int64_t)7) + 8 * d; }

// 16-byte register window over the lane's class stream: one dwordx4 load per 8 characters.  The stream of
// document d starts at element doc_off[d] of the (256-byte aligned) cls buffer; block t of the lane is the
// aligned 16-byte block ((doc_off[d] >> 3) + t) -- kept as base pointer + index so the loads stay global_load.
struct ClsWin {
    const uint4 *cls16; int64_t blk0; int shift; uint4 w; int tag;
    __device__ __forceinline__ void init(const uint16_t *cls_buf, int64_t elem_off)
    {
        cls16 = (const uint4 *)cls_buf; blk0 = elem_off >> 3; shift = (int)(elem_off & 7); tag = -1; w = make_uint4(0, 0, 0, 0);
    }
    __device__ __forceinline__ uint32_t operator()(int i)
    {
        const int a = i + shift, t = a >> 3;
        if (t != tag) { w = cls16[blk0 + t]; tag = t; }
        // element a & 7 of the window: pick the 8-byte half, then one v_perm_b32 extracts the 16-bit class
        // (selector bytes 2k, 2k+1 of the pair; 0x0c = constant zero)
        const bool up = (a & 4) != 0;
        const uint32_t d0 = up ? w.z : w.x, d1 = up ? w.w : w.y;
        return __builtin_amdgcn_perm(d1, d0, 0x0c0c0100u + 0x0202u * (uint32_t)(a & 3));
    }
    // refill for position i if it lies outside the window (issued early by LexLane::step(); not waited for here)
    __device__ __forceinline__ void prefetch(int i)
    {
        const int t = (i + shift) >> 3;
        if (t != tag) { w = cls16[blk0 + t]; tag = t; }
    }
    __device__ __forceinline__ bool has(int i) const { return ((i + shift) >> 3) == tag; }
    // number of consecutive elements flagged LX_C_LOOP starting at position i, as far as the window shows (0 .. 8)
    __device__ __forceinline__ int run(int i)
    {
        const int a = i + shift, t = a >> 3;
        if (t != tag) { w = cls16[blk0 + t]; tag = t; }
        // bit 14 of the eight 16-bit elements = bit 6 of the odd bytes: two v_perm_b32 gather them into one byte per element, the
        // run ends at the first element whose bit is clear (14 instructions instead of 25 for the shift-and-mask form)
        const uint32_t c0 = ~__builtin_amdgcn_perm(w.y, w.x, 0x07050301u) & 0x40404040u;
        const uint32_t c1 = ~__builtin_amdgcn_perm(w.w, w.z, 0x07050301u) & 0x40404040u;
        const int e = a & 7;
        const unsigned long long c = (((unsigned long long)c1 << 32) | c0) >> (8 * e);
        return c ? (__builtin_ctzll(c) >> 3) : 8 - e;
    }
};

} // namespace bfa
namespace wv {
__device__ __forceinline__ int lane() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ unsigned long long ballot(bool b) { return __ballot(b); }
__device__ __forceinline__ bool any(bool b) { return __ballot(b) != 0ull; }
__device__ __forceinline__ void sync() { bfa::wave_handoff(); }
template <class T> __device__ __forceinline__ T shfl(T v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ uint32_t bcast(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ unsigned long long bcast(unsigned long long v, int src)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ int64_t bcast(int64_t v, int src) { return (int64_t)bcast((unsigned long long)v, src); }
template <class T> __device__ __forceinline__ T shfl_up(T v, int delta) { return __shfl_up(v, (unsigned)delta, 64); }
template <class T> __device__ __forceinline__ T shfl_down(T v, int delta) { return __shfl_down(v, (unsigned)delta, 64); }
// inclusive prefix sum over the wave in six data-parallel-primitive adds (row shifts inside the rows of 16 lanes, then the two row broadcasts)
__device__ __forceinline__ int incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
    return v;
}
// the smallest value of the wave: the same six steps with min (lanes without a source keep their own value)
__device__ __forceinline__ uint32_t min_all(uint32_t v)
{
    int x = (int)v;
#define BF_WV_MIN_STEP(ctrl, rows) { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(x, x, ctrl, rows, 0xf, false); x = (int)((uint32_t)x < o ? (uint32_t)x : o); }
    BF_WV_MIN_STEP(0x111, 0xf) BF_WV_MIN_STEP(0x112, 0xf) BF_WV_MIN_STEP(0x114, 0xf) BF_WV_MIN_STEP(0x118, 0xf) BF_WV_MIN_STEP(0x142, 0xa) BF_WV_MIN_STEP(0x143, 0xc)
#undef BF_WV_MIN_STEP
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ unsigned long long atomic_add(unsigned long long *p, unsigned long long v) { return atomicAdd(p, v); }
__device__ __forceinline__ void atomic_or(int *p, int v) { atomicOr(p, v); }
__device__ __forceinline__ void atomic_add_i32(int32_t *p, int32_t v) { atomicAdd(p, v); }
// v_perm_b32: byte k of the result = byte sel[k] of {hi, lo} (0..3 = lo, 4..7 = hi); v_alignbit_b32: ({hi, lo} >> sh) [31:0]
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
__device__ __forceinline__ void lds_or(uint32_t *p, uint32_t v) { atomicOr(p, v); }      // a word in LDS, from divergent lanes
__device__ __forceinline__ unsigned long long clock() { return __builtin_readcyclecounter(); }
// a word that other lanes of the wave update with atomics (executed in the L2): read past the CU's vector cache
__device__ __forceinline__ uint32_t load_l2(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void atomic_or_u32(uint32_t *p, uint32_t v) { atomicOr(p, v); }
__device__ __forceinline__ void atomic_max_u32(uint32_t *p, uint32_t v) { atomicMax(p, v); }
__device__ __forceinline__ void lds_add(uint32_t *p, uint32_t v) { atomicAdd(p, v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// a wave-uniform value the compiler may not trace back to where it came from (it then lives in a scalar register of its own)
__device__ __forceinline__ int own(int v) { asm volatile("" : "+s"(v)); return v; }
// The four values are in their registers from here on: ONE s_waitcnt for the loads that fetch them, in straight-line code.  Without it the compiler
// places a wait in front of the first use of each -- and when those uses are stores in conditional blocks of their own (k_wp_wave's retire pass),
// every wait also waits for the store before it to be acknowledged (vmcnt counts stores on gfx9): four round trips to L2 instead of one.
__device__ __forceinline__ void arrived(int32_t &a, int32_t &b, int32_t &c, int32_t &d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ uint32_t mbcnt(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
} // namespace wv
namespace bfa {

template <class K>
static int wp_blocks_per_cu(K kernel, int &cached)
{
    if (cached <= 0) {
        int q = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kernel, 256, 0) != hipSuccess || q <= 0) q = 2;
        (void)hipGetLastError();
        cached = q;
    }
    return cached;
}

} // namespace bfa
