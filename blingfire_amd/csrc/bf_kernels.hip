// bf_kernels.hip -- HIP kernels for gfx950 (MI355X, wave64).  Integer / indexing work: no MFMA.
//
//  k_prep_wp   wave per document: strict UTF-8 validate + decode, BOM skip, fused charmap+class
//              lookup, stream compaction of the class stream (ballot / prefix scan).  Streaming,
//              coalesced: the HBM-shaped stage.                (reference: FAStrUtf8ToArray
//              cl/src/FAUtf8Utils.cpp:233-270, FAUtf8ToInt :121-196, FANormalize cl/inc/FAUtils_cl.h:311-369,
//              FAIwMap_pack::GetNewIw cl/inc/FAIwMap_pack.h:55-110)
//  k_lex_wp_*  one document per lane: DFA lexer + WordPiece post-pass (bf_lex.h).  Latency/gather bound:
//              one 4-byte gather per DFA transition into the displacement-packed table (L2 resident).
//  k_scan_*    exclusive scan of per-document id counts -> id offsets.
//  k_compact   wave-cooperative gather of the per-document staging slots into one contiguous id array.
#include <hip/hip_runtime.h>
#include "bf_kernels.h"

namespace bfa {

#define BF_WAVE 64

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }

__device__ __forceinline__ uint32_t cpmap_get(const DevCpMap &m, int cp)
{
    return m.pages[(uint32_t)m.l1[cp >> 8] * 256u + (uint32_t)(cp & 255)];
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
// k_prep_wp
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep_wp(WpPrepParams p)
{
    __shared__ uint16_t ascii_cls[128];      // fused map of U+0000..U+007F; 0xFFFE = needs the general path
    if (threadIdx.x < 128) {
        const uint32_t v = cpmap_get(p.cpmap, (int)threadIdx.x);
        ascii_cls[threadIdx.x] = (v & 0x80000000u) ? (uint16_t)0xFFFE : (uint16_t)v;
    }
    __syncthreads();
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t d = wave0; d < p.b.ndocs; d += nwaves) {
        const int64_t b = p.b.doc_off[d];
        const int64_t n64 = p.b.doc_off[d + 1] - b;
        if (n64 <= 0 || n64 > 1000000000) { if (lane == 0) p.nchars[d] = 0; continue; }   // tokdll:1121
        const int n = (int)n64;
        const uint8_t *s = p.b.text + b;
        uint16_t *out = p.cls + b;
        int pos = 0;
        if (n >= 3 && s[0] == 0xEF && s[1] == 0xBB && s[2] == 0xBF) pos = 3;            // FAUtf8Utils.cpp:247-252
        const int bom = pos;
        int outc = 0; bool bad = false;
        while (pos < n) {
            // ---- fast path: up to 256 bytes, all ASCII, all 1:1
            {
                const int q = pos + lane * 4;
                int nb = n - q; nb = nb < 0 ? 0 : (nb > 4 ? 4 : nb);
                uint32_t w = 0;
                if (nb == 4) __builtin_memcpy(&w, s + q, 4);
                else for (int k = 0; k < nb; ++k) w |= (uint32_t)s[q + k] << (8 * k);
                const uint16_t c0 = ascii_cls[w & 0x7f], c1 = ascii_cls[(w >> 8) & 0x7f], c2 = ascii_cls[(w >> 16) & 0x7f], c3 = ascii_cls[(w >> 24) & 0x7f];
                bool slow = (w & 0x80808080u) != 0;
                slow |= (nb > 0 && c0 == 0xFFFE) | (nb > 1 && c1 == 0xFFFE) | (nb > 2 && c2 == 0xFFFE) | (nb > 3 && c3 == 0xFFFE);
                if (!__any(slow)) {
                    uint16_t *o = out + outc + lane * 4;
                    if (nb == 4) { const uint64_t pk = (uint64_t)c0 | ((uint64_t)c1 << 16) | ((uint64_t)c2 << 32) | ((uint64_t)c3 << 48); __builtin_memcpy(o, &pk, 8); }
                    else { if (nb > 0) o[0] = c0; if (nb > 1) o[1] = c1; if (nb > 2) o[2] = c2; }
                    const int adv = (n - pos) < 256 ? (n - pos) : 256;
                    outc += adv; pos += adv;
                    continue;
                }
            }
            // ---- general path: 64 bytes, one byte position per lane
            const int q = pos + lane;
            const bool in = q < n;
            uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
            if (in) b0 = s[q];
            if (q + 1 < n) b1 = s[q + 1];
            if (q + 2 < n) b2 = s[q + 2];
            if (q + 3 < n) b3 = s[q + 3];
            const bool cont = (b0 & 0xC0) == 0x80;
            bool start = in && !cont;
            bool err = false;
            int cp = (int)b0;
            if (in && cont) {
                // a continuation byte must be covered by a preceding lead (sequential decoder would
                // otherwise meet it at a character start and reject it: FAUtf8Utils.cpp:152-165)
                uint32_t p1 = (q - 1 >= bom) ? s[q - 1] : 0x80u, p2 = (q - 2 >= bom) ? s[q - 2] : 0x80u, p3 = (q - 3 >= bom) ? s[q - 3] : 0x80u;
                bool ok;
                if ((p1 & 0xC0) != 0x80) ok = (q - 1 >= bom) && p1 >= 0xC0;                 // any multi-byte lead covers +1
                else if ((p2 & 0xC0) != 0x80) ok = (q - 2 >= bom) && p2 >= 0xE0;            // 3- or 4-byte lead covers +2
                else if ((p3 & 0xC0) != 0x80) ok = (q - 3 >= bom) && p3 >= 0xF0;            // 4-byte lead covers +3
                else ok = false;
                err = !ok;                       // invalid leads (F8..FF) are rejected by their own lane
            } else if (start && b0 >= 0x80) {
                int len;
                if ((b0 & 0xE0) == 0xC0) { len = 2; cp = (int)(b0 & 0x1F); }
                else if ((b0 & 0xF0) == 0xE0) { len = 3; cp = (int)(b0 & 0x0F); }
                else if ((b0 & 0xF8) == 0xF0) { len = 4; cp = (int)(b0 & 0x07); }
                else { len = 1; err = true; }
                if (q + len > n) err = true;                                                 // truncated tail (:167-171)
                if (len >= 2) { if ((b1 & 0xC0) != 0x80) err = true; cp = (cp << 6) | (int)(b1 & 0x3F); }
                if (len >= 3) { if ((b2 & 0xC0) != 0x80) err = true; cp = (cp << 6) | (int)(b2 & 0x3F); }
                if (len >= 4) { if ((b3 & 0xC0) != 0x80) err = true; cp = (cp << 6) | (int)(b3 & 0x3F); }
                const int need = cp <= 0x7F ? 1 : cp <= 0x7FF ? 2 : cp <= 0xFFFF ? 3 : cp <= 0x10FFFF ? 4 : 0;
                if (need != len) err = true;                                                 // overlong / > U+10FFFF (:185-188)
                if ((cp & 0xFFFFF800) == 0xD800) err = true;                                 // surrogate (:190-193)
                if (err) cp = 0;
            }
            if (__any(err)) bad = true;
            uint32_t v = 0xFFFFu; int w = 0;
            if (start && !err) {
                v = cpmap_get(p.cpmap, cp);
                w = (v & 0x80000000u) ? (int)p.multi_pool[v & 0x7FFFFFFFu] : 1;
            }
            int idx, total;
            if (!p.has_multi) {
                const unsigned long long m = __ballot(w != 0);
                idx = outc + __popcll(m & lanemask_lt());
                total = __popcll(m);
            } else {
                const int inc = wave_incl_scan(w);
                idx = outc + inc - w;
                total = __shfl(inc, 63, 64);
            }
            if (w == 1 && !(v & 0x80000000u)) { if (idx < n) out[idx] = (uint16_t)v; }
            else if (w > 0) {
                const uint16_t *rec = p.multi_pool + (v & 0x7FFFFFFFu) + 1;
                for (int k = 0; k < w; ++k) if (idx + k < n) out[idx + k] = rec[k];
            }
            outc += total;
            pos += 64;
        }
        if (lane == 0) p.nchars[d] = (bad || outc > n) ? 0 : outc;     // tokdll:1151-1153,1185-1187
    }
}

void launch_prep_wp(const WpPrepParams &p, hipStream_t s)
{
    int64_t blocks = (p.b.ndocs + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_prep_wp, dim3((unsigned)blocks), dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------
// k_lex_wp variant 0: one document per lane, static assignment, the sequential program of
// bf_lex.h executed as-is (the compiler's reconvergence gives "walk until every lane's walk
// ends, then handle matches").
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lex_wp_v0(WpLexParams p)
{
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= p.b.ndocs) return;
    const int64_t b = p.b.doc_off[d];
    const int64_t nbytes = p.b.doc_off[d + 1] - b;
    const int n = p.nchars[d];
    const uint16_t *cls = p.cls + b;
    int cap = p.max_ids; if ((int64_t)cap > nbytes) cap = (int)nbytes;
    if (cap < 0) cap = 0;
    const int c = lex_doc(p.L, [cls](int i) -> uint32_t { return cls[i]; }, n, p.ids_tmp + b, cap, p.unk);
    p.counts[d] = c;
}

void launch_lex_wp(const WpLexParams &p, int variant, hipStream_t s)
{
    (void)variant;
    const int64_t blocks = (p.b.ndocs + 255) / 256;
    hipLaunchKernelGGL(k_lex_wp_v0, dim3((unsigned)blocks), dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------
// scan: counts[ndocs] (int32) -> id_off[ndocs+1] (int64), three small kernels
// ------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 4, SCAN_THREADS = 256, SCAN_TILE = SCAN_ITEMS * SCAN_THREADS;

int scan_nblocks(int64_t ndocs) { return (int)((ndocs + SCAN_TILE - 1) / SCAN_TILE); }

__device__ __forceinline__ long long block_excl_scan(long long v, long long *total, long long *sh /*[4]*/)
{
    // exclusive scan of one value per thread across a 256-thread block
    const int lane = lane_id(), wv = (int)(threadIdx.x >> 6);
    long long inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { long long t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    long long base = 0, tot = 0;
    for (int i = 0; i < 4; ++i) { if (i < wv) base += sh[i]; tot += sh[i]; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_block_sums(ScanParams p)
{
    __shared__ long long sh[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    long long v = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < p.ndocs) v += p.counts[base + k];
    long long tot; block_excl_scan(v, &tot, sh);
    if (threadIdx.x == 0) p.block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_top(ScanParams p)
{
    // single block: exclusive scan of block_sums in place, total -> id_off[ndocs]
    __shared__ long long sh[4];
    long long carry = 0;
    for (int i0 = 0; i0 < p.nblocks; i0 += SCAN_THREADS) {
        const int i = i0 + (int)threadIdx.x;
        const long long v = i < p.nblocks ? p.block_sums[i] : 0;
        long long tot; const long long ex = block_excl_scan(v, &tot, sh);
        if (i < p.nblocks) p.block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) p.id_off[p.ndocs] = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(ScanParams p)
{
    __shared__ long long sh[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int c[SCAN_ITEMS]; long long v = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) { c[k] = base + k < p.ndocs ? p.counts[base + k] : 0; v += c[k]; }
    long long tot; long long ex = block_excl_scan(v, &tot, sh) + p.block_sums[blockIdx.x];
    for (int k = 0; k < SCAN_ITEMS; ++k) { if (base + k < p.ndocs) p.id_off[base + k] = ex; ex += c[k]; }
}

void launch_scan(const ScanParams &p, hipStream_t s)
{
    if (p.nblocks > 0) {
        hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)p.nblocks), dim3(SCAN_THREADS), 0, s, p);
    }
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(SCAN_THREADS), 0, s, p);
    if (p.nblocks > 0) {
        hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)p.nblocks), dim3(SCAN_THREADS), 0, s, p);
    }
}

// ------------------------------------------------------------------------------------------
// k_compact: wave per document, coalesced copy staging slot -> contiguous output
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact(CompactParams p)
{
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t d = wave0; d < p.b.ndocs; d += nwaves) {
        const int c = p.counts[d];
        const int32_t *src = p.ids_tmp + p.b.doc_off[d];
        const int64_t o = p.id_off[d];
        for (int i = lane; i < c; i += 64) {
            if (o + i < p.ids_cap) p.ids_out[o + i] = src[i];
            else if (i == lane) atomicOr(p.status, 1);
        }
    }
}

void launch_compact(const CompactParams &p, hipStream_t s)
{
    int64_t blocks = (p.b.ndocs + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_compact, dim3((unsigned)blocks), dim3(256), 0, s, p);
}

} // namespace bfa
