// bf_kernels.hip -- HIP kernels for gfx950 (MI355X, wave64), WordPiece branch.  Integer / indexing work: no MFMA.  (The _sp branch, the dictionary
// look-ups, scan, compaction and the string-assembly kernels: bf_kernels_sp.hip.)
//
//  k_wp_pre, k_wp_flat, k_wp_units, k_wp_hardlist, k_wp_count, k_wp_merge   the flat program (bf_flat.h, bf_flat_body.h): the batch as one byte
//              stream, whole words answered by a hash table keyed by the word, the others walked by a kernel of their own, ids (and byte
//              offsets) merged into place through LDS.  The path of flat-form lexers (every BERT model) for batches that fill the device.
//  k_wp_wave   the wave program (bf_wave.h, bf_wave_body.h): decode + lexer + post-pass of a unit-form lexer in one kernel, document by document:
//              small batches, the documents the flat program hands back (LIST instances), the offsets instance.
//  k_prep_wp_flat / _docs   prologue of the lane kernels in two passes: byte -> class translation of the whole text buffer (HBM
//              streaming) + per-document check; documents with multi-byte characters are redone by a wave (strict UTF-8
//              decode, BOM skip, fused charmap+class lookup, wave scan).  k_prep_wp: the one-pass form (offsets API).
//              (reference: FAStrUtf8ToArray cl/src/FAUtf8Utils.cpp:233-270, FAUtf8ToInt :121-196, FANormalize
//              cl/inc/FAUtils_cl.h:311-369, FAIwMap_pack::GetNewIw cl/inc/FAIwMap_pack.h:55-110)
//  k_lex_wp_*  one document per lane: DFA lexer + WordPiece post-pass (bf_lex.h): TextToWords / TextToSentences and lexers outside the unit
//              form.  One 8-byte gather per DFA transition into the displacement-packed table.
#include "bf_kernels_common.h"

namespace bfa {

// ------------------------------------------------------------------------------------------
// k_prep_wp
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void prep_ascii_table(const WpPrepParams &p, uint16_t *ascii_cls)
{
    if (threadIdx.x < 128) {
        const uint32_t v = cpmap_get(p.cpmap, (int)threadIdx.x);
        ascii_cls[threadIdx.x] = (v & 0x80000000u) ? (uint16_t)0xFFFE : (uint16_t)v;
    }
    __syncthreads();
}

// One document, the whole wave, 512 bytes per iteration: every lane owns 8 consecutive bytes (one 8-byte load) and sees
// the 3 bytes on either side through shuffles, so a strict UTF-8 decode of each byte position needs no further loads.
// Per byte position q (same rules as the sequential decoder, FAUtf8Utils.cpp:121-196,233-270):
//   continuation byte  -> no character; it must be covered by a lead at q-1 / q-2 / q-3 (else the decoder would meet
//                         it at a character start and reject it, :152-165)
//   lead byte          -> length from the lead, continuation bytes checked, truncated tail (:167-171), overlong and
//                         > U+10FFFF (:185-188), surrogates (:190-193)
// then the fused charmap+class map gives 0 / 1 / 2..10 stream elements per character, compacted with a wave scan.
// The pieces [pos0, pos1) of a document (pos0 a multiple of 512), stream position of the first one's first element = outc.  bom < 0:
// the range starts the document and finds the byte order mark itself; COUNT: nothing is written, outc counts the elements.
template <bool COUNT>
__device__ __forceinline__ void prep_wp_pieces(const WpPrepParams &p, int64_t b, int n, int lane, const uint16_t *ascii_cls, uint16_t *stage /* 512 elements of LDS, this wave's */,
                                               int pos0, int pos1, int bom, int &outc, bool &err_any)
{
    const uint8_t *s = p.b.text + b;
    uint16_t *out = p.cls + b;
    uint32_t carry = 0;
    if (pos0 > 0) carry = (uint32_t)s[pos0 - 3] | ((uint32_t)s[pos0 - 2] << 8) | ((uint32_t)s[pos0 - 1] << 16);
    for (int pos = pos0; pos < pos1; pos += 512) {
        const int q0 = pos + lane * 8;
        uint64_t own = 0;
        int nb = n - q0; nb = nb < 0 ? 0 : (nb > 8 ? 8 : nb);
        if (nb == 8) __builtin_memcpy(&own, s + q0, 8);
        else for (int k = 0; k < nb; ++k) own |= (uint64_t)s[q0 + k] << (8 * k);
        uint32_t nxt = __shfl_down((uint32_t)own, 1, 64);
        if (lane == 63) { nxt = 0; for (int k = 0; k < 3; ++k) if (q0 + 8 + k < n) nxt |= (uint32_t)s[q0 + 8 + k] << (8 * k); }
        const uint32_t tail3 = (uint32_t)(own >> 40);                     // own bytes 5, 6, 7
        uint32_t prv = __shfl_up(tail3, 1, 64);
        if (lane == 0) prv = carry;
        carry = __shfl(tail3, 63, 64);
        if (pos == 0 && bom < 0) {                                        // FAUtf8Utils.cpp:247-252
            const int has_bom = (n >= 3 && ((uint32_t)own & 0xFFFFFFu) == 0xBFBBEFu) ? 3 : 0;
            bom = __shfl(has_bom, 0, 64);
        }
        // X[i] = byte at q0 - 3 + i, i = 0..13
        uint32_t X[14];
        X[0] = prv & 0xFF; X[1] = (prv >> 8) & 0xFF; X[2] = (prv >> 16) & 0xFF;
#pragma unroll
        for (int k = 0; k < 8; ++k) X[3 + k] = (uint32_t)(own >> (8 * k)) & 0xFF;
        X[11] = nxt & 0xFF; X[12] = (nxt >> 8) & 0xFF; X[13] = (nxt >> 16) & 0xFF;
        uint32_t v[8]; int w[8]; int cnt = 0; bool err = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = q0 + k;
            const bool in = q >= bom && q < n;
            const uint32_t b0 = X[3 + k], b1 = X[4 + k], b2 = X[5 + k], b3 = X[6 + k];
            const bool cont = (b0 & 0xC0) == 0x80;
            const bool start = in && !cont;
            uint32_t vv = LX_CLS_NONE; int ww = 0;
            if (in && cont) {
                const uint32_t p1 = X[2 + k], p2 = X[1 + k], p3 = X[k];
                bool ok;
                if ((p1 & 0xC0) != 0x80 || q - 1 < bom) ok = (q - 1 >= bom) && p1 >= 0xC0;                 // any multi-byte lead covers +1
                else if ((p2 & 0xC0) != 0x80 || q - 2 < bom) ok = (q - 2 >= bom) && p2 >= 0xE0;            // 3- or 4-byte lead covers +2
                else if ((p3 & 0xC0) != 0x80 || q - 3 < bom) ok = (q - 3 >= bom) && p3 >= 0xF0;            // 4-byte lead covers +3
                else ok = false;
                err |= !ok;                       // invalid leads (F8..FF) are rejected at their own position
            } else if (start) {
                if (b0 < 0x80) {
                    vv = ascii_cls[b0];
                    if (vv == 0xFFFEu) vv = cpmap_get(p.cpmap, (int)b0);
                } else {
                    int len, cp; bool e = false;
                    if ((b0 & 0xE0) == 0xC0) { len = 2; cp = (int)(b0 & 0x1F); }
                    else if ((b0 & 0xF0) == 0xE0) { len = 3; cp = (int)(b0 & 0x0F); }
                    else if ((b0 & 0xF8) == 0xF0) { len = 4; cp = (int)(b0 & 0x07); }
                    else { len = 1; cp = 0; e = true; }
                    if (q + len > n) e = true;                                                 // truncated tail (:167-171)
                    if (len >= 2) { if ((b1 & 0xC0) != 0x80) e = true; cp = (cp << 6) | (int)(b1 & 0x3F); }
                    if (len >= 3) { if ((b2 & 0xC0) != 0x80) e = true; cp = (cp << 6) | (int)(b2 & 0x3F); }
                    if (len >= 4) { if ((b3 & 0xC0) != 0x80) e = true; cp = (cp << 6) | (int)(b3 & 0x3F); }
                    const int need = cp <= 0x7F ? 1 : cp <= 0x7FF ? 2 : cp <= 0xFFFF ? 3 : cp <= 0x10FFFF ? 4 : 0;
                    if (need != len) e = true;                                                 // overlong / > U+10FFFF (:185-188)
                    if ((cp & 0xFFFFF800) == 0xD800) e = true;                                 // surrogate (:190-193)
                    err |= e;
                    if (!e) vv = cpmap_get(p.cpmap, cp);
                    else ww = -1;
                }
                if (ww == 0) ww = (vv & 0x80000000u) ? (int)p.multi_pool[vv & 0x7FFFFFFFu] : 1;
                else ww = 0;
            }
            v[k] = vv; w[k] = ww; cnt += ww;
        }
        err_any |= err;
        const int inc = wave_incl_scan(cnt);
        int idx = outc + inc - cnt;
        const int total = __shfl(inc, 63, 64);
        if constexpr (COUNT) { outc += total; continue; }
        bool multi = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) multi |= w[k] > 0 && (v[k] & 0x80000000u) != 0;
        if (!p.src_off && !__any(multi)) {
            // the common case (every character gives one element): the window's elements are compacted in LDS and leave as
            // whole 128-byte rows (a lane's own elements are 16 bytes apart in the stream: direct 2-byte stores would be
            // 64 partial writes per instruction)
            int li = inc - cnt;
#pragma unroll
            for (int k = 0; k < 8; ++k) { if (w[k] == 1) stage[li] = (uint16_t)v[k]; li += w[k]; }
            wave_handoff();
            for (int t = lane; t < total; t += 64) if (outc + t < n) out[outc + t] = stage[t];
            wave_handoff();                                               // the next window overwrites the stage
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (w[k] == 1 && !(v[k] & 0x80000000u)) { if (idx < n) { out[idx] = (uint16_t)v[k]; if (p.src_off) p.src_off[b + idx] = q0 + k; } }
                else if (w[k] > 0) {
                    const uint16_t *rec = p.multi_pool + (v[k] & 0x7FFFFFFFu) + 1;
                    for (int t = 0; t < w[k]; ++t) if (idx + t < n) { out[idx + t] = rec[t]; if (p.src_off) p.src_off[b + idx + t] = q0 + k; }
                }
                idx += w[k];
            }
        }
        outc += total;
    }
}

// documents of more than this many bytes are decoded by sixteen waves (k_prep_wp_long) when the caller gives a list to put them on
constexpr int PREP_LONG_BYTES = 2048, PREP_LONG_WAVES = 16;

__device__ __forceinline__ void prep_wp_doc(const WpPrepParams &p, int64_t d, int64_t b, int64_t n64, int lane, const uint16_t *ascii_cls, uint16_t *stage /* 512 elements of LDS, this wave's */)
{
    if (n64 <= 0 || n64 > 1000000000) { if (lane == 0) p.nchars[d] = 0; return; }   // tokdll:1121
    if (b < 0 || b + n64 > p.b.total_bytes) {                                         // outside the caller's buffer: empty + status
        if (lane == 0) { p.nchars[d] = 0; atomicOr(p.b.status, BF_STATUS_BAD_OFFSETS); }
        return;
    }
    const int n = (int)n64;
    if (p.long_list && n > PREP_LONG_BYTES) {                                         // a long document: listed for k_prep_wp_long
        if (lane == 0) {
            const unsigned slot = atomicAdd(p.long_count, 1u);
            if ((int64_t)slot < p.long_cap) p.long_list[slot] = d;
        }
        return;
    }
    int outc = 0; bool err_any = false;
    prep_wp_pieces<false>(p, b, n, lane, ascii_cls, stage, 0, n, -1, outc, err_any);
    const bool bad = __any(err_any);
    if (lane == 0) p.nchars[d] = (bad || outc > n) ? 0 : outc;     // tokdll:1151-1153,1185-1187
}

// wave per document, every document through prep_wp_doc (used when the source offsets are wanted or the text is unaligned)
__global__ __launch_bounds__(256) void k_prep_wp(WpPrepParams p)
{
    __shared__ uint16_t ascii_cls[128];      // fused map of U+0000..U+007F; 0xFFFE = needs the general path
    __shared__ uint16_t stage_all[4 * 512];
    prep_ascii_table(p, ascii_cls);
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave_in_block();
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t d = wave0; d < p.b.ndocs; d += nwaves) { const int64_t b = p.b.doc_off[d]; prep_wp_doc(p, d, b, p.b.doc_off[d + 1] - b, lane, ascii_cls, stage_all + wave_in_block() * 512); }
}

// The listed documents, one workgroup of sixteen waves each: every wave owns a contiguous run of 512-byte pieces, counts its elements
// (the decode without the stores), the counts are summed in front of each wave, and the wave decodes its pieces again at that
// stream position.  (One wave took 7.6 ms for a 1 MB document: 3.7 us per piece, one after the other.)
__global__ __launch_bounds__(PREP_LONG_WAVES * 64) void k_prep_wp_long(WpPrepParams p)
{
    __shared__ uint16_t ascii_cls[128];
    __shared__ uint16_t stage_all[PREP_LONG_WAVES * 512];
    __shared__ int s_cnt[PREP_LONG_WAVES], s_err[PREP_LONG_WAVES];
    prep_ascii_table(p, ascii_cls);
    const int lane = lane_id(), wave = wave_in_block();
    int64_t nlist = (int64_t)*p.long_count; if (nlist > p.long_cap) nlist = p.long_cap;
    for (int64_t j = blockIdx.x; j < nlist; j += gridDim.x) {
        const int64_t d = p.long_list[j];
        const int64_t b = p.b.doc_off[d];
        const int n = (int)(p.b.doc_off[d + 1] - b);                                  // (checked by prep_wp_doc before the document was listed)
        const uint8_t *s = p.b.text + b;
        const int bom = (s[0] == 0xEF && s[1] == 0xBB && s[2] == 0xBF) ? 3 : 0;       // n > PREP_LONG_BYTES
        const int pieces = (n + 511) >> 9, per = (pieces + PREP_LONG_WAVES - 1) / PREP_LONG_WAVES;
        int pos0 = wave * per * 512, pos1 = pos0 + per * 512;
        if (pos0 > n) pos0 = n;
        if (pos1 > n) pos1 = n;
        int cnt = 0; bool err = false;
        prep_wp_pieces<true>(p, b, n, lane, ascii_cls, stage_all + wave * 512, pos0, pos1, bom, cnt, err);
        const bool err_wave = __any(err);
        if (lane == 0) { s_cnt[wave] = cnt; s_err[wave] = err_wave ? 1 : 0; }
        __syncthreads();
        int outc = 0, total = 0, bad = 0;
        for (int w = 0; w < PREP_LONG_WAVES; ++w) { if (w < wave) outc += s_cnt[w]; total += s_cnt[w]; bad |= s_err[w]; }
        bool err2 = false;
        if (!bad && total <= n) prep_wp_pieces<false>(p, b, n, lane, ascii_cls, stage_all + wave * 512, pos0, pos1, bom, outc, err2);
        if (wave == 0 && lane == 0) p.nchars[d] = (bad || total > n) ? 0 : total;     // tokdll:1151-1153,1185-1187
        __syncthreads();                                                              // (s_cnt / s_err are rewritten by the next document)
    }
}

// Two-pass form for the common case (no offsets wanted).  An all-ASCII document whose characters all map 1:1 has
// stream position == byte position, so its class stream is a position-preserving byte -> class translation that
// needs no document boundaries at all:
//   pass 1 (k_prep_wp_flat)  streams the WHOLE text buffer, 16 bytes per lane, through the 128-entry LDS table into
//                            cls[] and records one "dirty" bit per 16-byte chunk (a byte >= 0x80, or an ASCII
//                            character that the charmap deletes / expands);
//   pass 2 (k_prep_wp_docs)  lane per document: no dirty bit in the document's chunks -> nchars = nbytes, done;
//                            otherwise the wave redoes that document with prep_wp_doc (which overwrites only the
//                            document's own range).  A neighbour's dirty byte in a shared boundary chunk only costs
//                            a redundant redo.
__global__ __launch_bounds__(256) void k_prep_wp_flat(WpPrepParams p, int64_t total_bytes, unsigned long long *flags)
{
    __shared__ uint16_t ascii_cls[128];
    prep_ascii_table(p, ascii_cls);
    const int lane = lane_id();
    const int64_t nchunks = (total_bytes + 15) >> 4;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t c0 = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63u); c0 < nchunks; c0 += stride) {
        const int64_t c = c0 + lane;
        bool dirty = false;
        if (c < nchunks) {
            const int64_t q = c << 4;
            uint32_t w[4] = {0, 0, 0, 0};
            if (q + 16 <= total_bytes) { const uint4 v = *(const uint4 *)(p.b.text + q); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
            else for (int k = 0; q + k < total_bytes; ++k) w[k >> 2] |= (uint32_t)p.b.text[q + k] << (8 * (k & 3));
            uint32_t o[8]; uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t c0_ = ascii_cls[w[k] & 0x7f], c1_ = ascii_cls[(w[k] >> 8) & 0x7f], c2_ = ascii_cls[(w[k] >> 16) & 0x7f], c3_ = ascii_cls[(w[k] >> 24) & 0x7f];
                o[2 * k] = c0_ | (c1_ << 16); o[2 * k + 1] = c2_ | (c3_ << 16);
                acc |= c0_ | c1_ | c2_ | c3_;
            }
            dirty = (((w[0] | w[1] | w[2] | w[3]) & 0x80808080u) != 0) | ((acc & 0x8000u) != 0);     // 0xFFFE is the only entry with bit 15
            uint4 *dst = (uint4 *)(p.cls + q);
            dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
            dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
        }
        const unsigned long long m = __ballot(dirty);
        if (lane == 0) flags[c0 >> 6] = m;
    }
}

__global__ __launch_bounds__(256) void k_prep_wp_docs(WpPrepParams p, const unsigned long long *flags)
{
    __shared__ uint16_t ascii_cls[128];
    __shared__ uint16_t stage_all[4 * 512];
    prep_ascii_table(p, ascii_cls);
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave_in_block();
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t d0 = wave0 * 64; d0 < p.b.ndocs; d0 += nwaves * 64) {
        const int64_t d = d0 + lane;
        bool redo = false;
        int64_t b = 0, n64 = 0;
        if (d < p.b.ndocs) {
            b = p.b.doc_off[d];
            n64 = p.b.doc_off[d + 1] - b;
            if (n64 <= 0 || n64 > 1000000000) p.nchars[d] = 0;                                   // tokdll:1121
            else if (b < 0 || b + n64 > p.b.total_bytes) { p.nchars[d] = 0; atomicOr(p.b.status, BF_STATUS_BAD_OFFSETS); }
            else {
                const int64_t cf = b >> 4, cl = (b + n64 - 1) >> 4;
                bool any = false;
                for (int64_t wi = cf >> 6; wi <= (cl >> 6) && !any; ++wi) {
                    const int lo = wi == (cf >> 6) ? (int)(cf & 63) : 0, hi = wi == (cl >> 6) ? (int)(cl & 63) : 63;
                    const unsigned long long mask = (hi == 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull);
                    any = (flags[wi] & mask) != 0;
                }
                if (any) redo = true; else p.nchars[d] = (int)n64;
            }
        }
        unsigned long long m = __ballot(redo);
        while (m) {
            const int k = __ffsll((long long)m) - 1; m &= m - 1;
            const int64_t bk = __shfl((long long)b, k, 64), nk = __shfl((long long)n64, k, 64);
            prep_wp_doc(p, d0 + k, bk, nk, lane, ascii_cls, stage_all + wave_in_block() * 512);
        }
    }
}

static void launch_prep_wp_long(const WpPrepParams &p, hipStream_t s)
{
    if (!p.long_list) return;
    int64_t nb = p.long_cap < device_cus() ? p.long_cap : device_cus();       // (the number of listed documents is on the device: workgroups without one leave at once)
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_prep_wp_long, dim3((unsigned)nb), dim3(PREP_LONG_WAVES * 64), 0, s, p);
}

void launch_prep_wp(const WpPrepParams &p, int64_t total_bytes, unsigned long long *flags, hipStream_t s)
{
    if (flags && !p.src_off && ((uintptr_t)p.b.text & 15) == 0 && ((uintptr_t)p.cls & 31) == 0 && total_bytes > 0) {
        const int64_t nchunks = (total_bytes + 15) >> 4;
        int64_t b1 = (nchunks + 255) / 256; if (b1 > device_cus() * 16) b1 = device_cus() * 16;
        hipLaunchKernelGGL(k_prep_wp_flat, dim3((unsigned)b1), dim3(256), 0, s, p, total_bytes, flags);
        int64_t b2 = (p.b.ndocs + 255) / 256; if (b2 > device_cus() * 16) b2 = device_cus() * 16; if (b2 < 1) b2 = 1;
        hipLaunchKernelGGL(k_prep_wp_docs, dim3((unsigned)b2), dim3(256), 0, s, p, (const unsigned long long *)flags);
        launch_prep_wp_long(p, s);
        return;
    }
    int64_t blocks = (p.b.ndocs + 3) / 4;
    if (blocks > device_cus() * 16) blocks = device_cus() * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_prep_wp, dim3((unsigned)blocks), dim3(256), 0, s, p);
    launch_prep_wp_long(p, s);
}


// The same window over 32 bytes = 16 elements (two aligned 16-byte loads): half the refills, and a word that is read twice (top
// level, then the vocabulary function) usually stays inside it.  Costs four more VGPRs; used by the headline instance only.
struct ClsWin32 {
    const uint4 *cls16; int64_t blk0; int shift; uint4 w0, w1; int tag;
    __device__ __forceinline__ void init(const uint16_t *cls_buf, int64_t elem_off)
    {
        cls16 = (const uint4 *)cls_buf; blk0 = (elem_off >> 4) * 2; shift = (int)(elem_off & 15); tag = -1; w0 = w1 = make_uint4(0, 0, 0, 0);
    }
    __device__ __forceinline__ void load(int t) { const uint4 *q = cls16 + blk0 + 2 * (int64_t)t; w0 = q[0]; w1 = q[1]; tag = t; }
    __device__ __forceinline__ uint32_t operator()(int i)
    {
        const int a = i + shift, t = a >> 4;
        if (t != tag) load(t);
        const bool hi = (a & 8) != 0, up = (a & 4) != 0;
        const uint32_t x = hi ? w1.x : w0.x, y = hi ? w1.y : w0.y, z = hi ? w1.z : w0.z, w = hi ? w1.w : w0.w;
        const uint32_t d0 = up ? z : x, d1 = up ? w : y;
        return __builtin_amdgcn_perm(d1, d0, 0x0c0c0100u + 0x0202u * (uint32_t)(a & 3));
    }
    __device__ __forceinline__ void prefetch(int i)
    {
        const int t = (i + shift) >> 4;
        if (t != tag) load(t);
    }
    __device__ __forceinline__ bool has(int i) const { return ((i + shift) >> 4) == tag; }
    // number of consecutive elements flagged LX_C_LOOP starting at position i, as far as the window shows (0 .. 16)
    __device__ __forceinline__ int run(int i)
    {
        const int a = i + shift, t = a >> 4;
        if (t != tag) load(t);
        const unsigned long long lo = ((unsigned long long)(~__builtin_amdgcn_perm(w0.w, w0.z, 0x07050301u) & 0x40404040u) << 32) |
                                      (~__builtin_amdgcn_perm(w0.y, w0.x, 0x07050301u) & 0x40404040u);
        const unsigned long long hi = ((unsigned long long)(~__builtin_amdgcn_perm(w1.w, w1.z, 0x07050301u) & 0x40404040u) << 32) |
                                      (~__builtin_amdgcn_perm(w1.y, w1.x, 0x07050301u) & 0x40404040u);
        const int e = a & 15;
        if (e < 8) {
            const unsigned long long c = lo >> (8 * e);
            if (c) return __builtin_ctzll(c) >> 3;
            return (8 - e) + (hi ? (__builtin_ctzll(hi) >> 3) : 8);
        }
        const unsigned long long c = hi >> (8 * (e - 8));
        return c ? (__builtin_ctzll(c) >> 3) : 16 - e;
    }
};

// saved frames in LDS, structure-of-arrays (bank = lane): word (d, field) of lane t at [(d*12 + field) * nthreads + t]
struct FramesLds {
    int32_t *lds; int nthreads;
    __device__ __forceinline__ void save(int d, const LexFrame &f)
    {
        int32_t *q = lds + (size_t)d * LEX_FRAME_WORDS * nthreads + threadIdx.x;
        q[0] = (int32_t)f.ini; q[1 * nthreads] = f.off; q[2 * nthreads] = f.n; q[3 * nthreads] = f.from; q[4 * nthreads] = f.once;
        q[5 * nthreads] = f.a_idx; q[6 * nthreads] = f.a_end; q[7 * nthreads] = f.to2; q[8 * nthreads] = f.fn_once;
        q[9 * nthreads] = f.fp_r; q[10 * nthreads] = f.fn_from; q[11 * nthreads] = f.emit_mark;
    }
    __device__ __forceinline__ void load(int d, LexFrame &f) const
    {
        const int32_t *q = lds + (size_t)d * LEX_FRAME_WORDS * nthreads + threadIdx.x;
        f.ini = (uint32_t)q[0]; f.off = q[1 * nthreads]; f.n = q[2 * nthreads]; f.from = q[3 * nthreads]; f.once = q[4 * nthreads];
        f.a_idx = q[5 * nthreads]; f.a_end = q[6 * nthreads]; f.to2 = q[7 * nthreads]; f.fn_once = q[8 * nthreads];
        f.fp_r = q[9 * nthreads]; f.fn_from = q[10 * nthreads]; f.emit_mark = q[11 * nthreads];
    }
};

// id chunk buffer in LDS (8 ids per lane, structure-of-arrays): ids leave the lane as whole 32-byte sectors
struct IdOutLds {
    int32_t *slot; int32_t *buf; int nthreads; int cb;     // buf = &lds_idbuf[threadIdx.x]; word i at buf[i * nthreads]
    int32_t *spans;                                        // optional (offsets API): global, same indexing as the slot, 2 ints per id
    __device__ __forceinline__ void init(int32_t *s, int32_t *sp = nullptr) { slot = s; cb = -1; spans = sp; }
    __device__ __forceinline__ void span(int k, int from, int to) { if (spans) { spans[2 * k] = from; spans[2 * k + 1] = to; } }
    __device__ __forceinline__ void flush()
    {
        int4 *q = (int4 *)(slot + (int64_t)cb * 8);
        q[0] = make_int4(buf[0], buf[nthreads], buf[2 * nthreads], buf[3 * nthreads]);
        q[1] = make_int4(buf[4 * nthreads], buf[5 * nthreads], buf[6 * nthreads], buf[7 * nthreads]);
    }
    __device__ __forceinline__ void put(int k, int32_t v)
    {
        const int c = k >> 3;
        if (c != cb) {
            if (c > cb) { if (cb >= 0) flush(); }
            else {   // rewind into an earlier chunk (UNK after a long run of tentative sub-tokens): reload it
                const int4 *q = (const int4 *)(slot + (int64_t)c * 8);
                const int4 a = q[0], b = q[1];
                buf[0] = a.x; buf[nthreads] = a.y; buf[2 * nthreads] = a.z; buf[3 * nthreads] = a.w;
                buf[4 * nthreads] = b.x; buf[5 * nthreads] = b.y; buf[6 * nthreads] = b.z; buf[7 * nthreads] = b.w;
            }
            cb = c;
        }
        buf[(k & 7) * nthreads] = v;
    }
    __device__ __forceinline__ void finish(int count) { if (cb >= 0 && cb * 8 < count) flush(); }
};

// Divergence-aware driver.  Lanes are persistent and pull documents from a global
// counter; every loop iteration a lane in WALK mode makes exactly one DFA transition, while the heavier
// "event" code (match handling, calls/returns, next start position) and the document fetch run only when
// enough lanes of the wave are waiting for them (ballot vote), so that they execute with most lanes active.
// PLAIN: ids only (no TextToWords mode, no offsets): `words` and the span pointer become compile-time constants
template <int THREADS, class WIN, bool HAS_ANY, int UNROLL, bool STATS, bool TLDS = false, bool TWO = false, bool PLAIN = false>
__device__ __forceinline__ void lex_wp_flat_body(const WpLexParams &p)
{
    extern __shared__ int32_t lex_lds[];
    enum { M_NEED = 0, M_WALK = 1, M_EVENT = 2, M_EXIT = 3 };
    typedef typename std::conditional<TLDS, TabLds, TabDirect>::type TAB;
    WIN cls_at; cls_at.init(p.cls, 0);
    IdOutLds out; out.buf = lex_lds + (size_t)p.L.max_frames * LEX_FRAME_WORDS * THREADS + threadIdx.x; out.nthreads = THREADS;
    out.init(p.ids_tmp);
    FramesLds frames{lex_lds, THREADS};
    // the (tiny) action pool is staged in LDS so that match handling touches no global memory
    LexTables L = p.L;
    TAB tab;
    {
        int32_t *acts_lds = lex_lds + ((size_t)p.L.max_frames * LEX_FRAME_WORDS + 8) * THREADS;
        for (int i = threadIdx.x; i < p.acts_n; i += THREADS) acts_lds[i] = p.L.acts[i];
        if constexpr (TLDS) {
            // the whole transition table, staged once per workgroup (8-byte aligned behind the action pool)
            uint64_t *tab_lds = (uint64_t *)(acts_lds + ((p.acts_n + 1) & ~1));
            for (int i = threadIdx.x; i < p.table_n; i += THREADS) tab_lds[i] = p.L.T[i];
            tab.lds = tab_lds; tab.n = (uint32_t)p.table_n;
        } else tab.T = p.L.T;
        __syncthreads();
        L.acts = acts_lds;          // unconditionally LDS: the loads compile to ds_read, not flat_load
    }
    LexLane<WIN, IdOutLds, FramesLds, HAS_ANY, TAB> lane(L, cls_at, out, frames, tab);
    lane.init(0, 0, 0);
    int mode = M_NEED;
    int64_t doc = -1;
    const int ev_thresh = p.ev_thresh, fetch_thresh = p.fetch_thresh;
    // instrumentation (STATS builds only: BF_LEX_STATS=1): trips, useful lane-steps, event / fetch rounds, cycles per phase
    unsigned long long st_trips = 0, st_walk_lanes = 0, st_ev_rounds = 0, st_ev_lanes = 0, st_fetch_rounds = 0, st_need_lanes = 0;
    unsigned long long tk_walk = 0, tk_event = 0, tk_fetch = 0, tk0 = 0;
    for (;;) {
        if (STATS) tk0 = __builtin_readcyclecounter();
        // ---- walk: every walking lane makes UNROLL DFA transitions per trip, until enough lanes have a finished
        //      walk (or nobody walks any more).  `walk` lives in a scalar lane mask: the loop control costs no VALU.
        bool walk = mode == M_WALK;
        const unsigned long long m_entry = __ballot(walk);
        unsigned long long m_walk = m_entry;
        while (m_walk != 0 && __popcll(m_entry & ~m_walk) < ev_thresh) {
            if (STATS) { st_trips += UNROLL; st_walk_lanes += UNROLL * __popcll(m_walk); st_need_lanes += UNROLL * __popcll(__ballot(mode == M_NEED)); }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { if (walk) walk = lane.step_r(); }
            m_walk = __ballot(walk);
        }
        if (mode == M_WALK && !walk) mode = M_EVENT;
        const unsigned long long m_event = m_entry & ~m_walk;
        if (STATS) { st_ev_rounds += 1; st_ev_lanes += __popcll(m_event); const unsigned long long t1 = __builtin_readcyclecounter(); tk_walk += t1 - tk0; tk0 = t1; }
        // ---- events: match handling, calls / returns, next start position
        if (mode == M_EVENT) {
            bool more;
            if constexpr (TWO) { lane.after_walk2(); more = lane.prepare2(); }      // two-level lexers (every WordPiece model): no frame stack
            else { lane.after_walk(); more = lane.prepare(); }
            if (more) mode = M_WALK;
            else { p.counts[doc] = lane.finish(); mode = M_NEED; }
        }
        if (STATS) { const unsigned long long t1 = __builtin_readcyclecounter(); tk_event += t1 - tk0; tk0 = t1; }
        // ---- fetch documents for idle lanes
        const unsigned long long m_need = __ballot(mode == M_NEED);
        if (m_need) {
            const unsigned long long m_busy = __ballot(mode == M_WALK || mode == M_EVENT);
            if (__popcll(m_need) >= fetch_thresh || m_busy == 0) {
                if (STATS) st_fetch_rounds += 1;
                if (mode == M_NEED) {
                    const int cnt = __popcll(m_need);
                    const int leader = __ffsll((long long)m_need) - 1;
                    unsigned long long base = 0;
                    if (lane_id() == leader) base = atomicAdd(p.next_doc, (unsigned long long)cnt);
                    base = __shfl(base, leader, 64);
                    doc = (int64_t)base + __popcll(m_need & lanemask_lt());
                    if (doc >= p.b.ndocs) mode = M_EXIT;
                    else {
                        const int64_t b = p.b.doc_off[doc];
                        const int64_t nbytes = p.b.doc_off[doc + 1] - b;
                        int n = p.nchars[doc];
                        int cap = p.max_ids; if ((int64_t)cap > nbytes) cap = (int)nbytes;
                        if (cap < 0) cap = 0;
                        // words modes: a long document that k_lex_long_list took (its count is the mark) belongs to the long path; the lane
                        // sees an empty document, writes nothing and asks for the next one
                        bool taken = false;
                        if constexpr (!PLAIN) { taken = p.lg.thresh > 0 && n > p.lg.thresh && p.counts[doc] == -1; n = taken ? 0 : n; }
                        cls_at.init(p.cls, b);
                        if constexpr (PLAIN) { out.init(p.ids_tmp + ids_slot(b, doc), nullptr); lane.init(n, cap, p.unk, 0); }
                        else {
                            out.init(p.ids_tmp + ids_slot(b, doc), p.span_tmp ? p.span_tmp + 2 * ids_slot(b, doc) : nullptr); lane.init(n, cap, p.unk, p.words);
                            if (p.lg.cap_shift) lane.max_triples = n >> p.lg.cap_shift;      // (test knob of the words modes: bf_kernels.h LexLongParams)
                        }
                        bool more;
                        if constexpr (TWO) more = lane.prepare2(); else more = lane.prepare();
                        if (more) mode = M_WALK;
                        else if (!taken) p.counts[doc] = lane.finish();          // empty / invalid document: 0 ids, stay idle
                    }
                }
                if (__ballot(mode != M_EXIT) == 0) break;
            }
        }
        if (STATS) { const unsigned long long t1 = __builtin_readcyclecounter(); tk_fetch += t1 - tk0; }
    }
    if (STATS && lane_id() == 0) {
        atomicAdd(&p.stats[0], st_trips); atomicAdd(&p.stats[1], st_walk_lanes); atomicAdd(&p.stats[2], st_ev_rounds);
        atomicAdd(&p.stats[3], st_ev_lanes); atomicAdd(&p.stats[4], st_fetch_rounds); atomicAdd(&p.stats[5], st_need_lanes);
        atomicAdd(&p.stats[6], tk_walk); atomicAdd(&p.stats[7], tk_event); atomicAdd(&p.stats[8], tk_fetch);
    }
}

template <int THREADS, class WIN, bool HAS_ANY, int UNROLL, bool STATS, bool TLDS = false, bool TWO = false>
__global__ __launch_bounds__(THREADS) void k_lex_wp_flat(WpLexParams p)
{
    lex_wp_flat_body<THREADS, WIN, HAS_ANY, UNROLL, STATS, TLDS, TWO, false>(p);
}

// The headline instance (two-level lexer, ids only) at 8 waves per SIMD: the lane program needs 65 VGPRs as the compiler allocates it
// freely (7 waves) and fits 64 without scratch when asked to.  Measured on MI355X, default workload: tokenise 39.3 ms at 7 waves,
// 35.2 ms at 8 -- the kernel hides table-gather latency with resident waves.  The other instances would spill (12..44 B of scratch).
template <int UNROLL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_lex_wp_plain(WpLexParams p)
{
    lex_wp_flat_body<64, ClsWin32, false, UNROLL, false, false, true, true>(p);
}

static size_t lex_lds_bytes(const WpLexParams &p, int threads, bool tlds = false)
{
    return (((size_t)p.L.max_frames * LEX_FRAME_WORDS + 8) * threads + (size_t)((p.acts_n + 1) & ~1)) * 4 + (tlds ? (size_t)p.table_n * 8 : 0);
}
constexpr int LEX_TLDS_THREADS = 512;            // workgroup of the LDS-table variant: 8 waves share one copy of the table
constexpr size_t LEX_TLDS_MAX_BYTES = 64 * 1024; // LDS budget of one workgroup (two of them fit a CU's 160 KB)

// variant (experiments): bits 8..15 = event threshold, bits 16..19 = fetch threshold, bits 20..23 = transitions per vote (two-level
// ids-only instance: 3 or 5), bits 24..29 = resident waves per CU
void launch_lex_wp(const WpLexParams &p, int variant, hipStream_t s)
{
    WpLexParams q = p;
    q.ev_thresh = (variant >> 8) & 0xff; if (q.ev_thresh == 0) q.ev_thresh = 32;
    q.fetch_thresh = (variant >> 16) & 0xf; if (q.fetch_thresh == 0) q.fetch_thresh = 8;
    q.acts_n = p.acts_n;                                          // <= 4096 ints, checked at LoadModel
    int waves_per_cu = (variant >> 24) & 0x3f;
    if (waves_per_cu == 0) {                                      // persistent: exactly the resident waves
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lex_wp_flat<64, ClsWin, false, 3, false>, 64, lex_lds_bytes(q, 64)) != hipSuccess || per_cu <= 0) per_cu = 16;
        waves_per_cu = per_cu;
        (void)hipGetLastError();
    }
    int64_t blocks = (int64_t)device_cus() * (int64_t)waves_per_cu;
    const int64_t need = (p.b.ndocs + 63) / 64;
    if (blocks > need) blocks = need;
    if (blocks < 1) blocks = 1;
    const bool has_any = p.L.cls_any != LX_CLS_NONE;
    // small models (wbd.bin: TextToWords): the whole table lives in LDS, one copy per workgroup of 512 threads -- or of 256 when the
    // lanes' frames and id buffers leave no room for it beside 512 (wbd.bin: 41 + 27 KB; a table gather from L2 is 250+ ns, the lane
    // kernel of config 1 took 225 us with it)
    if (!p.stats && p.table_n > 0 && lex_lds_bytes(q, LEX_TLDS_THREADS / 2, true) <= LEX_TLDS_MAX_BYTES) {
        auto go = [&](auto th) {
            constexpr int TH = decltype(th)::value;
            const size_t lds = lex_lds_bytes(q, TH, true);
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lex_wp_flat<TH, ClsWin, false, 3, false, true>, TH, lds) != hipSuccess || per_cu <= 0) per_cu = 2;
            (void)hipGetLastError();
            int64_t nb = (int64_t)device_cus() * per_cu;
            const int64_t need_b = (p.b.ndocs + TH - 1) / TH;
            if (nb > need_b) nb = need_b;
            if (nb < 1) nb = 1;
            if (has_any) hipLaunchKernelGGL((k_lex_wp_flat<TH, ClsWin, true, 1, false, true>), dim3((unsigned)nb), dim3(TH), lds, s, q);
            else hipLaunchKernelGGL((k_lex_wp_flat<TH, ClsWin, false, 3, false, true>), dim3((unsigned)nb), dim3(TH), lds, s, q);
        };
        if (lex_lds_bytes(q, LEX_TLDS_THREADS, true) <= LEX_TLDS_MAX_BYTES) go(std::integral_constant<int, LEX_TLDS_THREADS>{});
        else go(std::integral_constant<int, LEX_TLDS_THREADS / 2>{});
        return;
    }
    if (p.L.two_level && !p.stats) {
        // two-level lexers: no saved frames in LDS, cheap events -> a lower event threshold pays (swept on MI355X)
        WpLexParams q2 = q; q2.L.max_frames = 0;
        if (((variant >> 8) & 0xff) == 0) q2.ev_thresh = 16;
        const size_t lds2 = lex_lds_bytes(q2, 64);
        // transitions per vote: swept on MI355X with the two-level event code (2: 8.50 ms, 3: 7.67, 4: 6.86 on the 1.25 M-doc shard)
        const int un = (variant >> 20) & 0xf; (void)un;
        const bool plain = !has_any && !q2.words && !q2.span_tmp;
        int per_cu = 0;
        const hipError_t oe = plain ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lex_wp_plain<4>, 64, lds2)
                                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lex_wp_flat<64, ClsWin, false, 4, false, false, true>, 64, lds2);
        if (oe != hipSuccess || per_cu <= 0) per_cu = 16;
        (void)hipGetLastError();
        if (((variant >> 24) & 0x3f) != 0) per_cu = (variant >> 24) & 0x3f;
        int64_t nb = (int64_t)device_cus() * per_cu;
        if (nb > need) nb = need;
        if (nb < 1) nb = 1;
        const dim3 g2((unsigned)nb), t2(64);
        if (has_any) hipLaunchKernelGGL((k_lex_wp_flat<64, ClsWin, true, 1, false, false, true>), g2, t2, lds2, s, q2);
#ifdef BF_EXPERIMENTS
        else if (plain && un == 5) hipLaunchKernelGGL(k_lex_wp_plain<5>, g2, t2, lds2, s, q2);
        else if (plain && un == 3) hipLaunchKernelGGL(k_lex_wp_plain<3>, g2, t2, lds2, s, q2);
#endif
        else if (plain) hipLaunchKernelGGL(k_lex_wp_plain<4>, g2, t2, lds2, s, q2);
        else hipLaunchKernelGGL((k_lex_wp_flat<64, ClsWin, false, 4, false, false, true>), g2, t2, lds2, s, q2);
        return;
    }
    const dim3 g((unsigned)blocks), t(64); const size_t lds = lex_lds_bytes(q, 64);
    if (has_any) hipLaunchKernelGGL((k_lex_wp_flat<64, ClsWin, true, 1, false>), g, t, lds, s, q);
    else if (p.stats) hipLaunchKernelGGL((k_lex_wp_flat<64, ClsWin, false, 3, true>), g, t, lds, s, q);
    else hipLaunchKernelGGL((k_lex_wp_flat<64, ClsWin, false, 3, false>), g, t, lds, s, q);     // three transitions per vote: swept on MI355X
}

// ------------------------------------------------------------------------------------------
// The long documents of the words modes (bf_kernels.h LexLongParams, bf_lex.h lex_one_start / lex_chain_visit).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lex_long_list(WpLexParams p)
{
    for (int64_t doc = (int64_t)blockIdx.x * 256 + threadIdx.x; doc < p.b.ndocs; doc += (int64_t)gridDim.x * 256) {
        const int n = p.nchars[doc];
        if (n <= p.lg.thresh) continue;
        const unsigned long long nch = ((unsigned long long)n + 1 + 63) >> 6;              // positions -1 .. n-1
        const unsigned long long old = atomicAdd(p.lg.hdr, (1ull << 32) | nch);
        const int64_t slot = (int64_t)(old >> 32), c0 = (int64_t)(old & LEX_LONG_CHUNK_MASK);
        const bool fits = slot < p.lg.cap_docs && c0 + (int64_t)nch <= p.lg.cap_chunks;
        if (slot < p.lg.cap_docs) p.lg.list[slot] = LexLongDoc{fits ? doc : -1, c0};
        p.counts[doc] = fits ? -1 : 0;               // -1: the lane kernel leaves the document to the long path
    }
}

constexpr int LEX_LONG_THREADS = 256;
static size_t lex_long_lds_bytes(const WpLexParams &p, bool tlds)
{
    return (((size_t)p.L.max_frames * LEX_FRAME_WORDS) * LEX_LONG_THREADS + (size_t)((p.acts_n + 1) & ~1)) * 4 + (tlds ? (size_t)p.table_n * 8 : 0);
}

// The chain inside one chunk, from cell `rel`, for the wave that holds the chunk's results in its lanes (to = the next cell relative to
// the chunk, >= 64: outside): the mask of the visited cells.  Scalar: v_readlane of the next cell, five instructions per hop.
__device__ __forceinline__ unsigned long long lex_chunk_chain(int to, int rel)
{
    unsigned long long V = 0;
    do { V |= 1ull << rel; rel = __builtin_amdgcn_readlane(to, rel); } while (rel < 64);
    return V;
}

// EMIT false: every cell of the chunk space runs its start position without output (spec), the wave resolves its chunk (jump).
// EMIT true:  the chunks the chain enters follow it inside themselves; the visited cells run again and write their tokens.
template <bool HAS_ANY, bool TLDS, bool EMIT>
__global__ __launch_bounds__(LEX_LONG_THREADS) void k_lex_long(WpLexParams p)
{
    extern __shared__ int32_t lex_lds[];
    constexpr int THREADS = LEX_LONG_THREADS, WAVES = THREADS / 64;
    typedef typename std::conditional<TLDS, TabLds, TabDirect>::type TAB;
    const unsigned long long hdr = *p.lg.hdr;
    int64_t nlist = (int64_t)(hdr >> 32), nchunks = (int64_t)(hdr & LEX_LONG_CHUNK_MASK);
    if (nlist > p.lg.cap_docs) nlist = p.lg.cap_docs;
    if (nchunks > p.lg.cap_chunks) nchunks = p.lg.cap_chunks;
    if ((int64_t)blockIdx.x * WAVES >= nchunks) return;                   // no chunk for this workgroup: before anything is staged
    LexTables L = p.L;
    TAB tab;
    {
        int32_t *acts_lds = lex_lds + (size_t)p.L.max_frames * LEX_FRAME_WORDS * THREADS;
        for (int i = threadIdx.x; i < p.acts_n; i += THREADS) acts_lds[i] = p.L.acts[i];
        if constexpr (TLDS) {
            uint64_t *tab_lds = (uint64_t *)(acts_lds + ((p.acts_n + 1) & ~1));
            for (int i = threadIdx.x; i < p.table_n; i += THREADS) tab_lds[i] = p.L.T[i];
            tab.lds = tab_lds; tab.n = (uint32_t)p.table_n;
        } else tab.T = p.L.T;
        __syncthreads();
        L.acts = acts_lds;
    }
    FramesLds frames{lex_lds, THREADS};
    const int lane = lane_id();
    for (int64_t c = (int64_t)blockIdx.x * WAVES + wave_in_block(); c < nchunks; c += (int64_t)gridDim.x * WAVES) {
        int64_t lo = 0, hi = nlist;                                       // the listed document whose chunks hold chunk c
        while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (p.lg.list[mid].chunk0 <= c) lo = mid; else hi = mid; }
        const LexLongDoc ld = p.lg.list[lo];
        if (ld.doc < 0) continue;
        const int n = p.nchars[ld.doc];
        const int cap = n >> p.lg.cap_shift;                              // triples the reference's buffer holds (tokdll:494-499)
        const int64_t b = p.b.doc_off[ld.doc];
        const int w0 = (int)(c - ld.chunk0) * 64;                         // the chunk's first cell in the document (cell = position + 1)
        const int pos = w0 + lane - 1;
        const int64_t cell = c * 64 + lane;
        const bool act = pos < n;
        ClsWin cls_at; cls_at.init(p.cls, b);
        if constexpr (!EMIT) {
            LexStart r; r.next = n; r.n_out = 0; r.n_emit = 0;
            int one = 0;                                                   // the span of the position's first token when it writes one or two and they pack (else 0)
            if (act) {
                IdOutFirst out;
                r = lex_one_start<HAS_ANY>(L, cls_at, n, pos, out, frames, tab, p.words, cap);
                if (r.n_out == 1) one = out.packed0(pos);
                else if (r.n_out == 2) {
                    const int two = out.packed1(pos);
                    one = two ? out.packed0(pos) : 0;
                    ((int2 *)p.lg.tok2)[cell] = make_int2(out.tag1, two);
                }
                ((int4 *)p.lg.spec)[cell] = make_int4(r.next, r.n_out, r.n_emit, out.tag0);
            }
            // from every cell: the first cell beyond the chunk and the counts on the way, by pointer doubling over the lanes (a chain
            // makes at most 63 hops inside a chunk).  J >= 0: the chain from here is at lane J after 2^k hops; J < 0: it has left, to E
            const bool end = !act || r.next >= n;
            const int to = r.next + 1;
            int J = (!end && to < w0 + 64) ? to - w0 : -1;
            int E = end ? LEX_CHAIN_END : to;
            int so = r.n_out, se = r.n_emit;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int j = J < 0 ? lane : J;
                const int Jj = __shfl(J, j, 64), Ej = __shfl(E, j, 64), soj = __shfl(so, j, 64), sej = __shfl(se, j, 64);
                if (J >= 0) {
                    so = so + soj < LEX_COUNT_SAT ? so + soj : LEX_COUNT_SAT; se = se + sej < LEX_COUNT_SAT ? se + sej : LEX_COUNT_SAT;
                    if (Jj < 0) { E = Ej; J = -1; } else J = Jj;
                }
            }
            ((int4 *)p.lg.jump)[cell] = make_int4(E, so, se, one);
            if (lane == 0) ((int4 *)p.lg.entry)[c] = make_int4(-1, 0, 0, 0);
        } else {
            const int4 ent = ((const int4 *)p.lg.entry)[c];
            const int rel0 = __builtin_amdgcn_readfirstlane(ent.x);
            if (rel0 < 0) continue;                                       // the chain does not come through this chunk
            const int ob = __builtin_amdgcn_readfirstlane(ent.y), eb = __builtin_amdgcn_readfirstlane(ent.z);
            int4 sp = make_int4(n, 0, 0, 0);
            if (act) sp = ((const int4 *)p.lg.spec)[cell];
            const int to = sp.x + 1 > n ? LEX_CHAIN_END : sp.x + 1 - w0;
            const unsigned long long V = lex_chunk_chain(to, rel0);
            const bool in = (V >> lane) & 1ull;
            const int no = in ? sp.y : 0, ne = in ? sp.z : 0;
            const int so = wv::incl_scan(no), se = wv::incl_scan(ne);
            // the visited cell at which the triple buffer fills (bf_lex.h lex_chain_visit) runs with the room that is left and ends the document
            const unsigned long long m_over = __ballot(in && eb + se > cap);
            const int first = m_over ? __ffsll((long long)m_over) - 1 : 64;
            // (a visited cell without a token has nothing to write: under sbd.bin the loop visits every position and one in a hundred ends a sentence)
            if (in && lane <= first && (no > 0 || lane == first)) {
                const int base = ob + so - no;
                const int64_t slot = ids_slot(b, ld.doc) + base;
                const int one = (no <= 2 && lane != first) ? p.lg.jump[4 * cell + 3] : 0;
                if (one < 0) {                                                 // the one or two tokens of this position are on record: no second walk
                    p.ids_tmp[slot] = sp.w; p.span_tmp[2 * slot] = pos + ((one >> 16) & 0x7fff); p.span_tmp[2 * slot + 1] = pos + (one & 0xffff);
                    if (no == 2) {
                        const int2 t2 = ((const int2 *)p.lg.tok2)[cell];
                        p.ids_tmp[slot + 1] = t2.x; p.span_tmp[2 * slot + 2] = pos + ((t2.y >> 16) & 0x7fff); p.span_tmp[2 * slot + 3] = pos + (t2.y & 0xffff);
                    }
                    continue;
                }
                IdOutDirect out{p.ids_tmp + slot, p.span_tmp + 2 * slot};
                const LexStart r = lex_one_start<HAS_ANY>(L, cls_at, n, pos, out, frames, tab, p.words, lane == first ? cap - (eb + se - ne) : cap);
                if (lane == first) p.counts[ld.doc] = base + r.n_out;
            }
        }
    }
}

// One workgroup per listed document goes from chunk to chunk along the chain: one hop per chunk through the results of k_lex_long<false>
// (jump), staged in LDS 2048 cells per round by all four waves -- the next round's loads in flight while the first wave hops through this one's.  (Following the chain cell by cell here, with
// the chunk's results in registers and v_readlane hops, cost 78 cycles per visited cell: 131 us for the 8,396-byte line of config 1 and
// 34 ms for a 1 MB document under sbd.bin, whose loop visits every position.)
constexpr int LEX_CHAIN_BLK = 2048, LEX_CHAIN_THREADS = 256, LEX_CHAIN_PER = LEX_CHAIN_BLK / LEX_CHAIN_THREADS;
__global__ __launch_bounds__(LEX_CHAIN_THREADS) void k_lex_long_chain(WpLexParams p)
{
    __shared__ int4 blk[LEX_CHAIN_BLK];
    __shared__ int s_next;                             // wave 0 -> the workgroup: the cell the chain goes on at (-1: the document is done)
    const unsigned long long hdr = *p.lg.hdr;
    int64_t nlist = (int64_t)(hdr >> 32);
    if (nlist > p.lg.cap_docs) nlist = p.lg.cap_docs;
    const int lane = lane_id(), wave = wave_in_block();
    for (int64_t j = blockIdx.x; j < nlist; j += gridDim.x) {
        const LexLongDoc ld = p.lg.list[j];
        if (ld.doc < 0) continue;
        const int n = __builtin_amdgcn_readfirstlane(p.nchars[ld.doc]);
        if (p.lg.big_cells > 0 && n + 1 > p.lg.big_cells) continue;      // (the two-level kernels below)
        const int cap = n >> p.lg.cap_shift;
        const int4 *jump = (const int4 *)p.lg.jump + ld.chunk0 * 64;
        int4 *entry = (int4 *)p.lg.entry + ld.chunk0;
        const int end_cells = (n & ~63) + 64;          // cells 0 .. n (cell = position + 1); the document owns whole chunks
        int q = 0, ob = 0, eb = 0, blk0 = 0;
        int4 t[LEX_CHAIN_PER];
        auto load = [&](int from) {
#pragma unroll
            for (int k = 0; k < LEX_CHAIN_PER; ++k) { const int i = from + k * LEX_CHAIN_THREADS + (int)threadIdx.x; t[k] = i < end_cells ? jump[i] : make_int4(0, 0, 0, 0); }
        };
        auto store = [&]() {
#pragma unroll
            for (int k = 0; k < LEX_CHAIN_PER; ++k) blk[k * LEX_CHAIN_THREADS + (int)threadIdx.x] = t[k];
        };
        load(0); store();
        __syncthreads();
        for (;;) {
            // the next round's cells are on their way while the first wave hops through this round's (the chain seldom jumps over a round)
            const bool more = blk0 + LEX_CHAIN_BLK < end_cells;
            if (more) load(blk0 + LEX_CHAIN_BLK);
            if (wave == 0) {
                int next = -1;
                for (;;) {
                    const int4 rec = blk[q - blk0];
                    const int E = __builtin_amdgcn_readfirstlane(rec.x), so = __builtin_amdgcn_readfirstlane(rec.y), se = __builtin_amdgcn_readfirstlane(rec.z);
                    if (lane == 0) entry[q >> 6] = make_int4(q & 63, ob, eb, 0);
                    if (eb + se > cap) break;                                       // the triple buffer fills inside this chunk: k_lex_long<true> ends the document there
                    ob += so; eb += se;
                    if (E == LEX_CHAIN_END) { if (lane == 0) p.counts[ld.doc] = ob; break; }
                    q = E;
                    if (q >= blk0 + LEX_CHAIN_BLK) { next = q; break; }
                }
                if (lane == 0) s_next = next;
            }
            __syncthreads();                           // the first wave is done with blk
            q = s_next;
            if (q < 0) break;
            if (q >= blk0 + 2 * LEX_CHAIN_BLK) { blk0 = q & ~63; load(blk0); }      // (a jump over the whole next round)
            else blk0 += LEX_CHAIN_BLK;
            store();
            __syncthreads();
        }
        __syncthreads();                               // (s_next and blk are rewritten by the next document)
    }
}

// ---- the chain of a very long document in two levels.  One hop per chunk is 16 k dependent hops for a 1 MB document (2.3 ms at 140 ns);
//      the hops of different super-chunks (64 chunks) do not depend on each other once every cell knows where the chain from it leaves its
//      super-chunk, and that is the same walk from every cell at once.
constexpr int LEX_SUPER_SHIFT = 12;             // cells per super-chunk: 4096

// every cell of a big document: follow the chunk-level jumps to the first cell beyond the cell's super-chunk, summing the counts
__global__ __launch_bounds__(256) void k_lex_long_jump2(WpLexParams p)
{
    const unsigned long long hdr = *p.lg.hdr;
    int64_t nlist = (int64_t)(hdr >> 32), nchunks = (int64_t)(hdr & LEX_LONG_CHUNK_MASK);
    if (nlist > p.lg.cap_docs) nlist = p.lg.cap_docs;
    if (nchunks > p.lg.cap_chunks) nchunks = p.lg.cap_chunks;
    const int lane = lane_id();
    for (int64_t c = (int64_t)blockIdx.x * 4 + wave_in_block(); c < nchunks; c += (int64_t)gridDim.x * 4) {
        int64_t lo = 0, hi = nlist;
        while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (p.lg.list[mid].chunk0 <= c) lo = mid; else hi = mid; }
        const LexLongDoc ld = p.lg.list[lo];
        if (ld.doc < 0) continue;
        const int n = p.nchars[ld.doc];
        if (n + 1 <= p.lg.big_cells) continue;
        const int4 *jump = (const int4 *)p.lg.jump + ld.chunk0 * 64;
        const int rel_chunk = (int)(c - ld.chunk0);
        const int q = rel_chunk * 64 + lane;
        const int sc_end = ((q >> LEX_SUPER_SHIFT) + 1) << LEX_SUPER_SHIFT;
        int4 r = jump[q];
        int E = r.x, so = r.y, se = r.z;
        while (E != LEX_CHAIN_END && E < sc_end) {
            r = jump[E];
            so = so + r.y < LEX_COUNT_SAT ? so + r.y : LEX_COUNT_SAT; se = se + r.z < LEX_COUNT_SAT ? se + r.z : LEX_COUNT_SAT;
            E = r.x;
        }
        ((int4 *)p.lg.jump2)[c * 64 + lane] = make_int4(E, so, se, 0);
        if (lane == 0 && (rel_chunk & 63) == 0) ((int4 *)p.lg.entry2)[c] = make_int4(-1, 0, 0, 0);
    }
}

// one wave per big document: from super-chunk to super-chunk
__global__ __launch_bounds__(64) void k_lex_long_chain2(WpLexParams p)
{
    const unsigned long long hdr = *p.lg.hdr;
    int64_t nlist = (int64_t)(hdr >> 32);
    if (nlist > p.lg.cap_docs) nlist = p.lg.cap_docs;
    const int lane = lane_id();
    for (int64_t j = blockIdx.x; j < nlist; j += gridDim.x) {
        const LexLongDoc ld = p.lg.list[j];
        if (ld.doc < 0) continue;
        const int n = __builtin_amdgcn_readfirstlane(p.nchars[ld.doc]);
        if (n + 1 <= p.lg.big_cells) continue;
        const int cap = n >> p.lg.cap_shift;
        const int4 *jump2 = (const int4 *)p.lg.jump2 + ld.chunk0 * 64;
        int4 *entry2 = (int4 *)p.lg.entry2 + ld.chunk0;
        int q = 0, ob = 0, eb = 0;
        for (;;) {
            const int4 rec = jump2[q];
            const int E = __builtin_amdgcn_readfirstlane(rec.x), so = __builtin_amdgcn_readfirstlane(rec.y), se = __builtin_amdgcn_readfirstlane(rec.z);
            if (lane == 0) entry2[(q >> LEX_SUPER_SHIFT) << (LEX_SUPER_SHIFT - 6)] = make_int4(q, ob, eb, 0);
            if (eb + se > cap) break;                   // the triple buffer fills inside this super-chunk: k_lex_long_chain3 stops at the chunk
            ob += so; eb += se;
            if (E == LEX_CHAIN_END) break;              // (k_lex_long_chain3 arrives there itself and writes the document's count)
            q = E;
        }
    }
}

// one wave per entered super-chunk of a big document: from chunk to chunk inside it (what k_lex_long_chain does for a whole document)
__global__ __launch_bounds__(256) void k_lex_long_chain3(WpLexParams p)
{
    const unsigned long long hdr = *p.lg.hdr;
    int64_t nlist = (int64_t)(hdr >> 32), nchunks = (int64_t)(hdr & LEX_LONG_CHUNK_MASK);
    if (nlist > p.lg.cap_docs) nlist = p.lg.cap_docs;
    if (nchunks > p.lg.cap_chunks) nchunks = p.lg.cap_chunks;
    const int lane = lane_id();
    const int64_t nsuper = (nchunks + 63) >> 6;          // (an upper bound of the super-chunks: a document's are spaced 64 chunks from ITS first chunk)
    for (int64_t c = (int64_t)blockIdx.x * 4 + wave_in_block(); c < nchunks; c += (int64_t)gridDim.x * 4) {
        (void)nsuper;
        int64_t lo = 0, hi = nlist;
        while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (p.lg.list[mid].chunk0 <= c) lo = mid; else hi = mid; }
        const LexLongDoc ld = p.lg.list[lo];
        if (ld.doc < 0 || ((c - ld.chunk0) & 63) != 0) continue;
        const int n = __builtin_amdgcn_readfirstlane(p.nchars[ld.doc]);
        if (n + 1 <= p.lg.big_cells) continue;
        const int4 ent = ((const int4 *)p.lg.entry2)[c];
        int q = __builtin_amdgcn_readfirstlane(ent.x);
        if (q < 0) continue;                              // the chain does not come through this super-chunk
        int ob = __builtin_amdgcn_readfirstlane(ent.y), eb = __builtin_amdgcn_readfirstlane(ent.z);
        const int cap = n >> p.lg.cap_shift;
        const int4 *jump = (const int4 *)p.lg.jump + ld.chunk0 * 64;
        int4 *entry = (int4 *)p.lg.entry + ld.chunk0;
        const int sc_end = ((q >> LEX_SUPER_SHIFT) + 1) << LEX_SUPER_SHIFT;
        for (;;) {
            const int4 rec = jump[q];
            const int E = __builtin_amdgcn_readfirstlane(rec.x), so = __builtin_amdgcn_readfirstlane(rec.y), se = __builtin_amdgcn_readfirstlane(rec.z);
            if (lane == 0) entry[q >> 6] = make_int4(q & 63, ob, eb, 0);
            if (eb + se > cap) break;                     // the triple buffer fills inside this chunk: k_lex_long<true> ends the document there
            ob += so; eb += se;
            if (E == LEX_CHAIN_END) { if (lane == 0) p.counts[ld.doc] = ob; break; }
            q = E;
            if (q >= sc_end) break;
        }
    }
}

void launch_lex_long_list(const WpLexParams &p, hipStream_t s)
{
    if (p.lg.thresh <= 0 || p.b.ndocs <= 0) return;
    int64_t nb = (p.b.ndocs + 255) / 256;
    const int64_t cap = (int64_t)device_cus() * 8;
    if (nb > cap) nb = cap;
    hipLaunchKernelGGL(k_lex_long_list, dim3((unsigned)nb), dim3(256), 0, s, p);
}

void launch_lex_long(const WpLexParams &p, hipStream_t s)
{
    if (p.lg.thresh <= 0 || p.b.ndocs <= 0) return;
    const bool has_any = p.L.cls_any != LX_CLS_NONE;
    const bool tlds = p.table_n > 0 && lex_long_lds_bytes(p, true) <= LEX_TLDS_MAX_BYTES;
    const size_t lds = lex_long_lds_bytes(p, tlds);
    // the number of chunks is on the device: a grid that fills the chip (the resident workgroups: LDS -- frames and the table -- bounds them),
    // workgroups without a chunk leave at once
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lex_long<false, false, false>, LEX_LONG_THREADS, lds) != hipSuccess || per_cu <= 0) per_cu = 4;
    (void)hipGetLastError();
    int64_t nb = (int64_t)device_cus() * per_cu;
    const int64_t most = (p.lg.cap_chunks + 3) / 4;
    if (nb > most) nb = most;
    if (nb < 1) nb = 1;
    int64_t nc = p.lg.cap_docs < (int64_t)device_cus() * 8 ? p.lg.cap_docs : (int64_t)device_cus() * 8;
    if (nc < 1) nc = 1;
    const dim3 g((unsigned)nb), t(LEX_LONG_THREADS);
#define BF_LONG(EMIT) \
    do { \
        if (has_any) { if (tlds) hipLaunchKernelGGL((k_lex_long<true, true, EMIT>), g, t, lds, s, p); else hipLaunchKernelGGL((k_lex_long<true, false, EMIT>), g, t, lds, s, p); } \
        else { if (tlds) hipLaunchKernelGGL((k_lex_long<false, true, EMIT>), g, t, lds, s, p); else hipLaunchKernelGGL((k_lex_long<false, false, EMIT>), g, t, lds, s, p); } \
    } while (0)
    BF_LONG(false);
    // (a document of more than big_cells cells has more bytes than that: batches that cannot hold one skip the three launches)
    const bool big = p.lg.big_cells > 0 && p.b.total_bytes + 1 > p.lg.big_cells;
    int64_t nb2 = (p.lg.cap_chunks + 3) / 4;
    if (nb2 > (int64_t)device_cus() * 8) nb2 = (int64_t)device_cus() * 8;
    if (nb2 < 1) nb2 = 1;
    if (big) hipLaunchKernelGGL(k_lex_long_jump2, dim3((unsigned)nb2), dim3(256), 0, s, p);
    hipLaunchKernelGGL(k_lex_long_chain, dim3((unsigned)nc), dim3(LEX_CHAIN_THREADS), 0, s, p);
    if (big) {
        hipLaunchKernelGGL(k_lex_long_chain2, dim3((unsigned)(nc < 256 ? nc : 256)), dim3(64), 0, s, p);
        hipLaunchKernelGGL(k_lex_long_chain3, dim3((unsigned)nb2), dim3(256), 0, s, p);
    }
    BF_LONG(true);
#undef BF_LONG
}

// ------------------------------------------------------------------------------------------
// k_wp_wave: the WordPiece path of unit-form lexers (bf_wave.h): wave-cooperative decode -> top-level tokens -> one word per
// lane slot -> ids in document order.  No class stream in HBM, no prep kernel.  namespace wv = the wave intrinsics of gfx950
// behind the names the wave program (bf_wave_body.h) uses; the test build supplies a simulator behind the same names.
// ------------------------------------------------------------------------------------------
} // namespace bfa
#include "bf_wave_body.h"
namespace bfa {

// WPE: waves per SIMD the register allocation is asked to allow (a workgroup is four waves, one per SIMD: WPE workgroups per CU)
template <class LDS, int NU, int STEPS, int WPE, bool STATS, int UMIN = 4, bool OFFS = false, int TRIM = 0, bool LIST = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_wp_wave(WpWaveParams p, int grab)
{
    __shared__ LDS lds[4];
    __shared__ uint16_t ascii[128];
    __shared__ int32_t acts[WV_ACTS_MAX];
    __shared__ WpWaveCold cold;
    if (LIST && *p.list_n == 0) return;                 // (the usual case of the LIST instance: the flat program handed nothing back)
    if (threadIdx.x == 0) cold = p.cold;
    wv_init_ascii(p.cold, ascii, (int)threadIdx.x, 256);
    for (int i = (int)threadIdx.x; i < p.acts_n; i += 256) acts[i] = p.acts[i];
    __syncthreads();
    // the wave number as a scalar: what a wave reads of its own LDS block at a wave-uniform index is then wave-uniform for the compiler too
    WpWave<LDS, NU, STATS, 0, STEPS, UMIN, 0, OFFS, TRIM, LIST> w(p, cold, lds[wave_in_block()], ascii, acts);
    w.run(grab, (int)(blockIdx.x * 4) + wave_in_block(), (int)(gridDim.x * 4));
}


// The instances the library carries: ids (ring of 1,024 elements -- the longest word + one chunk fit: bf_model.cpp "unit form" --, a queue
// of 256 tokens, a table of 8 open documents, eight workgroups per CU: 64 VGPRs, 20 KB of LDS per workgroup; every parameter swept on
// MI355X, profiles/r03_ab_runs.txt, r04_p_wp_trim.txt), its twin with counters (BfSetLexStats), the offsets instance (a span with every id:
// six workgroups per CU) and the LIST instance (the documents the flat program hands back).  variant bits 12..15 = documents per grab
// (0 = by batch size), bits 24..29 = workgroups per CU (measurements: fewer resident waves).
void launch_wp_wave(const WpWaveParams &p, int variant, hipStream_t s)
{
    int grab = (variant >> 12) & 0xf;
    if (grab == 0 || grab > WV_GRAB_MAX) {
        // documents a wave takes from the work counter at once: eight in a large batch (one atomic per 4 KB of text); a batch with fewer
        // documents than that would give every wave is spread over the waves one by one (a batch of 64 single sentences: 64 waves)
        const int64_t per_wave = p.ndocs / ((int64_t)device_cus() * 32);
        grab = per_wave >= WV_GRAB_MAX ? WV_GRAB_MAX : per_wave < 1 ? 1 : (int)per_wave;
    }
    const int per_cu_override = (variant >> 24) & 0x3f;
    typedef WvLds<1024, 256, 8> L;
    typedef WvLds<1024, 256, 8, true> LO;
    static int pc_ids = 0, pc_off = 0, pc_list = 0, pc_list_off = 0;
    int per_cu;
    if (p.doc_list && p.span_tmp) { per_cu = wp_blocks_per_cu(k_wp_wave<LO, 1, 3, 6, false, 4, true, 15, true>, pc_list_off); grab = 1; }
    else if (p.doc_list) { per_cu = wp_blocks_per_cu(k_wp_wave<L, 1, 3, 8, false, 4, false, 15, true>, pc_list); grab = 1; }
    else if (p.span_tmp) per_cu = wp_blocks_per_cu(k_wp_wave<LO, 1, 3, 6, false, 4, true, 15>, pc_off);
    else per_cu = wp_blocks_per_cu(k_wp_wave<L, 1, 3, 8, false, 4, false, 15>, pc_ids);
    if (per_cu_override > 0) per_cu = per_cu_override;
    int64_t blocks = (int64_t)device_cus() * per_cu;
    if (!p.doc_list) {                      // (the number of listed documents is known to the device only)
        const int64_t need = (p.ndocs + (int64_t)grab * 4 - 1) / ((int64_t)grab * 4);
        if (blocks > need) blocks = need;
    }
    if (blocks < 1) blocks = 1;
    const dim3 g((unsigned)blocks), t(256);
    if (p.doc_list && p.span_tmp) hipLaunchKernelGGL((k_wp_wave<LO, 1, 3, 6, false, 4, true, 15, true>), g, t, 0, s, p, grab);
    else if (p.doc_list) hipLaunchKernelGGL((k_wp_wave<L, 1, 3, 8, false, 4, false, 15, true>), g, t, 0, s, p, grab);
    else if (p.span_tmp) hipLaunchKernelGGL((k_wp_wave<LO, 1, 3, 6, false, 4, true, 15>), g, t, 0, s, p, grab);
    else if (p.cold.stats) hipLaunchKernelGGL((k_wp_wave<L, 1, 3, 8, true, 4, false, 15>), g, t, 0, s, p, grab);
    else hipLaunchKernelGGL((k_wp_wave<L, 1, 3, 8, false, 4, false, 15>), g, t, 0, s, p, grab);
}

} // namespace bfa
#include "bf_flat_body.h"
namespace bfa {

// ------------------------------------------------------------------------------------------
// The flat program (bf_flat.h): k_wp_pre -> k_wp_flat -> k_wp_hardlist -> k_wp_wave<LIST> -> k_wp_count -> scan -> k_wp_merge
// ------------------------------------------------------------------------------------------
// ranges of documents of about equal bytes (range r begins with the first document at or behind r / nranges of the text), and whether the
// batch is fit for the flat program at all: offsets in order and inside the buffer, no document of more than WF_DOC_MAX bytes
__global__ __launch_bounds__(256) void k_wp_pre(const int64_t *doc_off, int64_t ndocs, int64_t total_bytes, int nranges, int64_t *range_doc, int *unsafe)
{
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
    bool bad = false;
    for (int64_t d = tid; d < ndocs; d += nth) {
        const int64_t a = doc_off[d], b = doc_off[d + 1];
        bad |= a < 0 || b < a || b > total_bytes || b - a > WF_DOC_MAX;
    }
    if (bad) atomicOr(unsafe, 1);
    const int64_t first = doc_off[0], span = doc_off[ndocs] - first;
    for (int64_t r = tid; r <= nranges; r += nth) {
        int64_t lo = 0, hi = ndocs;                              // the first document whose first byte is >= target
        if (r == 0) hi = 0; else if (r == nranges) lo = ndocs;
        else {
            const int64_t target = first + (int64_t)((__int128)span * r / nranges);
            while (lo < hi) { const int64_t mid = lo + (hi - lo) / 2; if (doc_off[mid] >= target) hi = mid; else lo = mid + 1; }
        }
        range_doc[r] = lo;
    }
}

template <int WPE, bool STATS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_wp_flat(WfParams p)
{
    __shared__ WfLds lds[4];
    __shared__ alignas(16) uint32_t lut[WF_LUT];
    __shared__ WpWaveCold cold;
    if (threadIdx.x == 0) cold = p.cold;
    if (threadIdx.x < 128) lut[threadIdx.x] = wf_lut_value(p.cold, (int)threadIdx.x);
    else if (threadIdx.x < WF_LUT) lut[threadIdx.x] = wf_kmask_value((int)threadIdx.x - 128);
    __syncthreads();
    WfWave<STATS> w(p, lds[wave_in_block()], lut, cold);
    w.run((int)(blockIdx.x * 4) + wave_in_block(), (int)(gridDim.x * 4));
}

// the words of the list (bf_flat_body.h wf_units): a wave takes batches of 64 * NU records, batch number = wave number + k * waves
template <int WPE, int NU, bool STATS, bool OFFS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_wp_units(WfUnitParams p)
{
    __shared__ uint32_t lut[128];
    __shared__ uint16_t cbuf[4][64 * 16 + 8];             // the classes of a word's characters, made up front (the second list)
    if (threadIdx.x < 128) {
        WpWaveCold cold; cold.cpmap = p.cpmap; cold.kind = p.kind; cold.nclasses = p.nclasses; cold.status = nullptr; cold.stats = nullptr; cold.no_fast = 0;
        lut[threadIdx.x] = wf_lut_value(cold, (int)threadIdx.x);
    }
    __syncthreads();
    const int wave = (int)(blockIdx.x * 4) + wave_in_block(), nwaves = (int)(gridDim.x * 4);
    unsigned long long rounds = 0, batches = 0;
    for (int r = wave; r < p.nranges; r += nwaves) {
        const int64_t dlo = p.range_doc[r], dhi = p.range_doc[r + 1];
        if (dlo >= dhi) continue;
        const int64_t b0 = p.doc_off[dlo], b1 = p.doc_off[dhi];
        const unsigned long long nfast = (unsigned long long)p.wrec_cnt[2 * r], nslow = (unsigned long long)p.wrec_cnt[2 * r + 1];
        const uint32_t *fl = p.wrec + 4 * ((b0 + (1 << WF_REC_SHIFT) - 1) >> WF_REC_SHIFT), *sl = p.wrec + 4 * ((b1 >> WF_REC_SHIFT) - (int64_t)nslow);       // (the second list grows down from the end: its order does not matter)
        uint16_t *cb = cbuf[wave_in_block()];
        for (unsigned long long first = 0; first < nfast; first += 64 * NU) { wf_units<NU, STATS, 0, OFFS>(p, lut, cb, fl, b0, dlo, first, nfast, &rounds); ++batches; }
        for (unsigned long long first = 0; first < nslow; first += 64) { wf_units<1, STATS, 1, OFFS>(p, lut, cb, sl, b0, dlo, first, nslow, &rounds); wf_units<1, STATS, 2, OFFS>(p, lut, cb, sl, b0, dlo, first, nslow, &rounds); ++batches; }
    }
    if (STATS && p.stats && lane_id() == 0) { atomicAdd(&p.stats[8], rounds); atomicAdd(&p.stats[9], batches); }
}

// the documents the flat program hands back (all of them when the batch is not fit for it), in any order
__global__ __launch_bounds__(256) void k_wp_hardlist(const int32_t *dstat, const int *unsafe, int64_t ndocs, int32_t *list, unsigned int *list_n)
{
    const bool all = *unsafe != 0;
    for (int64_t d0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) & ~(int64_t)63; d0 < ndocs; d0 += (int64_t)gridDim.x * 256) {
        const int64_t d = d0 + lane_id();
        const bool take = d < ndocs && (all || (dstat[d] & WF_D_HARD));
        const unsigned long long m = __ballot(take);
        if (!m) continue;
        unsigned int base = 0;
        if (lane_id() == 0) base = atomicAdd(list_n, (unsigned int)__popcll(m));
        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
        if (take) list[base + (unsigned int)__popcll(m & lanemask_lt())] = (int32_t)d;
    }
}

__global__ __launch_bounds__(256) void k_wp_count(WfMergeParams p)
{
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave_in_block(), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t base = wave0 * 64; base < p.ndocs; base += nwaves * 64) wf_count_docs(p, base);
}

template <bool OFFS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OFFS ? 6 : 8, OFFS ? 6 : 8))) void k_wp_merge(WfMergeParams p)
{
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave_in_block(), nwaves = (int64_t)gridDim.x * 4;
    __shared__ WfMergeLds<OFFS> lds[4];
    bool over = false;
    for (int64_t base = wave0 * 64; base < p.ndocs; base += nwaves * 64) wf_merge_docs<OFFS>(p, base, over, lds[wave_in_block()]);
    if (over) atomicOr(p.status, 1);
}

int wp_flat_ranges(int64_t ndocs, int64_t total_bytes)
{
    // four ranges per resident wave, of at least 16 KB and at most WF_RANGE_MAX bytes
    int64_t r = (int64_t)device_cus() * 32 * 4;
    const int64_t most = total_bytes / 16384 + 1, least = total_bytes / WF_RANGE_MAX + 1;
    if (r > most) r = most;
    if (r < least) r = least;
    if (r > ndocs) r = ndocs;
    return (int)(r < 1 ? 1 : r);
}

void launch_wp_pre(const int64_t *doc_off, int64_t ndocs, int64_t total_bytes, int nranges, int64_t *range_doc, int *unsafe, hipStream_t s)
{
    int64_t blocks = (ndocs + 255) / 256;
    if (blocks > device_cus() * 8) blocks = device_cus() * 8;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_wp_pre, dim3((unsigned)blocks), dim3(256), 0, s, doc_off, ndocs, total_bytes, nranges, range_doc, unsafe);
}

template <int WPE>
static void launch_wp_flat_wpe(const WfParams &p, int per_cu_override, hipStream_t s)
{
    static int pc = 0;
    int per_cu = wp_blocks_per_cu(k_wp_flat<WPE, false>, pc);
    if (per_cu_override > 0) per_cu = per_cu_override;
    int64_t blocks = (int64_t)device_cus() * per_cu;
    const int64_t need = ((int64_t)p.nranges + 3) / 4;
    if (blocks > need) blocks = need;
    if (blocks < 1) blocks = 1;
    if (p.cold.stats) hipLaunchKernelGGL((k_wp_flat<WPE, true>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_wp_flat<WPE, false>), dim3((unsigned)blocks), dim3(256), 0, s, p);
}

// variant bits 16..19 (measurements): waves per SIMD the kernel is compiled for (0 = shipped), bits 24..29: workgroups per CU
void launch_wp_flat(const WfParams &p, int variant, hipStream_t s)
{
    const int per_cu_override = (variant >> 24) & 0x3f, wpe = (variant >> 16) & 0xf;
#ifdef BF_EXPERIMENTS
    if (wpe == 8) { launch_wp_flat_wpe<8>(p, per_cu_override, s); return; }
    if (wpe == 6) { launch_wp_flat_wpe<6>(p, per_cu_override, s); return; }
    if (wpe == 5) { launch_wp_flat_wpe<5>(p, per_cu_override, s); return; }
#endif
    (void)wpe;
    launch_wp_flat_wpe<7>(p, per_cu_override, s);          // seven waves per SIMD: 72 registers (swept on MI355X: 5 / 6 / 7 waves 3.09 / 2.92 / 2.83 ms per 2.5 M documents)
}

template <int WPE, int NU>
static void launch_wp_units_cfg(const WfUnitParams &p, int per_cu_override, hipStream_t s)
{
    static int pc = 0;
    int per_cu = wp_blocks_per_cu(k_wp_units<WPE, NU, false>, pc);
    if (per_cu_override > 0) per_cu = per_cu_override;
    const int64_t blocks = (int64_t)device_cus() * per_cu;
    if (p.hspan) hipLaunchKernelGGL((k_wp_units<WPE, NU, false, true>), dim3((unsigned)blocks), dim3(256), 0, s, p);      // the offsets API: the pieces carry their spans
    else if (p.stats) hipLaunchKernelGGL((k_wp_units<WPE, NU, true>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_wp_units<WPE, NU, false>), dim3((unsigned)blocks), dim3(256), 0, s, p);
}

// variant bits 8..11 (measurements, BF_EXPERIMENTS builds): 1 = eight waves per SIMD, 2 = one word per lane
void launch_wp_units(const WfUnitParams &p, int variant, hipStream_t s)
{
    const int per_cu_override = (variant >> 24) & 0x3f, cfg = (variant >> 8) & 0xf;
#ifdef BF_EXPERIMENTS
    if (cfg == 1) { launch_wp_units_cfg<8, 2>(p, per_cu_override, s); return; }
    if (cfg == 2) { launch_wp_units_cfg<8, 1>(p, per_cu_override, s); return; }
    if (cfg == 3) { launch_wp_units_cfg<4, 4>(p, per_cu_override, s); return; }
#endif
    (void)cfg;
    launch_wp_units_cfg<6, 2>(p, per_cu_override, s);
}

void launch_wp_hardlist(const int32_t *dstat, const int *unsafe, int64_t ndocs, int32_t *list, unsigned int *list_n, hipStream_t s)
{
    int64_t blocks = (ndocs + 255) / 256;
    if (blocks > device_cus() * 8) blocks = device_cus() * 8;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_wp_hardlist, dim3((unsigned)blocks), dim3(256), 0, s, dstat, unsafe, ndocs, list, list_n);
}

void launch_wp_count(const WfMergeParams &p, hipStream_t s)
{
    int64_t blocks = (p.ndocs + 255) / 256;
    static int pc = 0;
    const int64_t resident = (int64_t)device_cus() * wp_blocks_per_cu(k_wp_count, pc);
    if (blocks > resident) blocks = resident;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_wp_count, dim3((unsigned)blocks), dim3(256), 0, s, p);
}

void launch_wp_merge(const WfMergeParams &p, hipStream_t s)
{
    static int pc = 0;
    int64_t blocks = (p.ndocs + 255) / 256;
    static int pc_off = 0;
    const int64_t resident = (int64_t)device_cus() * (p.espan ? wp_blocks_per_cu(k_wp_merge<true>, pc_off) : wp_blocks_per_cu(k_wp_merge<false>, pc));      // one round of workgroups: every wave has the same share
    if (blocks > resident) blocks = resident;
    if (blocks < 1) blocks = 1;
    if (p.espan) hipLaunchKernelGGL(k_wp_merge<true>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_wp_merge<false>, dim3((unsigned)blocks), dim3(256), 0, s, p);
}

} // namespace bfa
