// bf_kernels_sp.hip -- HIP kernels for gfx950 (MI355X, wave64): the SentencePiece-style branch and what every branch shares.
//
//  k_prep_sp8 / k_prep_sp, k_sp_hist/_scan/_scatter   prologue (decode, dummy prefix, 1:n charmap, white-space collapse, symbol -> class) and the
//              order of the documents (longest first).
//  k_seg_unigram_lane, k_uni_back   Unigram-LM: forward pass (scores and the live window in LDS rings) and backward pass (bf_seg.h).
//  k_bpe_wave, k_bpe_seg, k_bpe_*   BPE: the wave program with a word unit (bf_bpe_wave_body.h), one wave per document for any input
//              (bf_bpe_seg_body.h), the lane kernels of plain BPE and the offsets API.
//  k_dict_*    FADictInterpreter_t::GetInfo for many keys.
//  k_scan_*    exclusive scan of per-document counts -> offsets.
//  k_compact*  gather of the per-document staging slots into one contiguous id array (+ byte offsets).
//  k_i2t_*, k_w2t_*, k_s2t_*, k_normsp, k_hash_*   IdsToText / TextToWords / TextToSentences string assembly,
//              NormalizeSpaces, TextToHashes: variable-length byte gathers and streaming kernels.
#include "bf_kernels_common.h"
#include "bf_bpe_wave_body.h"
namespace bfa {

// k_bpe_wave: the BPE wave program (bf_bpe_wave_body.h) on the class streams k_prep_sp wrote.  The program is validated in the test
// simulator (tests/test_bpe_wave_emu.py); the device path (bf_capi.cpp, behind BfSetVariant bit 0x40 until it has had its GPU parity
// and timing runs) redoes the documents it hands back (flags[d] = 1) with the lane-per-document kernels.
__host__ __device__ __forceinline__ int64_t sp_slot(int64_t doc_off_d, int64_t d, int mul) { return (int64_t)mul * (doc_off_d + d); }

template <class LDS, int WPE, int STEPS, int UMIN, bool HOME>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_bpe_wave(BpeWaveParams p, int grab)
{
    __shared__ LDS lds[4];
    BpeWave<LDS, STEPS, UMIN, HOME> w(p, lds[wave_in_block()]);
    w.run(grab, (int)(blockIdx.x * 4) + wave_in_block(), (int)(gridDim.x * 4));
}

template <int STEPS, int UMIN, int QCAP = 256, bool HOME = true>
static void launch_bpe_wave_cfg(const BpeWaveParams &p, hipStream_t s)
{
    typedef BwLds<1024, QCAP, 8> L;
    static int per_cu = 0;
    if (per_cu <= 0) {
        int q = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, k_bpe_wave<L, 4, STEPS, UMIN, HOME>, 256, 0) != hipSuccess || q <= 0) q = 2;
        (void)hipGetLastError();
        per_cu = q;
    }
    const int64_t per_wave = p.ndocs / ((int64_t)device_cus() * per_cu * 4);
    const int grab = per_wave >= WV_GRAB_MAX ? WV_GRAB_MAX : per_wave < 1 ? 1 : (int)per_wave;
    int64_t blocks = (int64_t)device_cus() * per_cu;
    const int64_t need = (p.ndocs + (int64_t)grab * 4 - 1) / ((int64_t)grab * 4);
    if (blocks > need) blocks = need;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((k_bpe_wave<L, 4, STEPS, UMIN, HOME>), dim3((unsigned)blocks), dim3(256), 0, s, p, grab);
}

// tune (BfSetVariant bits 8..11): 0 = shipped (the HOME form, four transitions per round, the units phase ends with fewer than 32 busy units, a queue
// of 256 words); 8: the in-order form of rounds 4..5 (ids retired through the queue; A/B runs and tests); experiments builds: 1 .. 3: the phase ends
// with fewer than 16 / 4 / 48; 6: six transitions per round (in-order form).  Queues of 512 / 1024 words cost a block per CU each: 22.9 / 31.5 ms against 18.8.
bool bpe_wave_home(int tune) { return tune != 8 && tune != 1 && tune != 2 && tune != 3 && tune != 6 && tune != 4 && tune != 5 && tune != 7; }
void launch_bpe_wave(const BpeWaveParams &p, int tune, hipStream_t s)
{
#ifdef BF_EXPERIMENTS
    if (tune == 1) { launch_bpe_wave_cfg<4, 16, 256, false>(p, s); return; }
    if (tune == 2) { launch_bpe_wave_cfg<4, 4, 256, false>(p, s); return; }
    if (tune == 3) { launch_bpe_wave_cfg<4, 48, 256, false>(p, s); return; }
    if (tune == 6) { launch_bpe_wave_cfg<6, 32, 256, false>(p, s); return; }
#endif
    if (tune == 8) { launch_bpe_wave_cfg<4, 32, 256, false>(p, s); return; }
    if (tune == 9) { launch_bpe_wave_cfg<4, 48, 256, true>(p, s); return; }
    if (tune == 10) { launch_bpe_wave_cfg<3, 32, 256, true>(p, s); return; }
    if (tune == 11) { launch_bpe_wave_cfg<4, 60, 256, true>(p, s); return; }
    launch_bpe_wave_cfg<4, 32, 256, true>(p, s);
}

// The HOME form of the BPE wave program leaves every id at its word's home -- a cell of the document's staging slot under the word's own elements,
// BW_HOME_NONE in the cells between -- with the document's count (the program adds it up as the words report); this puts them where the API wants
// them (tokdll:1512: the first max_ids of them, front to back):
//   k_bpe_home_gather  the cells that hold an id, in order, to ids_out + id_off[d]; a handed-back document's ids (k_bpe_seg) are dense at the slot's start
// A wave per document, 128 cells per trip, a ballot and a population count each: streaming, 4 bytes per stream element.
__global__ __launch_bounds__(256) void k_bpe_home_gather(BpeHomeParams p)
{
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave_in_block();
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    bool over = false;
    for (int64_t d = wave0; d < p.b.ndocs; d += nwaves) {
        int c = p.counts[d];
        if (c <= 0) continue;
        const int64_t o = p.id_off[d];
        if (o + c > p.ids_cap) { over = true; c = o < p.ids_cap ? (int)(p.ids_cap - o) : 0; }
        const int32_t *cell = p.ids_tmp + sp_slot(p.b.doc_off[d], d, p.slot_mul);
        int32_t *out = p.ids_out + o;
        if (p.flags[d]) { for (int i = lane; i < c; i += 64) out[i] = cell[i]; continue; }
        const int L = p.lens[d];
        int k = 0;
        for (int i0 = 0; i0 < L && k < c; i0 += 128) {
            const int ia = i0 + lane, ib = i0 + 64 + lane;
            const int32_t va = ia < L ? cell[ia] : BW_HOME_NONE, vb = ib < L ? cell[ib] : BW_HOME_NONE;
            const unsigned long long ma = __ballot(va != BW_HOME_NONE), mb = __ballot(vb != BW_HOME_NONE);
            const int ka = k + __popcll(ma & lanemask_lt()), kb = k + __popcll(ma) + __popcll(mb & lanemask_lt());
            if (va != BW_HOME_NONE && ka < c) out[ka] = va;
            if (vb != BW_HOME_NONE && kb < c) out[kb] = vb;
            k += __popcll(ma) + __popcll(mb);
        }
    }
    if (over) atomicOr(p.status, 1);
}
void launch_bpe_home_gather(const BpeHomeParams &p, hipStream_t s)
{
    int64_t blocks = (p.b.ndocs + 3) / 4;
    if (blocks > device_cus() * 16) blocks = device_cus() * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_bpe_home_gather, dim3((unsigned)blocks), dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------
// k_prep_sp: wave per document.  bytes / strict UTF-8 -> fused charmap+element-code map -> dummy prefix ->
// whitespace collapse (local keep-predicate) -> trailing trim  (tokdll:1367-1496).
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ bool sp_delimish(uint32_t code, uint32_t delim) { return code == 0xFFFDu || code == delim; }

// one 64-byte window of a document, lane = byte: what its characters map to.  Nothing here depends on the windows before it
// (the whitespace collapse does, and runs afterwards), so the loads and map gathers of several windows are in flight together.
struct SpWin { uint32_t v; int c; uint32_t first, last; bool ok; };

__device__ __forceinline__ SpWin sp_window(const SpPrepParams &p, const uint32_t *ascii_v, const uint8_t *s, int n, int bom, int q, int lane, bool &bad)
{
    const bool in = q < n;
    uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    if (in) b0 = s[q];
    bool start = in; bool err = false; int cp = (int)b0;
    // a window of plain ASCII (most of Latin-script text) needs no UTF-8 decoding at all: every byte is a character
    // (a continuation byte that straddles INTO the window from a previous lead is >= 0x80 itself, so it takes the full path)
    if (!p.use_bytes && __any(b0 >= 0x80u)) {
        // the three bytes on either side come from the neighbouring lanes (one byte load per lane instead of seven); only the
        // lanes at the edges of the 64-byte window load theirs
        b1 = __shfl_down(b0, 1, 64); b2 = __shfl_down(b0, 2, 64); b3 = __shfl_down(b0, 3, 64);
        uint32_t u1 = __shfl_up(b0, 1, 64), u2 = __shfl_up(b0, 2, 64), u3 = __shfl_up(b0, 3, 64);
        if (lane >= 61) {
            if (lane + 1 > 63) b1 = q + 1 < n ? s[q + 1] : 0u;
            if (lane + 2 > 63) b2 = q + 2 < n ? s[q + 2] : 0u;
            b3 = q + 3 < n ? s[q + 3] : 0u;
        }
        if (lane < 3) {
            if (lane < 1) u1 = q - 1 >= bom ? s[q - 1] : 0x80u;
            if (lane < 2) u2 = q - 2 >= bom ? s[q - 2] : 0x80u;
            u3 = q - 3 >= bom ? s[q - 3] : 0x80u;
        }
        const bool cont = (b0 & 0xC0) == 0x80;
        start = in && !cont;
        if (in && cont) {
            const uint32_t p1 = (q - 1 >= bom) ? u1 : 0x80u, p2 = (q - 2 >= bom) ? u2 : 0x80u, p3 = (q - 3 >= bom) ? u3 : 0x80u;
            bool ok;
            if ((p1 & 0xC0) != 0x80) ok = (q - 1 >= bom) && p1 >= 0xC0;
            else if ((p2 & 0xC0) != 0x80) ok = (q - 2 >= bom) && p2 >= 0xE0;
            else if ((p3 & 0xC0) != 0x80) ok = (q - 3 >= bom) && p3 >= 0xF0;
            else ok = false;
            err = !ok;
        } else if (start && b0 >= 0x80) {
            int len;
            if ((b0 & 0xE0) == 0xC0) { len = 2; cp = (int)(b0 & 0x1F); }
            else if ((b0 & 0xF0) == 0xE0) { len = 3; cp = (int)(b0 & 0x0F); }
            else if ((b0 & 0xF8) == 0xF0) { len = 4; cp = (int)(b0 & 0x07); }
            else { len = 1; err = true; }
            if (q + len > n) err = true;
            if (len >= 2) { if ((b1 & 0xC0) != 0x80) err = true; cp = (cp << 6) | (int)(b1 & 0x3F); }
            if (len >= 3) { if ((b2 & 0xC0) != 0x80) err = true; cp = (cp << 6) | (int)(b2 & 0x3F); }
            if (len >= 4) { if ((b3 & 0xC0) != 0x80) err = true; cp = (cp << 6) | (int)(b3 & 0x3F); }
            const int need = cp <= 0x7F ? 1 : cp <= 0x7FF ? 2 : cp <= 0xFFFF ? 3 : cp <= 0x10FFFF ? 4 : 0;
            if (need != len) err = true;
            if ((cp & 0xFFFFF800) == 0xD800) err = true;
            if (err) cp = 0;
        }
        if (__any(err)) bad = true;
    }
    // ---- elements of this character
    SpWin w; w.v = 0xFFFFu; w.c = 0; w.ok = start && !err;
    if (w.ok) {
        if (cp < 0x80) w.v = ascii_v[cp]; else w.v = cpmap_get(p.cpmap, cp);
        w.c = (w.v & 0x80000000u) ? (int)p.multi_pool[w.v & 0x7FFFFFFFu] : 1;
    }
    const uint16_t *rec = p.multi_pool + (w.v & 0x7FFFFFFFu) + 1;
    const bool multi = (w.v & 0x80000000u) != 0;
    w.first = w.c > 0 ? (multi ? (uint32_t)rec[0] : w.v) : 0u;
    w.last = w.c > 0 ? (multi ? (uint32_t)rec[w.c - 1] : w.v) : 0u;
    return w;
}

constexpr int SP_PREP_WINDOWS = 4;      // windows whose loads / map gathers are issued before the (sequential) collapse pass

__global__ __launch_bounds__(256) void k_prep_sp(SpPrepParams p)
{
    __shared__ uint32_t ascii_v[128];        // fused map of U+0000..U+007F (in bytes mode: of the bytes 0..127): no global gather for ASCII text
    if (threadIdx.x < 128) ascii_v[threadIdx.x] = cpmap_get(p.cpmap, (int)threadIdx.x);
    __syncthreads();
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave_in_block();
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const uint32_t D = p.delim_code;
    for (int64_t d = wave0; d < p.b.ndocs; d += nwaves) {
        const int64_t b = p.b.doc_off[d];
        const int64_t n64 = p.b.doc_off[d + 1] - b;
        if (n64 <= 0 || n64 > 1000000000) { if (lane == 0) p.lens[d] = 0; continue; }     // tokdll:1361-1363
        if (b < 0 || b + n64 > p.b.total_bytes) { if (lane == 0) { p.lens[d] = 0; atomicOr(p.b.status, BF_STATUS_BAD_OFFSETS); } continue; }
        const int n = (int)n64;
        const uint8_t *s = p.b.text + b;
        uint16_t *out = p.stream + sp_slot(b, d, p.slot_mul);
        const int cap = p.slot_mul * (n + 1);
        int pos = 0;
        if (n >= 3 && s[0] == 0xEF && s[1] == 0xBB && s[2] == 0xBF) pos = 3;             // BOM, also in bytes mode (FAUtf8Utils.cpp:330-335)
        const int bom = pos;
        // ---- state after the (normalised) dummy prefix: uniform across the wave
        int outc = 0, normc = 0, decoded = 0;
        uint32_t prev = 0; bool have_prev = false;
        for (int k = 0; k < p.prefix_n; ++k) {
            const uint32_t e = p.prefix[k];
            const bool ws = e == 0xFFFDu;
            const bool keep = !ws || !have_prev || !sp_delimish(prev, D);
            if (keep) { if (lane == 0 && outc < cap) { out[outc] = (uint16_t)(ws ? D : e); if (p.src_off) p.src_off[sp_slot(b, d, p.slot_mul) + outc] = -1; } ++outc; }
            prev = e; have_prev = true; ++normc;
        }
        bool bad = false;
        while (pos < n) {
            // ---- the next SP_PREP_WINDOWS windows: bytes -> characters -> map entries (independent of each other)
            SpWin win[SP_PREP_WINDOWS];
#pragma unroll
            for (int wi = 0; wi < SP_PREP_WINDOWS; ++wi) {
                win[wi].v = 0xFFFFu; win[wi].c = 0; win[wi].first = win[wi].last = 0; win[wi].ok = false;
                if (pos + 64 * wi < n) win[wi] = sp_window(p, ascii_v, s, n, bom, pos + 64 * wi + lane, lane, bad);
            }
            // ---- whitespace collapse and output, window after window (carry: prev element, output count)
#pragma unroll
            for (int wi = 0; wi < SP_PREP_WINDOWS; ++wi) {
                if (!(pos + 64 * wi < n)) break;
                const int q = pos + 64 * wi + lane;
                const uint32_t v = win[wi].v; const int c = win[wi].c; const uint32_t first = win[wi].first, last = win[wi].last;
                const uint16_t *rec = p.multi_pool + (v & 0x7FFFFFFFu) + 1;
                // previous element = last element of the nearest earlier lane that produced any, else the carry
                const unsigned long long m_has = __ballot(c > 0);
                const unsigned long long below = m_has & lanemask_lt();
                const int pl = below ? 63 - __clzll((long long)below) : 0;
                const uint32_t pl_last = __shfl(last, pl, 64);
                const uint32_t pe0 = below ? pl_last : prev;
                const bool hp0 = below ? true : have_prev;
                const bool any_multi = __any(c > 1);
                if (!any_multi) {
                    // the common window: every character is one element (or none) -- straight-line, one ballot for the output positions
                    const bool ws = first == 0xFFFDu;
                    const bool keep = c > 0 && (!ws || !hp0 || !sp_delimish(pe0, D));
                    const unsigned long long m_keep = __ballot(keep);
                    const int idx = outc + __popcll(m_keep & lanemask_lt());
                    if (keep && idx < cap) { out[idx] = (uint16_t)(ws ? D : first); if (p.src_off) p.src_off[sp_slot(b, d, p.slot_mul) + idx] = q; }
                    outc += __popcll(m_keep);
                    normc += __popcll(m_has);
                } else {
                    uint32_t pe = pe0; bool hp = hp0;
                    // keep flags
                    int kept = 0;
                    {
                        uint32_t e = first;
                        for (int k = 0; k < c; ++k) {
                            if (k > 0) e = rec[k];
                            const bool ws = e == 0xFFFDu;
                            if (!ws || !hp || !sp_delimish(pe, D)) ++kept;
                            pe = e; hp = true;
                        }
                    }
                    const int inc = wave_incl_scan(kept);
                    int idx = outc + inc - kept;
                    {
                        uint32_t pe2 = pe0; bool hp2 = hp0;
                        uint32_t e = first;
                        for (int k = 0; k < c; ++k) {
                            if (k > 0) e = rec[k];
                            const bool ws = e == 0xFFFDu;
                            if (!ws || !hp2 || !sp_delimish(pe2, D)) { if (idx < cap) { out[idx] = (uint16_t)(ws ? D : e); if (p.src_off) p.src_off[sp_slot(b, d, p.slot_mul) + idx] = q; } ++idx; }
                            pe2 = e; hp2 = true;
                        }
                    }
                    outc += __shfl(inc, 63, 64);
                    normc += __shfl(wave_incl_scan(c), 63, 64);
                }
                decoded += __popcll(__ballot(win[wi].ok));
                if (m_has) { const int hl = 63 - __clzll((long long)m_has); prev = __shfl(last, hl, 64); have_prev = true; }
            }
            pos += 64 * SP_PREP_WINDOWS;
        }
        int len = outc;
        if (len > 1 && have_prev && sp_delimish(prev, D)) --len;                          // tokdll:1491-1493
        if (bad || decoded <= 0) len = 0;                                                   // tokdll:1409-1411
        if (p.has_charmap && (normc <= 0 || normc > 2 * (n + 1))) len = 0;                // tokdll:1440-1444
        if (len > cap) len = 0;
        if (lane == 0) p.lens[d] = len;
    }
}

// ------------------------------------------------------------------------------------------
// k_prep_sp8: the same prologue with EIGHT bytes per lane (512 bytes per wave step), the decoder of the WordPiece wave program
// (bf_wave_body.h decode_chunk) in front of the _sp rules.  Measured on the multilingual corpus of config 4 (round 4): the byte-per-lane
// form issues 1,211 scalar + 1,220 vector instructions per 512-byte document -- as many as the whole WordPiece kernel -- because every
// 64-byte window pays the full set of ballots, shuffles and execution-mask bookkeeping.  Here a chunk costs: one 8-byte load per lane,
// one LDS look-up per ASCII byte, one trip per lead byte of the fullest lane (a lane of Cyrillic text holds four, of CJK three) with the
// two-level map gather, a lane-local pass over the lane's (at most eight) characters for the keep rule of the whitespace collapse, one
// prefix sum, and the stores.  Characters that expand 1 : n (or vanish) take a general form of the two lane-local passes.
// Per character the rules are those of k_prep_sp above (tokdll:1367-1496, FAUtf8Utils.cpp:121-196,233-270,316-345).
// ------------------------------------------------------------------------------------------
// WPE: waves per SIMD the register allocation is asked to allow (0: the compiler's own choice, 73 VGPRs = six waves)
template <int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE == 0 ? 1 : WPE, WPE == 0 ? 8 : WPE))) void k_prep_sp8(SpPrepParams p)
{
    __shared__ uint32_t ascii_v[128];
    __shared__ uint32_t stage_all[4][8 * 64];                  // per wave: map value of the character that starts at byte k of lane l at [k * 64 + l], SP8_NOCHAR: none
    constexpr uint32_t SP8_NOCHAR = 0xFFFFFFFEu;
    if (threadIdx.x < 128) ascii_v[threadIdx.x] = cpmap_get(p.cpmap, (int)threadIdx.x);
    __syncthreads();
    const int lane = lane_id();
    uint32_t *stage = stage_all[wave_in_block()];
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave_in_block();
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const uint32_t D = p.delim_code;
    for (int64_t d = wave0; d < p.b.ndocs; d += nwaves) {
        const int64_t b = p.b.doc_off[d];
        const int64_t n64 = p.b.doc_off[d + 1] - b;
        if (n64 <= 0 || n64 > 1000000000) { if (lane == 0) p.lens[d] = 0; continue; }     // tokdll:1361-1363
        if (b < 0 || b + n64 > p.b.total_bytes) { if (lane == 0) { p.lens[d] = 0; atomicOr(p.b.status, BF_STATUS_BAD_OFFSETS); } continue; }
        const int n = (int)n64;
        const uint8_t *s = p.b.text + b;
        const int64_t slot = sp_slot(b, d, p.slot_mul);
        uint16_t *out = p.stream + slot;
        const int cap = p.slot_mul * (n + 1);
        const int bom = (n >= 3 && s[0] == 0xEF && s[1] == 0xBB && s[2] == 0xBF) ? 3 : 0;   // also in bytes mode (FAUtf8Utils.cpp:330-335)
        int outc = 0, normc = 0, decoded = 0;
        uint32_t prev = 0; bool have_prev = false;
        for (int k = 0; k < p.prefix_n; ++k) {                                            // the (normalised) dummy prefix: uniform
            const uint32_t e = p.prefix[k];
            const bool ws = e == 0xFFFDu;
            const bool keep = !ws || !have_prev || !sp_delimish(prev, D);
            if (keep) { if (lane == 0 && outc < cap) { out[outc] = (uint16_t)(ws ? D : e); if (p.src_off) p.src_off[slot + outc] = -1; } ++outc; }
            prev = e; have_prev = true; ++normc;
        }
        bool bad = false;
        for (int pos = 0; pos < n; pos += 512) {
            const int q0 = pos + lane * 8;
            int nb = n - q0; nb = nb < 0 ? 0 : (nb > 8 ? 8 : nb);
            uint64_t own = 0;
            if (nb == 8) __builtin_memcpy(&own, s + q0, 8);
            else for (int k = 0; k < nb; ++k) own |= (uint64_t)s[q0 + k] << (8 * k);
            uint32_t vmask = nb >= 8 ? 0xFFu : ((1u << nb) - 1u);
            if (pos == 0 && lane == 0 && bom) vmask &= ~7u;
            // ---- which bytes start a character, and its map value into the stage
            uint32_t em;                                                                   // bytes of this lane that start a character
            const bool plain = p.use_bytes || !__any((own & 0x8080808080808080ull) != 0);
            if (plain && !p.use_bytes) {
                em = vmask;
#pragma unroll
                for (int k = 0; k < 8; ++k) stage[k * 64 + lane] = ((em >> k) & 1u) ? ascii_v[(uint32_t)(own >> (8 * k)) & 0x7f] : SP8_NOCHAR;
            } else if (p.use_bytes) {                                                      // every byte is a symbol (FAUtf8Utils.cpp:316-345)
                em = vmask;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t by = (uint32_t)(own >> (8 * k)) & 0xFFu;
                    uint32_t v = SP8_NOCHAR;
                    if ((em >> k) & 1u) v = by < 0x80u ? ascii_v[by] : cpmap_get(p.cpmap, (int)by);
                    stage[k * 64 + lane] = v;
                }
            } else {
                uint32_t nxt = __shfl_down((uint32_t)own, 1, 64);
                if (lane == 63) { nxt = 0; for (int k = 0; k < 3; ++k) if (q0 + 8 + k < n) nxt |= (uint32_t)s[q0 + 8 + k] << (8 * k); }
                const uint64_t h80 = own & 0x8080808080808080ull, h40 = (own << 1) & 0x8080808080808080ull;
                uint32_t m80 = 0, m40 = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) { m80 |= (uint32_t)((h80 >> (8 * k + 7)) & 1ull) << k; m40 |= (uint32_t)((h40 >> (8 * k + 7)) & 1ull) << k; }
                const uint32_t contm = m80 & ~m40 & vmask, leadm = m80 & m40 & vmask;
                em = vmask & ~contm;
#pragma unroll
                for (int k = 0; k < 8; ++k) stage[k * 64 + lane] = (((em & ~leadm) >> k) & 1u) ? ascii_v[(uint32_t)(own >> (8 * k)) & 0x7f] : SP8_NOCHAR;
                uint32_t cov = 0; bool e_any = false; uint32_t errm = 0;
                for (uint32_t lm = leadm; __any(lm != 0);) {                              // one lead byte per lane and trip
                    if (lm) {
                        const int k = __builtin_ctz(lm); lm &= lm - 1u;
                        const int q = q0 + k;
                        uint64_t w = own >> (8 * k);
                        if (k) w |= (uint64_t)nxt << (64 - 8 * k);
                        const uint32_t b0 = (uint32_t)w & 0xFF, b1 = (uint32_t)(w >> 8) & 0xFF, b2 = (uint32_t)(w >> 16) & 0xFF, b3 = (uint32_t)(w >> 24) & 0xFF;
                        int len, cp; bool er = false;
                        if ((b0 & 0xE0) == 0xC0) { len = 2; cp = (int)(b0 & 0x1F); }
                        else if ((b0 & 0xF0) == 0xE0) { len = 3; cp = (int)(b0 & 0x0F); }
                        else if ((b0 & 0xF8) == 0xF0) { len = 4; cp = (int)(b0 & 0x07); }
                        else { len = 1; cp = 0; er = true; }
                        if (q + len > n) er = true;
                        if (len >= 2) { if ((b1 & 0xC0) != 0x80) er = true; cp = (cp << 6) | (int)(b1 & 0x3F); }
                        if (len >= 3) { if ((b2 & 0xC0) != 0x80) er = true; cp = (cp << 6) | (int)(b2 & 0x3F); }
                        if (len >= 4) { if ((b3 & 0xC0) != 0x80) er = true; cp = (cp << 6) | (int)(b3 & 0x3F); }
                        const int need = cp <= 0x7F ? 1 : cp <= 0x7FF ? 2 : cp <= 0xFFFF ? 3 : cp <= 0x10FFFF ? 4 : 0;
                        if (need != len) er = true;
                        if ((cp & 0xFFFFF800) == 0xD800) er = true;
                        e_any |= er; if (er) errm |= 1u << k;
                        cov |= (((1u << len) - 1u) & ~1u) << k;
                        stage[k * 64 + lane] = er ? SP8_NOCHAR : (0x40000000u | (uint32_t)cp);     // the map gathers of all the lane's characters go out together below
                    }
                }
                uint32_t spill = __shfl_up(cov >> 8, 1, 64);
                if (lane == 0) {
                    spill = 0;
                    for (int back = 1; back <= 3 && pos - back >= bom; ++back) {
                        const uint32_t c0 = s[pos - back];
                        if ((c0 & 0xC0) == 0x80) continue;
                        const int len = (c0 & 0xE0) == 0xC0 ? 2 : (c0 & 0xF0) == 0xE0 ? 3 : (c0 & 0xF8) == 0xF0 ? 4 : 1;
                        if (c0 >= 0xC0 && len > back) spill = (1u << (len - back)) - 1u;
                        break;
                    }
                }
                if (contm & ~(cov | spill)) e_any = true;
                if (__any(e_any)) bad = true;
                em &= ~errm;                                                               // (the document yields nothing anyway)
            }
            wave_handoff();
            // ---- the lane's characters in order: elements, keep rule of the whitespace collapse (ws kept iff the element before it is
            //      neither ws nor the delimiter), counts
            uint32_t v[8]; int cnt_el = 0; uint32_t last = 0; bool any_multi = false;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = stage[k * 64 + lane];
            if (!plain) {                                                                  // decoded code points -> map values: up to eight two-level gathers in flight
                uint32_t pg[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { const bool pend = (v[k] >> 30) == 1u; pg[k] = pend ? (uint32_t)p.cpmap.l1[(v[k] & 0x1FFFFFu) >> 8] : 0u; }
#pragma unroll
                for (int k = 0; k < 8; ++k) { const bool pend = (v[k] >> 30) == 1u; if (pend) v[k] = p.cpmap.pages[pg[k] * 256u + (v[k] & 255u)]; }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (v[k] != SP8_NOCHAR) {
                    if (v[k] & 0x80000000u) { any_multi = true; const uint16_t *rec = p.multi_pool + (v[k] & 0x7FFFFFFFu); const int c = (int)rec[0]; cnt_el += c; if (c > 0) last = rec[c]; }
                    else { ++cnt_el; last = v[k]; }
                }
            }
            const int nchars = __popc(em);
            const unsigned long long m_has = __ballot(cnt_el > 0);
            const unsigned long long below = m_has & lanemask_lt();
            const int pl = below ? 63 - __clzll((long long)below) : 0;
            const uint32_t pl_last = __shfl(last, pl, 64);
            uint32_t pe = below ? pl_last : prev; bool hp = below ? true : have_prev;
            uint32_t keepm = 0; int kept = 0;                                              // common form: bit k = the (single) element of byte k is kept
            // (per lane: a lane that holds a character with 0 or several elements takes the general form of the two passes, the others --
            // nearly all -- the straight-line one; with NFKC charmaps about one chunk in two holds such a character somewhere)
            const bool multi = any_multi;
            if (!multi) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bool has = v[k] != SP8_NOCHAR;
                    const bool ws = v[k] == 0xFFFDu;
                    const bool keep = has && (!ws || !hp || !sp_delimish(pe, D));
                    keepm |= keep ? (1u << k) : 0u;
                    pe = has ? v[k] : pe; hp = hp || has;
                }
                kept = __popc(keepm);
            } else {
                for (int k = 0; k < 8; ++k) {
                    if (v[k] == SP8_NOCHAR) continue;
                    const bool mv = (v[k] & 0x80000000u) != 0;
                    const uint16_t *rec = p.multi_pool + (v[k] & 0x7FFFFFFFu);
                    const int c = mv ? (int)rec[0] : 1;
                    for (int j = 0; j < c; ++j) {
                        const uint32_t e = mv ? (uint32_t)rec[1 + j] : v[k];
                        const bool ws = e == 0xFFFDu;
                        if (!ws || !hp || !sp_delimish(pe, D)) ++kept;
                        pe = e; hp = true;
                    }
                }
            }
            const int inc = wave_incl_scan(kept);
            int idx = outc + inc - kept;
            if (!multi) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if ((keepm >> k) & 1u) {
                        if (idx < cap) { out[idx] = (uint16_t)(v[k] == 0xFFFDu ? D : v[k]); if (p.src_off) p.src_off[slot + idx] = q0 + k; }
                        ++idx;
                    }
                }
            } else {
                uint32_t pe2 = below ? pl_last : prev; bool hp2 = below ? true : have_prev;
                for (int k = 0; k < 8; ++k) {
                    if (v[k] == SP8_NOCHAR) continue;
                    const bool mv = (v[k] & 0x80000000u) != 0;
                    const uint16_t *rec = p.multi_pool + (v[k] & 0x7FFFFFFFu);
                    const int c = mv ? (int)rec[0] : 1;
                    for (int j = 0; j < c; ++j) {
                        const uint32_t e = mv ? (uint32_t)rec[1 + j] : v[k];
                        const bool ws = e == 0xFFFDu;
                        if (!ws || !hp2 || !sp_delimish(pe2, D)) { if (idx < cap) { out[idx] = (uint16_t)(ws ? D : e); if (p.src_off) p.src_off[slot + idx] = q0 + k; } ++idx; }
                        pe2 = e; hp2 = true;
                    }
                }
            }
            outc += __shfl(inc, 63, 64);
            normc += __shfl(wave_incl_scan(cnt_el), 63, 64);
            decoded += __shfl(wave_incl_scan(nchars), 63, 64);
            if (m_has) { const int hl = 63 - __clzll((long long)m_has); prev = __shfl(last, hl, 64); have_prev = true; }
            wave_handoff();                                                                // the stage is reused by the next chunk
        }
        int len = outc;
        if (len > 1 && have_prev && sp_delimish(prev, D)) --len;                          // tokdll:1491-1493
        if (bad || decoded <= 0) len = 0;                                                   // tokdll:1409-1411
        if (p.has_charmap && (normc <= 0 || normc > 2 * (n + 1))) len = 0;                // tokdll:1440-1444
        if (len > cap) len = 0;
        if (lane == 0) p.lens[d] = len;
    }
}

void launch_prep_sp(const SpPrepParams &p, hipStream_t s)
{
    int64_t blocks = (p.b.ndocs + 3) / 4;
    if (blocks > device_cus() * 16) blocks = device_cus() * 16;
    if (blocks < 1) blocks = 1;
    if (p.old_form) hipLaunchKernelGGL(k_prep_sp, dim3((unsigned)blocks), dim3(256), 0, s, p);      // the byte-per-lane form (A/B runs: BfSetVariant bit 0x80)
    // shipped: compiled for eight waves per SIMD (64 VGPRs, 8 bytes of scratch).  Config 4, 10 M documents (profiles/r04_u_prep_waves.txt): six waves
    // (the compiler's own choice, 73 VGPRs) 31.7 ms, seven (72 VGPRs) 30.5, eight 27.3.  BfSetVariant bits 24..27 = 6 / 7: the other instances
#ifdef BF_EXPERIMENTS
    else if (p.waves == 6) hipLaunchKernelGGL(k_prep_sp8<0>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    else if (p.waves == 7) hipLaunchKernelGGL(k_prep_sp8<7>, dim3((unsigned)blocks), dim3(256), 0, s, p);
#endif
    else hipLaunchKernelGGL(k_prep_sp8<8>, dim3((unsigned)blocks), dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------
// k_seg_sp: one document per lane, the sequential programs of bf_seg.h with lane-private global scratch.
// (First correct version: static assignment, compiler-managed divergence.)
// ------------------------------------------------------------------------------------------
// length bucketing: key = min(len / 4, 1023)
__device__ __forceinline__ int sp_len_bucket(int len) { int k = len >> 2; return k > 1023 ? 1023 : k; }

constexpr int SP_SORT_ITEMS = 8;          // documents per thread in the counting-sort kernels

// per-block histogram in LDS, one global atomic per non-empty bucket and block (1.25 M threads hammering 1024 global counters
// cost 0.9 ms per kernel)
__global__ __launch_bounds__(256) void k_sp_hist(SpSegParams p)
{
    __shared__ unsigned int h[1024];
    for (int k = threadIdx.x; k < 1024; k += 256) h[k] = 0;
    __syncthreads();
    const int64_t d0 = (int64_t)blockIdx.x * 256 * SP_SORT_ITEMS;
    for (int it = 0; it < SP_SORT_ITEMS; ++it) {
        const int64_t d = d0 + (int64_t)it * 256 + threadIdx.x;
        if (d < p.b.ndocs) atomicAdd(&h[sp_len_bucket(p.lens[d])], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 1024; k += 256) if (h[k]) atomicAdd(&p.hist[k], h[k]);
}
__global__ __launch_bounds__(1024) void k_sp_hist_scan(SpSegParams p)
{
    // exclusive scan of the 1024 bucket counts, longest documents first (they start first: better tail)
    __shared__ unsigned int sh[1024];
    const int t = threadIdx.x;
    sh[t] = p.hist[1023 - t];
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { unsigned int v = t >= o ? sh[t - o] : 0; __syncthreads(); sh[t] += v; __syncthreads(); }
    p.hist[1024 + (1023 - t)] = t ? sh[t - 1] : 0;      // cursor of bucket (1023 - t)
}
__global__ __launch_bounds__(256) void k_sp_scatter(SpSegParams p)
{
    // the block counts its documents per bucket, reserves one range per non-empty bucket, then places the documents
    __shared__ unsigned int cnt[1024], base[1024];
    for (int k = threadIdx.x; k < 1024; k += 256) cnt[k] = 0;
    __syncthreads();
    const int64_t d0 = (int64_t)blockIdx.x * 256 * SP_SORT_ITEMS;
    int bucket[SP_SORT_ITEMS];
#pragma unroll
    for (int it = 0; it < SP_SORT_ITEMS; ++it) {
        const int64_t d = d0 + (int64_t)it * 256 + threadIdx.x;
        bucket[it] = d < p.b.ndocs ? sp_len_bucket(p.lens[d]) : -1;
        if (bucket[it] >= 0) atomicAdd(&cnt[bucket[it]], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 1024; k += 256) { base[k] = cnt[k] ? atomicAdd(&p.hist[1024 + k], cnt[k]) : 0u; cnt[k] = 0; }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < SP_SORT_ITEMS; ++it) {
        if (bucket[it] >= 0) {
            const unsigned int pos = base[bucket[it]] + atomicAdd(&cnt[bucket[it]], 1u);
            p.perm[pos] = (int32_t)(d0 + (int64_t)it * 256 + threadIdx.x);
        }
    }
}

// Unigram-LM: the sequential program per lane (documents in length order)
__global__ __launch_bounds__(64) void k_seg_unigram(SpSegParams p)
{
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= p.b.ndocs) return;
    const int64_t d = p.perm[i];
    const int64_t slot = sp_slot(p.b.doc_off[d], d, p.slot_mul);
    ClsWin cls_at; cls_at.init(p.stream, slot);
    IdOutDirect out{p.ids_tmp + slot, p.span_tmp ? p.span_tmp + 2 * slot : nullptr};
    p.counts[d] = seg_unigram_doc(p.S, cls_at, p.lens[d], p.best + slot, out, p.max_ids, p.unk);
}

// BPE phase A: collect arcs, one document per lane
__global__ __launch_bounds__(64) void k_bpe_collect(SpSegParams p)
{
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= p.b.ndocs) return;
    const int64_t d = p.perm[i];
    const int64_t b = p.b.doc_off[d];
    const int64_t slot = sp_slot(b, d, p.slot_mul);
    const int L = p.lens[d];
    const int64_t nbytes = p.b.doc_off[d + 1] - b;
    const int arc_cap = 6 * (int)(p.slot_mul * (nbytes + 1)) + 32;
    ClsWin cls_at; cls_at.init(p.stream, slot);
    p.narcs[d] = L > 0 ? seg_bpe_collect(p.S, cls_at, L, p.arcs + 6 * slot + 32 * d, arc_cap, p.unk) : 0;
}

// the same for the documents the arc-free k_bpe_fused handed over (narcs == BPE_COLLECT on the fallback list)
__global__ __launch_bounds__(64) void k_bpe_collect_list(SpSegParams p)
{
    const unsigned int nfb = *p.fb_count;
    for (unsigned int idx = blockIdx.x * 64u + threadIdx.x; idx < nfb; idx += gridDim.x * 64u) {
        const int64_t d = p.fb_list[idx];
        if (p.narcs[d] != -3 /* BPE_COLLECT */) continue;
        const int64_t b = p.b.doc_off[d];
        const int64_t slot = sp_slot(b, d, p.slot_mul);
        const int arc_cap = 6 * (int)(p.slot_mul * (p.b.doc_off[d + 1] - b + 1)) + 32;
        ClsWin cls_at; cls_at.init(p.stream, slot);
        p.narcs[d] = seg_bpe_collect(p.S, cls_at, p.lens[d], p.arcs + 6 * slot + 32 * d, arc_cap, p.unk);
    }
}

// BPE phase B1: sort the arcs of one document per 256-thread block (bitonic sort in LDS on the integer keys of
// bf_seg.h; the order is total, so the result equals the reference's qsort).  Documents with more arcs than the
// LDS holds are sorted in place in global memory by the same block (merge exchange).
constexpr int BPE_NMAX = 4096;
__global__ __launch_bounds__(256) void k_bpe_sort(SpSegParams p)
{
    __shared__ uint32_t k_hi[BPE_NMAX];
    __shared__ uint64_t k_lo[BPE_NMAX];
    __shared__ uint32_t k_val[BPE_NMAX];
    __shared__ unsigned long long s_doc;
    const bool merges = p.S.kind == SG_KIND_BPE_MERGES;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_doc = atomicAdd(p.next_doc, 1ull);
        __syncthreads();
        const unsigned long long i = s_doc;
        if (i >= (p.fb_list ? (unsigned long long)*p.fb_count : (unsigned long long)p.b.ndocs)) break;
        const int64_t d = p.fb_list ? p.fb_list[i] : p.perm[i];
        const int n = p.narcs[d];
        if (n <= 1) continue;
        SegArc *arcs = p.arcs + 6 * sp_slot(p.b.doc_off[d], d, p.slot_mul) + 32 * d;
        if (n > BPE_NMAX) {
            // too many arcs for the LDS: Batcher's merge exchange (Knuth 5.2.2 M) in place in global memory, by the whole block.
            // It sorts any n without padding and the compare-exchanges of one step are independent of each other.
            int t = 0; while ((1 << t) < n) ++t;
            for (int pp = 1 << (t - 1); pp > 0; pp >>= 1) {
                int q = 1 << (t - 1), r = 0, dd = pp;
                while (dd > 0) {
                    for (int i = threadIdx.x; i < n - dd; i += 256) {
                        if ((i & pp) == r) {
                            const SegArc a = arcs[i], b = arcs[i + dd];
                            const uint32_t ah = sg_key_hi(a, merges), bh = sg_key_hi(b, merges);
                            const uint64_t al = sg_key_lo(a), bl = sg_key_lo(b);
                            if (ah > bh || (ah == bh && al > bl)) { arcs[i] = b; arcs[i + dd] = a; }
                        }
                    }
                    __threadfence_block();
                    __syncthreads();
                    dd = q - pp; q >>= 1; r = pp;
                }
            }
            continue;
        }
        int n2 = 256; while (n2 < n) n2 <<= 1;
        for (int k = threadIdx.x; k < n2; k += 256) {
            if (k < n) { const SegArc a = arcs[k]; k_hi[k] = sg_key_hi(a, merges); k_lo[k] = sg_key_lo(a); k_val[k] = (uint32_t)a.end; }
            else { k_hi[k] = 0xFFFFFFFFu; k_lo[k] = ~0ull; k_val[k] = 0; }
        }
        __syncthreads();
        for (int k2 = 2; k2 <= n2; k2 <<= 1) {
            for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                for (int t = threadIdx.x; t < (n2 >> 1); t += 256) {
                    const int lo_i = ((t / j2) * (j2 << 1)) + (t % j2), hi_i = lo_i + j2;
                    const bool up = (lo_i & k2) == 0;
                    const uint32_t ah = k_hi[lo_i], bh = k_hi[hi_i];
                    const uint64_t al = k_lo[lo_i], bl = k_lo[hi_i];
                    const bool a_gt_b = ah > bh || (ah == bh && al > bl);
                    if (a_gt_b == up) {
                        k_hi[lo_i] = bh; k_hi[hi_i] = ah; k_lo[lo_i] = bl; k_lo[hi_i] = al;
                        const uint32_t av = k_val[lo_i]; k_val[lo_i] = k_val[hi_i]; k_val[hi_i] = av;
                    }
                }
                __syncthreads();
            }
        }
        for (int k = threadIdx.x; k < n; k += 256) {
            SegArc a; a.start = (int32_t)(uint32_t)k_lo[k]; a.end = (int32_t)k_val[k];
            a.id = (int32_t)((uint32_t)(k_lo[k] >> 32) ^ 0x80000000u); a.rank_bits = 0;
            arcs[k] = a;
        }
    }
}

// BPE phase B2: apply the sorted arcs and emit ids, one document per lane
__global__ __launch_bounds__(64) void k_bpe_apply(SpSegParams p)
{
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= p.b.ndocs) return;
    const int64_t d = p.perm[i];
    const int64_t slot = sp_slot(p.b.doc_off[d], d, p.slot_mul);
    const int L = p.lens[d];
    const int n = p.narcs[d];
    IdOutDirect out{p.ids_tmp + slot, p.span_tmp ? p.span_tmp + 2 * slot : nullptr};
    int r = 0;
    if (n < 0) { atomicOr(p.status, 2); }
    else if (L > 0) {
        r = seg_bpe_finish(p.S, L, p.arcs + 6 * slot + 32 * d, n, p.tos + slot, p.idsv + slot, p.inter + slot, out, p.max_ids, p.unk, true);
        if (r < 0) { atomicOr(p.status, 2); r = 0; }
    }
    p.counts[d] = r;
}

constexpr int BPE_DONE = -2;     // narcs[d]: the document was finished by k_bpe_fused
constexpr int BPE_COLLECT = -3;  // narcs[d]: the arc-free k_bpe_fused hands the document to the full path, which collects its arcs first (k_bpe_collect_list)

// pointer of lane `o` (uniform) to every lane, through v_readlane
__device__ __forceinline__ const void *bcast_ptr(const void *q, int o)
{
    const unsigned long long v = (unsigned long long)q;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, o), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), o);
    return (const void *)(((unsigned long long)hi << 32) | lo);
}

// BPE phase B2, divergence-free form: lane per document, persistent lanes, one sorted arc (or one emitted token) per trip.
// State of the reference's three work arrays (..._bpe_t.h:258-296) kept as two lane-private bitmaps instead of 9 bytes per
// position: `inter` (position is inside an applied arc) and `applied` (an arc was applied from this position); the end of
// the token that starts at s is the position before the next non-interior one (equal to pTos[s]: an applied arc keeps its
// end a boundary for as long as its start is one), so pTos is not stored at all and nothing is initialised per position.
// Bitmap words of document d start at word (slot_d >> 5) + d of each bitmap (room for its capacity + 1 bits).
__global__ __launch_bounds__(64) void k_bpe_apply_flat(SpSegParams p)
{
    enum { M_NEED = 0, M_ARCS = 1, M_EMIT = 2, M_EXIT = 3 };
    int mode = M_NEED;
    int64_t doc = 0; const SegArc *arcs = nullptr; uint32_t *bmi = nullptr, *bma = nullptr; int32_t *idsv = nullptr, *ids = nullptr, *spans = nullptr;
    int L = 0, n = 0, k = 0, start = 0, cnt = 0; bool err = false;
    SegArc cur; cur.start = cur.end = cur.id = 0; cur.rank_bits = 0;
    for (unsigned long long trip = 0; trip < (1ull << 34); ++trip) {
        const unsigned long long m_need = __ballot(mode == M_NEED);
        if (m_need) {
            const unsigned long long m_busy = __ballot(mode == M_ARCS || mode == M_EMIT);
            if (__popcll(m_need) >= 8 || m_busy == 0) {
                if (mode == M_NEED) {
                    const int c = __popcll(m_need);
                    const int leader = __ffsll((long long)m_need) - 1;
                    unsigned long long base = 0;
                    if (lane_id() == leader) base = atomicAdd(p.next_doc, (unsigned long long)c);
                    base = __shfl(base, leader, 64);
                    const int64_t idx = (int64_t)base + __popcll(m_need & lanemask_lt());
                    if (idx >= (p.fb_list ? (int64_t)*p.fb_count : p.b.ndocs)) mode = M_EXIT;
                    else {
                        doc = p.fb_list ? p.fb_list[idx] : p.perm[idx];
                        const int64_t slot = sp_slot(p.b.doc_off[doc], doc, p.slot_mul);
                        L = p.lens[doc]; n = p.narcs[doc];
                        if (n == BPE_DONE) {}                                        // finished by k_bpe_fused
                        else if (n < 0) p.counts[doc] = 0;                           // arc reserve exceeded: k_bpe_big takes the document
                        else if (L <= 0) p.counts[doc] = 0;
                        else {
                            arcs = p.arcs + 6 * slot + 32 * doc;
                            const int64_t w0 = (slot >> 5) + doc;
                            bmi = (uint32_t *)p.tos + w0; bma = (uint32_t *)p.tos + p.bm_words + w0;
                            idsv = p.idsv + slot; ids = p.ids_tmp + slot; spans = p.span_tmp ? p.span_tmp + 2 * slot : nullptr;
                            const int nw = (L + 32) >> 5;                      // bits 0 .. L
                            for (int q = 0; q < nw; ++q) { bmi[q] = 0; bma[q] = 0; }
                            k = 0; start = 0; cnt = 0; err = false;
                            if (n > 0) { cur = arcs[0]; mode = M_ARCS; } else mode = M_EMIT;
                        }
                    }
                }
                if (__ballot(mode != M_EXIT) == 0) break;
            }
        }
        if (mode == M_ARCS) {                                           // ..._bpe_t.h:274-296, arc k of the sorted list
            const int s_ = cur.start, e_ = cur.end, e1 = e_ + 1;
            const uint32_t ws = bmi[s_ >> 5], we = bmi[e1 >> 5];
            const SegArc nxt = arcs[k + 1 < n ? k + 1 : k];              // the next arc travels with the bitmap words
            const bool free_s = ((ws >> (s_ & 31)) & 1u) == 0;
            const bool free_e = e1 == L || ((we >> (e1 & 31)) & 1u) == 0;
            if (free_s && free_e) {
                for (int w = (s_ + 1) >> 5; w <= (e_ >> 5) && s_ < e_; ++w) {   // interior bits s+1 .. e
                    const int lo = w == ((s_ + 1) >> 5) ? ((s_ + 1) & 31) : 0, hi = w == (e_ >> 5) ? (e_ & 31) : 31;
                    bmi[w] |= (hi == 31 ? ~0u : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
                }
                idsv[s_] = cur.id;
                bma[s_ >> 5] |= 1u << (s_ & 31);
            }
            cur = nxt;
            if (++k >= n) mode = M_EMIT;
        } else if (mode == M_EMIT) {                                    // ..._bpe_t.h:299-313 + tokdll:1512-1516
            const bool applied = (bma[start >> 5] >> (start & 31)) & 1u;
            int e_ = start, id = p.unk;
            if (applied) {
                id = idsv[start];
                int q = start + 1;                                       // first non-interior position after start (bit L is never set)
                for (;;) {
                    const uint32_t inv = ~bmi[q >> 5] & ~((1u << (q & 31)) - 1u);
                    if (inv) { q = (q & ~31) + (__ffs((int)inv) - 1); break; }
                    q = (q & ~31) + 32;
                }
                e_ = q - 1;
            } else if (start > 0) err = true;                           // pTos[start] == 0 < start: the reference would walk backwards
            else e_ = 0;                                                 // pTos[0] == 0, pIds[0] == UnkId
            if (cnt < p.max_ids) { ids[cnt] = id + p.S.id_offset; if (spans) { spans[2 * cnt] = start; spans[2 * cnt + 1] = e_; } }
            ++cnt;
            start = e_ + 1;
            if (start >= L || err) {
                if (err) { atomicOr(p.status, 2); cnt = 0; }
                p.counts[doc] = cnt < p.max_ids ? cnt : p.max_ids;
                mode = M_NEED;
            }
        }
    }
    if (mode != M_EXIT) atomicOr(p.status, 2);      // trip limit hit: never expected
}

// ------------------------------------------------------------------------------------------
// k_bpe_fused: BPE (all three flavours) in ONE pass for the common case.
//
// The reference collects every arc of the document, sorts them by merge priority and applies them in that order
// (..._bpe_t.h:151-313).  Whether an arc [s, e] is applied depends only on the interior marks of positions s and
// e + 1, and those are only ever set by arcs that cover them.  So at a CUT -- a start position s that no earlier arc
// reaches (s > max end so far) -- the arc list splits into independent SEGMENTS, each of which can be sorted and
// applied on its own, in position order, as soon as it is complete.  With the bpe-opt whole-token shortcut almost every
// segment of real text is a single arc (= one token, always applied); the others are one out-of-vocabulary word wide.
//
//  * lane per document, persistent lanes, one trie transition per trip (the arc list itself is built exactly like
//    seg_bpe_collect builds it, in the same global buffer, so that a document can still fall back to the full path);
//  * a 1-arc segment is emitted by its lane; a longer one (<= 64 arcs, <= 63 positions) waits for a vote and is then
//    solved by the WHOLE WAVE: lane j takes arc j, ranks come from an all-pairs key comparison through shuffles,
//    the arcs are applied in rank order against a 64-bit interior mask held in a uniform register, and the tokens
//    are emitted by the lanes that sit on the segment's non-interior positions (ballot prefix);
//  * a segment can only be closed once the next start is known not to extend it (an unknown position merges into a
//    preceding unknown arc, ..._bpe_t.h:217-225);
//  * anything larger marks the document for the full path (k_bpe_sort + k_bpe_apply_flat redo it from its arc list).
// narcs[d] on exit: -2 = finished here, -1 = arc capacity exceeded (loud error), >= 0 = arc count for the full path.
// ------------------------------------------------------------------------------------------
// lane-local arc window of the open segment (LDS, structure-of-arrays): entry = sort key, ascending == the reference's
// comparator order.  plain BPE: [id:20 | start - seg_begin:6 | end - seg_begin:6]; with merges: [rank key:32 | the same 32 bits]
template <bool MERGES> struct BpeLocal { typedef uint32_t Key; enum { CAP = 32 }; };
template <> struct BpeLocal<true> { typedef uint64_t Key; enum { CAP = 16 }; };
constexpr int BPE_LOCAL_ID_BITS = 20;

// mirror arc `a` (index k of the arc list) in the lane's LDS window; `local` drops to false when the open segment no
// longer fits (more than CAP arcs, a position more than 62 past the segment start, an id that needs more than 20 bits)
// returns false when the arc does not fit (the window holds CAP arcs of the open segment, positions up to 62 past its
// start, ids below 2^20)
template <class LKey, int LCAP>
__device__ __forceinline__ bool lds_put_t(LKey *ring, int k, int seg_begin, const SegArc &a, int seg_first, bool merges)
{
    const int s_rel = a.start - seg_begin, e_rel = a.end - seg_begin;
    if (k - seg_first >= LCAP || e_rel > 62 || s_rel < 0 || (uint32_t)a.id >= (1u << BPE_LOCAL_ID_BITS)) return false;
    const uint32_t pk = ((uint32_t)a.id << 12) | ((uint32_t)s_rel << 6) | (uint32_t)e_rel;
    LKey key = (LKey)pk;
    if (sizeof(LKey) == 8) key |= (LKey)((unsigned long long)sg_key_hi(a, merges) << 32);
    ring[(k - seg_first) * 64] = key;
    return true;
}
#define lds_put(ring, k, seg_begin, a, seg_first) lds_put_t<LKey, LCAP>(ring, k, seg_begin, a, seg_first, merges)

// ARCS = false (the default when the lane-local window can be used): the arc list is written to global memory only where somebody
// reads it.  It used to be 16 bytes per arc for every document (29.8 GB per 1 M documents of BASELINE config 3,
// profiles/r02_final_config3.txt) for the sake of the segments that outgrow the lane's LDS window (the wave-cooperative solve reads
// them from global memory) and of the few documents (17 of 200,000) that fall back.  Now a lane starts writing when its open segment
// stops fitting the window (`spill`: the arcs the window holds are written out once, the later ones as they come) and stops at the
// next segment that fits; a document that falls back goes on the list with narcs = BPE_COLLECT and the full path collects its arcs
// itself (k_bpe_collect_list).
template <bool MERGES, bool LOCAL, bool ARCS = true>
__global__ __launch_bounds__(64) void k_bpe_fused(SpSegParams p)
{
    static_assert(ARCS || LOCAL, "the arc-free form needs the lane-local window");
    bool spill = ARCS;                                  // arcs of the open segment (and of the current start) go to global memory
    typedef typename BpeLocal<MERGES>::Key LKey;
    constexpr int LCAP = BpeLocal<MERGES>::CAP;
    extern __shared__ unsigned char bpe_lds_raw[];
    LKey *ring = (LKey *)bpe_lds_raw + lane_id();          // entry of arc index k at ring[(k - seg_first) * 64]
    const int lthresh = p.tune ? p.tune : 16;
    // writes the arcs the LDS window holds ([first, first + n), keyed relative to `base`) to the document's arc list, in the reference's form
    auto spill_window = [&](SegArc *dst, int first, int n, int base, bool mg) {
        for (int t = 0; t < n; ++t) {
            const LKey key = ring[t * 64];
            const uint32_t pk = (uint32_t)key;
            SegArc a; a.start = base + (int)((pk >> 6) & 63u); a.end = base + (int)(pk & 63u); a.id = (int32_t)(pk >> 12); a.rank_bits = 0;
            if (mg && sizeof(LKey) == 8) { const uint32_t asc = ~(uint32_t)((unsigned long long)key >> 32); a.rank_bits = (asc & 0x80000000u) ? (asc & 0x7FFFFFFFu) : ~asc; }   // inverse of sg_key_hi
            dst[first + t] = a;
        }
    };
    bool seg_local = LOCAL, cs_ok = true;                   // the open segment / the arcs of the current start are fully mirrored in the LDS window
    enum { M_NEED = 0, M_WALK = 1, M_SOLVE = 2, M_POST = 3, M_EXIT = 4, M_LSOLVE = 5 };
    const bool merges = MERGES;
    const bool fast = p.S.kind == SG_KIND_BPE_OPT || merges;              // m_fFastBpe (..._bpe_t.h:110, ..._with_merges_t.h:113)
    const int lane = lane_id();
    int mode = M_NEED;
    int64_t doc = 0; SegArc *arcs = nullptr; int32_t *ids = nullptr, *spans = nullptr;
    ClsWin cls_at; cls_at.init(p.stream, 0);
    int L = 0, arc_cap = 0, start = 0, i = 0, sum = 0, narcs = 0, count_at_start = 0, ff = 0, cnt = 0;
    uint32_t state = 0; bool unknown = true, token_start = false, fallback = false, closing_last = false;
    int last_id = 0;                                   // id of arcs[narcs - 1]
    int seg_first = 0, seg_n = 0, seg_maxend = -1;     // current segment: arcs [seg_first, seg_first + seg_n), reach seg_maxend
    int one_s = 0, one_e = 0, one_id = 0;              // copy of arcs[seg_first] (start, end, id)
    int cs_first_e = 0, cs_first_id = 0, cs_last_e = 0;   // arcs of the current start: first (end, id) and largest end
    for (unsigned long long trip = 0; trip < (1ull << 34); ++trip) {
        const unsigned long long m_need = __ballot(mode == M_NEED);
        if (m_need) {
            const unsigned long long m_busy = __ballot(mode == M_WALK || mode == M_SOLVE || mode == M_POST || mode == M_LSOLVE);
            if (__popcll(m_need) >= 8 || m_busy == 0) {
                if (mode == M_NEED) {
                    const int c = __popcll(m_need);
                    const int leader = __ffsll((long long)m_need) - 1;
                    unsigned long long base = 0;
                    if (lane == leader) base = atomicAdd(p.next_doc, (unsigned long long)c);
                    base = __shfl(base, leader, 64);
                    const int64_t idx = (int64_t)base + __popcll(m_need & lanemask_lt());
                    if (idx >= p.b.ndocs) mode = M_EXIT;
                    else {
                        doc = p.perm[idx];
                        const int64_t b = p.b.doc_off[doc];
                        const int64_t slot = sp_slot(b, doc, p.slot_mul);
                        L = p.lens[doc];
                        arc_cap = 6 * (int)(p.slot_mul * (p.b.doc_off[doc + 1] - b + 1)) + 32;
                        arcs = p.arcs + 6 * slot + 32 * doc; ids = p.ids_tmp + slot; spans = p.span_tmp ? p.span_tmp + 2 * slot : nullptr;
                        cls_at.init(p.stream, slot);
                        if (L <= 0) { p.counts[doc] = 0; p.narcs[doc] = BPE_DONE; }
                        else {
                            start = 0; i = 0; state = p.S.initial; sum = 0; unknown = true; narcs = 0; count_at_start = 0; ff = 0; cnt = 0;
                            token_start = cls_at(0) == p.S.cls_delim; fallback = false; closing_last = false; last_id = 0;
                            seg_first = 0; seg_n = 0; seg_maxend = -1; seg_local = LOCAL; cs_ok = true; spill = ARCS;
                            mode = M_WALK;
                        }
                    }
                }
                if (__ballot(mode != M_EXIT) == 0) break;
            }
        }
        // ---- wave-cooperative solve of the waiting segments (their lanes continue at M_POST)
        {
            unsigned long long ms = __ballot(mode == M_SOLVE);
            if (ms && (__popcll(ms) >= 16 || __ballot(mode == M_WALK || mode == M_POST || mode == M_LSOLVE) == 0)) {
                __threadfence_block();                                    // the arcs the lanes appended are visible to the wave
                while (ms) {
                    const int o = __builtin_amdgcn_readfirstlane(__ffsll((long long)ms) - 1); ms &= ms - 1;
                    const SegArc *oa = (const SegArc *)bcast_ptr(arcs + seg_first, o);
                    const int n = __builtin_amdgcn_readlane(seg_n, o), begin = __builtin_amdgcn_readlane(one_s, o), Ls = __builtin_amdgcn_readlane(seg_maxend, o) - begin + 1;
                    const int ocnt = __builtin_amdgcn_readlane(cnt, o);
                    int32_t *oids = (int32_t *)bcast_ptr(ids, o), *ospans = (int32_t *)bcast_ptr(spans, o);
                    SegArc a; a.start = begin; a.end = begin; a.id = 0; a.rank_bits = 0;
                    if (lane < n) a = oa[lane];
                    // all-pairs ranking through v_readlane (uniform source lane): the order is total; equal keys broken by index
                    const uint32_t khi = sg_key_hi(a, merges); const uint64_t klo = sg_key_lo(a);
                    const uint32_t kl0 = (uint32_t)klo, kl1 = (uint32_t)(klo >> 32);
                    int rank = 0;
                    for (int t = 0; t < n; ++t) {
                        const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)kl1, t), t0 = (uint32_t)__builtin_amdgcn_readlane((int)kl0, t);
                        const uint32_t th = merges ? (uint32_t)__builtin_amdgcn_readlane((int)khi, t) : 0u;
                        const bool lt = th < khi || (th == khi && (t1 < kl1 || (t1 == kl1 && (t0 < kl0 || (t0 == kl0 && t < lane)))));
                        rank += lt ? 1 : 0;
                    }
                    if (lane >= n) rank = lane;                           // ranks form a permutation of the 64 lanes
                    // lane r receives the arc of rank r
                    const int sa_start = __builtin_amdgcn_ds_permute(rank << 2, a.start), sa_end = __builtin_amdgcn_ds_permute(rank << 2, a.end),
                              sa_id = __builtin_amdgcn_ds_permute(rank << 2, a.id);
                    unsigned long long inter = 0;                         // bit q: position begin + q is inside an applied arc (bit Ls is never set)
                    int pos_id = p.unk; bool pos_applied = false;
                    for (int r = 0; r < n; ++r) {                         // ..._bpe_t.h:274-296 in sorted order, on uniform registers
                        const int s_rel = __builtin_amdgcn_readlane(sa_start, r) - begin, e_rel = __builtin_amdgcn_readlane(sa_end, r) - begin;
                        if (!((inter >> s_rel) & 1ull) && !((inter >> (e_rel + 1)) & 1ull)) {
                            if (e_rel > s_rel) inter |= ((1ull << (e_rel + 1)) - 1ull) & ~((1ull << (s_rel + 1)) - 1ull);
                            const int id_r = __builtin_amdgcn_readlane(sa_id, r);
                            if (lane == s_rel) { pos_id = id_r; pos_applied = true; }
                        }
                    }
                    // tokens = the non-interior positions of the segment, in order (..._bpe_t.h:299-313 + tokdll:1512-1516)
                    const bool is_tok = lane < Ls && !((inter >> lane) & 1ull);
                    const unsigned long long mt = __ballot(is_tok);
                    const bool bad = is_tok && !pos_applied && (begin + lane) > 0;     // pTos[start] == 0 < start: the reference walks backwards here
                    const int kk = ocnt + __popcll(mt & lanemask_lt());
                    if (is_tok && kk < p.max_ids) {
                        oids[kk] = pos_id + p.S.id_offset;
                        if (ospans) {
                            const unsigned long long above = (mt | (1ull << Ls)) >> (lane + 1);
                            ospans[2 * kk] = begin + lane; ospans[2 * kk + 1] = pos_applied ? begin + lane + __ffsll((long long)above) - 1 : begin + lane;
                        }
                    }
                    const bool any_bad = __any(bad);
                    if (lane == o) { cnt += __popcll(mt); if (any_bad) fallback = true; mode = M_POST; }
                }
            }
        }

        // ---- lane-local solve: every waiting lane sorts (selection by key) and applies the arcs of ITS segment from its LDS
        //      window against a 64-bit interior mask, then emits the tokens; up to 64 segments per pass
        if (LOCAL) {
            const unsigned long long ml = __ballot(mode == M_LSOLVE);
            if (ml && (__popcll(ml) >= lthresh || __ballot(mode == M_WALK || mode == M_POST) == 0)) {
                if (mode == M_LSOLVE) {
                    // Batcher's odd-even merge sort, in place in the LDS window: every compare-exchange moves the smaller key to the
                    // lower index, so entries at or past seg_n (the arcs of the current start) act as +infinity and are left alone;
                    // all offsets are compile-time constants, the network of one phase runs with its LDS reads in flight together
#pragma unroll
                    for (int pp = 1; pp < LCAP; pp <<= 1) {
#pragma unroll
                        for (int kq = pp; kq >= 1; kq >>= 1) {
#pragma unroll
                            for (int j = kq % pp; j <= LCAP - 1 - kq; j += 2 * kq) {
#pragma unroll
                                for (int i2 = 0; i2 <= (kq - 1 < LCAP - j - kq - 1 ? kq - 1 : LCAP - j - kq - 1); ++i2) {
                                    if ((i2 + j) / (2 * pp) == (i2 + j + kq) / (2 * pp)) {
                                        const int lo_i = i2 + j, hi_i = i2 + j + kq;
                                        if (hi_i < seg_n) {
                                            const LKey x = ring[lo_i * 64], y = ring[hi_i * 64];
                                            if (y < x) { ring[lo_i * 64] = y; ring[hi_i * 64] = x; }
                                        }
                                    }
                                }
                            }
                        }
                    }
                    unsigned long long inter = 0;                         // bit q: position seg_begin + q is inside an applied arc
#pragma unroll
                    for (int t = 0; t < LCAP; ++t) {                      // ..._bpe_t.h:274-296 in sorted order
                        if (t < seg_n) {
                            const uint32_t pk = (uint32_t)ring[t * 64]; const int s_rel = (int)((pk >> 6) & 63u), e_rel = (int)(pk & 63u);
                            if (!((inter >> s_rel) & 1ull) && !((inter >> (e_rel + 1)) & 1ull) && e_rel > s_rel)
                                inter |= ((1ull << (e_rel + 1)) - 1ull) & ~((1ull << (s_rel + 1)) - 1ull);
                        }
                    }
                    // the token that starts at a non-interior position s is the arc [s, next non-interior position - 1]; an arc
                    // shows exactly that pattern iff it was applied and never swallowed (arcs are unique by their span)
                    const int Ls = seg_maxend - one_s + 1;
                    const unsigned long long bound = ~inter & ((1ull << Ls) - 1ull);
                    int found = 0;
#pragma unroll
                    for (int t = 0; t < LCAP; ++t) {                      // ..._bpe_t.h:299-313 + tokdll:1512-1516
                        if (t < seg_n) {
                            const uint32_t pk = (uint32_t)ring[t * 64];
                            const int s_rel = (int)((pk >> 6) & 63u), e_rel = (int)(pk & 63u);
                            const unsigned long long body = e_rel > s_rel ? (((1ull << (e_rel + 1)) - 1ull) & ~((1ull << (s_rel + 1)) - 1ull)) : 0ull;
                            if (!((inter >> s_rel) & 1ull) && !((inter >> (e_rel + 1)) & 1ull) && (inter & body) == body) {
                                const int kk = cnt + __popcll(bound & ((1ull << s_rel) - 1ull));
                                if (kk < p.max_ids) { ids[kk] = (int)(pk >> 12) + p.S.id_offset; if (spans) { spans[2 * kk] = one_s + s_rel; spans[2 * kk + 1] = one_s + e_rel; } }
                                ++found;
                            }
                        }
                    }
                    const int ntok = __popcll(bound);
                    if (found != ntok) fallback = true;                   // a token start without an applied arc: the full path reproduces the reference there
                    else cnt += ntok;
                    mode = M_POST;
                }
            }
        }
        bool post = mode == M_POST, post_cut = true, post_merged = false;
        if (mode == M_WALK) {
            bool walk_ends = true, over = false;                           // over: the arc buffer is full (loud error, the document is dropped)
            if (!closing_last) {
                // ---- one transition of the walk from `start` (seg_bpe_collect, ..._bpe_t.h:151-232)
                const uint64_t e = sg_lookup(p.S, state, cls_at(i));
                walk_ends = e == SG_MISS;
                if (!walk_ends) {
                    state = (uint32_t)((e >> SG_NEXT_SHIFT) & SG_NEXT_MASK);
                    sum += (int)(e >> SG_OW_SHIFT);
                    if (e & SG_FINAL) {
                        const SegInfo r = p.S.info[sum];
                        const bool whole = fast && token_start && ((i < L - 1) ? cls_at(i + 1) == p.S.cls_delim : true) && count_at_start < narcs;
                        SegArc a; a.start = start; a.end = i; a.id = r.id; a.rank_bits = r.score_bits;
                        if (!whole) {
                            if (narcs >= arc_cap) over = true;
                            else {
                                if (narcs == count_at_start) { cs_first_e = i; cs_first_id = r.id; }
                                if (LOCAL) cs_ok = cs_ok && lds_put(ring, narcs, seg_n > 0 ? one_s : start, a, seg_first);
                                if (!ARCS && !cs_ok && !spill) { spill_window(arcs, seg_first, narcs - seg_first, seg_n > 0 ? one_s : start, merges); spill = true; }
                                if (spill) arcs[narcs] = a;
                                ++narcs;
                            }
                        } else {                                            // whole-token arc replaces the pieces (..._bpe_t.h:189-206)
                            if (LOCAL) cs_ok = lds_put(ring, count_at_start, seg_n > 0 ? one_s : start, a, seg_first);
                            if (!ARCS && !cs_ok && !spill) { spill_window(arcs, seg_first, count_at_start - seg_first, seg_n > 0 ? one_s : start, merges); spill = true; }
                            if (spill) arcs[count_at_start] = a;
                            narcs = count_at_start + 1; ff = i; cs_first_e = i; cs_first_id = r.id;
                        }
                        cs_last_e = i; last_id = r.id;
                        unknown = false;
                    }
                    ++i;
                    walk_ends = i >= L;
                }
            }
            if (walk_ends && !over && !closing_last && unknown && !(0 < narcs && p.unk == last_id) && narcs >= arc_cap) over = true;
            if (over) { p.counts[doc] = 0; p.narcs[doc] = -1; p.fb_list[atomicAdd(p.fb_count, 1u)] = (int32_t)doc; mode = M_NEED; }
            else if (walk_ends) {
                // ---- the walk from `start` is over: unknown arc, then the cut test
                bool merged = false;
                if (!closing_last && unknown) {                             // ..._bpe_t.h:212-225
                    if (0 < narcs && p.unk == last_id) {
                        if (spill) arcs[narcs - 1].end = start;
                        merged = true;
                        if (narcs - 1 == seg_first) one_e = start;
                        if (seg_maxend < start) seg_maxend = start;
                        if (LOCAL) {                                        // the same extension in the LDS window
                            if (start - one_s > 62) {
                                if (!ARCS && seg_local && !spill) {      // the window still holds the segment with the arc's old end: write it out, then the new end
                                    spill_window(arcs, seg_first, narcs - seg_first, one_s, merges); arcs[narcs - 1].end = start; spill = true;
                                }
                                seg_local = false;
                            }
                            else if (seg_local) { LKey *q = ring + (narcs - 1 - seg_first) * 64; *q = (*q & ~(LKey)63) | (LKey)(start - one_s); }
                        }
                    } else {
                        SegArc a; a.start = start; a.end = start; a.id = p.unk; a.rank_bits = 0;
                        if (LOCAL) cs_ok = cs_ok && lds_put(ring, narcs, seg_n > 0 ? one_s : start, a, seg_first);
                        if (!ARCS && !cs_ok && !spill) { spill_window(arcs, seg_first, narcs - seg_first, seg_n > 0 ? one_s : start, merges); spill = true; }
                        if (spill) arcs[narcs] = a;
                        ++narcs; cs_first_e = start; cs_first_id = p.unk; cs_last_e = start; last_id = p.unk;
                    }
                }
                const bool cut = seg_n > 0 && (closing_last || (!merged && start > seg_maxend));
                if (cut && !fallback) {                                     // the previous segment is complete
                    if (seg_n == 1) {                                       // one arc = one token (always applied)
                        if (cnt < p.max_ids) { ids[cnt] = one_id + p.S.id_offset; if (spans) { spans[2 * cnt] = one_s; spans[2 * cnt + 1] = one_e; } }
                        ++cnt;
                    } else if (LOCAL && seg_local) mode = M_LSOLVE;
                    else if (seg_n <= 64 && seg_maxend - one_s + 1 <= 63) mode = M_SOLVE;
                    else fallback = true;
                }
                if (mode != M_SOLVE && mode != M_LSOLVE) { post = true; post_cut = cut; post_merged = merged; }
            }
        }
        if (post) {
            // ---- after the closure decision: the arcs of `start` open / join a segment; next start or end of document
            if (closing_last) {
                p.counts[doc] = fallback ? 0 : (cnt < p.max_ids ? cnt : p.max_ids);
                p.narcs[doc] = fallback ? (ARCS ? narcs : BPE_COLLECT) : BPE_DONE;
                if (fallback) p.fb_list[atomicAdd(p.fb_count, 1u)] = (int32_t)doc;       // the full path redoes it from its arc list
                mode = M_NEED;
            } else {
                if (!post_merged) {
                    if (post_cut || seg_n == 0) {
                        if (LOCAL) {
                            // the arcs of `start` were keyed relative to the old segment: re-base them to the new one (all of them start at `start`)
                            const int d0 = seg_n > 0 ? start - one_s : 0, added = narcs - count_at_start;
                            seg_local = cs_ok && added <= LCAP;
                            if (!ARCS) spill = !seg_local;                   // a segment that fits the window again: nothing more is written
                            const int old_n = count_at_start - seg_first;     // they sit behind the closed segment's entries: move them to the front
                            if (seg_local && old_n > 0) for (int t = 0; t < added; ++t) ring[t * 64] = ring[(old_n + t) * 64] - (LKey)(d0 * 65);
                        }
                        seg_first = count_at_start; seg_n = narcs - count_at_start; one_s = start; one_e = cs_first_e; one_id = cs_first_id; seg_maxend = cs_last_e;
                    }
                    else { seg_n = narcs - seg_first; if (seg_maxend < cs_last_e) seg_maxend = cs_last_e; seg_local = seg_local && cs_ok; }
                }
                if (fast) start = ff;                                       // ..._bpe_t.h:228-230
                ++start;
                if (start < L) {
                    i = start; state = p.S.initial; sum = 0; unknown = true; count_at_start = narcs; ff = start; cs_ok = true;
                    token_start = cls_at(start) == p.S.cls_delim;
                } else closing_last = true;                                 // one more trip closes the final segment
                mode = M_WALK;
            }
        }
    }
    if (mode != M_EXIT) atomicOr(p.status, 2);      // trip limit hit: never expected
}

// 32-byte register window (two aligned 16-byte blocks) over a lane's class stream, positioned by seek(start): the walk from
// `start` reads start .. start + depth - 1 and then jumps back to start + 1, so a window that follows `start` serves almost
// every read from registers; seek() shifts it by one block when `start` crosses a block (one load, nothing waits for it).
// Elements past the window (walks longer than 9..16 elements) are read directly.
struct ClsWin2 {
    const uint4 *cls16; int64_t blk0; int shift; uint4 w0, w1; int t0;
    __device__ __forceinline__ void init(const uint16_t *cls_buf, int64_t elem_off)
    {
        cls16 = (const uint4 *)cls_buf; blk0 = elem_off >> 3; shift = (int)(elem_off & 7); t0 = -4; w0 = w1 = make_uint4(0, 0, 0, 0);
    }
    __device__ __forceinline__ void seek(int start)
    {
        const int t = (start + shift) >> 3;
        if (t == t0) return;
        if (t == t0 + 1) { w0 = w1; w1 = cls16[blk0 + t + 1]; }
        else { w0 = cls16[blk0 + t]; w1 = cls16[blk0 + t + 1]; }
        t0 = t;
    }
    __device__ __forceinline__ uint32_t operator()(int i) const
    {
        const int a = i + shift, r = a - (t0 << 3);
        if ((unsigned)r >= 16u) return ((const uint16_t *)cls16)[(blk0 << 3) + a];
        const bool hi = (r & 8) != 0, up = (r & 4) != 0;
        const uint32_t x = hi ? w1.x : w0.x, y = hi ? w1.y : w0.y, z = hi ? w1.z : w0.z, w = hi ? w1.w : w0.w;
        const uint32_t d0 = up ? z : x, d1 = up ? w : y;
        return __builtin_amdgcn_perm(d1, d0, 0x0c0c0100u + 0x0202u * (uint32_t)(r & 3));
    }
};

// The same window fed from registers instead of memory: two 64-byte sector buffers follow it (sb0 = the sector of the window's second
// block, sb1 = the sector after it), so a lane touches every sector of its stream ONCE, and the loads of sb1 are issued for the whole wave at
// one time (refill(), called where the wave is converged): vector memory loads return in order, so a stream load that misses L2 holds up
// every table gather issued behind it -- with one 16-byte load per lane and eight positions, some lane of the wave had one in flight on
// 93 % of the steps.
struct ClsWinS {
    static __device__ __forceinline__ uint4 sel4(bool c, const uint4 x, const uint4 y) { return make_uint4(c ? x.x : y.x, c ? x.y : y.y, c ? x.z : y.z, c ? x.w : y.w); }
    const uint4 *cls16; int64_t blk0; int shift; uint4 w0, w1; int t0;
    uint4 a0, a1, a2, a3, b0, b1, b2, b3; bool have_b;
    __device__ __forceinline__ void init(const uint16_t *cls_buf, int64_t elem_off)
    {
        cls16 = (const uint4 *)cls_buf; blk0 = elem_off >> 3; shift = (int)(elem_off & 7); t0 = -4; w0 = w1 = make_uint4(0, 0, 0, 0);
        a0 = a1 = a2 = a3 = b0 = b1 = b2 = b3 = w0; have_b = true;
    }
    __device__ __forceinline__ void seek(int start)
    {
        const int t = (start + shift) >> 3;
        if (t == t0) return;
        const int64_t ab = blk0 + t + 1;                    // the block that enters the window
        if (t == t0 + 1) {
            w0 = w1;
            if ((ab & 3) == 0) { a0 = b0; a1 = b1; a2 = b2; a3 = b3; have_b = false; }      // refill() keeps have_b true one block ahead of this
            const int k = (int)(ab & 3);
            const bool k1 = (k & 1) != 0, k2 = (k & 2) != 0;
            const uint4 lo = sel4(k1, a1, a0), hi = sel4(k1, a3, a2);        // component-wise: a select between whole structs goes through memory
            w1 = sel4(k2, hi, lo);
        } else {
            const int64_t sec = ab >> 2;
            w0 = cls16[blk0 + t]; w1 = cls16[ab];
            a0 = cls16[sec * 4]; a1 = cls16[sec * 4 + 1]; a2 = cls16[sec * 4 + 2]; a3 = cls16[sec * 4 + 3];
            b0 = cls16[sec * 4 + 4]; b1 = cls16[sec * 4 + 5]; b2 = cls16[sec * 4 + 6]; b3 = cls16[sec * 4 + 7];
            have_b = true;
        }
        t0 = t;
    }
    // seek(start) for a start that is the one before, or the one after it (the cut form's walks: no loads here, refill() brings the sectors)
    __device__ __forceinline__ void advance(int start)
    {
        const int t = (start + shift) >> 3;
        if (t == t0) return;
        const int64_t ab = blk0 + t + 1;
        w0 = w1;
        if ((ab & 3) == 0) { a0 = b0; a1 = b1; a2 = b2; a3 = b3; have_b = false; }
        const int k = (int)(ab & 3);
        const bool k1 = (k & 1) != 0, k2 = (k & 2) != 0;
        const uint4 lo = sel4(k1, a1, a0), hi = sel4(k1, a3, a2);
        w1 = sel4(k2, hi, lo);
        t0 = t;
    }
    // wave-converged: when some walking lane comes within a block of needing sb1, every lane that lacks it loads it
    __device__ __forceinline__ void refill(bool walking)
    {
        const int64_t ab = blk0 + t0 + 1;
        const bool lacks = walking && !have_b;
        if (__ballot(lacks && (ab & 3) >= 2) == 0) return;
        if (lacks) {
            const int64_t sec = (ab >> 2) + 1;
            b0 = cls16[sec * 4]; b1 = cls16[sec * 4 + 1]; b2 = cls16[sec * 4 + 2]; b3 = cls16[sec * 4 + 3];
            have_b = true;
        }
    }
    __device__ __forceinline__ uint32_t operator()(int i) const
    {
        const int a = i + shift, r = a - (t0 << 3);
        if ((unsigned)r >= 16u) return ((const uint16_t *)cls16)[(blk0 << 3) + a];
        const bool hi = (r & 8) != 0, up = (r & 4) != 0;
        const uint32_t x = hi ? w1.x : w0.x, y = hi ? w1.y : w0.y, z = hi ? w1.z : w0.z, w = hi ? w1.w : w0.w;
        const uint32_t d0 = up ? z : x, d1 = up ? w : y;
        return __builtin_amdgcn_perm(d1, d0, 0x0c0c0100u + 0x0202u * (uint32_t)(r & 3));
    }
};

// End2BestArc entries of the live window: per-lane rings in LDS, structure-of-arrays (scores: bank pair = lane; records: bank = lane)
struct RingLds {
    double *sc; uint32_t *rc; int mask, n;
    __device__ __forceinline__ double score(int pos) const { return sc[(pos & mask) * 64]; }
    __device__ __forceinline__ uint32_t rec(int pos) const { return rc[(pos & mask) * 64]; }
    __device__ __forceinline__ void set(int pos, double v, uint32_t r) { sc[(pos & mask) * 64] = v; rc[(pos & mask) * 64] = r; }
    __device__ __forceinline__ void fill(double v) { for (int k = 0; k < n; ++k) { sc[k * 64] = v; rc[k * 64] = UNI_REC_NONE; } }
};

// Unigram-LM, default form: bf_seg.h UniLane per lane, persistent lanes pulling documents (longest first).  Every trip of
// the loop a walking lane makes UNROLL trie transitions and a lane in its backward pass makes one hop whose record was
// requested BEFORE the walk steps (its latency hides behind them); finished lanes fetch new documents by vote.
template <int UNROLL, bool SPLIT = false, int QN = 4>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k_seg_unigram_lane(SpSegParams p, int ring_n)
{
    extern __shared__ double seg_ring[];            // [ring_n][64] scores, then [ring_n][64] packed records
    enum { M_NEED = 0, M_WALK = 1, M_BACK = 2, M_EXIT = 3 };
    const int lane = lane_id();
    RingLds ring{seg_ring + lane, (uint32_t *)(seg_ring + (size_t)ring_n * 64) + lane, ring_n - 1, ring_n};
    typedef typename std::conditional<SPLIT, ClsWinS, ClsWin2>::type Win;
    Win cls_at; cls_at.init(p.stream, 0);
    UniLane<Win, RingLds, QN> ul(p.S, cls_at, ring);
    ul.L = 0; ul.depth = p.trie_depth; ul.start = ul.i = ul.sum = 0; ul.state = 0; ul.unknown = true; ul.pend = false; ul.prev = 0; ul.pend_i = 0;
    ul.pend_score = 0; ul.pend_key = 0; ul.end = 0; ul.cnt = 0; ul.unk_run = 0; ul.qn = 0; ul.abs0 = 0;
#pragma unroll
    for (int k = 0; k < QN; ++k) ul.q[k] = 0;
    int mode = M_NEED;
    int64_t doc = 0; int32_t *ids = nullptr; int32_t *spans = nullptr; int cap = 0;
    int32_t g0 = 0, g1 = 0, g2 = 0, g3 = 0; int gn = 0;     // ids of the backward pass waiting for their 16-byte group
    for (unsigned long long trip = 0; trip < (1ull << 34); ++trip) {
        // ---- documents for idle lanes
        const unsigned long long m_need = __ballot(mode == M_NEED);
        if (m_need) {
            const unsigned long long m_busy = __ballot(mode == M_WALK || mode == M_BACK);
            if (__popcll(m_need) >= 8 || m_busy == 0) {
                if (mode == M_NEED) {
                    const int c = __popcll(m_need);
                    const int leader = __ffsll((long long)m_need) - 1;
                    unsigned long long base = 0;
                    if (lane == leader) base = atomicAdd(p.next_doc, (unsigned long long)c);
                    base = __shfl(base, leader, 64);
                    const int64_t idx = (int64_t)base + __popcll(m_need & lanemask_lt());
                    if (idx >= p.b.ndocs) mode = M_EXIT;
                    else {
                        doc = p.perm[idx];
                        const int64_t b = p.b.doc_off[doc];
                        const int64_t slot = sp_slot(b, doc, p.slot_mul);
                        cap = p.slot_mul * (int)(p.b.doc_off[doc + 1] - b + 1);
                        const int L = p.lens[doc];
                        ids = p.ids_tmp + slot; spans = p.span_tmp ? p.span_tmp + 2 * slot : nullptr;
                        if (L <= 0) { p.counts[doc] = 0; p.narcs[doc] = 0; }
                        else { cls_at.init(p.stream, slot); ul.init(L, p.trie_depth, (uint32_t *)p.best + slot, slot); mode = M_WALK; }
                    }
                }
                if (__ballot(mode != M_EXIT) == 0) break;
            }
        }
        // ---- backward pass: request this trip's record now, use it after the walk steps
        uint32_t br = 0;
        const bool back = !SPLIT && mode == M_BACK;
        if (back) br = ul.recs[ul.end];
        if constexpr (SPLIT) cls_at.refill(mode == M_WALK);
        // ---- forward pass: UNROLL trie transitions
        if (mode == M_WALK) {
            bool walk = true;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { if (walk) walk = ul.wstep(); }
            if (!walk) {
                if (SPLIT) mode = M_NEED;            // the records are complete; k_uni_back reads them
                else { ul.begin_back(); mode = M_BACK; }
            }
        }
        if (back) {
            // ids leave in descending address order (right-aligned in the slot): they are queued and stored as whole aligned
            // 16-byte groups (g0 = newest = lowest address), single words only at the two ends of the sequence
            int32_t *dst = nullptr;
            auto put = [&](int k, int id, int from, int to) {
                g3 = g2; g2 = g1; g1 = g0; g0 = id; ++gn;
                dst = ids + (cap - 1 - k);
                if (spans) { spans[2 * (cap - 1 - k)] = from; spans[2 * (cap - 1 - k) + 1] = to; }
            };
            const bool more = ul.bstep(br, put, p.unk);
            if (!more || (((uintptr_t)dst >> 2) & 3) == 0) {
                if (gn == 4 && (((uintptr_t)dst >> 2) & 3) == 0) *(int4 *)dst = make_int4(g0, g1, g2, g3);
                else { dst[0] = g0; if (gn > 1) dst[1] = g1; if (gn > 2) dst[2] = g2; if (gn > 3) dst[3] = g3; }
                gn = 0;
            }
            if (!more) {
                p.counts[doc] = ul.cnt < p.max_ids ? ul.cnt : p.max_ids;
                p.narcs[doc] = cap - ul.cnt;
                mode = M_NEED;
            }
        }
    }
    if (mode != M_EXIT) atomicOr(p.status, 2);      // trip limit hit: never expected
}

// The backward pass of the lane program as its own kernel (SPLIT instance of k_seg_unigram_lane): a lane per document hops from record to
// record (…_1best_t.h:237-265).  Every hop is a load that misses L2 (the records were written once, long ago); inside the forward kernel such
// a load holds up the table gathers issued behind it (vector memory loads return in order), here nothing else waits.
__global__ __launch_bounds__(256) void k_uni_back(SpSegParams p)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.b.ndocs) return;
    const int64_t doc = idx;
    const int L = p.lens[doc];
    if (L <= 0) return;                               // counts / narcs were written by the forward kernel
    const int64_t b = p.b.doc_off[doc];
    const int64_t slot = sp_slot(b, doc, p.slot_mul);
    const int cap = p.slot_mul * (int)(p.b.doc_off[doc + 1] - b + 1);
    int32_t *ids = p.ids_tmp + slot; int32_t *spans = p.span_tmp ? p.span_tmp + 2 * slot : nullptr;
    struct NoCls { __device__ void seek(int) {} __device__ uint32_t operator()(int) const { return 0; } } nocls;
    struct NoRing { __device__ double score(int) const { return 0; } __device__ uint32_t rec(int) const { return 0; } __device__ void set(int, double, uint32_t) {} __device__ void fill(double) {} } noring;
    UniLane<NoCls, NoRing> ul(p.S, nocls, noring);
    ul.L = L; ul.recs = (uint32_t *)p.best + slot;
    ul.begin_back();
    int32_t g0 = 0, g1 = 0, g2 = 0, g3 = 0; int gn = 0;
    // the records are read a 64-byte sector at a time (16 positions, ~4 hops) and kept in registers: with half a million documents in flight
    // nothing survives in L2 between two hops of one lane, so a 4-byte load per hop fetched every sector four times over
    const uint4 *rec16 = (const uint4 *)p.best;
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0; int64_t cur = -1;
    for (;;) {
        const int64_t a = slot + ul.end, sec = a >> 4;
        if (sec != cur) { r0 = rec16[sec * 4]; r1 = rec16[sec * 4 + 1]; r2 = rec16[sec * 4 + 2]; r3 = rec16[sec * 4 + 3]; cur = sec; }
        const int k = (int)(a & 15);
        const bool k4 = (k & 4) != 0, k8 = (k & 8) != 0;
        const uint32_t x = k8 ? (k4 ? r3.x : r2.x) : (k4 ? r1.x : r0.x), y = k8 ? (k4 ? r3.y : r2.y) : (k4 ? r1.y : r0.y);
        const uint32_t z = k8 ? (k4 ? r3.z : r2.z) : (k4 ? r1.z : r0.z), w = k8 ? (k4 ? r3.w : r2.w) : (k4 ? r1.w : r0.w);
        const uint32_t br = (k & 2) ? ((k & 1) ? w : z) : ((k & 1) ? y : x);
        int32_t *dst = nullptr;
        auto put = [&](int kk, int id, int from, int to) {
            g3 = g2; g2 = g1; g1 = g0; g0 = id; ++gn;
            dst = ids + (cap - 1 - kk);
            if (spans) { spans[2 * (cap - 1 - kk)] = from; spans[2 * (cap - 1 - kk) + 1] = to; }
        };
        const bool more = ul.bstep(br, put, p.unk);
        if (!more || (((uintptr_t)dst >> 2) & 3) == 0) {
            if (gn == 4 && (((uintptr_t)dst >> 2) & 3) == 0) *(int4 *)dst = make_int4(g0, g1, g2, g3);
            else { dst[0] = g0; if (gn > 1) dst[1] = g1; if (gn > 2) dst[2] = g2; if (gn > 3) dst[3] = g3; }
            gn = 0;
        }
        if (!more) break;
    }
    p.counts[doc] = ul.cnt < p.max_ids ? ul.cnt : p.max_ids;
    p.narcs[doc] = cap - ul.cnt;
}

// Unigram-LM, cut form (round 6): bf_seg.h UniCut per lane -- the forward pass of the lane program above, one BYTE of record per position in
// an LDS ring of W positions instead of four in memory, the tokens read off the ring at the cuts (positions the reference's backward pass
// is certain to land on) and written in forward order, left-aligned in the document's slot, one word per token (key / symbols to walk / unknown);
// k_uni_ids turns them into ids.  No record array traffic, no backward kernel.  LDS per lane: ring_n scores (8 B) + W record bytes: 10 KB
// per wave, sixteen waves per CU (the lane program above: 12 KB, thirteen) -- and the kernel's time is latency x waves (bf_seg.h).
// Emission is a phase the whole wave enters together -- when a lane has finished its document or filled its ring, and every `period` trips --
// so that its loops run with most lanes active instead of a few lanes every trip.
struct RingCut {
    double *sc; uint8_t *rc; int smask, rmask, sn;
    __device__ __forceinline__ double score(int pos) const { return sc[(pos & smask) * 64]; }
    __device__ __forceinline__ uint32_t rec(int pos) const { return rc[(pos & rmask) * 64]; }
    __device__ __forceinline__ void set(int pos, double v, uint32_t r) { sc[(pos & smask) * 64] = v; rc[(pos & rmask) * 64] = (uint8_t)r; }
    __device__ __forceinline__ void setrec(int pos, uint32_t r) { rc[(pos & rmask) * 64] = (uint8_t)r; }
    __device__ __forceinline__ void setscore(int pos, double v) { sc[(pos & smask) * 64] = v; }
    __device__ __forceinline__ void fill(double v) { for (int k = 0; k < sn; ++k) sc[k * 64] = v; }      // (a record slot is written before it is read: bf_seg.h)
};

template <int UNROLL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_uni_cut(SpSegParams p, int ring_n, int W, unsigned period_mask, int quick)
{
    extern __shared__ double seg_ring[];            // [ring_n][64] scores, then [W][64] record bytes
    enum { M_NEED = 0, M_WALK = 1, M_FLUSH = 2, M_EXIT = 3 };
    const int lane = lane_id();
    RingCut ring{seg_ring + lane, (uint8_t *)(seg_ring + (size_t)ring_n * 64) + lane, ring_n - 1, W - 1, ring_n};
    ClsWinS cls_at; cls_at.init(p.stream, 0);
    UniCut<ClsWinS, RingCut> uc(p.S, cls_at, ring);
    uc.L = 0; uc.depth = p.trie_depth; uc.W = W; uc.start = uc.i = uc.sum = 0; uc.state = 0; uc.unknown = true; uc.pend = false; uc.walking = false; uc.prev = 0; uc.pend_i = 0;
    uc.pend_score = 0; uc.unk_run = 0; uc.reach = -1; uc.rk = 0; uc.rs = 0; uc.ck = 0; uc.cs = -1; uc.cut0 = 0; uc.lastcut = -1; uc.ring_lo = 0; uc.nout = 0; uc.tok_align = 0;
    int mode = M_NEED;
    int64_t doc = 0; int32_t *toks = nullptr;
    for (unsigned long long trip = 0; trip < (1ull << 34); ++trip) {
        // ---- documents for idle lanes
        const unsigned long long m_need = __ballot(mode == M_NEED);
        if (m_need) {
            const unsigned long long m_busy = __ballot(mode == M_WALK || mode == M_FLUSH);
            if (__popcll(m_need) >= 8 || m_busy == 0) {
                if (mode == M_NEED) {
                    const int c = __popcll(m_need);
                    const int leader = __ffsll((long long)m_need) - 1;
                    unsigned long long base = 0;
                    if (lane == leader) base = atomicAdd(p.next_doc, (unsigned long long)c);
                    base = __shfl(base, leader, 64);
                    const int64_t idx = (int64_t)base + __popcll(m_need & lanemask_lt());
                    if (idx >= p.b.ndocs) mode = M_EXIT;
                    else {
                        doc = p.perm[idx];
                        const int64_t b = p.b.doc_off[doc];
                        const int64_t slot = sp_slot(b, doc, p.slot_mul);
                        const int L = p.lens[doc];
                        toks = p.ids_tmp + slot;
                        if (L <= 0) { p.counts[doc] = 0; p.narcs[doc] = 0; }
                        else { cls_at.init(p.stream, slot); uc.init(L, p.trie_depth, W, (uint8_t *)p.best + slot); uc.tok_align = (int)(slot & 3); mode = M_WALK; }
                    }
                }
                if (__ballot(mode != M_EXIT) == 0) break;
            }
        }
        cls_at.refill(mode == M_WALK);
        // ---- forward pass: UNROLL trie transitions
        struct TokPut {
            int32_t *toks; int max_ids;
            __device__ __forceinline__ void operator()(int k, uint32_t v) const { if (k < max_ids) toks[k] = (int32_t)v; }
            __device__ __forceinline__ void quad(int k, uint32_t a, uint32_t b, uint32_t c, uint32_t d) const
            {
                if (k + 3 < max_ids) *(uint4 *)(toks + k) = make_uint4(a, b, c, d);
                else { (*this)(k, a); (*this)(k + 1, b); (*this)(k + 2, c); (*this)(k + 3, d); }
            }
        } put{toks, p.max_ids};
        // a trip: UNROLL transitions of the lane's walk (a walk that is over waits), then -- once, for all the lanes whose walk is over -- the
        // end of the start position; the short way out for the chunks that are complete
        const bool can = mode == M_WALK && uc.room(UNROLL);
        const bool stall = mode == M_WALK && !can;
        if (can) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { if (uc.walking) uc.step(); }
            if (!uc.walking) { if (uc.finish_start() == UC_DONE) mode = M_FLUSH; }
            if (quick) uc.quick(put);                // (A/B runs, BfSetVariant bit 0x20: chunks of one or two tokens leave here with their key, bf_seg.h)
        }
        // ---- emission phase: the chunks of several tokens in front of every lane's latest cut
        if (__ballot(mode == M_FLUSH || stall) != 0 || ((unsigned)trip & period_mask) == period_mask) {
            if (mode == M_WALK || mode == M_FLUSH) {
                if (uc.pending()) uc.emit(put);
                else if (stall) uc.spill(UNROLL);
            }
            if (mode == M_FLUSH) {
                p.counts[doc] = uc.nout < p.max_ids ? uc.nout : p.max_ids;
                p.narcs[doc] = uc.nout;
                mode = M_NEED;
            }
        }
    }
    if (mode != M_EXIT) atomicOr(p.status, 2);      // trip limit hit: never expected
}

// The ids of the token words the cut form left behind (tokdll:1512-1516 on the Unigram path; bf_seg.h uni_token_id): the id column of I2Info
// at the word's key; UnkId for a word flagged unknown; for a word that names its symbols (one token in eight) the walk over them first -- one
// transition per symbol, the output weights add up to the key; + IdOffset in every case.  The ids go straight to their place in the caller's
// array: this kernel is the compaction of the path too.
// A wave takes 64 consecutive documents (count, slot, place: one document per lane, handed round with readlane as in k_compact_ids) and their
// tokens 64 per trip: keys and unknowns are answered at once (64 independent look-ups); the words to walk go to a list in LDS and are walked
// 128 at a time, two per lane, whatever documents they come from -- every lane of a walk has tokens, and a walk's symbols are read eight per load.
struct __attribute__((packed, aligned(2))) Sym8 { uint32_t x, y, z, w; };      // eight stream elements at a 2-byte aligned address: one 16-byte load
constexpr int UNI_WM = 2;                         // words a lane walks at a time (independent chains: the dependent gathers of one hide behind the other's)
constexpr int UNI_WL = 64 * UNI_WM * 2;           // the wave's list of words to walk (a ring: a batch waits while up to 64 more arrive)
__global__ __launch_bounds__(256) void k_uni_ids(UniIdsParams p)
{
    __shared__ int64_t wl_src[4][UNI_WL], wl_out[4][UNI_WL];
    __shared__ int32_t wl_len[4][UNI_WL];
    const int lane = lane_id(), wv_i = wave_in_block();
    int64_t *l_src = wl_src[wv_i], *l_out = wl_out[wv_i]; int32_t *l_len = wl_len[wv_i];
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wv_i;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    bool over = false;
    uint32_t head = 0, tail = 0;                    // wave-uniform
    auto walk_batch = [&](int n) {                  // the first n (<= 64 * UNI_WM) words of the list: lane l takes the words l, l + 64, ...
        int64_t src[UNI_WM], out[UNI_WM]; int len[UNI_WM], sum[UNI_WM]; uint32_t state[UNI_WM]; uint64_t blo[UNI_WM], bhi[UNI_WM];      // (the eight symbols at hand: two 64-bit halves)
        int maxlen = 0;
#pragma unroll
        for (int m = 0; m < UNI_WM; ++m) {
            src[m] = 0; out[m] = 0; len[m] = 0; sum[m] = 0; state[m] = p.initial; blo[m] = bhi[m] = 0;
            if (lane + 64 * m < n) { const uint32_t e = (head + (uint32_t)(lane + 64 * m)) & (UNI_WL - 1); src[m] = l_src[e]; out[m] = l_out[e]; len[m] = l_len[e]; }
            maxlen = len[m] > maxlen ? len[m] : maxlen;
        }
        for (int j = 0; __any(j < maxlen); ++j) {   // j is the same for every lane: the symbol's place in the block is a scalar
            uint64_t e[UNI_WM];
#pragma unroll
            for (int m = 0; m < UNI_WM; ++m) {
                if ((j & 7) == 0 && j < len[m]) { const Sym8 v = *(const Sym8 *)(p.stream + src[m] + j); blo[m] = (uint64_t)v.x | ((uint64_t)v.y << 32); bhi[m] = (uint64_t)v.z | ((uint64_t)v.w << 32); }     // (the element slot is 2-byte aligned only; the stream buffer is padded)
            }
#pragma unroll
            for (int m = 0; m < UNI_WM; ++m) {
                const uint32_t c = (uint32_t)(((j & 4) ? bhi[m] : blo[m]) >> (16 * (j & 3))) & 0xFFFFu;
                e[m] = j < len[m] ? p.T[state[m] + c] : 0ull;
            }
#pragma unroll
            for (int m = 0; m < UNI_WM; ++m) {
                if (j < len[m]) { state[m] = (uint32_t)((e[m] >> SG_NEXT_SHIFT) & SG_NEXT_MASK); sum[m] += (int)(e[m] >> SG_OW_SHIFT); }
            }
        }
        int id[UNI_WM];
#pragma unroll
        for (int m = 0; m < UNI_WM; ++m) id[m] = lane + 64 * m < n ? p.ids[sum[m]] : -1;
#pragma unroll
        for (int m = 0; m < UNI_WM; ++m) if (lane + 64 * m < n) p.ids_out[out[m]] = (id[m] != -1 ? id[m] : p.unk) + p.id_offset;
        head += (uint32_t)n;
    };
    for (int64_t base = wave0 * 64; base < p.b.ndocs; base += nwaves * 64) {
        const int64_t d = base + lane;
        int c = 0; int64_t slot = 0, o = 0;
        if (d < p.b.ndocs) {
            c = p.counts[d];
            slot = sp_slot(p.b.doc_off[d], d, p.slot_mul);
            o = p.id_off[d];
            if (o + c > p.ids_cap) { over = true; c = o < p.ids_cap ? (int)(p.ids_cap - o) : 0; }
        }
        const int nd = p.b.ndocs - base < 64 ? (int)(p.b.ndocs - base) : 64;
        for (int k = 0; k < nd; ++k) {
            const int ck = __builtin_amdgcn_readlane(c, k);
            if (ck <= 0) continue;
            const int64_t sk = wv::bcast(slot, k), ok = wv::bcast(o, k);
            const uint32_t *tok = (const uint32_t *)p.toks + sk;
            for (int t0 = 0; t0 < ck; t0 += 64) {
                const int t = t0 + lane;
                const bool act = t < ck;
                const uint32_t v = act ? tok[t] : UC_TOK_UNK;
                const bool walk = act && (v & (UC_TOK_UNK | UC_TOK_KEY)) == 0u;
                if (act && !walk) {
                    const int id = (v & UC_TOK_KEY) ? p.ids[v & 0xFFFFFu] : -1;
                    p.ids_out[ok + t] = (id != -1 ? id : p.unk) + p.id_offset;
                }
                const unsigned long long wm = __ballot(walk);
                if (wm) {
                    if (walk) {
                        const uint32_t e = (tail + (uint32_t)__popcll(wm & lanemask_lt())) & (UNI_WL - 1);
                        l_src[e] = sk + (int64_t)(v & ((1u << UC_BEGIN_BITS) - 1u)); l_out[e] = ok + t; l_len[e] = (int)(v >> UC_BEGIN_BITS) + 1;
                    }
                    tail += (uint32_t)__popcll(wm);
                    wave_handoff();
                    if (tail - head >= 64u * UNI_WM) { walk_batch(64 * UNI_WM); wave_handoff(); }
                }
            }
        }
    }
    if (tail != head) walk_batch((int)(tail - head));
    if (over) atomicOr(p.status, 1);
}

void launch_uni_ids(const UniIdsParams &p, hipStream_t s)
{
    int64_t blocks = (p.b.ndocs + 255) / 256;
    if (blocks > device_cus() * 8) blocks = device_cus() * 8;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_uni_ids, dim3((unsigned)blocks), dim3(256), 0, s, p);
}

// BPE, documents whose arcs exceed the per-document reserve (narcs == -1 on the fallback list; a long run of one character whose
// run-length tokens are all in the vocabulary): one wave per document, every step wave-cooperative, arcs in a block claimed from the
// batch's pool at its exact size (bf_bpe_seg_body.h; the same source runs in the test simulator against the oracle).  A pool that
// runs out costs the document (count 0, BF_STATUS_POOL), not the batch; the host grows the pool and runs the batch again.
} // namespace bfa
#include "bf_bpe_seg_body.h"
namespace bfa {

__global__ __launch_bounds__(64) void k_bpe_seg(BpeSegParams p)
{
    __shared__ BsLds lds;
    BpeSeg<BsLds> w(p, lds);
    w.run();
}

static BpeSegParams bpe_seg_params(const SpSegParams &p)
{
    BpeSegParams q;
    q.T = p.S.T; q.info = p.S.info; q.initial = p.S.initial; q.cls_delim = p.S.cls_delim; q.id_offset = p.S.id_offset; q.kind = p.S.kind;
    q.prio = p.bpe_prio; q.place_id = p.bpe_place_id; q.unk_prio = p.bpe_unk_prio; q.prio_bits = p.bpe_prio_bits;
    q.stream = p.stream; q.lens = p.lens; q.doc_off = p.b.doc_off; q.slot_mul = p.slot_mul;
    q.list = p.fb_list; q.list_n = p.fb_count; q.narcs = p.narcs; q.narcs_want = -1; q.ndocs = p.b.ndocs;
    q.ids_tmp = p.ids_tmp; q.span_tmp = p.span_tmp; q.counts = p.counts; q.max_ids = p.max_ids; q.unk = p.unk;
    q.next_doc = p.next_doc; q.status = p.status;
    q.pool = p.big_pool; q.pool_bytes = p.big_cap; q.pool_used = p.big_used; q.pool_need = p.big_need; q.stats = p.seg_stats;
    return q;
}

// the documents the BPE wave program handed back (flags[d] != 0) as a list for k_bpe_seg
__global__ __launch_bounds__(256) void k_bpe_flag_list(const int32_t *flags, int64_t ndocs, int32_t *list, unsigned int *count)
{
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (d < ndocs && flags[d]) list[atomicAdd(count, 1u)] = (int32_t)d;
}

// BPE wave program models: the handed-back documents go straight to the one-wave-per-document program (no lane kernels: one lane
// walks one document at ~13 us per byte, a wave at ~1 us per 64 bytes)
void launch_bpe_seg_flags(const SpSegParams &p_in, const int32_t *flags, int32_t *list, unsigned int *count, hipStream_t s)
{
    SpSegParams p = p_in;
    (void)hipMemsetAsync(count, 0, sizeof(unsigned int), s);
    (void)hipMemsetAsync(p.next_doc, 0, sizeof(unsigned long long), s);
    hipLaunchKernelGGL(k_bpe_flag_list, dim3((unsigned)((p.b.ndocs + 255) / 256)), dim3(256), 0, s, flags, p.b.ndocs, list, count);
    p.fb_list = list; p.fb_count = count;
    BpeSegParams q = bpe_seg_params(p);
    q.narcs = nullptr;
    hipLaunchKernelGGL(k_bpe_seg, dim3((unsigned)device_cus() * 2u), dim3(64), 0, s, q);
}

void launch_seg_sp(const SpSegParams &p_in, hipStream_t s)
{
    const SpSegParams &p = p_in;
    const unsigned bsort = (unsigned)((p.b.ndocs + 256 * SP_SORT_ITEMS - 1) / (256 * SP_SORT_ITEMS));
    const unsigned b64 = (unsigned)((p.b.ndocs + 63) / 64);
    (void)hipMemsetAsync(p.hist, 0, 2048 * sizeof(unsigned int), s);
    hipLaunchKernelGGL(k_sp_hist, dim3(bsort), dim3(256), 0, s, p);
    hipLaunchKernelGGL(k_sp_hist_scan, dim3(1), dim3(1024), 0, s, p);
    hipLaunchKernelGGL(k_sp_scatter, dim3(bsort), dim3(256), 0, s, p);
    if (p.S.kind == SG_KIND_UNIGRAM) {
        // sequential form (experiments, and models outside the lane program's limits: entries longer than 32 symbols or ids >= 2^20 - 2)
        if (!p.lane_ok) hipLaunchKernelGGL(k_seg_unigram, dim3(b64), dim3(64), 0, s, p);
        else {
            int ring = 1; while (ring < p.trie_depth) ring <<= 1;
            const size_t lds = (size_t)ring * 64 * (sizeof(double) + sizeof(uint32_t));
            int per_cu = 0;
#ifdef BF_EXPERIMENTS
            const bool one_kernel = (p.variant & 0x20) != 0;     // A/B runs: forward and backward pass in one kernel (the form of rounds 2..3)
            const int unroll = p.tune ? p.tune : 3;
            auto kern = one_kernel ? (const void *)k_seg_unigram_lane<3> : unroll == 4 ? (const void *)k_seg_unigram_lane<4, true, 8> : (const void *)k_seg_unigram_lane<3, true, 8>;
#else
            const bool one_kernel = false; const int unroll = 3;
            auto kern = (const void *)k_seg_unigram_lane<3, true, 8>;
#endif
            if (p.uni_cut) {
                // the cut form (ids only): records in LDS, tokens out at the cuts, no backward kernel
                int W = 32; while (W < p.trie_depth + UC_SPILL + 4) W <<= 1;
                const size_t lds_c = (size_t)ring * 64 * sizeof(double) + (size_t)W * 64;
                int pc = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, (const void *)k_uni_cut<3>, 64, lds_c) != hipSuccess || pc <= 0) pc = 8;
                (void)hipGetLastError();
                if (p.tune2 > 0 && p.tune2 < pc) pc = p.tune2;
                unsigned blocks = (unsigned)device_cus() * (unsigned)pc;
                if ((int64_t)blocks > (int64_t)b64) blocks = b64;
                unsigned period_mask = 31u;
                if (p.tune > 0) { period_mask = 1u; while (period_mask + 1u < (unsigned)p.tune) period_mask = period_mask * 2u + 1u; if (p.tune == 1) period_mask = 0u; }
                if (p.ev_dom0) (void)hipEventRecord((hipEvent_t)p.ev_dom0, s);
                if (p.variant & 0x10) hipLaunchKernelGGL(k_uni_cut<2>, dim3(blocks), dim3(64), lds_c, s, p, ring, W, period_mask, (p.variant & 0x20) ? 1 : 0);
                else hipLaunchKernelGGL(k_uni_cut<3>, dim3(blocks), dim3(64), lds_c, s, p, ring, W, period_mask, (p.variant & 0x20) ? 1 : 0);
                if (p.ev_dom1) (void)hipEventRecord((hipEvent_t)p.ev_dom1, s);
                return;
            }
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64, lds) != hipSuccess || per_cu <= 0) per_cu = 8;
            (void)hipGetLastError();
            if (p.tune2 > 0 && p.tune2 < per_cu) per_cu = p.tune2;
            unsigned blocks = (unsigned)device_cus() * (unsigned)per_cu;
            if ((int64_t)blocks > (int64_t)b64) blocks = b64;
#ifdef BF_EXPERIMENTS
            if (one_kernel) {
                if (p.ev_dom0) (void)hipEventRecord((hipEvent_t)p.ev_dom0, s);
                hipLaunchKernelGGL(k_seg_unigram_lane<3>, dim3(blocks), dim3(64), lds, s, p, ring);
                if (p.ev_dom1) (void)hipEventRecord((hipEvent_t)p.ev_dom1, s);
            } else if (unroll == 4) {
                if (p.ev_dom0) (void)hipEventRecord((hipEvent_t)p.ev_dom0, s);
                hipLaunchKernelGGL((k_seg_unigram_lane<4, true, 8>), dim3(blocks), dim3(64), lds, s, p, ring);
                if (p.ev_dom1) (void)hipEventRecord((hipEvent_t)p.ev_dom1, s);
                hipLaunchKernelGGL(k_uni_back, dim3((unsigned)((p.b.ndocs + 255) / 256)), dim3(256), 0, s, p);
            } else
#endif
            {
                // forward pass (persistent lanes, records out), then the backward pass over every document
                (void)one_kernel; (void)unroll;
                if (p.ev_dom0) (void)hipEventRecord((hipEvent_t)p.ev_dom0, s);
                hipLaunchKernelGGL((k_seg_unigram_lane<3, true, 8>), dim3(blocks), dim3(64), lds, s, p, ring);
                if (p.ev_dom1) (void)hipEventRecord((hipEvent_t)p.ev_dom1, s);
                hipLaunchKernelGGL(k_uni_back, dim3((unsigned)((p.b.ndocs + 255) / 256)), dim3(256), 0, s, p);
            }
        }
    } else {
        SpSegParams p = p_in;
        if (p.variant == 2) { p.fb_list = nullptr; p.fb_count = nullptr; hipLaunchKernelGGL(k_bpe_collect, dim3(b64), dim3(64), 0, s, p); }
        else {
            p.fb_count = p.hist; p.fb_list = p.tos + 2 * p.bm_words;    // behind the two bitmaps of k_bpe_apply_flat
            (void)hipMemsetAsync(p.fb_count, 0, sizeof(unsigned int), s);
            const bool mg = p.S.kind == SG_KIND_BPE_MERGES;
            const bool local = p.variant != 4 && p.unk >= 0 && p.unk < (1 << BPE_LOCAL_ID_BITS);
            const size_t lds = local ? (size_t)8 * 1024 : 0;           // CAP * sizeof(Key) * 64 lanes, both flavours
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bpe_fused<false, true>, 64, lds) != hipSuccess || per_cu <= 0) per_cu = 12;
            (void)hipGetLastError();
            unsigned blocks = (unsigned)device_cus() * (unsigned)per_cu;
            if ((int64_t)blocks > (int64_t)b64) blocks = b64;
            const bool arcs = p.variant == 5;                          // variant 5 (A/B runs): the instances that write every document's arc list
            if (mg) {
                if (local && !arcs) hipLaunchKernelGGL((k_bpe_fused<true, true, false>), dim3(blocks), dim3(64), lds, s, p);
                else if (local) hipLaunchKernelGGL((k_bpe_fused<true, true>), dim3(blocks), dim3(64), lds, s, p);
                else hipLaunchKernelGGL((k_bpe_fused<true, false>), dim3(blocks), dim3(64), lds, s, p);
            } else {
                if (local && !arcs) hipLaunchKernelGGL((k_bpe_fused<false, true, false>), dim3(blocks), dim3(64), lds, s, p);
                else if (local) hipLaunchKernelGGL((k_bpe_fused<false, true>), dim3(blocks), dim3(64), lds, s, p);
                else hipLaunchKernelGGL((k_bpe_fused<false, false>), dim3(blocks), dim3(64), lds, s, p);
            }
            if (local && !arcs) hipLaunchKernelGGL(k_bpe_collect_list, dim3(64), dim3(64), 0, s, p);
            (void)hipMemsetAsync(p.next_doc, 0, sizeof(unsigned long long), s);
        }
        unsigned sort_blocks = p.fb_list ? 64 : (unsigned)device_cus() * 2;                 // the fallback list is normally empty
        if ((int64_t)sort_blocks > p.b.ndocs) sort_blocks = (unsigned)p.b.ndocs;
        hipLaunchKernelGGL(k_bpe_sort, dim3(sort_blocks), dim3(256), 0, s, p);
        if (p.variant == 2) hipLaunchKernelGGL(k_bpe_apply, dim3(b64), dim3(64), 0, s, p);
        else {
            (void)hipMemsetAsync(p.next_doc, 0, sizeof(unsigned long long), s);
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bpe_apply_flat, 64, 0) != hipSuccess || per_cu <= 0) per_cu = 16;
            (void)hipGetLastError();
            unsigned blocks = p.fb_list ? (unsigned)device_cus() : (unsigned)device_cus() * (unsigned)per_cu;
            if ((int64_t)blocks > (int64_t)b64) blocks = b64;
            hipLaunchKernelGGL(k_bpe_apply_flat, dim3(blocks), dim3(64), 0, s, p);
            (void)hipMemsetAsync(p.next_doc, 0, sizeof(unsigned long long), s);
            hipLaunchKernelGGL(k_bpe_seg, dim3((unsigned)device_cus()), dim3(64), 0, s, bpe_seg_params(p));      // idle unless the list holds a document with narcs == -1
        }
    }
}

// ------------------------------------------------------------------------------------------
// IdsToText (reference tokdll:1689-1745), batch form: a variable-length byte gather.  Wave per sequence, 64 ids per
// iteration.  The reference's one sequential rule -- "no space in the leading position": while nothing has been written,
// a token loses one leading 0x20 -- only ever touches the tokens up to and including the first one that is neither
// skipped, empty nor exactly " " (the ones before it vanish, it loses its leading space if it has one), so a uniform
// `found` flag carried across iterations reproduces it.
//   k_i2t_len   bytes per sequence (0 if it contains an unknown id)         -> scan -> text offsets
//   k_i2t_copy  per iteration the 64 token lengths are prefix-summed in LDS; then byte b of the iteration's output is
//               copied by lane b mod 64 (binary search of its token in the 64 prefix sums): stores are whole rows
// ------------------------------------------------------------------------------------------
struct I2tTok { int len; uint32_t src; bool bad; };

// token `i` of the sequence [b, e): its length / source offset after the skip rule and the leading-space rule
__device__ __forceinline__ I2tTok i2t_token(const I2tParams &p, int64_t i, int64_t e, bool &found)
{
    I2tTok t; t.len = 0; t.src = 0; t.bad = false;
    bool solid = false, sp = false;
    if (i < e) {
        const int id = p.ids[i];
        const bool skip = p.skip_special && (id < p.min_id || id > p.max_id);          // tokdll:1712-1714
        if (!skip) {
            if (id < 0 || id >= p.ntok) t.bad = true;                                     // unknown id: the call returns 0 (tokdll:1719-1721)
            else {
                const uint32_t o0 = p.tok_off[id], o1 = p.tok_off[id + 1];
                t.len = (int)(o1 - o0); t.src = o0;
                sp = t.len > 0 && p.tok_data[o0] == 0x20;
                solid = t.len > 0 && !(t.len == 1 && sp);
            }
        }
    }
    // leading-space rule (tokdll:1724-1728), resolved across the wave
    const unsigned long long ms = __ballot(solid);
    if (!found) {
        const int first = ms ? __ffsll((long long)ms) - 1 : 64;
        const int l = lane_id();
        if (l < first) t.len = 0;                                   // empty or " " before anything was written: nothing
        else if (l == first && sp) { t.len -= 1; t.src += 1; }
        if (ms) found = true;
    }
    return t;
}

__global__ __launch_bounds__(256) void k_i2t_len(I2tParams p)
{
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave_in_block(), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t d = wave0; d < p.nseq; d += nwaves) {
        const int64_t b = p.id_off[d], e = p.id_off[d + 1];
        bool found = false, bad = false; long long total = 0;
        for (int64_t i0 = b; i0 < e; i0 += 64) {
            const I2tTok t = i2t_token(p, i0 + lane, e, found);
            bad |= t.bad; total += t.len;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
        const bool any_bad = __any(bad);
        if (lane == 0) {
            if (any_bad) atomicOr(p.status, 4);
            p.lens[d] = (any_bad || total > 0x7ffffff0ll) ? 0 : (int32_t)total;
        }
    }
}

__global__ __launch_bounds__(256) void k_i2t_copy(I2tParams p)
{
    __shared__ int s_pre[4][65];
    __shared__ uint32_t s_src[4][64];
    const int lane = lane_id(), wv = wave_in_block();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    for (int64_t d = wave0; d < p.nseq; d += nwaves) {
        const int64_t b = p.id_off[d], e = p.id_off[d + 1];
        int64_t out = p.text_off[d];
        const int64_t out_end = p.text_off[d + 1];
        if (out_end <= out) continue;                                  // empty, failed or all-skipped sequence
        bool found = false;
        for (int64_t i0 = b; i0 < e; i0 += 64) {
            const I2tTok t = i2t_token(p, i0 + lane, e, found);
            const int inc = wave_incl_scan(t.len);
            const int total = __shfl(inc, 63, 64);
            s_pre[wv][lane] = inc - t.len; s_src[wv][lane] = t.src;
            if (lane == 63) s_pre[wv][64] = total;
            wave_handoff();
            for (int q = lane; q < total; q += 64) {
                int lo = 0, hi = 63;                                   // last token whose prefix is <= q (empty tokens share a prefix: the last one wins)
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_pre[wv][mid] <= q) lo = mid; else hi = mid - 1; }
                if (out + q < p.text_cap) p.text[out + q] = p.tok_data[s_src[wv][lo] + (uint32_t)(q - s_pre[wv][lo])];
            }
            out += total;
            wave_handoff();                                            // the next iteration overwrites s_pre / s_src
        }
    }
}

// Documents of more than W2T_LONG_TOKENS tokens leave the wave-per-document copy kernels (one wave assembles 64 tokens at a time, one
// block after the other: 2.2 us per block, 96 us for the 2,800 words of config 1's longest line, 10.6 ms for a 1 MB document of
// sentences) for k_w2t_copy_long: sixteen waves per document, each round 1024 tokens.
constexpr int W2T_LONG_TOKENS = 1024, W2T_LONG_WAVES = 16;
__device__ __forceinline__ bool w2t_list_long(const W2tParams &p, int64_t d, int64_t ntok, int lane)
{
    if (!p.long_list || ntok <= W2T_LONG_TOKENS) return false;
    if (lane == 0) {
        const unsigned slot = atomicAdd(p.long_count, 1u);
        if ((int64_t)slot < p.long_cap) p.long_list[slot] = d;
    }
    return true;
}


// come from the caller's text, every word but the first of its document is preceded by one ' '.
__global__ __launch_bounds__(256) void k_w2t_len(W2tParams p)
{
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave_in_block(), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t d = wave0; d < p.ndocs; d += nwaves) {
        const int64_t b = p.word_off[d], e = p.word_off[d + 1];
        if (w2t_list_long(p, d, e - b, lane)) continue;
        long long total = 0;
        for (int64_t i = b + lane; i < e; i += 64) total += (long long)(p.ends[i] - p.starts[i] + 1) + (i > b ? 1 : 0);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
        if (lane == 0) p.lens[d] = total > 0x7ffffff0ll ? 0 : (int32_t)total;
    }
}

__global__ __launch_bounds__(256) void k_w2t_copy(W2tParams p)
{
    __shared__ int s_pre[4][65];
    __shared__ int s_src[4][64];
    const int lane = lane_id(), wv = wave_in_block();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    for (int64_t d = wave0; d < p.ndocs; d += nwaves) {
        const int64_t b = p.word_off[d], e = p.word_off[d + 1];
        int64_t out = p.text_off[d];
        if (p.text_off[d + 1] <= out) continue;
        if (w2t_list_long(p, d, e - b, lane)) continue;
        const uint8_t *src = p.text + p.doc_off[d];
        for (int64_t i0 = b; i0 < e; i0 += 64) {
            const int64_t i = i0 + lane;
            int len = 0, st = 0;
            if (i < e) { st = p.starts[i]; len = p.ends[i] - st + 1 + (i > b ? 1 : 0); }     // the separator is counted with the word it precedes
            const int inc = wave_incl_scan(len);
            const int total = __shfl(inc, 63, 64);
            s_pre[wv][lane] = inc - len; s_src[wv][lane] = st - ((i > b) ? 1 : 0);
            if (lane == 63) s_pre[wv][64] = total;
            wave_handoff();
            for (int q = lane; q < total; q += 64) {
                int lo = 0, hi = 63;
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_pre[wv][mid] <= q) lo = mid; else hi = mid - 1; }
                const int k = q - s_pre[wv][lo];
                uint8_t c;
                if (k == 0 && i0 + lo > b) c = ' ';                                         // tokdll:529-531
                else { c = src[s_src[wv][lo] + k]; if (c == ' ' || c == 0) c = '_'; }       // tokdll:482,543
                if (out + q < p.out_cap) p.out[out + q] = c;
            }
            out += total;
            wave_handoff();                                            // the next iteration overwrites s_pre / s_src
        }
    }
}

__device__ __forceinline__ bool dev_is_ws(int c)      // blingfiretokdll.h:17-21 __FAIsWhiteSpace__
{
    return c <= 0x20 || c == 0xa0 || (c >= 0x2000 && c <= 0x200f) || c == 0x202f || c == 0x205f || c == 0x2060 || c == 0x2420 || c == 0x2424 ||
           c == 0x3000 || c == 0xfeff;
}

// TextToSentences assembly (reference tokdll:257-339).  Sentence k of a document = bytes (end of token k-1) + 1 .. end of token k,
// plus one more from the last token to the end of the document; its lane skips the leading white space (FAGetFirstNonWhiteSpace,
// tokdll:138-150, on the valid UTF-8 the lexer accepted) and drops the sentence if nothing is left.  Every emitted sentence
// but the first is preceded by '\n' (counted with it, like the separator of k_w2t_*).
struct S2tTok { int len; int src; };
__device__ __forceinline__ S2tTok s2t_sentence(const W2tParams &p, const uint8_t *src, int n, int bom, int64_t b, int64_t e, int64_t i, bool &any_before, unsigned long long *emit_mask = nullptr)
{
    S2tTok t; t.len = 0; t.src = 0;
    bool emit = false;
    if (i <= e) {                                                        // i == e: the rest of the document (tokdll:307-311)
        const int from = i == b ? bom : p.ends[i - 1] + 1;
        const int to = i < e ? p.ends[i] : n - 1;
        int q = from;
        if (i < e || from < n) {
            while (q <= to) {
                const unsigned b0 = src[q]; const int len = b0 < 0x80 ? 1 : b0 < 0xE0 ? 2 : b0 < 0xF0 ? 3 : 4; int cp = (int)b0;
                if (len == 2) cp = ((b0 & 0x1F) << 6) | (src[q + 1] & 0x3F);
                else if (len == 3) cp = ((b0 & 0x0F) << 12) | ((src[q + 1] & 0x3F) << 6) | (src[q + 2] & 0x3F);
                else if (len == 4) cp = ((b0 & 0x07) << 18) | ((src[q + 1] & 0x3F) << 12) | ((src[q + 2] & 0x3F) << 6) | (src[q + 3] & 0x3F);
                if (!dev_is_ws(cp)) break;                               // (U+0000 counts as U+0020, tokdll:233)
                q += len;
            }
            if (q <= to) { emit = true; t.len = to - q + 1; t.src = q; }
        }
    }
    const unsigned long long me = __ballot(emit);
    if (emit_mask) *emit_mask = me;
    const bool sep = emit && (any_before || (me & lanemask_lt()) != 0);
    if (sep) { t.len += 1; t.src -= 1; }                                // the '\n' in front of it
    else if (emit) t.src = -t.src - 2;                                  // marks "no separator" (decoded in the copy loop)
    if (me) any_before = true;
    return t;
}

__global__ __launch_bounds__(256) void k_s2t_len(W2tParams p)
{
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave_in_block(), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t d = wave0; d < p.ndocs; d += nwaves) {
        const int64_t b = p.word_off[d], e = p.word_off[d + 1];
        const int64_t n64 = p.doc_off[d + 1] - p.doc_off[d];
        if (n64 > 0 && n64 <= 1000000000 && w2t_list_long(p, d, e - b, lane)) continue;
        const uint8_t *src = p.text + p.doc_off[d];
        long long total = 0;
        if (n64 > 0 && n64 <= 1000000000) {
            const int n = (int)n64;
            const int bom = (n >= 3 && src[0] == 0xEF && src[1] == 0xBB && src[2] == 0xBF) ? 3 : 0;
            bool any_before = false;
            for (int64_t i0 = b; i0 <= e; i0 += 64) { const S2tTok t = s2t_sentence(p, src, n, bom, b, e, i0 + lane, any_before); total += t.len; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
        // a document the single call rejects (invalid UTF-8: no characters decoded) yields nothing: its lens entry is forced to 0 by nvalid
        if (lane == 0) p.lens[d] = (total > 0x7ffffff0ll || (p.nvalid && p.nvalid[d] <= 0)) ? 0 : (int32_t)total;
    }
}

__global__ __launch_bounds__(256) void k_s2t_copy(W2tParams p)
{
    __shared__ int s_pre[4][65];
    __shared__ int s_src[4][64];
    const int lane = lane_id(), wv = wave_in_block();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    for (int64_t d = wave0; d < p.ndocs; d += nwaves) {
        const int64_t b = p.word_off[d], e = p.word_off[d + 1];
        int64_t out = p.text_off[d];
        if (p.text_off[d + 1] <= out) continue;
        if (w2t_list_long(p, d, e - b, lane)) continue;
        const int n = (int)(p.doc_off[d + 1] - p.doc_off[d]);
        const uint8_t *src = p.text + p.doc_off[d];
        const int bom = (n >= 3 && src[0] == 0xEF && src[1] == 0xBB && src[2] == 0xBF) ? 3 : 0;
        bool any_before = false;
        for (int64_t i0 = b; i0 <= e; i0 += 64) {
            const S2tTok t = s2t_sentence(p, src, n, bom, b, e, i0 + lane, any_before);
            const int inc = wave_incl_scan(t.len);
            const int total = __shfl(inc, 63, 64);
            s_pre[wv][lane] = inc - t.len; s_src[wv][lane] = t.src;
            if (lane == 63) s_pre[wv][64] = total;
            wave_handoff();
            for (int q = lane; q < total; q += 64) {
                int lo = 0, hi = 63;
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_pre[wv][mid] <= q) lo = mid; else hi = mid - 1; }
                const int k = q - s_pre[wv][lo];
                const int sv = s_src[wv][lo];
                uint8_t c;
                if (sv >= -1) { if (k == 0) c = '\n'; else { c = src[sv + k]; if (c == '\n' || c == 0) c = ' '; } }     // tokdll:291-296
                else { c = src[(-sv - 2) + k]; if (c == '\n' || c == 0) c = ' '; }
                if (out + q < p.out_cap) p.out[out + q] = c;
            }
            out += total;
            wave_handoff();                                            // the next iteration overwrites s_pre / s_src
        }
    }
}

// the output length of a listed document, sixteen waves: the sum does not depend on the order (a sentence's '\n' is counted with every emitted
// one and taken back once for the document's first)
template <bool SENT>
__global__ __launch_bounds__(W2T_LONG_WAVES * 64) void k_w2t_len_long(W2tParams p)
{
    __shared__ long long s_sum[W2T_LONG_WAVES];
    __shared__ int s_any[W2T_LONG_WAVES];
    const int lane = lane_id(), wv = wave_in_block();
    int64_t nlist = (int64_t)*p.long_count; if (nlist > p.long_cap) nlist = p.long_cap;
    for (int64_t j = blockIdx.x; j < nlist; j += gridDim.x) {
        const int64_t d = p.long_list[j];
        const int64_t b = p.word_off[d], e = p.word_off[d + 1];
        const int n = (int)(p.doc_off[d + 1] - p.doc_off[d]);
        const uint8_t *src = p.text + p.doc_off[d];
        const int bom = (SENT && n >= 3 && src[0] == 0xEF && src[1] == 0xBB && src[2] == 0xBF) ? 3 : 0;
        const int64_t last = SENT ? e : e - 1;
        long long total = 0; bool any = false;
        for (int64_t i0 = b + 64 * wv; i0 <= last; i0 += 64 * W2T_LONG_WAVES) {
            const int64_t i = i0 + lane;
            if constexpr (SENT) {
                bool all_before = true; unsigned long long me = 0;
                const S2tTok t = s2t_sentence(p, src, n, bom, b, e, i, all_before, &me);
                total += t.len; any = any || me != 0;
            } else if (i < e) total += (long long)(p.ends[i] - p.starts[i] + 1) + (i > b ? 1 : 0);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
        if (lane == 0) { s_sum[wv] = total; s_any[wv] = any ? 1 : 0; }
        __syncthreads();
        if (threadIdx.x == 0) {
            long long t = 0; int a = 0;
            for (int w = 0; w < W2T_LONG_WAVES; ++w) { t += s_sum[w]; a |= s_any[w]; }
            if (SENT && a) t -= 1;
            p.lens[d] = (t > 0x7ffffff0ll || (SENT && p.nvalid && p.nvalid[d] <= 0)) ? 0 : (int32_t)t;
        }
        __syncthreads();
    }
}

template <bool SENT>
static void launch_w2t_len_long(const W2tParams &p, hipStream_t s)
{
    if (!p.long_list) return;
    int64_t nb = p.long_cap < device_cus() ? p.long_cap : device_cus();
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_w2t_len_long<SENT>, dim3((unsigned)nb), dim3(W2T_LONG_WAVES * 64), 0, s, p);
}

void launch_s2t_len(const W2tParams &p, hipStream_t s)
{
    int64_t blocks = (p.ndocs + 3) / 4; if (blocks > device_cus() * 16) blocks = device_cus() * 16; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_s2t_len, dim3((unsigned)blocks), dim3(256), 0, s, p);
    launch_w2t_len_long<true>(p, s);
}
template <bool SENT>
__global__ __launch_bounds__(W2T_LONG_WAVES * 64) void k_w2t_copy_long(W2tParams p)
{
    __shared__ int s_pre[W2T_LONG_WAVES][65];
    __shared__ int s_src[W2T_LONG_WAVES][64];
    __shared__ int s_tot[W2T_LONG_WAVES], s_emit[W2T_LONG_WAVES];
    const int lane = lane_id(), wv = wave_in_block();
    int64_t nlist = (int64_t)*p.long_count; if (nlist > p.long_cap) nlist = p.long_cap;
    for (int64_t j = blockIdx.x; j < nlist; j += gridDim.x) {
        const int64_t d = p.long_list[j];
        const int64_t b = p.word_off[d], e = p.word_off[d + 1];
        int64_t out0 = p.text_off[d];
        const int n = (int)(p.doc_off[d + 1] - p.doc_off[d]);
        const uint8_t *src = p.text + p.doc_off[d];
        const int bom = (SENT && n >= 3 && src[0] == 0xEF && src[1] == 0xBB && src[2] == 0xBF) ? 3 : 0;
        bool doc_any_before = false;                                       // SENT: a sentence was emitted in an earlier round
        const int64_t last = SENT ? e : e - 1;                             // SENT: index e is the rest of the document (tokdll:307-311)
        for (int64_t r0 = b; r0 <= last; r0 += 64 * W2T_LONG_WAVES) {
            const int64_t i0 = r0 + 64 * wv, i = i0 + lane;
            int len = 0, sv = 0;
            if constexpr (SENT) {
                // every emitted sentence with its '\n' in front; the document's first one gives it back below
                bool all_before = true; unsigned long long me = 0;
                S2tTok t = s2t_sentence(p, src, n, bom, b, e, i, all_before, &me);
                if (lane == 0) s_emit[wv] = me != 0;
                __syncthreads();
                int first_wave = -1;
                for (int w = W2T_LONG_WAVES - 1; w >= 0; --w) if (s_emit[w]) first_wave = w;
                if (!doc_any_before && wv == first_wave && lane == __ffsll((long long)me) - 1) { t.len -= 1; t.src = -(t.src + 1) - 2; }
                doc_any_before = doc_any_before || first_wave >= 0;
                len = t.len; sv = t.src;
            } else if (i < e) { const int st = p.starts[i]; len = p.ends[i] - st + 1 + (i > b ? 1 : 0); sv = st - (i > b ? 1 : 0); }
            const int inc = wave_incl_scan(len);
            const int total = __shfl(inc, 63, 64);
            s_pre[wv][lane] = inc - len; s_src[wv][lane] = sv;
            if (lane == 63) s_pre[wv][64] = total;
            if (lane == 0) s_tot[wv] = total;
            __syncthreads();
            int64_t out = out0; int round_total = 0;
            for (int w = 0; w < W2T_LONG_WAVES; ++w) { if (w < wv) out += s_tot[w]; round_total += s_tot[w]; }
            for (int q = lane; q < total; q += 64) {
                int lo = 0, hi = 63;
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_pre[wv][mid] <= q) lo = mid; else hi = mid - 1; }
                const int k = q - s_pre[wv][lo];
                const int v = s_src[wv][lo];
                uint8_t c;
                if constexpr (SENT) {
                    if (v >= -1) { if (k == 0) c = '\n'; else { c = src[v + k]; if (c == '\n' || c == 0) c = ' '; } }     // tokdll:291-296
                    else { c = src[(-v - 2) + k]; if (c == '\n' || c == 0) c = ' '; }
                } else {
                    if (k == 0 && i0 + lo > b) c = ' ';                                         // tokdll:529-531
                    else { c = src[v + k]; if (c == ' ' || c == 0) c = '_'; }                   // tokdll:482,543
                }
                if (out + q < p.out_cap) p.out[out + q] = c;
            }
            out0 += round_total;
            __syncthreads();                                               // the next round overwrites s_pre / s_src / s_tot / s_emit
        }
    }
}

template <bool SENT>
static void launch_w2t_copy_long(const W2tParams &p, hipStream_t s)
{
    if (!p.long_list) return;
    int64_t nb = p.long_cap < device_cus() ? p.long_cap : device_cus();       // (the number of listed documents is on the device)
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_w2t_copy_long<SENT>, dim3((unsigned)nb), dim3(W2T_LONG_WAVES * 64), 0, s, p);
}

void launch_s2t_copy(const W2tParams &p, hipStream_t s)
{
    int64_t blocks = (p.ndocs + 3) / 4; if (blocks > device_cus() * 16) blocks = device_cus() * 16; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_s2t_copy, dim3((unsigned)blocks), dim3(256), 0, s, p);
    launch_w2t_copy_long<true>(p, s);
}

void launch_w2t_len(const W2tParams &p, hipStream_t s)
{
    int64_t blocks = (p.ndocs + 3) / 4; if (blocks > device_cus() * 16) blocks = device_cus() * 16; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_w2t_len, dim3((unsigned)blocks), dim3(256), 0, s, p);
    launch_w2t_len_long<false>(p, s);
}
void launch_w2t_copy(const W2tParams &p, hipStream_t s)
{
    int64_t blocks = (p.ndocs + 3) / 4; if (blocks > device_cus() * 16) blocks = device_cus() * 16; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_w2t_copy, dim3((unsigned)blocks), dim3(256), 0, s, p);
    launch_w2t_copy_long<false>(p, s);
}

// ------------------------------------------------------------------------------------------
// NormalizeSpaces (reference tokdll:629-679), wave per document, 64 bytes per iteration.  The sequential rule "a white-space
// character becomes uSpace unless nothing was written yet or the last written character equals uSpace" is local: a white-space
// character is written iff the character before it exists, is not white space and is not uSpace itself.  One trailing uSpace
// is trimmed (when more than one character was written).  Pass 1 sizes, pass 2 (WRITE) stores at the scanned offsets.
// ------------------------------------------------------------------------------------------

template <bool WRITE>
__global__ __launch_bounds__(256) void k_normsp(NormSpParams p)
{
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave_in_block(), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t d = wave0; d < p.ndocs; d += nwaves) {
        const int64_t b = p.doc_off[d];
        const int64_t n64 = p.doc_off[d + 1] - b;
        if (n64 <= 0 || n64 > 1000000000) { if (!WRITE && lane == 0) { p.lens[d] = 0; p.aux[d] = 1; } continue; }   // tokdll:634-636
        const int n = (int)n64;
        const uint8_t *s = p.text + b;
        uint8_t *out = WRITE ? p.out + p.out_off[d] : nullptr;
        const int64_t room = WRITE ? p.out_off[d + 1] - p.out_off[d] : 0;
        if (WRITE && room <= 0) continue;
        const int bom = (n >= 3 && s[0] == 0xEF && s[1] == 0xBB && s[2] == 0xBF) ? 3 : 0;    // FAUtf8Utils.cpp:247-252
        bool bad = false, c_exists = false, c_ws = false, c_eq = false;    // the last character of the previous window
        int nchars = 0, nwritten = 0, nspaces = 0, last_len = 0; bool last_usp = false, last_norm = false;
        long long total = 0;
        for (int pos = bom; pos < n; pos += 64) {
            const int q = pos + lane;
            const bool in = q < n;
            uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
            if (in) b0 = s[q];
            if (q + 1 < n) b1 = s[q + 1];
            if (q + 2 < n) b2 = s[q + 2];
            if (q + 3 < n) b3 = s[q + 3];
            const bool cont = (b0 & 0xC0) == 0x80;
            const bool start = in && !cont;
            bool err = false; int cp = (int)b0, len = 1;
            if (in && cont) {                                            // must be covered by a lead (FAUtf8Utils.cpp:152-165)
                const uint32_t p1 = (q - 1 >= bom) ? s[q - 1] : 0x80u, p2 = (q - 2 >= bom) ? s[q - 2] : 0x80u, p3 = (q - 3 >= bom) ? s[q - 3] : 0x80u;
                bool ok;
                if ((p1 & 0xC0) != 0x80) ok = (q - 1 >= bom) && p1 >= 0xC0;
                else if ((p2 & 0xC0) != 0x80) ok = (q - 2 >= bom) && p2 >= 0xE0;
                else if ((p3 & 0xC0) != 0x80) ok = (q - 3 >= bom) && p3 >= 0xF0;
                else ok = false;
                err = !ok;
            } else if (start && b0 >= 0x80) {
                if ((b0 & 0xE0) == 0xC0) { len = 2; cp = (int)(b0 & 0x1F); }
                else if ((b0 & 0xF0) == 0xE0) { len = 3; cp = (int)(b0 & 0x0F); }
                else if ((b0 & 0xF8) == 0xF0) { len = 4; cp = (int)(b0 & 0x07); }
                else { len = 1; err = true; }
                if (q + len > n) err = true;
                if (len >= 2) { if ((b1 & 0xC0) != 0x80) err = true; cp = (cp << 6) | (int)(b1 & 0x3F); }
                if (len >= 3) { if ((b2 & 0xC0) != 0x80) err = true; cp = (cp << 6) | (int)(b2 & 0x3F); }
                if (len >= 4) { if ((b3 & 0xC0) != 0x80) err = true; cp = (cp << 6) | (int)(b3 & 0x3F); }
                const int need = cp <= 0x7F ? 1 : cp <= 0x7FF ? 2 : cp <= 0xFFFF ? 3 : cp <= 0x10FFFF ? 4 : 0;
                if (need != len) err = true;
                if ((cp & 0xFFFFF800) == 0xD800) err = true;
            }
            bad |= err;
            const bool ws = start && dev_is_ws(cp), eq = start && cp == p.u_space;
            const unsigned long long m_start = __ballot(start);
            const unsigned long long below = m_start & lanemask_lt();
            const int prev = below ? 63 - __clzll((long long)below) : -1;
            const int pws = __shfl((int)ws, prev < 0 ? 0 : prev, 64), peq = __shfl((int)eq, prev < 0 ? 0 : prev, 64);
            const bool prev_exists = prev >= 0 || c_exists, prev_ws = prev >= 0 ? pws != 0 : c_ws, prev_eq = prev >= 0 ? peq != 0 : c_eq;
            const bool norm = ws && prev_exists && !prev_ws && !prev_eq;                // this white space becomes one uSpace
            const bool written = start && (!ws || norm);
            const int ob = !written ? 0 : (ws ? p.usp_len : len);
            const int inc = wave_incl_scan(ob);
            const unsigned long long m_w = __ballot(written);
            if (m_w) {
                const int lw = 63 - __clzll((long long)m_w);
                last_usp = __shfl((int)(ws ? (int)norm : (int)eq), lw, 64) != 0;        // value of the last written character == uSpace
                last_norm = __shfl((int)norm, lw, 64) != 0;
                last_len = __shfl(ob, lw, 64);
            }
            if (WRITE && written) {
                const int64_t o = total + inc - ob;
                if (!ws) { for (int k = 0; k < len; ++k) if (o + k < room) out[o + k] = s[q + k]; }
                else for (int k = 0; k < p.usp_len; ++k) if (o + k < room) out[o + k] = (uint8_t)(p.usp_bytes >> (8 * k));
            }
            total += __shfl(inc, 63, 64);
            nchars += __popcll(m_start); nwritten += __popcll(m_w); nspaces += __popcll(__ballot(norm));
            if (m_start) { const int ls = 63 - __clzll((long long)m_start); c_exists = true; c_ws = __shfl((int)ws, ls, 64) != 0; c_eq = __shfl((int)eq, ls, 64) != 0; }
        }
        if (nwritten > 1 && last_usp) { total -= last_len; if (last_norm) --nspaces; }  // tokdll:667-669
        const bool any_bad = __any(bad) || nchars <= 0;                                  // tokdll:646-648
        if (!WRITE && lane == 0) {
            p.lens[d] = (any_bad || total > 0x7ffffff0ll) ? 0 : (int32_t)total;
            p.aux[d] = (any_bad ? 1 : 0) | ((nspaces > 0x3fffffff ? 0x3fffffff : (nspaces < 0 ? 0 : nspaces)) << 1);
        }
    }
}

void launch_normsp(const NormSpParams &p, bool write, hipStream_t s)
{
    int64_t blocks = (p.ndocs + 3) / 4; if (blocks > device_cus() * 16) blocks = device_cus() * 16; if (blocks < 1) blocks = 1;
    if (write) hipLaunchKernelGGL(k_normsp<true>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_normsp<false>, dim3((unsigned)blocks), dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------
// TextToHashes (reference tokdll:683-815): tokens are the byte strings between single spaces; hash of a token = the
// fasttext FNV-1a variant (bytes sign-extended); word n-grams are chained from the unigram hashes (sign-extended to 64 bits)
// and stored modulo the bucket count in blocks of `tokens` entries behind the unigrams.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hash_count(HashParams p)
{
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave_in_block(), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t d = wave0; d < p.ndocs; d += nwaves) {
        const int64_t b = p.doc_off[d], n = p.doc_off[d + 1] - b;
        long long sp = 0;
        for (int64_t q = lane; q < n; q += 64) sp += p.text[b + q] == ' ';
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sp += __shfl_xor(sp, o, 64);
        const long long cnt = (sp + 1) * (long long)p.ngrams;
        if (lane == 0) p.lens[d] = (n < 0 || cnt > 0x7ffffff0ll) ? 0 : (int32_t)cnt;
    }
}

__global__ __launch_bounds__(256) void k_hash_fill(HashParams p)
{
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave_in_block(), nwaves = (int64_t)gridDim.x * 4;
    // hash of "</s>" (tokdll:696)
    uint32_t eh = 2166136261u;
    { const char e4[4] = {'<', '/', 's', '>'}; for (int k = 0; k < 4; ++k) { eh ^= (uint32_t)(int8_t)e4[k]; eh *= 16777619u; } }
    const unsigned long long eos64 = (unsigned long long)(long long)(int32_t)eh;
    for (int64_t d = wave0; d < p.ndocs; d += nwaves) {
        const int64_t b = p.doc_off[d], n = p.doc_off[d + 1] - b;
        const int64_t o0 = p.out_off[d], room = p.out_off[d + 1] - o0;
        if (n < 0 || room <= 0) continue;
        const uint8_t *s = p.text + b;
        int32_t *out = p.out + o0;
        const int64_t tc = room / p.ngrams;                              // tokens = spaces + 1
        // unigrams: token 0 starts at 0, token k at the byte after the k-th space; its lane walks to the next space
        int64_t base = 0;                                                // spaces before this window
        for (int64_t pos = 0; pos < n || pos == 0; pos += 64) {
            const int64_t q = pos + lane;
            const bool sp = q < n && s[q] == ' ';
            const unsigned long long m = __ballot(sp);
            for (int r = 0; r < 2; ++r) {
                const bool mine = r == 0 ? sp : (q == 0);                // the token after my space; lane of position 0 also owns token 0
                if (mine) {
                    const int64_t idx = r == 0 ? base + __popcll(m & lanemask_lt()) + 1 : 0;
                    uint32_t h = 2166136261u;
                    for (int64_t k = r == 0 ? q + 1 : 0; k < n && s[k] != ' '; ++k) { h ^= (uint32_t)(int8_t)s[k]; h *= 16777619u; }   // tokdll:684-692
                    if (idx < tc) out[idx] = (int32_t)h;
                }
            }
            base += __popcll(m);
            if (n == 0) break;
        }
        __threadfence_block();                                           // the unigrams are read by other lanes below
        wave_handoff();
        for (int64_t i = lane; i < tc; i += 64) {                        // tokdll:699-714
            unsigned long long h = (unsigned long long)(long long)out[i];
            for (int j = 1; j < p.ngrams; ++j) {
                const unsigned long long t = (i + j < tc) ? (unsigned long long)(long long)out[i + j] : eos64;
                h = h * 116049371ull + t;
                out[(int64_t)j * tc + i] = (int32_t)(h % (unsigned long long)(long long)p.bucket);
            }
        }
    }
}

void launch_hash_count(const HashParams &p, hipStream_t s)
{
    int64_t blocks = (p.ndocs + 3) / 4; if (blocks > device_cus() * 16) blocks = device_cus() * 16; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_hash_count, dim3((unsigned)blocks), dim3(256), 0, s, p);
}
void launch_hash_fill(const HashParams &p, hipStream_t s)
{
    int64_t blocks = (p.ndocs + 3) / 4; if (blocks > device_cus() * 16) blocks = device_cus() * 16; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_hash_fill, dim3((unsigned)blocks), dim3(256), 0, s, p);
}

void launch_i2t_len(const I2tParams &p, hipStream_t s)
{
    int64_t blocks = (p.nseq + 3) / 4; if (blocks > device_cus() * 16) blocks = device_cus() * 16; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_i2t_len, dim3((unsigned)blocks), dim3(256), 0, s, p);
}
void launch_i2t_copy(const I2tParams &p, hipStream_t s)
{
    int64_t blocks = (p.nseq + 3) / 4; if (blocks > device_cus() * 16) blocks = device_cus() * 16; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_i2t_copy, dim3((unsigned)blocks), dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------
// Dictionary key -> info (reference FADictInterpreter_t<int>::GetInfo, cl/inc/FADictInterpreter_t.h:369-390): one key per lane
// (bf_seg.h dict_info_id), then the I2Info row of the id (FAMultiMap_pack_fixed::Get); values are gathered after a scan.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dict_ids(DictParams p)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < p.nkeys; k += stride) {
        const int64_t b = p.key_off[k], n64 = p.key_off[k + 1] - b;
        int id = -1;
        if (n64 > 0 && n64 <= DICT_MAX_WORD && b >= 0) id = dict_info_id(p.D, p.keys + b, (int)n64);
        int r = -1;
        if (id != -1 && id >= p.min_key && id - p.min_key < p.nrows) r = p.rows[(int64_t)(id - p.min_key) * p.stride];
        p.info_ids[k] = id; p.ret[k] = r; p.counts[k] = r > 0 ? r : 0;
    }
}

__global__ __launch_bounds__(256) void k_dict_fill(DictParams p)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < p.nkeys; k += stride) {
        const int c = p.counts[k];
        if (c <= 0) continue;
        const int32_t *row = p.rows + (int64_t)(p.info_ids[k] - p.min_key) * p.stride + 1;
        const int64_t o = p.val_off[k];
        for (int q = 0; q < c; ++q) if (o + q < p.vals_cap) p.vals[o + q] = row[q];
    }
}

void launch_dict_ids(const DictParams &p, hipStream_t s)
{
    int64_t blocks = (p.nkeys + 255) / 256; if (blocks > device_cus() * 16) blocks = device_cus() * 16; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_dict_ids, dim3((unsigned)blocks), dim3(256), 0, s, p);
}
void launch_dict_fill(const DictParams &p, hipStream_t s)
{
    int64_t blocks = (p.nkeys + 255) / 256; if (blocks > device_cus() * 16) blocks = device_cus() * 16; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_dict_fill, dim3((unsigned)blocks), dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------
// scan: counts[ndocs] (int32) -> id_off[ndocs+1] (int64), three small kernels
// ------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 4, SCAN_THREADS = 256, SCAN_TILE = SCAN_ITEMS * SCAN_THREADS;

int scan_nblocks(int64_t ndocs) { return (int)((ndocs + SCAN_TILE - 1) / SCAN_TILE); }

__device__ __forceinline__ long long block_excl_scan(long long v, long long *total, long long *sh /*[4]*/)
{
    // exclusive scan of one value per thread across a 256-thread block
    const int lane = lane_id(), wv = wave_in_block();
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
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave_in_block();
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t d = wave0; d < p.b.ndocs; d += nwaves) {
        const int c = p.counts[d];
        const int64_t b = p.b.doc_off[d];
        const int64_t stream_slot = p.slot_mul > 0 ? sp_slot(b, d, p.slot_mul) : b;
        const int64_t id_slot = (p.slot_mul > 0 ? stream_slot : ids_slot(b, d)) + (p.first ? p.first[d] : 0);
        const int32_t *src = p.ids_tmp + id_slot;
        const int64_t o = p.id_off[d];
        for (int i = lane; i < c; i += 64) {
            if (o + i < p.ids_cap) {
                p.ids_out[o + i] = src[i];
                if (p.starts_out) {
                    // stream positions -> byte offsets of the source characters; the end offset is inclusive and
                    // covers the whole last character (FAUtf8Size of its first byte; tokdll:1270-1272,1527-1528)
                    const int from = p.span_tmp[2 * (id_slot + i)], to = p.span_tmp[2 * (id_slot + i) + 1];
                    const int so = from >= 0 ? p.src_off[stream_slot + from] : -1;
                    const int eo = to >= 0 ? p.src_off[stream_slot + to] : -1;
                    int sz = 0;
                    if (eo >= 0) { const uint32_t ch = p.b.text[b + eo]; sz = (ch & 0x80) == 0 ? 1 : (ch & 0xE0) == 0xC0 ? 2 : (ch & 0xF0) == 0xE0 ? 3 : (ch & 0xF8) == 0xF0 ? 4 : 0; }
                    p.starts_out[o + i] = so; p.ends_out[o + i] = eo + (sz > 0 ? sz - 1 : 0);
                }
            } else if (i == lane) atomicOr(p.status, 1);
        }
    }
}

// k_compact_text: the copy of k_compact for the offsets instance of the wave program, which stages [first, last] CHARACTER of every id and
// no character -> byte stream: the byte offsets come from the text itself.  A character starts at every byte that is not a continuation byte
// (a leading BOM is none, FAUtf8Utils.cpp:247-252; documents with invalid UTF-8 have no ids), so a 64-byte block's start bytes are one ballot;
// the ids of a document are in text order, so the ones whose first (last) character lies in the block are the next few of the list:
// one per lane, the k-th start byte of the block by a six-step search over population counts.  End offset = the last character's first
// byte + its UTF-8 size - 1 (tokdll:1270-1272).  Wave per document.
__device__ __forceinline__ int select_bit64(unsigned long long m, int k)       // position of the k-th (0-based) set bit of m; k < popcount(m)
{
    int pos = 0;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
        const int c = __popcll((m >> pos) & ((1ull << step) - 1ull));
        if (k >= c) { pos += step; k -= c; }
    }
    return pos;
}

__global__ __launch_bounds__(256) void k_compact_text(CompactParams p)
{
    if (p.only_if && *p.only_if == 0u) return;
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave_in_block();
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    bool over = false;
    // a wave takes 64 consecutive documents: count, text range and place in the output are read once, one document per lane (as k_compact_ids does)
    for (int64_t dbase = wave0 * 64; dbase < p.b.ndocs; dbase += nwaves * 64) {
      int c_l = 0, n_l = 0; int64_t b_l = 0, o_l = 0;
      if (dbase + lane < p.b.ndocs) {
          c_l = p.counts[dbase + lane]; b_l = p.b.doc_off[dbase + lane]; n_l = (int)(p.b.doc_off[dbase + lane + 1] - b_l); o_l = p.id_off[dbase + lane];
          if (o_l + c_l > p.ids_cap) { over = true; c_l = 0; }
      }
      const int nd = p.b.ndocs - dbase < 64 ? (int)(p.b.ndocs - dbase) : 64;
      for (int dk = 0; dk < nd; ++dk) {
        const int c = __builtin_amdgcn_readlane(c_l, dk);
        if (c <= 0) continue;
        const int64_t d = dbase + dk, b = wv::bcast(b_l, dk), o = wv::bcast(o_l, dk);
        const int n = __builtin_amdgcn_readlane(n_l, dk);
        const int64_t id_slot = ids_slot(b, d);
        const int32_t *src = p.ids_tmp + id_slot;
        const int2 *span = (const int2 *)(p.span_tmp + 2 * id_slot);
        const uint8_t *t = p.b.text + b;
        uint32_t v = lane < n ? (uint32_t)t[lane] : 0x80u;           // the first block travels with the spans
        const int bom = (n >= 3 && t[0] == 0xEF && t[1] == 0xBB && t[2] == 0xBF) ? 3 : 0;
        int blk = 0, cbase = 0;                                        // the block under the lanes, characters before it
        // the ids in groups of 128, two per lane: their spans are read once, before the blocks they fall into are looked at (what depends on
        // what: nothing but the running character count -- the block loads are issued one block ahead)
        for (int g0 = 0; g0 < c; g0 += 128) {
            const int i0 = g0 + lane, i1 = g0 + 64 + lane;
            const int2 s0 = i0 < c ? span[i0] : make_int2(0x7fffffff, 0x7fffffff), s1 = i1 < c ? span[i1] : make_int2(0x7fffffff, 0x7fffffff);
            if (i0 < c) p.ids_out[o + i0] = src[i0];
            if (i1 < c) p.ids_out[o + i1] = src[i1];
            int gmax = max(i0 < c ? s0.y : -1, i1 < c ? s1.y : -1);           // the group's last character
            for (int sh = 32; sh >= 1; sh >>= 1) gmax = max(gmax, __shfl_xor(gmax, sh, 64));
            int so0 = -1, so1 = -1, eo0 = -1, eo1 = -1;                        // the byte offsets of the lane's two ids: stored once, whole rows
            while (blk < n) {
                const int qn = blk + 64 + lane;
                const uint32_t vn = qn < n ? (uint32_t)t[qn] : 0x80u;        // the next block
                const int q = blk + lane;
                const unsigned long long M = __ballot(q < n && q >= bom && (v & 0xC0u) != 0x80u);
                const int nchar = __popcll(M);
                const bool dense = __ballot(q < n && q >= bom && v < 0x80u) == (n - blk >= 64 ? ~0ull : ((1ull << (n - blk)) - 1ull));     // plain ASCII: character k of the block is byte k, one byte long
                const int lo = cbase, hi = cbase + nchar;
                // (dense is wave-uniform: the six-step search runs only for blocks that hold a multi-byte character or the BOM)
#define BF_CT_END(ct, EO, POS) { const bool in = (ct) >= lo && (ct) < hi; const int pos = in ? (POS) : 0; const uint32_t ch = __shfl(v, pos, 64); \
                                 const int sz = (ch & 0x80) == 0 ? 1 : (ch & 0xE0) == 0xC0 ? 2 : (ch & 0xF0) == 0xE0 ? 3 : (ch & 0xF8) == 0xF0 ? 4 : 0; if (in) EO = blk + pos + (sz > 0 ? sz - 1 : 0); }
                if (dense) {
                    if (s0.x >= lo && s0.x < hi) so0 = blk + s0.x - lo;
                    if (s1.x >= lo && s1.x < hi) so1 = blk + s1.x - lo;
                    if (s0.y >= lo && s0.y < hi) eo0 = blk + s0.y - lo;        // an ASCII character is one byte
                    if (s1.y >= lo && s1.y < hi) eo1 = blk + s1.y - lo;
                } else {
                    if (s0.x >= lo && s0.x < hi) so0 = blk + select_bit64(M, s0.x - lo);
                    if (s1.x >= lo && s1.x < hi) so1 = blk + select_bit64(M, s1.x - lo);
                    BF_CT_END(s0.y, eo0, select_bit64(M, s0.y - lo)) BF_CT_END(s1.y, eo1, select_bit64(M, s1.y - lo))
                }
#undef BF_CT_END
                if (gmax < hi) break;                                   // every id of the group is placed: the next group goes on in this block
                cbase = hi; blk += 64; v = vn;
            }
            if (i0 < c) { p.starts_out[o + i0] = so0; p.ends_out[o + i0] = eo0; }
            if (i1 < c) { p.starts_out[o + i1] = so1; p.ends_out[o + i1] = eo1; }
        }
      }
    }
    if (over) atomicOr(p.status, 1);
}

// k_compact_ids: the same copy when only ids are asked for (no offsets).  A wave takes 64 consecutive documents: their count, slot and
// place in the output are read once, one document per lane (three coalesced loads instead of three dependent loads per document),
// then handed round with readlane; the copies of two documents are in flight together.  Measured on the 10 M x 512 B workload
// (profiles/r03_*): 3.56 ms with the wave-per-document form, whose waves spend most of their time waiting for those three loads.
__global__ __launch_bounds__(256) void k_compact_ids(CompactParams p)
{
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave_in_block();
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int32_t *__restrict__ tmp = p.ids_tmp;
    int32_t *__restrict__ out = p.ids_out;
    bool over = false;
    for (int64_t base = wave0 * 64; base < p.b.ndocs; base += nwaves * 64) {
        const int64_t d = base + lane;
        int c = 0; int64_t slot = 0, o = 0;
        if (d < p.b.ndocs) {
            c = p.counts[d];
            const int64_t b = p.b.doc_off[d];
            slot = (p.slot_mul > 0 ? sp_slot(b, d, p.slot_mul) : ids_slot(b, d)) + (p.first ? p.first[d] : 0);
            o = p.id_off[d];
            if (o + c > p.ids_cap) { over = true; c = o < p.ids_cap ? (int)(p.ids_cap - o) : 0; }
        }
        const int nd = p.b.ndocs - base < 64 ? (int)(p.b.ndocs - base) : 64;
        for (int k = 0; k < nd; k += 2) {
            const int k1 = k + 1 < nd ? k + 1 : k;
            const int c0 = __builtin_amdgcn_readlane(c, k), c1 = k + 1 < nd ? __builtin_amdgcn_readlane(c, k1) : 0;
            const int64_t s0 = wv::bcast(slot, k), s1 = wv::bcast(slot, k1), o0 = wv::bcast(o, k), o1 = wv::bcast(o, k1);
            int32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
            if (lane < c0) a0 = tmp[s0 + lane];
            if (lane + 64 < c0) a1 = tmp[s0 + lane + 64];
            if (lane < c1) b0 = tmp[s1 + lane];
            if (lane + 64 < c1) b1 = tmp[s1 + lane + 64];
            if (lane < c0) out[o0 + lane] = a0;
            if (lane + 64 < c0) out[o0 + lane + 64] = a1;
            if (lane < c1) out[o1 + lane] = b0;
            if (lane + 64 < c1) out[o1 + lane + 64] = b1;
            for (int i = lane + 128; i < c0; i += 64) out[o0 + i] = tmp[s0 + i];
            for (int i = lane + 128; i < c1; i += 64) out[o1 + i] = tmp[s1 + i];
        }
    }
    if (over) atomicOr(p.status, 1);
}

void launch_compact(const CompactParams &p, hipStream_t s)
{
    if (!p.starts_out) {
        int64_t blocks = (p.b.ndocs + 255) / 256;
        if (blocks > device_cus() * 8) blocks = device_cus() * 8;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k_compact_ids, dim3((unsigned)blocks), dim3(256), 0, s, p);
        return;
    }
    int64_t blocks = (p.b.ndocs + 3) / 4;
    if (blocks > device_cus() * 16) blocks = device_cus() * 16;
    if (blocks < 1) blocks = 1;
    if (!p.src_off) hipLaunchKernelGGL(k_compact_text, dim3((unsigned)blocks), dim3(256), 0, s, p);      // the wave program's offsets instance: spans in characters, no character -> byte stream
    else hipLaunchKernelGGL(k_compact, dim3((unsigned)blocks), dim3(256), 0, s, p);
}

} // namespace bfa
