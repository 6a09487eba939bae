// bf_wave_body.h -- the wave program of bf_wave.h (see there).  Include AFTER a definition of namespace wv:
//   bf_kernels.hip   wave intrinsics of gfx950
//   tests/hosttest   wave_emu.h, the 64-fibre simulator (test only)
#pragma once
#include "bf_wave.h"

namespace bfa {

// LDS of one wave
template <int RING_, int QCAP_, int DTN_>
struct WvLds {
    static constexpr int RING = RING_, QCAP = QCAP_, DTN = DTN_;
    alignas(16) uint16_t ring[RING];
    int64_t dt_slot[DTN], dt_doc[DTN];
    uint32_t q0[QCAP], q1[QCAP];     // token: ring position | length << 16; action info
    int32_t rcnt[QCAP], rid[4 * QCAP];
    int32_t dt_cap[DTN], dt_cnt[DTN]; uint32_t dt_flags[DTN];
    uint16_t q2[QCAP];               // token: document table entry
};

template <class LDS, int UNROLL = 2, bool STATS = false>
struct WpWave {
    static constexpr int RING = LDS::RING, QCAP = LDS::QCAP, DTN = LDS::DTN;
    static constexpr uint32_t RMASK = RING - 1;
    static_assert((RING & (RING - 1)) == 0 && RING >= 1024 && RING <= 32768, "ring size");

    const WpWaveParams &p; LDS &S; const uint16_t *ascii;
    int lane;
    // ---- wave-uniform state
    uint32_t rhi, rlo;               // absolute ring positions: next element to write / oldest element still needed
    int qn, dn;
    // current document
    const uint8_t *s; int n; uint32_t rbase; int dec_bytes, dec, done, bom, curk;
    bool err;                        // per lane: this lane saw invalid UTF-8 in the current document
    unsigned long long st_win, st_slow, st_flush, st_tok, st_trips, st_steps, st_rewalk;

    BF_WVD WpWave(const WpWaveParams &p_, LDS &S_, const uint16_t *ascii_) : p(p_), S(S_), ascii(ascii_)
    {
        lane = wv::lane(); rhi = rlo = 0; qn = dn = 0; s = nullptr; n = 0; rbase = 0; dec_bytes = dec = done = bom = curk = 0; err = false;
        st_win = st_slow = st_flush = st_tok = st_trips = st_steps = st_rewalk = 0;
    }

    BF_WVD int ring_free() const { return RING - (int)(rhi - rlo); }
    BF_WVD uint32_t ring_at(uint32_t abs_pos) const { return S.ring[abs_pos & RMASK]; }

    // ------------------------------------------------------------------------------------------------------------------
    // decode: the next WV_CHUNK bytes of the current document -> ring elements.  Strict UTF-8, the rules of the sequential
    // decoder restated per byte position (FAUtf8Utils.cpp:121-196,233-270; same scheme as k_prep_wp): a continuation byte is
    // no character and must be covered by a lead 1..3 bytes before it; a lead byte gives length, checks its continuation
    // bytes, truncation (:167-171), overlong / > U+10FFFF (:185-188), surrogates (:190-193).  One leading BOM is skipped.
    // ------------------------------------------------------------------------------------------------------------------
    BF_WVD void decode_chunk()
    {
        const int pos = dec_bytes;
        const int q0 = pos + lane * 8;
        uint64_t own = 0;
        int nb = n - q0; nb = nb < 0 ? 0 : (nb > 8 ? 8 : nb);
        if (nb == 8) __builtin_memcpy(&own, s + q0, 8);
        else for (int k = 0; k < nb; ++k) own |= (uint64_t)s[q0 + k] << (8 * k);
        const uint32_t w0 = rbase + (uint32_t)dec;                  // absolute ring position of this chunk's first element
        if (!wv::any((own & 0x8080808080808080ull) != 0)) {
            // plain ASCII (a BOM is not): stream position == byte position, every byte through the 128-entry table
            uint32_t e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = ascii[(uint32_t)(own >> (8 * k)) & 0x7f];
            const uint32_t r = w0 + (uint32_t)(lane * 8);
            if ((w0 & 7) == 0 && nb == 8) {
                uint32_t *dst = (uint32_t *)(S.ring + (r & RMASK));  // 8 elements = one 16-byte row, never wraps (RING % 8 == 0)
                dst[0] = e[0] | (e[1] << 16); dst[1] = e[2] | (e[3] << 16); dst[2] = e[4] | (e[5] << 16); dst[3] = e[6] | (e[7] << 16);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k < nb) S.ring[(r + (uint32_t)k) & RMASK] = (uint16_t)e[k];
            }
            const int total = n - pos < WV_CHUNK ? n - pos : WV_CHUNK;
            dec += total; dec_bytes = pos + WV_CHUNK; rhi = rbase + (uint32_t)dec;
            wv::sync();
            return;
        }
        uint32_t nxt = wv::shfl_down((uint32_t)own, 1);
        if (lane == 63) { nxt = 0; for (int k = 0; k < 3; ++k) if (q0 + 8 + k < n) nxt |= (uint32_t)s[q0 + 8 + k] << (8 * k); }
        const uint32_t tail3 = (uint32_t)(own >> 40);                     // own bytes 5, 6, 7
        uint32_t prv = wv::shfl_up(tail3, 1);
        if (lane == 0) { prv = 0; for (int k = 0; k < 3; ++k) if (pos - 3 + k >= 0) prv |= (uint32_t)s[pos - 3 + k] << (8 * k); }
        if (pos == 0) {                                                   // FAUtf8Utils.cpp:247-252
            const int has_bom = (n >= 3 && ((uint32_t)own & 0xFFFFFFu) == 0xBFBBEFu) ? 3 : 0;
            bom = wv::bcast(has_bom, 0);
        }
        uint32_t X[14];                                                   // X[i] = byte at q0 - 3 + i
        X[0] = prv & 0xFF; X[1] = (prv >> 8) & 0xFF; X[2] = (prv >> 16) & 0xFF;
#pragma unroll
        for (int k = 0; k < 8; ++k) X[3 + k] = (uint32_t)(own >> (8 * k)) & 0xFF;
        X[11] = nxt & 0xFF; X[12] = (nxt >> 8) & 0xFF; X[13] = (nxt >> 16) & 0xFF;
        uint32_t v[8]; int cnt = 0; uint32_t wm = 0; bool e_any = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = q0 + k;
            const bool in = q >= bom && q < n;
            const uint32_t b0 = X[3 + k], b1 = X[4 + k], b2 = X[5 + k], b3 = X[6 + k];
            const bool cont = (b0 & 0xC0) == 0x80;
            uint32_t vv = 0; bool has = false;
            if (in && cont) {
                const uint32_t p1 = X[2 + k], p2 = X[1 + k], p3 = X[k];
                bool ok;
                if ((p1 & 0xC0) != 0x80 || q - 1 < bom) ok = (q - 1 >= bom) && p1 >= 0xC0;                 // any multi-byte lead covers +1
                else if ((p2 & 0xC0) != 0x80 || q - 2 < bom) ok = (q - 2 >= bom) && p2 >= 0xE0;            // 3- or 4-byte lead covers +2
                else if ((p3 & 0xC0) != 0x80 || q - 3 < bom) ok = (q - 3 >= bom) && p3 >= 0xF0;            // 4-byte lead covers +3
                else ok = false;
                e_any |= !ok;                     // invalid leads (F8..FF) are rejected at their own position
            } else if (in) {
                if (b0 < 0x80) { vv = ascii[b0]; has = true; }
                else {
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
                    e_any |= e;
                    if (!e) { vv = wv_element(p, cp); has = true; }
                }
            }
            v[k] = vv; if (has) { wm |= 1u << k; ++cnt; }
        }
        err |= e_any;
        const int inc = wv::incl_scan(cnt);
        uint32_t r = w0 + (uint32_t)(inc - cnt);
#pragma unroll
        for (int k = 0; k < 8; ++k) if (wm & (1u << k)) { S.ring[r & RMASK] = (uint16_t)v[k]; ++r; }
        const int total = wv::bcast(inc, 63);
        dec += total; dec_bytes = pos + WV_CHUNK; rhi = rbase + (uint32_t)dec;
        wv::sync();
    }

    // ------------------------------------------------------------------------------------------------------------------
    // phase A: one window of up to 64 start positions, the first of which (`done`) is a start position of the reference's
    // loop (FALexTools_t.h:229).  Queues the tokens whose extent is known and moves `done` behind them.  Returns false when
    // the token at `done` itself needs elements that are not decoded yet.  `fully`: the whole document is decoded.
    // ------------------------------------------------------------------------------------------------------------------
    BF_WVD bool phase_a(bool fully)
    {
        const int w0 = done;
        const int nv = dec - w0 < 64 ? dec - w0 : 64;
        const int maxtok = p.max_token_length;
        const bool valid = lane < nv;
        const int pos = w0 + lane;
        const uint32_t el = valid ? ring_at(rbase + (uint32_t)pos) : (WK_NOMATCH << WK_SHIFT);
        const uint32_t kind = el >> WK_SHIFT, cls = el & LX_T_CLS_MASK;
        const unsigned long long M = wv::ballot(valid && kind == WK_LOOP);
        const unsigned long long G = wv::ballot(valid && kind == WK_GENERAL);
        const bool at_end = fully && w0 + nv == dec;                    // the window reaches the end of the document
        if (STATS) ++st_win;
        // ---- per lane: the token a walk from this position finds (has / len / info), whether its extent is certain, next start
        bool has = false, complete = true; int len = 1; uint32_t info = 0;
        if (kind == WK_LOOP) {
            // the run of WK_LOOP elements from here (closed, final loop state: FALexTools_t.h:255-277 stays in it, fp follows)
            const unsigned long long x = ~(M >> lane);                  // bits above 63 - lane of M >> lane are 0: r <= 64 - lane
            const int r = x ? __builtin_ctzll(x) : 64;
            has = true; info = p.loop_info;
            len = r < maxtok ? r : maxtok;
            complete = r >= maxtok || lane + r < nv || at_end;          // a run that touches the end of the window may go on
        } else if (kind == WK_SOLO) {
            has = true; info = (uint32_t)(p.T[p.initial + cls] >> 32);
        }
        // a run that fills the whole window: look further (words of 64 characters and more)
        if (M == ~0ull && nv < maxtok) {
            int ext = nv; bool term = false;
            while (ext < maxtok && w0 + ext < dec) {
                const int nv2 = dec - (w0 + ext) < 64 ? dec - (w0 + ext) : 64;
                const uint32_t e2 = lane < nv2 ? ring_at(rbase + (uint32_t)(w0 + ext + lane)) : 0u;
                const unsigned long long m2 = wv::ballot(lane < nv2 && (e2 >> WK_SHIFT) == WK_LOOP);
                const int rr = m2 == ~0ull ? 64 : __builtin_ctzll(~m2);      // <= nv2: the lanes behind nv2 are clear
                ext += rr;
                if (rr < nv2) { term = true; break; }
            }
            if (lane == 0) {
                len = ext < maxtok ? ext : maxtok;
                complete = term || ext >= maxtok || (fully && w0 + ext >= dec);
            }
        }
        if (G) {
            if (kind == WK_GENERAL && valid) {
                // the automaton itself (FALexTools_t.h:255-277 without anchors: none apply, bf_model.cpp "unit form")
                uint32_t state = p.initial; int j = pos, fp = -1; uint32_t finfo = 0;
                const int b = pos + maxtok;
                for (;;) {
                    if (j >= b) break;
                    if (j >= dec) { complete = fully; break; }
                    const uint32_t c = ring_at(rbase + (uint32_t)j) & LX_T_CLS_MASK;
                    const uint64_t e64 = p.T[state + c];
                    const uint32_t e = (uint32_t)e64;
                    if ((e & LX_T_CLS_MASK) != c) break;
                    if ((int32_t)e < 0) { fp = j; finfo = (uint32_t)(e64 >> 32); }
                    state = (e >> LX_T_NEXT_SHIFT) & LX_T_NEXT_MASK;
                    ++j;
                }
                has = fp >= 0; len = has ? fp - pos + 1 : 1; info = finfo;
            }
        }
        const int nxt = pos + len;
        // ---- the chain of start positions inside the window
        const unsigned long long V = nv == 64 ? ~0ull : ((1ull << nv) - 1ull);
        unsigned long long St;
        const bool odd = (kind == WK_GENERAL && has && len > 1) || (kind == WK_LOOP && len == maxtok && lane + len < nv && ((M >> (lane + len)) & 1ull));
        if (wv::ballot(valid && odd) == 0) {
            St = (V & ~M) | (M & ~(M << 1));            // every non-run element, and the first element of every run
        } else {
            // a multi-element general token or a run cut by max-length: follow the chain (FALexTools_t.h:229, 390-393)
            if (STATS) ++st_slow;
            St = 0; int cur = w0;
            while (cur < w0 + nv) { St |= 1ull << (cur - w0); cur = wv::bcast(nxt, cur - w0); }
        }
        // ---- cut at the first start whose token is not certain yet
        const unsigned long long inc = wv::ballot(valid && !complete) & St;
        int new_done;
        if (inc) {
            const int fi = __builtin_ctzll(inc);
            if (fi == 0) return false;
            St &= (1ull << fi) - 1ull;
            new_done = w0 + fi;
        } else new_done = wv::bcast(nxt, 63 - __builtin_clzll(St));
        // ---- queue the tokens
        const unsigned long long TK = St & wv::ballot(has);
        if ((TK >> lane) & 1ull) {
            const int t = qn + __builtin_popcountll(TK & ((1ull << lane) - 1ull));
            S.q0[t] = ((rbase + (uint32_t)pos) & RMASK) | ((uint32_t)len << 16);
            S.q1[t] = info; S.q2[t] = (uint16_t)curk;
        }
        qn += __builtin_popcountll(TK);
        done = new_done;
        wv::sync();
        return true;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // phase B: the function frame of Process_int (FALexTools_t.h:229-393 at depth 1) on one word, and the post-pass on its
    // sub-tokens (tokdll:1239-1301): from = -1 takes the left anchor (resolved at load), a walk that matches emits a piece
    // and continues behind it, a walk that fails at from >= 0 leaves a gap, so the pieces cannot tile the word: UNK.
    // ------------------------------------------------------------------------------------------------------------------
    struct Unit {
        int tok;                         // queue index, -1: idle
        uint32_t rs; int L; uint32_t ini, ini_l;
        int from, j, lim; uint32_t state; int fp; uint32_t ftag;
        int cnt; int32_t id0, id1, id2, id3;
        int32_t *dst; int room;          // emit form (phase C, words of more than four pieces): pieces go straight to dst[0 .. room)
    };

    BF_WVD bool unit_setup(Unit &u) const
    {
        for (;;) {
            if (u.from >= u.L) return false;
            const int b = u.from + p.max_token_length;
            u.lim = b < u.L ? b : u.L;                               // no right-anchor step inside a function (fn_no_ra)
            if (u.from < 0) {
                if (u.ini_l == LX_NO_STATE || !(0 < u.lim)) { u.from = 0; continue; }
                u.state = u.ini_l; u.j = 0;
            } else { u.state = u.ini; u.j = u.from; }
            u.fp = -1;
            return true;
        }
    }
    BF_WVD void unit_finish(Unit &u)
    {
        if (!u.dst) {
            const int t = u.tok;
            S.rcnt[t] = u.cnt; S.rid[t] = u.id0; S.rid[QCAP + t] = u.id1; S.rid[2 * QCAP + t] = u.id2; S.rid[3 * QCAP + t] = u.id3;
        }
        u.tok = -1;
    }
    // starts the unit of queue entry t; false: finished at once (not a word / a word without a vocabulary call)
    BF_WVD bool unit_begin(Unit &u, int t)
    {
        u.tok = t; u.dst = nullptr; u.room = 0;
        const uint32_t w0 = S.q0[t], info = S.q1[t];
        u.rs = w0 & 0xFFFFu; u.L = (int)(w0 >> 16); u.cnt = 0; u.id0 = u.id1 = u.id2 = u.id3 = 0;
        int tag; bool call = false;
        if (info & LX_INFO_SIMPLE) tag = (int)(info & 0x7FFFFFFFu);
        else { const int32_t *a = p.acts + info; tag = a[2]; u.ini = (uint32_t)a[5]; u.ini_l = (uint32_t)a[6]; call = true; }
        if (tag != WBD_WORD_TAG) { unit_finish(u); return false; }              // tags 2..4: neither a word nor a sub-token
        if (!call) { u.cnt = 1; u.id0 = p.unk; unit_finish(u); return false; }  // a word without sub-tokens (tokdll:1282-1301)
        u.from = -1;
        if (!unit_setup(u)) { u.cnt = 1; u.id0 = p.unk; unit_finish(u); return false; }
        return true;
    }
    // one transition (FALexTools_t.h:255-277); when the walk ends, its result and the next walk's start
    BF_WVD void unit_step(Unit &u)
    {
        const uint32_t c = (uint32_t)S.ring[(u.rs + (uint32_t)u.j) & RMASK] & LX_T_CLS_MASK;
        const uint64_t e64 = p.T[u.state + c];
        const uint32_t e = (uint32_t)e64;
        const bool hit = (e & LX_T_CLS_MASK) == c;
        if (hit && (int32_t)e < 0) { u.fp = u.j; u.ftag = (uint32_t)(e64 >> 32); }
        if (hit) { u.state = (e >> LX_T_NEXT_SHIFT) & LX_T_NEXT_MASK; ++u.j; }
        if (hit && u.j < u.lim) return;
        // ---- the walk is over
        if (u.fp < 0) {
            if (u.from >= 0) { u.cnt = 1; u.id0 = p.unk; unit_finish(u); return; }      // a gap: the word is UNK whatever follows
            u.from = 0;                                                                  // the anchored walk found nothing (FALexTools_t.h:293)
        } else {
            const int32_t tag = (int32_t)(u.ftag & 0x7FFFFFFFu);
            if (u.dst) { if (u.cnt < u.room) u.dst[u.cnt] = tag; }
            else { if (u.cnt == 0) u.id0 = tag; else if (u.cnt == 1) u.id1 = tag; else if (u.cnt == 2) u.id2 = tag; else if (u.cnt == 3) u.id3 = tag; }
            ++u.cnt;
            u.from = u.fp + 1;
        }
        if (!unit_setup(u)) unit_finish(u);                                              // from == L: the pieces tile the word
    }
    BF_WVD void unit_refill(Unit &u, int &head)
    {
        const unsigned long long m = wv::ballot(u.tok < 0);
        if (m == 0 || head >= qn) return;
        const int t = head + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (u.tok < 0 && t < qn) unit_begin(u, t);
        const int k = __builtin_popcountll(m);
        head = head + k < qn ? head + k : qn;
    }

    BF_WVD void phase_b()
    {
        Unit a, b; a.tok = b.tok = -1; a.dst = b.dst = nullptr;
        int head = 0;
        for (;;) {
            unit_refill(a, head); unit_refill(b, head);
            if (!wv::any(a.tok >= 0 || b.tok >= 0)) { if (head >= qn) break; continue; }
            if (STATS) { ++st_trips; st_steps += (unsigned long long)__builtin_popcountll(wv::ballot(a.tok >= 0)) + (unsigned long long)__builtin_popcountll(wv::ballot(b.tok >= 0)); }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (a.tok >= 0) unit_step(a);
                if (b.tok >= 0) unit_step(b);
            }
        }
        wv::sync();
    }

    // ------------------------------------------------------------------------------------------------------------------
    // phase C: ids in queue order.  Position of a unit's ids = ids its document has so far + ids of the units before it in the
    // document (segmented prefix sum over the queue); nothing is written at or behind the document's cap (tokdll:1308-1310 is
    // then the prefix rule: what a full output array cuts off does not change what came before).
    // ------------------------------------------------------------------------------------------------------------------
    BF_WVD void phase_c()
    {
        for (int base = 0; base < qn; base += 64) {
            const int t = base + lane;
            const bool act = t < qn;
            const int cnt = act ? S.rcnt[t] : 0;
            const int k = act ? (int)S.q2[t] : -1;
            const int inc = wv::incl_scan(cnt), exc = inc - cnt;
            const int kp = wv::shfl_up(k, 1), kn = wv::shfl_down(k, 1);
            const unsigned long long hm = wv::ballot(lane == 0 || k != kp);
            const int head = 63 - __builtin_clzll(hm & ((2ull << lane) - 1ull));
            const int segbase = wv::shfl(exc, head);
            int pos = 0; int64_t slot = 0; int cap = 0;
            if (act) { pos = S.dt_cnt[k] + (exc - segbase); slot = S.dt_slot[k]; cap = S.dt_cap[k]; }
            wv::sync();                                                   // every lane has read its document's count
            if (act && (lane == 63 || k != kn)) S.dt_cnt[k] = pos + cnt;
            wv::sync();
            if (act && cnt > 0 && pos < cap) {
                int32_t *dst = p.ids_tmp + slot + pos;
                const int room = cap - pos;
                if (cnt <= 4) {
                    dst[0] = S.rid[t];
                    if (cnt > 1 && room > 1) dst[1] = S.rid[QCAP + t];
                    if (cnt > 2 && room > 2) dst[2] = S.rid[2 * QCAP + t];
                    if (cnt > 3 && room > 3) dst[3] = S.rid[3 * QCAP + t];
                } else {
                    // more than four pieces: walk the word again, this time straight into its place
                    if (STATS) ++st_rewalk;
                    Unit u;
                    unit_begin(u, t);                                     // a word with a call: never finishes at once
                    u.dst = dst; u.room = room;
                    while (u.tok >= 0) unit_step(u);
                }
            }
        }
    }

    // tokenises what is queued, writes the counts of the documents that are closed, frees the queue and the ring behind `done`
    BF_WVD void flush()
    {
        if (STATS) { ++st_flush; st_tok += (unsigned long long)qn; }
        if (qn > 0) { phase_b(); phase_c(); }
        wv::sync();
        for (int k = lane; k < dn; k += 64) {
            const uint32_t f = S.dt_flags[k];
            if (f & WV_DT_CLOSED) { const int c = S.dt_cnt[k], cap = S.dt_cap[k]; p.counts[S.dt_doc[k]] = (f & WV_DT_BAD) ? 0 : (c < cap ? c : cap); }
        }
        const bool open = dn > 0 && !(S.dt_flags[dn - 1] & WV_DT_CLOSED);
        wv::sync();
        if (open) {
            if (lane == 0 && dn > 1) { S.dt_slot[0] = S.dt_slot[dn - 1]; S.dt_doc[0] = S.dt_doc[dn - 1]; S.dt_cap[0] = S.dt_cap[dn - 1]; S.dt_cnt[0] = S.dt_cnt[dn - 1]; S.dt_flags[0] = 0; }
            dn = 1; curk = 0; rlo = rbase + (uint32_t)done;
        } else { dn = 0; rlo = rhi; }
        qn = 0;
        wv::sync();
    }

    // starts document d; false: nothing to tokenise (its count is written here)
    BF_WVD bool open_document(int64_t d)
    {
        const int64_t b = p.b.doc_off[d];
        const int64_t n64 = p.b.doc_off[d + 1] - b;
        if (n64 <= 0 || n64 > 1000000000) { if (lane == 0) p.counts[d] = 0; return false; }                                // tokdll:1121
        if (b < 0 || b + n64 > p.b.total_bytes) { if (lane == 0) { p.counts[d] = 0; wv::atomic_or(p.b.status, BF_STATUS_BAD_OFFSETS); } return false; }
        n = (int)n64; s = p.b.text + b;
        int cap = p.max_ids; if ((int64_t)cap > n64) cap = n; if (cap < 0) cap = 0;
        curk = dn++;
        if (lane == 0) { S.dt_slot[curk] = wv_ids_slot(b, d); S.dt_doc[curk] = d; S.dt_cap[curk] = cap; S.dt_cnt[curk] = 0; S.dt_flags[curk] = 0; }
        rhi = (rhi + 7u) & ~7u; rbase = rhi;
        dec_bytes = dec = done = bom = 0; err = false;
        wv::sync();
        return true;
    }
    BF_WVD void close_document()
    {
        const bool bad = wv::any(err);
        if (lane == 0) S.dt_flags[curk] = WV_DT_CLOSED | (bad ? WV_DT_BAD : 0u);
        wv::sync();
    }

    // One loop, every phase in it exactly once (the program is inlined into the kernel: what is called from two places is there
    // twice): each trip does the one thing that can be done next -- fetch a document, decode a chunk, resolve a window of start
    // positions, or flush (tokenise what is queued) when the queue, the document table or the ring is full.
    BF_WVD void run(int grab)
    {
        int64_t dnext = 0, dend = 0;
        bool have_doc = false, exiting = false;
        for (;;) {
            bool do_flush = false;
            if (!have_doc) {
                if (exiting) { if (qn == 0 && dn == 0) break; do_flush = true; }
                else if (dn == DTN) do_flush = true;
                else {
                    if (dnext >= dend) {
                        unsigned long long base = 0;
                        if (lane == 0) base = wv::atomic_add(p.next_doc, (unsigned long long)grab);
                        base = wv::bcast(base, 0);
                        dnext = (int64_t)base; dend = dnext + grab < p.b.ndocs ? dnext + grab : p.b.ndocs;
                        if (dnext >= p.b.ndocs) { exiting = true; continue; }
                    }
                    have_doc = open_document(dnext++);
                    continue;
                }
            } else {
                if (dec_bytes < n && ring_free() >= WV_CHUNK) { decode_chunk(); continue; }
                const bool fully = dec_bytes >= n;
                if (done < dec) {
                    if (qn + 64 > QCAP) do_flush = true;
                    else if (phase_a(fully)) continue;
                    else do_flush = true;                       // the token at `done` reaches past what is decoded and the ring has no room
                } else if (fully) { close_document(); have_doc = false; continue; }
                else do_flush = true;                           // everything decoded is resolved, the ring has no room for the next chunk
                if (do_flush && qn == 0 && rlo == rbase + (uint32_t)done) {
                    // nothing to flush and nothing to free: cannot happen (max token length <= RING - chunk - 16, checked at load)
                    if (lane == 0) wv::atomic_or(p.b.status, BF_STATUS_INTERNAL);
                    close_document(); have_doc = false; continue;
                }
            }
            if (do_flush) flush();
        }
        if (STATS && lane == 0) {
            wv::atomic_add(&p.stats[0], st_win); wv::atomic_add(&p.stats[1], st_slow); wv::atomic_add(&p.stats[2], st_flush); wv::atomic_add(&p.stats[3], st_tok);
            wv::atomic_add(&p.stats[4], st_trips); wv::atomic_add(&p.stats[5], st_steps);
        }
        if (STATS) { const unsigned long long r = st_rewalk; if (r) wv::atomic_add(&p.stats[6], r); }
    }
};

} // namespace bfa
