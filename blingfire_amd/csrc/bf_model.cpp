// bf_model.cpp -- see bf_model.h.  Host only (no HIP): parse the .bin, decode the automata,
// build the displacement-packed tables and code-point maps the kernels consume.
#include "bf_model.h"
#include "bf_layout.h"
#include "bf_tolower.h"
#include "bf_flat_key.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <unordered_map>

namespace bfa {

namespace {

struct Reader {
    const uint8_t *p; size_t n;
    bool ok(size_t off, size_t len) const { return off <= n && len <= n - off; }
    int32_t i32(size_t off) const { int32_t v = 0; if (ok(off, 4)) memcpy(&v, p + off, 4); return v; }
    uint32_t u32(size_t off) const { uint32_t v = 0; if (ok(off, 4)) memcpy(&v, p + off, 4); return v; }
    uint32_t le(size_t off, int size) const {  // little-endian unsigned of 1/2/4 bytes
        uint32_t v = 0; if (ok(off, (size_t)size)) memcpy(&v, p + off, (size_t)size); return v;
    }
    int32_t sle(size_t off, int size) const {  // signed little-endian of 1/2/4 bytes
        if (!ok(off, (size_t)size)) return 0;
        if (size == 1) return (int8_t)p[off];
        if (size == 2) { int16_t v; memcpy(&v, p + off, 2); return v; }
        int32_t v; memcpy(&v, p + off, 4); return v;
    }
    uint32_t be(size_t off, int size) const {  // big-endian unsigned of 1..4 bytes
        uint32_t v = 0; if (!ok(off, (size_t)size)) return 0;
        for (int i = 0; i < size; ++i) v = (v << 8) | p[off + i];
        return v;
    }
};

// Length-prefixed int arrays (format: reference FAChains_pack_triv.h:81-163): header {SizeOfValue, MaxCount}
struct Chains {
    Reader r; int sov = 4;
    void set(const Reader &rr) { r = rr; sov = r.i32(0); }
    bool valid() const { return sov == 1 || sov == 2 || sov == 4; }
    int count(size_t off) const { return r.sle(off, sov); }
    int value(size_t off, int idx) const { return r.sle(off + (size_t)sov * (1 + (size_t)idx), sov); }
};

// key -> int[] map (format: reference FAMultiMap_pack.cpp:22-53)
struct TrivMap {
    Reader r; uint32_t max_key = 0; int soo = 0; size_t offs = 8; Chains vals; bool set_ = false;
    bool set(const Reader &rr) {
        r = rr; max_key = r.u32(0); soo = (int)r.u32(4);
        if (soo < 1 || soo > 4) return false;
        size_t off = 8 + (size_t)soo * (1 + (size_t)max_key);
        if (off % 4) off += 4 - off % 4;
        if (!r.ok(off, 8)) return false;
        vals.set(Reader{r.p + off, r.n - off});
        set_ = vals.valid();
        return set_;
    }
    bool get(int key, std::vector<int> &out) const {
        out.clear();
        if (!set_ || key < 0 || (uint32_t)key > max_key) return false;
        uint32_t vo = r.be(offs + (size_t)soo * (size_t)key, soo);
        if (vo == 0) return false;
        int c = vals.count(vo - 1);
        if (c < 0 || c > 4096) return false;
        for (int i = 0; i < c; ++i) out.push_back(vals.value(vo - 1, i));
        return true;
    }
};

// fixed-stride key -> int[] map (format: reference FAMultiMap_pack_fixed.cpp:25-58)
struct FixedMap {
    Reader r; int sov = 0, max_count = 0, min_key = 0, max_key = -1; uint32_t stride = 0; bool set_ = false;
    bool set(const Reader &rr) {
        r = rr; sov = (int)r.u32(0); max_count = r.i32(4); min_key = r.i32(8); max_key = r.i32(12);
        if (!(sov == 1 || sov == 2 || sov == 4) || max_count <= 0 || min_key < 0 || max_key < min_key) return false;
        stride = (uint32_t)(max_count + 1) * (uint32_t)sov;
        if (!r.ok(16, (size_t)stride * (size_t)(max_key - min_key + 1))) return false;
        set_ = true; return true;
    }
    // returns the reference Get(Key,pValues,MaxCount=cap) count: -1 = no entry; values filled only if 0 <= count <= cap
    int get(int key, int *out, int cap) const {
        if (!set_ || key < min_key || key > max_key) return -1;
        size_t off = 16 + (size_t)((uint32_t)stride * (uint32_t)(key - min_key));
        int c = r.sle(off, sov);
        if (c > max_count) return -1;
        if (c >= 0 && c <= cap) for (int i = 0; i < c; ++i) out[i] = r.sle(off + (size_t)sov * (1 + (size_t)i), sov);
        return c;
    }
};

uint32_t crc32_update(const uint8_t *p, size_t n, uint32_t crc)
{
    static uint32_t table[256]; static bool init = false;
    if (!init) {
        for (uint32_t k = 0; k < 256; ++k) { uint32_t c = k; for (int j = 0; j < 8; ++j) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[k] = c; }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return ~crc;
}

bool is_boolean_param(int p)
{   // reference FALDB.cpp:119-132
    return p == 10 || p == PARAM_NO_TR || p == PARAM_IGNORE_CASE || p == 31 || p == 35 || p == 37 || p == 40 || p == 46 ||
           p == PARAM_VERIFY_LDB_BIN;
}

// ---- automaton decoding (format spec: reference blingfirecompile.library/inc/FADfaPack_triv.h:27-96)
struct DfaHeader { int dst_size = 3; int ows_offset = 0; bool remap = false; size_t initial = 0; };

bool decode_dfa(const Reader &r, bool mealy, RawDfa &out, std::string &err)
{
    DfaHeader h;
    size_t off = 0;
    h.dst_size = r.i32(0); if (h.dst_size < 1 || h.dst_size > 4) h.dst_size = 3;
    h.ows_offset = r.i32(4);
    uint32_t iwc = r.u32(8); off = 12;
    h.remap = (iwc & 0x80000000u) != 0;
    iwc &= 0x7fffffffu;
    if (iwc == 0 || (iwc & 1) || !r.ok(off, 4 * (size_t)iwc)) { err = "bad DFA alphabet header"; return false; }
    off += 4 * (size_t)iwc;
    out.remap = h.remap;
    if (h.remap) {
        if (mealy) { err = "Mealy DFA with remapped Iws is not a legal model"; return false; }
        int sz = r.i32(off); off += 4;
        // Iw map image (format: reference FAIwMap_pack.cpp:44-61)
        Reader m{r.p + off, r.n - off};
        int son = m.i32(0), ic = m.i32(4);
        if (son < 1 || son > 4 || ic < 0 || !m.ok(8, 12 * (size_t)ic)) { err = "bad Iw map"; return false; }
        size_t from_off = 8, to_off = 8 + 4 * (size_t)ic, data_off = 8 + 12 * (size_t)ic;
        for (int i = 0; i < ic; ++i) {
            int from = m.i32(from_off + 4 * (size_t)i), to = m.i32(to_off + 8 * (size_t)i), ioff = m.i32(to_off + 8 * (size_t)i + 4);
            if (to < from || ioff < 0 || (long)to - from > 0x200000) { err = "bad Iw map interval"; return false; }
            out.iw_from.push_back(from); out.iw_to.push_back(to);
            std::vector<int> cls((size_t)(to - from + 1));
            for (int k = 0; k <= to - from; ++k) {
                uint32_t v = m.be(data_off + (size_t)ioff + (size_t)son * (size_t)k, son);
                cls[(size_t)k] = v ? (int)v - 1 : -1;
            }
            out.iw_cls.push_back(std::move(cls));
        }
        off += (size_t)sz;
    }
    h.initial = off;

    Chains ows;
    if (mealy) {
        if (h.ows_offset == 0 || !r.ok((size_t)h.ows_offset, 8)) { err = "Mealy DFA without Ows section"; return false; }
        ows.set(Reader{r.p + h.ows_offset, r.n - (size_t)h.ows_offset});
        if (!ows.valid()) { err = "bad Ows section"; return false; }
    }

    // BFS over states identified by byte offset
    std::unordered_map<int, int> idx_of;
    std::vector<int> queue;
    auto intern = [&](int soff) -> int {
        auto it = idx_of.find(soff);
        if (it != idx_of.end()) return it->second;
        int id = (int)out.state_off.size();
        idx_of.emplace(soff, id); out.state_off.push_back(soff); queue.push_back(soff);
        return id;
    };
    intern((int)h.initial);
    out.initial = 0;
    struct Tr { int sym, dst, ow; };
    std::vector<std::vector<Tr>> trs;
    const int ds = h.dst_size;
    const uint32_t dead_mark = ds == 4 ? 0xffffffffu : ((1u << (8 * ds)) - 1);
    size_t total_tr = 0;
    for (size_t qi = 0; qi < queue.size(); ++qi) {
        const int s = queue[qi];
        if (s <= 0 || !r.ok((size_t)s, 1)) { err = "state offset out of range"; return false; }
        const uint32_t info = r.p[s];
        const int iw_size = (int)((info & 0x18) >> 3) + 1; // 1,2,3(unused),4
        const int tr = (int)(info & 7), ow_code = (int)((info & 0x60) >> 5);
        const int ow_size = ow_code == 3 ? 4 : ow_code;
        if (iw_size == 3) { err = "unsupported IwSize 3"; return false; }
        std::vector<Tr> list;
        size_t p = (size_t)s + 1, ow_pos = 0;
        auto dst_of = [&](size_t base, size_t idx) -> int {
            uint32_t v = r.be(base + (size_t)ds * idx, ds);
            return v == dead_mark ? DFA_DEAD_STATE : (int)v;
        };
        if (tr == TRS_PARA) {
            size_t cnt = 1 + (size_t)r.le(p, iw_size); p += (size_t)iw_size;
            size_t iws = p, dsts = p + cnt * (size_t)iw_size;
            if (!r.ok(dsts, cnt * (size_t)ds)) { err = "truncated PARA state"; return false; }
            for (size_t i = 0; i < cnt; ++i) list.push_back({(int)r.le(iws + i * (size_t)iw_size, iw_size), dst_of(dsts, i), (int)i});
            ow_pos = dsts + cnt * (size_t)ds;
        } else if (tr == TRS_IWIA) {
            if (mealy) { err = "IWIA state in a Mealy DFA"; return false; }
            uint32_t base = r.le(p, iw_size); p += (size_t)iw_size;
            uint32_t mx = r.le(p, iw_size); p += (size_t)iw_size;
            if (mx < base || mx - base > 0x200000 || !r.ok(p, (size_t)(mx - base + 1) * (size_t)ds)) { err = "bad IWIA state"; return false; }
            for (uint32_t k = 0; k <= mx - base; ++k) {
                int d = dst_of(p, k);
                if (d != 0) list.push_back({(int)(base + k), d, 0});
            }
            ow_pos = p + (size_t)(mx - base + 1) * (size_t)ds;
        } else if (tr == TRS_RANGE) {
            if (mealy) { err = "RANGE state in a Mealy DFA"; return false; }
            size_t cnt = 1 + (size_t)r.le(p, iw_size); p += (size_t)iw_size;
            size_t froms = p, tos = p + cnt * (size_t)iw_size, dsts = tos + cnt * (size_t)iw_size;
            if (!r.ok(dsts, cnt * (size_t)ds)) { err = "truncated RANGE state"; return false; }
            for (size_t i = 0; i < cnt; ++i) {
                uint32_t f = r.le(froms + i * (size_t)iw_size, iw_size), t = r.le(tos + i * (size_t)iw_size, iw_size);
                // the reference picks the LAST range whose From <= Iw and then requires Iw <= To
                uint32_t next_from = i + 1 < cnt ? r.le(froms + (i + 1) * (size_t)iw_size, iw_size) : 0xffffffffu;
                if (t >= next_from) t = next_from - 1;
                if (t < f) continue;
                if ((uint64_t)t - f > 0x20000 || total_tr + list.size() > 8000000) { err = "RANGE state too wide to expand (unsupported packing)"; return false; }
                int d = dst_of(dsts, i);
                for (uint32_t k = f; k <= t; ++k) list.push_back({(int)k, d, 0});
            }
            ow_pos = dsts + cnt * (size_t)ds;
        } else if (tr == TRS_IMPL) {
            int iw = (int)r.le(p, iw_size);
            list.push_back({iw, s + 1 + iw_size + ow_size, 0});
            ow_pos = p + (size_t)iw_size;
        } else if (tr == TRS_NONE) {
            ow_pos = p;
        } else { err = "unknown transition type"; return false; }

        int state_ow = -1;
        if (ow_code != 0) state_ow = r.sle(ow_pos, ow_size);
        if (mealy) {
            for (auto &t : list) {
                if (ow_code == 0) { err = "Mealy transition without output weight"; return false; }
                if (state_ow < 0 || !ows.r.ok((size_t)state_ow, (size_t)ows.sov)) { err = "bad Ows offset"; return false; }
                int c = ows.count((size_t)state_ow);
                if (t.ow >= c) { err = "Ows record shorter than the transition list"; return false; }
                t.ow = ows.value((size_t)state_ow, t.ow);
                if (t.ow < 0) { err = "negative Mealy output weight"; return false; }
            }
        }
        for (auto &t : list) if (t.dst != DFA_DEAD_STATE) t.dst = intern(t.dst);
        std::sort(list.begin(), list.end(), [](const Tr &a, const Tr &b) { return a.sym < b.sym; });
        for (size_t i = 1; i < list.size(); ++i) if (list[i].sym == list[i - 1].sym) { err = "duplicate symbol in a state"; return false; }
        total_tr += list.size();
        trs.push_back(std::move(list));
        out.is_final.push_back((info & 0x80) ? 1 : 0);
        out.ow.push_back(mealy ? -1 : state_ow);
        if (trs.size() > 4000000) { err = "automaton too large"; return false; }
    }
    out.tr_begin.assign(1, 0);
    for (auto &l : trs) {
        for (auto &t : l) { out.tr_sym.push_back(t.sym); out.tr_dst.push_back(t.dst); if (mealy) out.tr_ow.push_back(t.ow); }
        out.tr_begin.push_back((uint32_t)out.tr_sym.size());
    }
    out.off_index.clear();
    for (size_t i = 0; i < out.state_off.size(); ++i) out.off_index.push_back({out.state_off[i], (int)i});
    std::sort(out.off_index.begin(), out.off_index.end());
    return true;
}

// ---- displacement packing: first-fit, unique base per state
bool pack_dfa(const RawDfa &raw, bool wide, const std::vector<int> &extra_syms, PackedDfa &out, std::string &err)
{
    const size_t ns = raw.state_off.size();
    // class space
    std::vector<int> cls_of_tr(raw.tr_sym.size());
    if (raw.remap) {
        int mx = -1;
        for (auto &v : raw.iw_cls) for (int c : v) mx = std::max(mx, c);
        for (size_t i = 0; i < raw.tr_sym.size(); ++i) { cls_of_tr[i] = raw.tr_sym[i]; mx = std::max(mx, raw.tr_sym[i]); }
        out.nclasses = mx + 1;
    } else {
        std::vector<int> syms(raw.tr_sym);
        syms.insert(syms.end(), extra_syms.begin(), extra_syms.end());
        std::sort(syms.begin(), syms.end());
        syms.erase(std::unique(syms.begin(), syms.end()), syms.end());
        out.sym_of_class = syms;
        out.nclasses = (int)syms.size();
        for (size_t i = 0; i < raw.tr_sym.size(); ++i)
            cls_of_tr[i] = (int)(std::lower_bound(syms.begin(), syms.end(), raw.tr_sym[i]) - syms.begin());
    }
    for (int c : cls_of_tr) if (c < 0) { err = "negative symbol in automaton"; return false; }
    const uint32_t cls_limit = wide ? (uint32_t)T64_CLS_MASK : T32_CLS_MASK;
    if ((uint32_t)out.nclasses >= cls_limit) { err = "too many symbol classes for the table entry format"; return false; }
    out.wide = wide;

    const size_t ntr = raw.tr_sym.size();
    size_t cap = ntr + ntr / 4 + ns + (size_t)out.nclasses + 4096;
    std::vector<int32_t> slot_state(cap, -1);   // owner state index per slot
    std::vector<uint8_t> base_used(cap, 0);
    std::vector<uint32_t> base(ns + 1, 0xffffffffu);
    auto grow = [&](size_t need) {
        if (need <= cap) return;
        size_t nc = std::max(need + need / 8 + 1024, cap * 2);
        slot_state.resize(nc, -1); base_used.resize(nc, 0); cap = nc;
    };
    std::vector<uint32_t> order(ns);
    for (size_t i = 0; i < ns; ++i) order[i] = (uint32_t)i;
    auto fan = [&](uint32_t s) { return raw.tr_begin[s + 1] - raw.tr_begin[s]; };
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return fan(a) > fan(b); });
    // next_free[i]: smallest free slot >= i (union-find with path halving), so the first-fit search
    // only ever visits free slots
    std::vector<uint32_t> next_free(cap + 1);
    for (size_t i = 0; i <= cap; ++i) next_free[i] = (uint32_t)i;
    auto grow_nf = [&]() {
        const size_t old = next_free.size();
        if (old >= cap + 1) return;
        next_free.resize(cap + 1);
        for (size_t i = old; i <= cap; ++i) next_free[i] = (uint32_t)i;
    };
    auto find_free = [&](size_t i) -> size_t {
        while (next_free[i] != i) { next_free[i] = next_free[next_free[i]]; i = next_free[i]; }
        return i;
    };
    size_t max_slot = 0;
    for (uint32_t s : order) {
        const uint32_t b0 = raw.tr_begin[s], b1 = raw.tr_begin[s + 1];
        if (b0 == b1) break;  // leaves are last
        const size_t c0 = (size_t)cls_of_tr[b0], clast = (size_t)cls_of_tr[b1 - 1];
        size_t f = find_free(c0);          // candidate slot for the first symbol (base >= 0)
        size_t b;
        for (;;) {
            if (f + (clast - c0) + 2 >= cap) { grow(f + (clast - c0) + 2); grow_nf(); }
            b = f - c0;
            bool ok = !base_used[b];
            for (uint32_t t = b0 + 1; ok && t < b1; ++t) if (slot_state[b + (size_t)cls_of_tr[t]] != -1) ok = false;
            if (ok) break;
            f = find_free(f + 1);
        }
        base[s] = (uint32_t)b; base_used[b] = 1;
        for (uint32_t t = b0; t < b1; ++t) {
            const size_t sl = b + (size_t)cls_of_tr[t];
            slot_state[sl] = (int32_t)s; next_free[sl] = (uint32_t)(sl + 1);
        }
        max_slot = std::max(max_slot, b + clast);
    }
    // leaves + the synthetic dead state take unused base values (they own no slot)
    size_t bcur = 0;
    auto next_free_base = [&]() -> uint32_t {
        for (;; ++bcur) { grow(bcur + 1); if (!base_used[bcur]) { base_used[bcur] = 1; return (uint32_t)bcur++; } }
    };
    if (wide) {
        // a dictionary has a handful of leaves (a minimised Mealy automaton merges equivalent ones): they take CONSECUTIVE unused base values, so
        // that a walk knows by its state alone that it is over (base - leaf_lo < leaf_n: bf_seg.h, one gather less per walk that ends on a leaf)
        size_t nleaf = 0;
        for (uint32_t s : order) if (fan(s) == 0) ++nleaf;
        size_t b = 0;
        for (;; ++b) {
            grow(b + nleaf + 2);
            bool ok = true;
            for (size_t k = 0; ok && k < nleaf; ++k) if (base_used[b + k]) ok = false;
            if (ok) break;
        }
        out.leaf_lo = (uint32_t)b; out.leaf_n = (uint32_t)nleaf;
        for (uint32_t s : order) if (fan(s) == 0) { base[s] = (uint32_t)b; base_used[b] = 1; ++b; }
    } else
        for (uint32_t s : order) if (fan(s) == 0) base[s] = next_free_base();
    base[ns] = next_free_base();
    uint32_t max_base = 0;
    for (uint32_t v : base) max_base = std::max(max_base, v);
    const size_t table_len = std::max(max_slot + 1, (size_t)max_base + 1) + (wide ? (size_t)out.nclasses + 1 : (size_t)T32_CLS_MASK + 1); // any base + any class (or the lexer's "none" probe at +0x1FFF) stays in range
    const uint64_t next_limit = wide ? (T64_NEXT_MASK + 1) : (1ull << (32 - T32_NEXT_SHIFT));
    if (table_len >= next_limit) { err = "automaton too large for the table entry format"; return false; }

    out.state_base.assign(base.begin(), base.begin() + (long)ns);
    out.dead_base = base[ns];
    out.initial_base = base[(size_t)raw.initial];
    if (wide) out.t64.assign(table_len, T64_CLS_MASK); else out.t32.assign(table_len, T32_CLS_MASK);
    for (size_t s = 0; s < ns; ++s) {
        for (uint32_t t = raw.tr_begin[s]; t < raw.tr_begin[s + 1]; ++t) {
            const int d = raw.tr_dst[t];
            const uint32_t db = d == DFA_DEAD_STATE ? out.dead_base : base[(size_t)d];
            const bool fin = d != DFA_DEAD_STATE && raw.is_final[(size_t)d];
            const size_t slot = (size_t)base[s] + (size_t)cls_of_tr[t];
            if (wide) {
                uint64_t ow = raw.tr_ow.empty() ? 0 : (uint64_t)raw.tr_ow[t];
                if (ow >= (1ull << (64 - T64_OW_SHIFT))) { err = "output weight too large for the table entry format"; return false; }
                out.t64[slot] = (uint64_t)cls_of_tr[t] | (fin ? T64_FINAL_BIT : 0) | ((uint64_t)db << T64_NEXT_SHIFT) | (ow << T64_OW_SHIFT);
            } else {
                out.t32[slot] = (uint32_t)cls_of_tr[t] | (fin ? T32_FINAL_BIT : 0) | (db << T32_NEXT_SHIFT);
            }
        }
    }
    return true;
}

} // namespace

int RawDfa::class_of(int iw) const
{
    if (!remap) return iw;
    // last interval with from <= iw (reference FAIwMap_pack.h:72-110)
    auto it = std::upper_bound(iw_from.begin(), iw_from.end(), iw);
    if (it == iw_from.begin()) return -1;
    size_t i = (size_t)(it - iw_from.begin()) - 1;
    if (iw > iw_to[i]) return -1;
    return iw_cls[i][(size_t)(iw - iw_from[i])];
}

int RawDfa::find_state(int off) const
{
    auto it = std::lower_bound(off_index.begin(), off_index.end(), std::make_pair(off, -1));
    return (it != off_index.end() && it->first == off) ? it->second : -1;
}

void TwoLevelMap::init(uint32_t default_value)
{
    def = default_value;
    l1.assign(0x1100, 0);
    pages.assign(256, default_value);
}

void TwoLevelMap::set(int cp, uint32_t v)
{
    if (cp < 0 || cp > 0x10FFFF) return;
    uint16_t &pg = l1[(size_t)(cp >> 8)];
    if (pg == 0) {
        if (v == def) return;
        pg = (uint16_t)(pages.size() / 256);
        pages.resize(pages.size() + 256, def);
    }
    pages[(size_t)pg * 256 + (size_t)(cp & 255)] = v;
}

long PackedDfa::step(uint32_t base, uint32_t cls, int *final_out, int *ow_out) const
{
    if (wide) {
        if (cls >= T64_CLS_MASK) return -1;
        uint64_t e = t64[(size_t)base + cls];
        if ((e & T64_CLS_MASK) != cls) return -1;
        if (final_out) *final_out = (e & T64_FINAL_BIT) != 0;
        if (ow_out) *ow_out = (int)(e >> T64_OW_SHIFT);
        return (long)((e >> T64_NEXT_SHIFT) & T64_NEXT_MASK);
    }
    if (cls >= T32_CLS_MASK) return -1;
    uint32_t e = t32[(size_t)base + cls];
    if ((e & T32_CLS_MASK) != cls) return -1;
    if (final_out) *final_out = (e & T32_FINAL_BIT) != 0;
    if (ow_out) *ow_out = 0;
    return (long)(e >> T32_NEXT_SHIFT);
}

bool load_file(const char *path, std::vector<uint8_t> &out)
{
    FILE *f = path ? fopen(path, "rb") : nullptr;
    if (!f) return false;
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    if (sz <= 0) { fclose(f); return false; }
    out.resize((size_t)sz);
    bool ok = fread(out.data(), 1, (size_t)sz, f) == (size_t)sz;
    fclose(f);
    return ok;
}

static bool fail(Model &m, const std::string &e) { m.error = e; return false; }

// Flat form (bf_flat.h, bf_model.h Model::flat_tab): the one-piece words of the vocabulary function as a hash table keyed by the word.
// What a hit must mean, restated from bf_wave_body.h unit_call / unit_step / unit_event (themselves FALexTools_t.h:229-393 at depth 1 and
// tokdll:1239-1301): the FIRST walk of the frame -- from the state behind the left anchor when the function has one, else from its plain
// initial state -- reads the word's L characters (L <= max-token-length - 1: the walk's limit is the word's end), never misses, and its
// last transition enters a final state: fp = L - 1, one sub-token that tiles the word, its tag is the id.  The walk is the device's own
// (the entries of wbd_t2), so table and kernel cannot disagree about a transition.
struct FlatWord { uint64_t k0; uint32_t k1, id; };
// two-choice (cuckoo) placement at a load of at most 40 %; new multipliers when an insertion does not settle.  tab: entry e = [2e] k0, [2e + 1] k1 | id << 32
static bool place_words(const std::vector<FlatWord> &words, std::vector<uint64_t> &out, int &out_bits, uint32_t &out_m0, uint32_t &out_m1, uint32_t &out_m2)
{
    int bits = 10; while ((size_t)1 << bits < words.size() * 5 / 2 + 16) ++bits;
    uint64_t seed = 0x9E3779B97F4A7C15ull;
    auto next_odd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return (uint32_t)(seed >> 16) | 1u; };
    for (int attempt = 0; attempt < 64; ++attempt) {
        if (attempt == 32) ++bits;
        const uint32_t m0 = next_odd(), m1 = next_odd(), m2 = next_odd();
        std::vector<uint64_t> tab((size_t)2 << bits, 0);
        bool ok = true;
        for (size_t w = 0; w < words.size() && ok; ++w) {
            uint64_t k0 = words[w].k0, ki = (uint64_t)words[w].k1 | ((uint64_t)words[w].id << 32);
            uint32_t x = wf_mix(k0, (uint32_t)ki, m0), at = wf_h(x, m1, bits);
            ok = false;
            for (int kick = 0; kick < 512; ++kick) {
                uint64_t &t0 = tab[2 * (size_t)at], &t1 = tab[2 * (size_t)at + 1];
                if (t0 == 0) { t0 = k0; t1 = ki; ok = true; break; }
                if (t0 == k0 && (uint32_t)t1 == (uint32_t)ki) { ok = t1 == ki; break; }              // the same word twice
                std::swap(k0, t0); std::swap(ki, t1);
                x = wf_mix(k0, (uint32_t)ki, m0);
                const uint32_t h1 = wf_h(x, m1, bits), h2 = wf_h(x, m2, bits);
                at = at == h1 ? h2 : h1;
            }
        }
        if (!ok) continue;
        out.swap(tab); out_bits = bits; out_m0 = m0; out_m1 = m1; out_m2 = m2;
        return true;
    }
    return false;
}

static void build_flat_table(Model &m)
{
    m.flat_ok = false; m.flat_tab.clear(); m.flat_words = 0;
    if (!m.wave_ok || (m.loop_info & INFO_SIMPLE_BIT) || (m.wave_solo_info & INFO_SIMPLE_BIT)) return;
    if ((size_t)m.loop_info + 7 > m.acts_pool.size() || (size_t)m.wave_solo_info + 7 > m.acts_pool.size()) return;
    const int32_t *a = m.acts_pool.data() + m.loop_info, *c = m.acts_pool.data() + m.wave_solo_info;
    if (a[2] != 1 /* WBD_WORD_TAG */ || c[2] != 1 || a[5] != c[5] || a[6] != c[6]) return;      // bf_wave_body.h WpWave::fast_ok
    if (m.max_token_length < WF_RUN_MAX + 2) return;                  // every run the program resolves lies inside one walk limit
    m.flat_ini = (uint32_t)a[5]; m.flat_ini_l = (uint32_t)a[6];
    const bool anchored = m.flat_ini_l != 0xFFFFFFFFu && m.max_token_length > 1;
    const uint32_t start = anchored ? m.flat_ini_l : m.flat_ini;
    const std::vector<uint64_t> &T = m.wbd_t2;
    std::vector<FlatWord> words;
    auto step = [&](uint32_t state, uint32_t cls, uint32_t &next, bool &fin, uint32_t &tag) -> bool {
        const size_t at = (size_t)state + cls;
        if (at >= T.size()) return false;
        const uint64_t e64 = T[at]; const uint32_t e = (uint32_t)e64;
        if ((e & LX_T_CLS_MASK) != cls) return false;
        fin = (int32_t)e < 0; tag = (uint32_t)(e64 >> 32); next = (e >> LX_T_NEXT_SHIFT) & LX_T_NEXT_MASK;
        return true;
    };
    // runs: every word over the classes that have a code (a WK_LOOP class below 127), depth first
    std::vector<uint32_t> codable;
    for (int k = 0; k < m.wbd.nclasses && k < 127; ++k) if (m.wave_kind[(size_t)k] == 1 /* WK_LOOP */) codable.push_back((uint32_t)k);
    struct Fr { uint32_t state; uint64_t k0; uint32_t k1; int depth; };
    std::vector<Fr> stack; stack.push_back({start, 0, 0, 0});
    while (!stack.empty()) {
        const Fr f = stack.back(); stack.pop_back();
        for (uint32_t k : codable) {
            uint32_t nx = 0, tag = 0; bool fin = false;
            if (!step(f.state, k, nx, fin, tag)) continue;
            const uint64_t k0 = f.depth < 8 ? f.k0 | ((uint64_t)wf_code(k, 1u) << (8 * f.depth)) : f.k0;
            const uint32_t k1 = f.depth < 8 ? 0u : f.k1 | (wf_code(k, 1u) << (8 * (f.depth - 8)));
            if (fin) {
                if (!(tag & INFO_SIMPLE_BIT)) return;                 // cannot be (unit form: vocabulary tags are SIMPLE); no table then
                if ((tag & 0x7FFFFFFFu) > WF_ROW_ID_MASK) return;        // (an id of more than 24 bits: no table)
                words.push_back({k0, k1, (tag & 0x7FFFFFFFu) | ((uint32_t)(f.depth + 1) << WF_ROW_LEN_SHIFT)});
            }
            if (f.depth + 1 < WF_KEY_CHARS) stack.push_back({nx, k0, k1, f.depth + 1});
            if (words.size() > (1u << 22)) return;
        }
    }
    // one-element tokens: by class
    for (int k = 0; k < m.wbd.nclasses; ++k) {
        if (m.wave_kind[(size_t)k] != 3 /* WK_SOLO */) continue;
        uint32_t nx = 0, tag = 0; bool fin = false;
        if (step(start, (uint32_t)k, nx, fin, tag) && fin && (tag & INFO_SIMPLE_BIT)) {
            if ((tag & 0x7FFFFFFFu) > WF_ROW_ID_MASK) return;
            words.push_back({WF_KEY_SOLO | ((uint64_t)k << WF_KEY_SOLO_SHIFT), 0u, tag & 0x7FFFFFFFu});        // (characters: 0)
        }
    }
    if (place_words(words, m.flat_tab, m.flat_bits, m.flat_m0, m.flat_m1, m.flat_m2)) { m.flat_words = (int)words.size(); m.flat_ok = true; }
}

// ---- the word table of the BPE wave program (bf_bpe_wave_body.h, round 6).  A word of the text -- U+2581 and the symbols up to the next U+2581 --
// that IS a dictionary entry, with a shorter entry in front of it (the U+2581 alone, usually), is taken whole by the collection loop of both
// BPE flavours with m_fFastBpe: the whole-token arc replaces the arcs collected from that start and the walk jumps behind the word
// (..._1best_bpe_t.h:176,189-206,228-230: `token_start && next is U+2581 or the end && count_at_start < narcs`).  Whether that happens is a
// function of the word alone -- no rank is compared --, so every such word of <= 12 symbols behind the U+2581 is found here, by a walk of the
// device's own transition table from the state behind U+2581, and keyed by its symbols: one byte per symbol, its class (classes below 256; a
// word with another symbol has no key and takes the walk; the row's length field tells a class 0 from the padding), the U+2581 itself is not part of the key.  Rows and hash as the flat program's
// (bf_flat_key.h): {12 code bytes | id + IdOffset | symbols << 24}, two candidate rows per key.  gpt2.bin: 31 k words in 2^17 rows (2 MB).
static void build_bpe_word_table(Model &m)
{
    m.bpe_tab.clear(); m.bpe_tab_words = 0; m.bpe_tab_bits = 0;
    if (!m.bpe_wave_ok || m.sp_delim_code >= 0xFFFEu) return;
    const std::vector<uint64_t> &T = m.dict.t64;
    auto step = [&](uint32_t state, uint32_t cls, uint32_t &next, bool &fin, uint32_t &ow) -> bool {
        const size_t at = (size_t)state + cls;
        if (at >= T.size()) return false;
        const uint64_t e = T[at];
        if ((e & T64_CLS_MASK) != cls) return false;
        fin = (e & T64_FINAL_BIT) != 0; next = (uint32_t)((e >> T64_NEXT_SHIFT) & T64_NEXT_MASK); ow = (uint32_t)(e >> T64_OW_SHIFT);
        return true;
    };
    uint32_t s1 = 0, ow0 = 0; bool fin0 = false;
    if (!step(m.dict.initial_base, m.sp_delim_code, s1, fin0, ow0)) return;
    std::vector<FlatWord> words;
    const uint32_t ncode = (uint32_t)std::min(m.dict.nclasses, 256);
    struct Fr { uint32_t state, sum; uint64_t k0; uint32_t k1; int depth; bool seen; };
    std::vector<Fr> stack; stack.push_back({s1, ow0, 0, 0, 0, fin0});
    while (!stack.empty()) {
        const Fr f = stack.back(); stack.pop_back();
        for (uint32_t k = 0; k < ncode; ++k) {
            if (k == m.sp_delim_code) continue;
            uint32_t nx = 0, ow = 0; bool fin = false;
            if (!step(f.state, k, nx, fin, ow)) continue;
            const uint32_t sum = f.sum + ow;
            const uint64_t k0 = f.depth < 8 ? f.k0 | ((uint64_t)k << (8 * f.depth)) : f.k0;
            const uint32_t k1 = f.depth < 8 ? 0u : f.k1 | (k << (8 * (f.depth - 8)));
            if (fin && f.seen && k0 != 0) {                                                      // (k0 == 0 marks an empty row: a word of class-0 symbols alone takes the walk)
                if (sum >= m.i2info_id.size() || !m.i2info_valid[sum]) return;
                const int64_t id = (int64_t)m.i2info_id[sum] + m.id_offset;
                if (id < 0 || id > (int64_t)WF_ROW_ID_MASK) return;                            // (an id of more than 24 bits: no table)
                words.push_back({k0, k1, (uint32_t)id | ((uint32_t)(f.depth + 1) << WF_ROW_LEN_SHIFT)});
            }
            if (f.depth + 1 < WF_KEY_CHARS) stack.push_back({nx, sum, k0, k1, f.depth + 1, f.seen || fin});
            if (words.size() > (1u << 22)) return;
        }
    }
    if (words.empty()) return;
    if (place_words(words, m.bpe_tab, m.bpe_tab_bits, m.bpe_tab_m0, m.bpe_tab_m1, m.bpe_tab_m2)) m.bpe_tab_words = (int)words.size();
    else m.bpe_tab.clear();
}

bool build_model(Model &m, const uint8_t *img, size_t size)
{
    if (!img || size < 8) return fail(m, "empty model image");
    m.image.assign(img, img + size);
    m.image.resize(size + 16, 0);   // slack so that bounded unaligned reads at the tail are defined
    Reader R{m.image.data(), size};

    // ---- LDB container (format: reference FALDB.cpp:24-64)
    const int count = R.i32(0);
    if (count <= 0 || count > 3 * 64 || !R.ok(4, 4 * (size_t)count)) return fail(m, "bad LDB dump count");
    for (int i = 0; i < count; ++i) {
        int off = R.i32(4 + 4 * (size_t)i);
        if (off < 0 || (size_t)off >= size) return fail(m, "bad LDB dump offset");
        m.dump_off.push_back((size_t)off);
    }
    auto dump = [&](int i) -> Reader {
        if (i < 0 || i >= count) return Reader{m.image.data(), 0};
        size_t b = m.dump_off[(size_t)i];
        return Reader{m.image.data() + b, size - b};
    };
    TrivMap conf;
    if (!conf.set(dump(0))) return fail(m, "bad LDB configuration dump");
    std::vector<int> vals;

    // optional CRC32 verification (reference FALDB.cpp:67-116)
    if (conf.get(FUNC_GLOBAL, vals)) {
        for (size_t i = 0; i < vals.size(); ++i) {
            if (vals[i] == PARAM_VERIFY_LDB_BIN) {
                if (count < 2) return fail(m, "verify-ldb-bin without a validation dump");
                Reader v = dump(count - 1);
                if (v.u32(0) == 0) {
                    uint32_t dsize = 0, crc = 0;
                    for (int k = 0; k < count - 1; ++k) {
                        if (m.dump_off[(size_t)k + 1] < m.dump_off[(size_t)k]) return fail(m, "Invalid LDB binary file detected.");
                        size_t sz = m.dump_off[(size_t)k + 1] - m.dump_off[(size_t)k];
                        dsize += (uint32_t)sz; crc = crc32_update(m.image.data() + m.dump_off[(size_t)k], sz, crc);
                    }
                    if (dsize != v.u32(4) || crc != v.u32(8)) return fail(m, "Invalid LDB binary file detected.");
                }
            } else if (!is_boolean_param(vals[i])) ++i;
        }
    }

    // ---- [wbd] section (semantics: reference FAWbdConfKeeper.cpp:56-232)
    if (conf.get(FUNC_WBD, vals)) {
        m.has_wbd = true;
        int fsm_dump = -1, acts_dump = -1, charmap_dump = -1, fsm_type = TYPE_MOORE_DFA;
        for (size_t i = 0; i < vals.size(); ++i) {
            const int p = vals[i];
            if (p == PARAM_IGNORE_CASE) { m.ignore_case = true; continue; }
            if (i + 1 >= vals.size()) return fail(m, "truncated [wbd] parameters");
            const int v = vals[++i];
            switch (p) {
            case PARAM_MAP_MODE: if (v != MODE_PACK_TRIV) return fail(m, "[wbd] map mode must be triv"); break;
            case PARAM_DEPTH: m.max_depth = v; break;
            case PARAM_MAX_LENGTH: m.max_token_length = v; break;
            case PARAM_FSM_TYPE: fsm_type = v; break;
            case PARAM_FSM: fsm_dump = v; break;
            case PARAM_MULTI_MAP: acts_dump = v; break;
            case PARAM_CHARMAP: charmap_dump = v; break;
            default: break;   // tag ids (word/xword/seg/ignore/punkt/eos/eop/max-tag) and act-data: unused on this path
            }
        }
        if (fsm_dump >= 0) {
            if (fsm_type != TYPE_MOORE_DFA && fsm_type != TYPE_MOORE_MULTI_DFA) return fail(m, "[wbd] fsm-type must be moore-dfa or moore-multi-dfa");      // FAWbdConfKeeper.cpp:224-227
            if (acts_dump < 0) return fail(m, "[wbd] without an action map");
            if (m.max_depth < 0 || m.max_token_length < 0) return fail(m, "bad [wbd] limits");
            if (fsm_type == TYPE_MOORE_MULTI_DFA) {
                // FAWbdConfKeeper.cpp:219-224 gives such a model a State2Ows map and NO State2Ow one, and FALexTools_t has only the latter
                // (FALexTools_t.h:134): Process() returns -1 for every input (:412-414), so TextToIds answers 0 ids (tokdll:1202-1205:
                // -1 % 3 != 0) and TextToWords / TextToSentences -1 (tokdll:247-250, 499-502).  Reproduced by an automaton that has an
                // initial state and nothing else (no walk finds a token: 0 ids from every kernel) and lexer_void for the words forms.
                m.lexer_void = true;
                RawDfa &v = m.wbd_raw;
                v.initial = 0; v.remap = false; v.state_off.assign(1, 0); v.is_final.assign(1, 0); v.ow.assign(1, -1); v.tr_begin.assign(2, 0);
                v.off_index.assign(1, std::make_pair(0, 0));
            } else if (!decode_dfa(dump(fsm_dump), false, m.wbd_raw, m.error)) return false;
            if (!pack_dfa(m.wbd_raw, false, {IW_ANY, IW_L_ANCHOR, IW_R_ANCHOR, IW_EPSILON}, m.wbd, m.error)) return false;
            auto cls_sym = [&](int iw) -> uint32_t {   // symbol -> class of the packed table
                int c = m.wbd_raw.class_of(iw);
                if (c < 0) return CLS_NONE;
                if (!m.wbd_raw.remap) {
                    auto &S = m.wbd.sym_of_class;
                    auto it = std::lower_bound(S.begin(), S.end(), c);
                    if (it == S.end() || *it != c) return CLS_NONE;
                    c = (int)(it - S.begin());
                }
                return (uint32_t)c;
            };
            m.cls_any = cls_sym(IW_ANY); m.cls_l = cls_sym(IW_L_ANCHOR); m.cls_r = cls_sym(IW_R_ANCHOR);

            // actions (format: reference FAMultiMap_pack.cpp; semantics FALexTools_t.h:158-202, FAWbdConfKeeper.cpp:246-314)
            TrivMap acts;
            if (!acts.set(dump(acts_dump))) return fail(m, "bad action map");
            std::vector<std::vector<int>> actions;
            std::vector<int> a;
            int max_fn = -1;
            for (int id = 0; acts.get(id, a); ++id) {
                if (a.size() < 3) return fail(m, "invalid lexer action");
                actions.push_back(a);
                size_t i = 2;
                for (; i < a.size(); ++i) if (a[i] == 0 && i + 1 < a.size()) { ++i; break; }
                for (; i < a.size(); ++i) { if (a[i] < 0) return fail(m, "bad function id"); max_fn = std::max(max_fn, a[i]); }
            }
            if (m.lexer_void) { actions.clear(); max_fn = -1; }      // no state of the empty automaton is final: no action is ever looked up
            // Fn2Ini[f] = GetDest(GetDest(Initial, IW_R_ANCHOR), f)
            std::vector<long> fn2ini((size_t)std::max(max_fn + 1, 1), -1);
            fn2ini[0] = m.wbd.initial_base;
            if (max_fn >= 1) {
                long sr = m.cls_r == CLS_NONE ? -1 : m.wbd.step(m.wbd.initial_base, m.cls_r, nullptr, nullptr);
                for (int f = 1; f <= max_fn; ++f) {
                    uint32_t c = cls_sym(f);
                    fn2ini[(size_t)f] = (sr < 0 || c == CLS_NONE) ? -1 : m.wbd.step((uint32_t)sr, c, nullptr, nullptr);
                }
            }
            // destination of the left anchor from a frame's initial state, IW_ANY retry included (FALexTools_t.h:244-252, 265-270)
            auto after_l = [&](uint32_t ini) -> uint32_t {
                long d = m.cls_l == CLS_NONE ? -1 : m.wbd.step(ini, m.cls_l, nullptr, nullptr);
                if (d < 0 && m.cls_any != CLS_NONE) d = m.wbd.step(ini, m.cls_any, nullptr, nullptr);
                return d < 0 ? 0xFFFFFFFFu : (uint32_t)d;
            };
            m.initial_l = after_l(m.wbd.initial_base);
            std::vector<uint32_t> act_info(actions.size());
            for (size_t id = 0; id < actions.size(); ++id) {
                const auto &v = actions[id];
                const int left = v[0], right = v[1], tag = v[2];
                if (left < -65535 || left > 65535 || right < -65535 || right > 65535) return fail(m, "invalid lexer action context");
                size_t fi = v.size();
                if (v.size() == 3 && tag != 0) fi = 3;
                else if (v.size() > 3 && tag == 0) fi = 3;
                else if (v.size() > 4 && v[3] == 0) fi = 4;
                else return fail(m, "invalid lexer action");
                const size_t nfn = v.size() - fi;
                if (nfn == 0 && left == 0 && right == 0 && tag > 0) { act_info[id] = INFO_SIMPLE_BIT | (uint32_t)tag; continue; }
                act_info[id] = (uint32_t)m.acts_pool.size();
                m.acts_pool.push_back(left); m.acts_pool.push_back(right); m.acts_pool.push_back(tag); m.acts_pool.push_back((int)nfn);
                for (size_t k = fi; k < v.size(); ++k) {
                    const int fn = v[k];
                    if (fn < 0 || fn > max_fn || fn2ini[(size_t)fn] < 0) return fail(m, "lexer action calls an unknown function");
                    m.acts_pool.push_back(fn); m.acts_pool.push_back((int)fn2ini[(size_t)fn]);
                    m.acts_pool.push_back((int)after_l((uint32_t)fn2ini[(size_t)fn]));
                }
            }
            // static call depth: which functions can a function's rules call?  (bounds the frame stack of the lane program)
            {
                const RawDfa &rw = m.wbd_raw;
                const int r_sym = rw.remap ? rw.class_of(IW_R_ANCHOR) : IW_R_ANCHOR;
                std::vector<std::vector<int>> callees((size_t)max_fn + 1);
                std::vector<int> base2idx(m.wbd.table_len(), -1);
                for (size_t s = 0; s < rw.state_off.size(); ++s) base2idx[m.wbd.state_base[s]] = (int)s;
                for (int f = 0; f <= max_fn; ++f) {
                    if (fn2ini[(size_t)f] < 0) continue;
                    std::vector<uint8_t> seen(rw.state_off.size(), 0);
                    std::vector<int> stack{base2idx[(size_t)fn2ini[(size_t)f]]};
                    std::vector<uint8_t> called((size_t)max_fn + 1, 0);
                    auto note_final = [&](int st) {
                        if (!rw.is_final[(size_t)st]) return;
                        const int ow = rw.ow[(size_t)st];
                        if (ow < 0 || (size_t)ow >= actions.size()) return;
                        const auto &v = actions[(size_t)ow];
                        size_t fi = (v.size() > 3 && v[2] == 0) ? 3 : (v.size() > 4 && v[3] == 0) ? 4 : v.size();
                        for (size_t k = fi; k < v.size(); ++k) if (v[k] >= 0 && v[k] <= max_fn) called[(size_t)v[k]] = 1;
                    };
                    if (stack[0] < 0) continue;
                    seen[(size_t)stack[0]] = 1;
                    while (!stack.empty()) {
                        const int st = stack.back(); stack.pop_back();
                        note_final(st);
                        for (uint32_t t = rw.tr_begin[(size_t)st]; t < rw.tr_begin[(size_t)st + 1]; ++t) {
                            const int dst = rw.tr_dst[t];
                            if (dst < 0) continue;
                            if (rw.tr_sym[t] == r_sym) { note_final(dst); continue; }   // the right anchor is always the last symbol fed
                            if (!seen[(size_t)dst]) { seen[(size_t)dst] = 1; stack.push_back(dst); }
                        }
                    }
                    for (int g = 0; g <= max_fn; ++g) if (called[(size_t)g]) callees[(size_t)f].push_back(g);
                }
                std::vector<int> depth((size_t)max_fn + 1, -1); std::vector<uint8_t> on((size_t)max_fn + 1, 0);
                std::function<int(int)> dep = [&](int f) -> int {
                    if (on[(size_t)f]) return 1 << 20;                 // recursion: bounded only by max-depth
                    if (depth[(size_t)f] >= 0) return depth[(size_t)f];
                    on[(size_t)f] = 1; int dmax = 1;
                    for (int g : callees[(size_t)f]) dmax = std::max(dmax, std::min(1 << 20, 1 + dep(g)));
                    on[(size_t)f] = 0; depth[(size_t)f] = dmax; return dmax;
                };
                const int call_depth = max_fn >= 0 ? dep(0) : 1;
                m.lex_frames = std::max(0, std::min(m.max_depth, call_depth) - 1);
            }
            m.wbd_info.assign(m.wbd.table_len(), 0);
            for (size_t s = 0; s < m.wbd_raw.state_off.size(); ++s) {
                if (!m.wbd_raw.is_final[s]) continue;
                const int ow = m.wbd_raw.ow[s];
                if (ow < 0 || (size_t)ow >= actions.size()) return fail(m, "final state without a valid action");
                m.wbd_info[m.wbd.state_base[s]] = act_info[(size_t)ow];
            }

            // device entries (bf_layout.h): low word [final:1 | next:18 | cls:13], high word = action info of a final destination
            if (m.wbd.nclasses >= (int)LX_CLS_NONE) return fail(m, "too many symbol classes for the lexer table entry format");
            m.wbd_t2.resize(m.wbd.t32.size());
            for (size_t i = 0; i < m.wbd.t32.size(); ++i) {
                const uint32_t e = m.wbd.t32[i];
                if ((e & T32_CLS_MASK) == T32_CLS_MASK) { m.wbd_t2[i] = LX_T_CLS_MASK; continue; }     // empty slot
                const uint32_t next = e >> T32_NEXT_SHIFT;
                const bool fin = (e & T32_FINAL_BIT) != 0;
                const uint32_t lo = (e & T32_CLS_MASK) | (next << LX_T_NEXT_SHIFT) | (fin ? LX_T_FINAL : 0u);
                m.wbd_t2[i] = (uint64_t)lo | ((uint64_t)(fin ? m.wbd_info[next] : 0u) << 32);
            }

            // two-level shape (bf_lex.h LexTables::two_level): (1) every action with functions is [0, 0, tag != 0, one function],
            // (2) from a called function's initial states (plain and after the left anchor) only final states with SIMPLE actions
            // can be reached -- whatever symbols are fed.  Then the call depth is exactly 1 and the caller is the top-level frame.
            {
                bool ok = m.max_depth >= 2 && !m.acts_pool.empty();
                std::vector<uint32_t> fn_starts;
                for (size_t id = 0; ok && id < actions.size(); ++id) {
                    if (act_info[id] & INFO_SIMPLE_BIT) continue;
                    const int32_t *a = m.acts_pool.data() + act_info[id];
                    if (a[0] != 0 || a[1] != 0 || a[2] == 0 || a[3] != 1) { ok = false; break; }
                    fn_starts.push_back((uint32_t)a[5]);
                    if ((uint32_t)a[6] != 0xFFFFFFFFu) fn_starts.push_back((uint32_t)a[6]);
                }
                if (ok && !fn_starts.empty()) {
                    const RawDfa &rw = m.wbd_raw;
                    std::vector<int> base2idx(m.wbd.table_len(), -1);
                    for (size_t st = 0; st < rw.state_off.size(); ++st) base2idx[m.wbd.state_base[st]] = (int)st;
                    std::vector<uint8_t> seen(rw.state_off.size(), 0);
                    std::vector<int> stack;
                    for (uint32_t b : fn_starts) { const int st = b < base2idx.size() ? base2idx[b] : -1; if (st >= 0 && !seen[(size_t)st]) { seen[(size_t)st] = 1; stack.push_back(st); } }
                    while (ok && !stack.empty()) {
                        const int st = stack.back(); stack.pop_back();
                        if (rw.is_final[(size_t)st] && !(m.wbd_info[m.wbd.state_base[(size_t)st]] & INFO_SIMPLE_BIT)) { ok = false; break; }
                        for (uint32_t t = rw.tr_begin[(size_t)st]; t < rw.tr_begin[(size_t)st + 1]; ++t) {
                            const int dst = rw.tr_dst[t];
                            if (dst >= 0 && !seen[(size_t)dst]) { seen[(size_t)dst] = 1; stack.push_back(dst); }
                        }
                    }
                } else ok = false;          // no calling action at all: the general machine is already flat
                m.two_level = ok;
            }
            // fn_no_ra: from the initial states of EVERY function an action can call (plain and after the left anchor), following
            // all transitions, no state has a transition on the right anchor
            {
                const RawDfa &rw = m.wbd_raw;
                const int r_sym = rw.remap ? rw.class_of(IW_R_ANCHOR) : IW_R_ANCHOR;
                std::vector<uint32_t> starts;
                for (size_t id = 0; id < actions.size(); ++id) {
                    if (act_info[id] & INFO_SIMPLE_BIT) continue;
                    const int32_t *a = m.acts_pool.data() + act_info[id];
                    for (int k = 0; k < a[3]; ++k) {            // per function: (id, initial state, initial state after the left anchor)
                        starts.push_back((uint32_t)a[5 + 3 * k]);
                        if ((uint32_t)a[6 + 3 * k] != 0xFFFFFFFFu) starts.push_back((uint32_t)a[6 + 3 * k]);
                    }
                }
                bool ok = !starts.empty();
                if (ok && r_sym >= 0) {
                    std::vector<int> base2idx(m.wbd.table_len(), -1);
                    for (size_t st = 0; st < rw.state_off.size(); ++st) base2idx[m.wbd.state_base[st]] = (int)st;
                    std::vector<uint8_t> seen(rw.state_off.size(), 0);
                    std::vector<int> stack;
                    for (uint32_t b : starts) { const int st = b < base2idx.size() ? base2idx[b] : -1; if (st >= 0 && !seen[(size_t)st]) { seen[(size_t)st] = 1; stack.push_back(st); } }
                    while (ok && !stack.empty()) {
                        const int st = stack.back(); stack.pop_back();
                        for (uint32_t t = rw.tr_begin[(size_t)st]; t < rw.tr_begin[(size_t)st + 1]; ++t) {
                            if (rw.tr_sym[t] == r_sym) { ok = false; break; }
                            const int dst = rw.tr_dst[t];
                            if (dst >= 0 && !seen[(size_t)dst]) { seen[(size_t)dst] = 1; stack.push_back(dst); }
                        }
                    }
                }
                m.fn_no_ra = ok;
            }

            // loop state: the state with the most transitions to itself (ties: the first).  Only text classes matter: the anchors
            // and IW_ANY never occur in a class stream.
            m.loop_cls.assign((size_t)m.wbd.nclasses, 0);
            {
                const RawDfa &rw = m.wbd_raw;
                int best_s = -1; uint32_t best_n = 0;
                for (size_t st = 0; st < rw.state_off.size(); ++st) {
                    uint32_t n = 0;
                    for (uint32_t t = rw.tr_begin[st]; t < rw.tr_begin[st + 1]; ++t) n += rw.tr_dst[t] == (int)st;
                    if (n > best_n) { best_n = n; best_s = (int)st; }
                }
                if (best_s >= 0 && best_n >= 8) {
                    for (uint32_t t = rw.tr_begin[(size_t)best_s]; t < rw.tr_begin[(size_t)best_s + 1]; ++t) {
                        if (rw.tr_dst[t] != best_s) continue;
                        int c = rw.tr_sym[t];                               // class if remap, raw symbol otherwise
                        if (!rw.remap) {
                            auto &S = m.wbd.sym_of_class;
                            auto it = std::lower_bound(S.begin(), S.end(), c);
                            if (it == S.end() || *it != c) continue;
                            c = (int)(it - S.begin());
                        }
                        if (c >= 0 && c < m.wbd.nclasses && (uint32_t)c != m.cls_any && (uint32_t)c != m.cls_l && (uint32_t)c != m.cls_r) m.loop_cls[(size_t)c] = 1;
                    }
                    m.loop_base = m.wbd.state_base[(size_t)best_s];
                    m.loop_final = rw.is_final[(size_t)best_s] != 0;
                    m.loop_info = m.loop_final ? m.wbd_info[m.loop_base] : 0;
                }
            }
            auto flagged = [&](uint32_t k) -> uint32_t { return (k < m.loop_cls.size() && m.loop_cls[k]) ? (k | LX_C_LOOP) : k; };

            // fused code point -> class map (charmap semantics: reference FAUtils_cl.h:311-369 + FAMultiMap_pack_fixed.cpp:67-137)
            FixedMap cm;
            if (charmap_dump >= 0) { if (!cm.set(dump(charmap_dump))) return fail(m, "bad charmap"); m.wbd_has_charmap = true; }
            m.wbd_cpmap.init(CLS_NONE);
            m.words_cpmap.init(CLS_NONE);
            for (int cp = 0; cp <= 0x10FFFF; ++cp) {
                // a letter on its way into GetDest (FALexTools_t.h:256-264): "Iw < IW_EPSILON -> DefSubIw", then -- ignore-case lexers -- the fold
                // (only letters take this way: anchors and function ids are fed as they are)
                auto cls_text = [&](int o) -> uint32_t { o = o < IW_EPSILON ? IW_EPSILON : o; return cls_sym(m.ignore_case ? bf_tolower(o) : o); };
                { const int wcp = cp == 0 ? 0x20 : cp;                          // tokdll:482
                  const uint32_t k = cls_text(wcp); if (k != CLS_NONE) m.words_cpmap.set(cp, flagged(k)); }
                int norm[10]; int c = cm.set_ ? cm.get(cp, norm, 10) : -1;
                if (c == -1) { uint32_t k = cls_text(cp); if (k != CLS_NONE) m.wbd_cpmap.set(cp, flagged(k)); }
                else if (c == 1) { uint32_t k = cls_text(norm[0]); if (k != CLS_NONE) m.wbd_cpmap.set(cp, flagged(k)); }
                else {
                    if (c < 0 || c > 10) c = 0;    // silently dropped by FANormalize
                    m.wbd_charmap_multi = true;
                    if (m.wbd_multi_pool.size() > 0x7fff0000u) return fail(m, "charmap too large");
                    m.wbd_cpmap.set(cp, FUSED_MULTI | (uint32_t)m.wbd_multi_pool.size());
                    m.wbd_multi_pool.push_back((uint16_t)c);
                    for (int k = 0; k < c; ++k) m.wbd_multi_pool.push_back((uint16_t)flagged(cls_text(norm[k])));
                }
            }

            // ---- unit form (bf_wave.h): which conditions hold, and what a walk that starts on each class does
            {
                const RawDfa &rw = m.wbd_raw;
                const int nst = (int)rw.state_off.size();
                std::vector<int> base2idx(m.wbd.table_len(), -1);
                for (int st = 0; st < nst; ++st) base2idx[m.wbd.state_base[(size_t)st]] = st;
                auto cls_of_tr = [&](uint32_t t) -> int {          // class of transition t in the packed table's numbering, -1 if none
                    int c = rw.tr_sym[t];
                    if (!rw.remap) { auto &S = m.wbd.sym_of_class; auto it = std::lower_bound(S.begin(), S.end(), c); if (it == S.end() || *it != c) return -1; c = (int)(it - S.begin()); }
                    return c;
                };
                auto is_text = [&](int c) { return c >= 0 && (uint32_t)c != m.cls_any && (uint32_t)c != m.cls_l && (uint32_t)c != m.cls_r; };
                std::string why;
                if (!m.two_level) why = "not a two-level lexer";
                else if (m.cls_any != CLS_NONE) why = "IW_ANY is in the alphabet";
                else if (m.initial_l != 0xFFFFFFFFu) why = "the top level takes the left anchor";
                else if (m.wbd_charmap_multi) why = "the charmap deletes or expands characters";
                else if (!m.fn_no_ra) why = "a function rule uses the right anchor";
                else if (m.max_depth < 2) why = "max-depth < 2";
                else if (m.max_token_length < 1 || m.max_token_length > 496) why = "max token length outside 1..496";
                else if (m.acts_pool.size() > 64) why = "more than 64 ints of action records";
                // C = states behind at least one letter from the initial state (text classes only): what a top-level walk can be in
                std::vector<uint8_t> inC((size_t)nst, 0);
                if (why.empty()) {
                    std::vector<int> stack;
                    auto push_text_dsts = [&](int st) {
                        for (uint32_t t = rw.tr_begin[(size_t)st]; t < rw.tr_begin[(size_t)st + 1]; ++t) {
                            const int dst = rw.tr_dst[t];
                            if (dst < 0 || !is_text(cls_of_tr(t))) continue;
                            if (!inC[(size_t)dst]) { inC[(size_t)dst] = 1; stack.push_back(dst); }
                        }
                    };
                    push_text_dsts(rw.initial);
                    while (!stack.empty()) { const int st = stack.back(); stack.pop_back(); push_text_dsts(st); }
                    for (int st = 0; st < nst && why.empty(); ++st) {
                        if (!inC[(size_t)st]) continue;
                        for (uint32_t t = rw.tr_begin[(size_t)st]; t < rw.tr_begin[(size_t)st + 1]; ++t)
                            if (m.cls_r != CLS_NONE && cls_of_tr(t) == (int)m.cls_r) { why = "a top-level rule uses the right anchor"; break; }
                        if (!why.empty() || !rw.is_final[(size_t)st]) continue;
                        const uint32_t inf = m.wbd_info[m.wbd.state_base[(size_t)st]];
                        const int tag = (inf & INFO_SIMPLE_BIT) ? (int)(inf & 0x7FFFFFFFu) : m.acts_pool[(size_t)inf + 2];
                        if (tag < 1 || tag > 4) why = "a top-level tag outside 1..4";
                    }
                }
                // every final state below a function's initial states carries a vocabulary tag (> 4: never a word, always a sub-token)
                if (why.empty()) {
                    std::vector<uint8_t> seen((size_t)nst, 0); std::vector<int> stack;
                    for (size_t id = 0; id < actions.size(); ++id) {
                        if (act_info[id] & INFO_SIMPLE_BIT) continue;
                        const int32_t *a = m.acts_pool.data() + act_info[id];
                        for (uint32_t b : {(uint32_t)a[5], (uint32_t)a[6]}) { const int st = b < base2idx.size() ? base2idx[b] : -1; if (st >= 0 && !seen[(size_t)st]) { seen[(size_t)st] = 1; stack.push_back(st); } }
                    }
                    while (!stack.empty() && why.empty()) {
                        const int st = stack.back(); stack.pop_back();
                        if (rw.is_final[(size_t)st]) {
                            const uint32_t inf = m.wbd_info[m.wbd.state_base[(size_t)st]];
                            if (!(inf & INFO_SIMPLE_BIT) || (int)(inf & 0x7FFFFFFFu) <= 4) why = "a vocabulary tag <= 4";
                        }
                        for (uint32_t t = rw.tr_begin[(size_t)st]; t < rw.tr_begin[(size_t)st + 1]; ++t) { const int dst = rw.tr_dst[t]; if (dst >= 0 && !seen[(size_t)dst]) { seen[(size_t)dst] = 1; stack.push_back(dst); } }
                    }
                }
                m.wave_ok = why.empty(); m.wave_why = why;
                m.wave_kind.assign((size_t)m.wbd.nclasses, 0 /* WK_GENERAL */);
                if (m.wave_ok) {
                    // the loop state is usable as a run when it is closed (nothing but its self-loops), final, and every class that
                    // loops there also ENTERS it from the initial state (so "run of flagged elements" == "what the walk consumes")
                    const int ls = m.loop_base == 0xFFFFFFFFu ? -1 : base2idx[m.loop_base];
                    bool loop_ok = ls >= 0 && m.loop_final;
                    if (loop_ok) for (uint32_t t = rw.tr_begin[(size_t)ls]; t < rw.tr_begin[(size_t)ls + 1]; ++t) if (rw.tr_dst[t] != ls || !is_text(cls_of_tr(t))) { loop_ok = false; break; }
                    if (loop_ok) for (int c = 0; c < m.wbd.nclasses; ++c) if (m.loop_cls[(size_t)c] && m.wbd.step(m.wbd.initial_base, (uint32_t)c, nullptr, nullptr) != (long)m.loop_base) { loop_ok = false; break; }
                    for (int c = 0; c < m.wbd.nclasses; ++c) {
                        if (!is_text(c)) { m.wave_kind[(size_t)c] = 2; continue; }           // never in a class stream
                        int fin = 0;
                        const long d = m.wbd.step(m.wbd.initial_base, (uint32_t)c, &fin, nullptr);
                        if (d < 0) { m.wave_kind[(size_t)c] = 2 /* WK_NOMATCH */; continue; }
                        const int st = base2idx[(size_t)d];
                        if (loop_ok && d == (long)m.loop_base && m.loop_cls[(size_t)c]) m.wave_kind[(size_t)c] = 1 /* WK_LOOP */;
                        else if (st >= 0 && fin && rw.tr_begin[(size_t)st] == rw.tr_begin[(size_t)st + 1]) m.wave_kind[(size_t)c] = 3 /* WK_SOLO */;
                    }
                    // WK_SOLO tokens carry ONE action (the most frequent among the solo destinations; the BERT lexers have one);
                    // a solo class with another action is left to the general walk
                    std::vector<std::pair<uint32_t, int>> freq;
                    auto solo_inf = [&](int c) { return m.wbd_info[(size_t)m.wbd.step(m.wbd.initial_base, (uint32_t)c, nullptr, nullptr)]; };
                    for (int c = 0; c < m.wbd.nclasses; ++c) {
                        if (m.wave_kind[(size_t)c] != 3) continue;
                        const uint32_t inf = solo_inf(c);
                        size_t k = 0; for (; k < freq.size(); ++k) if (freq[k].first == inf) break;
                        if (k == freq.size()) freq.push_back({inf, 0});
                        ++freq[k].second;
                    }
                    int best = -1; for (size_t k = 0; k < freq.size(); ++k) if (best < 0 || freq[k].second > freq[(size_t)best].second) best = (int)k;
                    m.wave_solo_info = best >= 0 ? freq[(size_t)best].first : 0;
                    for (int c = 0; c < m.wbd.nclasses; ++c) if (m.wave_kind[(size_t)c] == 3 && solo_inf(c) != m.wave_solo_info) m.wave_kind[(size_t)c] = 0;
                }
                if (m.wave_ok) build_flat_table(m);
            }
        }
    }

    // ---- [pos-dict] section (semantics: reference FADictConfKeeper.cpp:57-228)
    if (conf.get(FUNC_POS_DICT, vals)) {
        m.has_seg = true;
        int fsm_dump = -1, i2info_dump = -1, charmap_dump = -1, k2i_dump = -1, fsm_type = TYPE_MOORE_DFA, mode = MODE_PACK_TRIV;
        for (size_t i = 0; i < vals.size(); ++i) {
            const int p = vals[i];
            if (p == PARAM_NO_TR) continue;
            if (p == PARAM_IGNORE_CASE) { m.dict_ignore_case = true; continue; }
            if (p == PARAM_USE_BYTE_ENCODING) { m.use_bytes = true; continue; }
            if (p == PARAM_NO_DUMMY_PREFIX) { m.no_dummy_prefix = true; continue; }
            if (i + 1 >= vals.size()) return fail(m, "truncated [pos-dict] parameters");
            const int v = vals[++i];
            switch (p) {
            case PARAM_DIRECTION: m.dict_direction = v; break;
            case PARAM_TOKENIZATION_TYPE: m.tok_algo = v; break;
            case PARAM_ID_OFFSET: m.id_offset = v; break;
            case PARAM_FSM_TYPE: fsm_type = v; break;
            case PARAM_MAP_MODE: mode = v; break;
            case PARAM_FSM: if (fsm_type != TYPE_MEALY_DFA) return fail(m, "[pos-dict] fsm must be a Mealy DFA (fsm-type must precede fsm)"); fsm_dump = v; break;
            case PARAM_ARRAY: k2i_dump = v; break;
            case PARAM_CHARMAP: charmap_dump = v; break;
            case PARAM_MULTI_MAP: if (mode != MODE_PACK_FIXED) return fail(m, "[pos-dict] multi-map must be fixed-dump"); i2info_dump = v; break;
            default: return fail(m, "unknown [pos-dict] parameter");
            }
        }
        if (fsm_dump < 0 || i2info_dump < 0 || k2i_dump < 0) return fail(m, "[pos-dict] is missing fsm / array / multi-map");
        if (dump(k2i_dump).i32(12) <= 0) return fail(m, "[pos-dict] K2I array is empty");   // reference 1best_t.h:113
        m.kind = m.tok_algo == TOKENIZE_BPE ? KIND_BPE : m.tok_algo == TOKENIZE_BPE_OPT ? KIND_BPE_OPT :
                 m.tok_algo == TOKENIZE_BPE_OPT_WITH_MERGES ? KIND_BPE_MERGES : KIND_UNIGRAM;
        if (!decode_dfa(dump(fsm_dump), true, m.dict_raw, m.error)) return false;
        if (!pack_dfa(m.dict_raw, true, {}, m.dict, m.error)) return false;
        m.dict_clsmap.init(CLS_NONE_W);
        for (size_t c = 0; c < m.dict.sym_of_class.size(); ++c) m.dict_clsmap.set(m.dict.sym_of_class[c], (uint32_t)c);
        for (int s : m.dict.sym_of_class) if (s < 0 || s > 0x10FFFF) return fail(m, "dictionary symbol outside the code point range");

        FixedMap info;
        if (!info.set(dump(i2info_dump)) || info.sov != 4) return fail(m, "bad I2Info map");
        const size_t rows = (size_t)(info.max_key - info.min_key + 1);
        m.i2info_min_key = info.min_key;
        m.i2info_id.assign(rows, 0); m.i2info_score.assign(rows, 0); m.i2info_valid.assign(rows, 0);
        for (size_t k = 0; k < rows; ++k) {
            int v[2] = {0, 0}; int c = info.get(info.min_key + (int)k, v, 2);
            if (c < 1 || c > 2) continue;
            m.i2info_id[k] = v[0]; m.i2info_score[k] = (uint32_t)v[1]; m.i2info_valid[k] = (uint8_t)c;
            if (v[0] > m.max_info_id) m.max_info_id = v[0];
            if (v[0] < -1) m.max_info_id = 0x7fffffff;            // negative ids: outside the packed record format
        }
        // general form of the same rows + the K2I array, for the key -> info lookup (FADictInterpreter_t::GetInfo)
        m.info_stride = info.max_count + 1; m.info_min_key = info.min_key;
        if ((double)rows * (double)m.info_stride > 2.0e8) return fail(m, "I2Info too large");
        m.info_rows.assign(rows * (size_t)m.info_stride, 0);
        {
            std::vector<int> tmp((size_t)info.max_count + 1);
            for (size_t k = 0; k < rows; ++k) {
                const int c = info.get(info.min_key + (int)k, tmp.data(), info.max_count);
                int32_t *row = m.info_rows.data() + k * (size_t)m.info_stride;
                row[0] = c < 0 ? -1 : c;
                for (int q = 0; q < c && q < info.max_count; ++q) row[1 + q] = tmp[(size_t)q];
            }
        }
        {   // K2I (format: reference FAArray_pack.cpp:27-65; GetAt :68-95)
            const Reader kr = dump(k2i_dump);
            const int M = kr.i32(0), soi = kr.i32(4), sov = kr.i32(8), count = kr.i32(12);
            if (M < 1 || M > 8 || soi < 0 || soi > 4 || sov < 1 || sov > 4 || count <= 0 || count > 100000000 || (M == 1) != (soi == 0)) return fail(m, "bad K2I array");
            const size_t index_bytes = M == 1 ? 0 : (size_t)((count + M - 1) / M) * (size_t)soi;
            auto dec = [&](size_t base, size_t idx, int size) -> uint32_t { return kr.be(base + idx * (size_t)size, size); };   // FADecode_1_2_3_4_idx: big-endian (FAEncodeUtils.h:418-450)
            m.k2i.resize((size_t)count);
            for (int i = 0; i < count; ++i) {
                if (M == 1) { if (!kr.ok(16 + (size_t)i * (size_t)sov, (size_t)sov)) return fail(m, "truncated K2I array"); m.k2i[(size_t)i] = (int32_t)dec(16, (size_t)i, sov); }
                else {
                    const uint32_t chain = dec(16, (size_t)(i / M), soi);
                    const size_t base = 16 + index_bytes + (size_t)chain * (size_t)(M * sov);
                    if (!kr.ok(base, (size_t)(M * sov))) return fail(m, "truncated K2I array");
                    m.k2i[(size_t)i] = (int32_t)dec(base, (size_t)(i % M), sov);
                }
            }
        }
        // every reachable final state's MPH index must have a usable row (else the reference LogAsserts at run time)
        {
            const auto &rw = m.dict_raw; const size_t ns = rw.state_off.size();
            // longest path + reachable ow sums are bounded by construction; verify keys lazily on the GPU would be silent,
            // so verify all finals here by DFS with accumulated ow
            std::vector<int> depth(ns, -1); std::vector<uint8_t> onstack(ns, 0);
            bool cyclic = false;
            std::function<int(int)> dfs = [&](int s) -> int {
                if (depth[(size_t)s] >= 0) return depth[(size_t)s];
                if (onstack[(size_t)s]) { cyclic = true; return 0; }
                onstack[(size_t)s] = 1; int d = 0;
                for (uint32_t t = rw.tr_begin[(size_t)s]; t < rw.tr_begin[(size_t)s + 1]; ++t)
                    if (rw.tr_dst[t] >= 0) d = std::max(d, 1 + dfs(rw.tr_dst[t]));
                onstack[(size_t)s] = 0; depth[(size_t)s] = d; return d;
            };
            // iterative-safe: tries are shallow (<= 128), recursion depth == path length
            m.trie_max_depth = dfs(rw.initial);
            if (cyclic) m.trie_max_depth = 0x7fffffff;
            // bf_bpe_wave_body.h: an arc never crosses a word when no state but the initial one has a transition on U+2581
            bool inner_delim = false;
            for (size_t s = 0; s < ns && !inner_delim; ++s)
                if ((int)s != rw.initial)
                    for (uint32_t t = rw.tr_begin[s]; t < rw.tr_begin[s + 1]; ++t) if (rw.tr_dst[t] >= 0 && rw.tr_sym[t] == 0x2581) { inner_delim = true; break; }
            m.bpe_wave_ok = (m.kind == KIND_BPE_OPT || m.kind == KIND_BPE_MERGES) && !cyclic && !inner_delim;       // both collect with m_fFastBpe (..._bpe_t.h:110, ..._with_merges_t.h:113)
        }
        const int need = m.kind == KIND_UNIGRAM || m.kind == KIND_BPE_MERGES ? 2 : 1;
        size_t nvalid = 0; for (uint8_t v : m.i2info_valid) if (v >= need) ++nvalid;
        if (nvalid == 0) return fail(m, "I2Info has no usable rows");

        if (charmap_dump >= 0) {
            FixedMap cm;
            if (!cm.set(dump(charmap_dump))) return fail(m, "bad charmap");
            m.dict_has_charmap = true;
            m.dict_charmap.init(NORM_NONE);
            for (int cp = cm.min_key; cp <= cm.max_key && cp <= 0x10FFFF; ++cp) {
                int norm[10]; int c = cm.get(cp, norm, 10);
                if (c == -1) continue;
                if (c < 0 || c > 10) c = 0;
                if (c == 1 && norm[0] >= 0 && norm[0] < (1 << 24)) { m.dict_charmap.set(cp, (1u << 24) | (uint32_t)norm[0]); continue; }
                if (m.dict_norm_pool.size() >= (1u << 24)) return fail(m, "charmap too large");
                m.dict_charmap.set(cp, ((uint32_t)c << 24) | (uint32_t)m.dict_norm_pool.size());
                if (c == 1) { m.dict_charmap.set(cp, (11u << 24) | (uint32_t)m.dict_norm_pool.size()); }
                for (int k = 0; k < c; ++k) m.dict_norm_pool.push_back(norm[k]);
            }
            if (cm.max_key > 0x10FFFF) return fail(m, "charmap keys beyond the code point range");
        }
        if (m.dict_ignore_case) {
            // key normalisation of an ignore-case dictionary (FADictInterpreter_t.h:203-205, 231-247): every symbol is folded, THEN the
            // charmap (if any) is applied -- in either direction.  One map does both: symbol -> charmap entry of its fold, or the fold itself
            m.dict_lookup_map.init(NORM_NONE);
            for (int cp = 0; cp <= 0x10FFFF; ++cp) {
                const int l = bf_tolower(cp);
                const uint32_t e = m.dict_has_charmap ? m.dict_charmap.get(l) : NORM_NONE;
                if (e != NORM_NONE) m.dict_lookup_map.set(cp, e);
                else if (l != cp) m.dict_lookup_map.set(cp, (1u << 24) | (uint32_t)l);
            }
        }
    }
    if (m.has_seg) {
        // ---- device forms of the dictionary side
        if (m.dict.nclasses >= 0xFFF0) return fail(m, "dictionary alphabet too large for the 16-bit class stream");
        if (m.i2info_min_key != 0) return fail(m, "I2Info keys must start at 0");
        {   // the MPH index of every accepted string must have a usable I2Info row (the reference LogAsserts otherwise)
            const auto &rw = m.dict_raw; const size_t ns = rw.state_off.size();
            std::vector<double> paths(ns, -1.0);
            std::function<double(int)> cnt = [&](int s) -> double {
                if (paths[(size_t)s] >= 0) return paths[(size_t)s];
                double c = rw.is_final[(size_t)s] ? 1.0 : 0.0;
                paths[(size_t)s] = 0;   // cycle guard
                for (uint32_t t = rw.tr_begin[(size_t)s]; t < rw.tr_begin[(size_t)s + 1]; ++t) if (rw.tr_dst[t] >= 0) c += cnt(rw.tr_dst[t]);
                paths[(size_t)s] = c; return c;
            };
            const double nstrings = m.trie_max_depth == 0x7fffffff ? 1e18 : cnt(rw.initial);
            const int need = (m.kind == KIND_UNIGRAM || m.kind == KIND_BPE_MERGES) ? 2 : 1;
            if (nstrings > (double)m.i2info_id.size()) return fail(m, "dictionary accepts more strings than I2Info has rows");
            for (size_t k = 0; k < (size_t)nstrings; ++k) if (m.i2info_valid[k] < need) return fail(m, "I2Info row without id/score");
        }
        m.seg_info.resize(m.i2info_id.size());
        for (size_t k = 0; k < m.i2info_id.size(); ++k) m.seg_info[k] = (uint64_t)(uint32_t)m.i2info_id[k] | ((uint64_t)m.i2info_score[k] << 32);
        m.seg_score = m.i2info_score;
        if (m.kind == KIND_BPE || m.kind == KIND_BPE_OPT || m.kind == KIND_BPE_MERGES) {
            bool ids_ok = true;
            for (size_t k = 0; k < m.i2info_id.size(); ++k) if (m.i2info_valid[k] && (m.i2info_id[k] < 0 || m.i2info_id[k] >= (1 << 20))) ids_ok = false;
            m.bpe_seg_ok = ids_ok && m.trie_max_depth > 0 && m.trie_max_depth <= 256;
            m.bpe_prio_bits = 22;
            if (m.kind == KIND_BPE_MERGES) {
                // the order of ..._with_merges_t.h:242-262 over the entries: bigger ranks first (float comparison), then smaller ids
                std::vector<uint32_t> order;
                for (size_t k = 0; k < m.i2info_id.size(); ++k) if (m.i2info_valid[k]) order.push_back((uint32_t)k);
                auto rank_of = [&](uint32_t k) { float f; const uint32_t b = m.i2info_score[k]; memcpy(&f, &b, 4); return f; };
                std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
                    const float ra = rank_of(a), rb = rank_of(b);
                    if (ra > rb) return true;
                    if (!(ra == rb)) return false;
                    return m.i2info_id[a] < m.i2info_id[b];
                });
                m.bpe_prio.assign(m.i2info_id.size(), 0u); m.bpe_place_id.resize(order.size()); m.bpe_place_rank.resize(order.size());
                // entries that compare equal (same rank, same id) share a place: the reference orders their arcs by start alone
                uint32_t place = 0;
                for (size_t q = 0; q < order.size(); ++q) {
                    if (q > 0) { const uint32_t a = order[q - 1], b = order[q]; if (!(rank_of(a) == rank_of(b) && m.i2info_id[a] == m.i2info_id[b])) ++place; }
                    m.bpe_prio[order[q]] = 2u * place + 1u; m.bpe_place_id[place] = m.i2info_id[order[q]]; m.bpe_place_rank[place] = m.i2info_score[order[q]];
                }
                m.bpe_place_id.resize(order.empty() ? 0 : place + 1); m.bpe_place_rank.resize(m.bpe_place_id.size());
                if (2ull * (m.bpe_place_id.size() + 1) >= (1ull << 22)) m.bpe_seg_ok = false;
                for (size_t k = 0; k < m.i2info_id.size(); ++k) if (m.i2info_valid[k] && rank_of((uint32_t)k) != rank_of((uint32_t)k)) m.bpe_seg_ok = false;   // NaN ranks: no order
            }
        }
        auto is_ws = [](int c) { return c <= 0x20 || c == 0xa0 || (c >= 0x2000 && c <= 0x200f) || c == 0x202f || c == 0x205f || c == 0x2060 ||
                                        c == 0x2420 || c == 0x2424 || c == 0x3000 || c == 0xfeff; };   // blingfiretokdll.h:17-21
        auto code_of = [&](int c) -> uint16_t {
            if (is_ws(c)) return SP_WS;
            uint32_t k = m.dict_clsmap.get(c);
            if (k != CLS_NONE_W) return (uint16_t)k;
            return c == 0x2581 ? SP_DELIM_ABSENT : SP_NONE;
        };
        m.sp_delim_code = code_of(0x2581);
        m.sp_cpmap.init(SP_NONE);
        const int cp_max = m.use_bytes ? 255 : 0x10FFFF;
        auto fused = [&](int cp, std::vector<uint16_t> &codes) {
            codes.clear();
            uint32_t v = m.dict_has_charmap ? m.dict_charmap.get(cp) : NORM_NONE;
            if (v == NORM_NONE) { codes.push_back(code_of(cp)); return; }
            const uint32_t cntv = v >> 24, pay = v & 0xffffffu;
            if (cntv == 1) codes.push_back(code_of((int)pay));
            else if (cntv == 11) codes.push_back(code_of(m.dict_norm_pool[pay]));
            else for (uint32_t k = 0; k < cntv; ++k) codes.push_back(code_of(m.dict_norm_pool[pay + k]));
        };
        std::vector<uint16_t> codes;
        for (int cp = 0; cp <= cp_max; ++cp) {
            fused(cp, codes);
            if (codes.size() == 1) { if (codes[0] != SP_NONE) m.sp_cpmap.set(cp, codes[0]); continue; }
            m.sp_has_multi = true;
            m.sp_cpmap.set(cp, FUSED_MULTI | (uint32_t)m.sp_multi_pool.size());
            m.sp_multi_pool.push_back((uint16_t)codes.size());
            for (uint16_t c : codes) m.sp_multi_pool.push_back(c);
        }
        fused(0x2581, m.sp_prefix);
        build_bpe_word_table(m);
    }
    // ---- [i2w] (reference tokdll:998-1045): string array dump + the range of regular token ids
    if (conf.get(35 /* FUNC_I2W */, vals)) {
        for (size_t i = 0; i < vals.size(); ++i) {
            if (vals[i] == 75 /* PARAM_STRING_ARRAY */ && i + 1 < vals.size()) {
                Reader d = dump(vals[++i]);
                const int cnt = d.i32(0);
                if (d.n < 8 || cnt < 0 || !d.ok(4, 4 * ((size_t)cnt + 1))) return fail(m, "bad [i2w] string array");
                m.i2w_off.resize((size_t)cnt + 1);
                for (int k = 0; k <= cnt; ++k) m.i2w_off[(size_t)k] = d.u32(4 + 4 * (size_t)k);
                for (int k = 0; k < cnt; ++k) if (m.i2w_off[(size_t)k + 1] < m.i2w_off[(size_t)k]) return fail(m, "bad [i2w] string array");
                const size_t base = 4 + 4 * ((size_t)cnt + 1);
                if (!d.ok(base, m.i2w_off[(size_t)cnt])) return fail(m, "bad [i2w] string array");
                m.i2w_data.assign(d.p + base, d.p + base + m.i2w_off[(size_t)cnt]);
                m.has_i2w = true;
            } else if (vals[i] == 76 /* PARAM_TOKENID_MIN */ && i + 1 < vals.size()) m.min_token_id = vals[++i];
            else if (vals[i] == 77 /* PARAM_TOKENID_MAX */ && i + 1 < vals.size()) m.max_token_id = vals[++i];
        }
    }
    if (!m.has_seg) {
        m.kind = KIND_WP;
        if (!m.has_wbd || m.wbd.table_len() == 0) {
            if (!m.has_i2w) return fail(m, "model has neither a [wbd] lexer, a [pos-dict] dictionary nor an [i2w] string array");
            m.kind = KIND_I2W;
        }
    }
    return true;
}

} // namespace bfa

namespace bfa {
uint32_t bpe_unk_prio(const Model &m, int unk)
{
    if (m.kind != KIND_BPE_MERGES) {
        // ids order the arcs (..._bpe_t.h:238-255); an entry's priority is 2 * id + 1
        if (unk < 0) return 0u;
        if (unk >= (1 << 20)) return 1u << 21;
        return 2u * (uint32_t)unk + 1u;
    }
    // (rank 0.0f, id unk) among the places: places before it have a bigger rank, or rank 0.0f and a smaller id
    auto before = [&](size_t place) {
        float r; const uint32_t b = m.bpe_place_rank[place]; memcpy(&r, &b, 4);
        if (r > 0.0f) return true;
        if (!(r == 0.0f)) return false;
        return m.bpe_place_id[place] < unk;
    };
    size_t lo = 0, hi = m.bpe_place_id.size();
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (before(mid)) lo = mid + 1; else hi = mid; }
    if (lo < m.bpe_place_id.size()) {
        float r; const uint32_t b = m.bpe_place_rank[lo]; memcpy(&r, &b, 4);
        if (r == 0.0f && m.bpe_place_id[lo] == unk) return 2u * (uint32_t)lo + 1u;      // compares equal to that entry: ordered by start among its arcs
    }
    return 2u * (uint32_t)lo;
}
} // namespace bfa
