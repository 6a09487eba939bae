// wave_emu.h -- TEST-ONLY: a 64-lane wavefront simulator for kernels written against namespace wv (bf_wave.h).
//
// Every lane is a fibre (its own stack, switched by a dozen instructions of x86-64 assembly); a wv:: collective is a rendezvous
// of the 64 fibres of one wave: the last lane to arrive publishes everybody's operand, then each lane computes its own result.
// Lanes run one after the other between collectives, so a missing wv::sync() between an LDS write and another lane's read shows
// up as a stale read for half of the lane pairs.  The simulator checks the rule of the house: all 64 lanes must arrive at the SAME
// call site (source line) -- a collective reached in divergent control flow aborts with a message instead of deadlocking.
// Several waves can be interleaved (one collective of each wave per sweep) so that the shared work counter is really contended.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <functional>

#if !defined(__x86_64__)
#error "wave_emu.h needs x86-64 (fibre switch in assembly)"
#endif

namespace wvemu {

extern "C" void wvemu_switch(void **save_sp, void *load_sp);
__asm__(".text\n.globl wvemu_switch\n.type wvemu_switch,@function\nwvemu_switch:\n"
        "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
        "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
        "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
        ".size wvemu_switch,.-wvemu_switch\n");

struct Wave;
struct Fiber {
    void *sp = nullptr; std::vector<uint8_t> stack; Wave *wave = nullptr; int lane = 0; bool done = false;
    unsigned long long wait_gen = 0; bool waiting = false;
};
struct Wave {
    Fiber f[64];
    int arrived = 0; unsigned long long gen = 0;
    uint64_t slot[64], snap[64]; long site[64];
    std::function<void()> body;
    int finished = 0;
};

static Fiber *g_cur = nullptr;
static void *g_sched_sp = nullptr;

static void fiber_entry()
{
    Fiber *me = g_cur;
    me->wave->body();
    me->done = true; me->wave->finished++;
    wvemu_switch(&me->sp, g_sched_sp);
    abort();
}

static void fiber_init(Fiber &f, Wave *w, int lane, size_t stack_bytes)
{
    f.wave = w; f.lane = lane; f.done = false; f.waiting = false;
    f.stack.assign(stack_bytes, 0);
    uintptr_t top = ((uintptr_t)f.stack.data() + stack_bytes) & ~(uintptr_t)15;
    void **sp = (void **)(top - 16);
    *sp = (void *)fiber_entry;          // `ret` lands here with rsp = top - 8 (what a call would leave)
    sp -= 6;                            // r15 r14 r13 r12 rbx rbp
    for (int i = 0; i < 6; ++i) sp[i] = nullptr;
    f.sp = sp;
}

static void yield_to_sched() { Fiber *me = g_cur; wvemu_switch(&me->sp, g_sched_sp); }

// the rendezvous: returns a pointer to the 64 published operands
static const uint64_t *collective(uint64_t v, long site)
{
    Fiber *me = g_cur; Wave *w = me->wave;
    if (w->finished) { fprintf(stderr, "wave_emu: lane %d reached a collective after %d lanes left the kernel\n", me->lane, w->finished); abort(); }
    w->slot[me->lane] = v; w->site[me->lane] = site;
    const unsigned long long my = w->gen;
    if (++w->arrived == 64) {
        for (int i = 1; i < 64; ++i) if (w->site[i] != w->site[0]) { fprintf(stderr, "wave_emu: collective reached in divergent control flow (lane %d at line %ld, lane 0 at line %ld)\n", i, w->site[i], w->site[0]); abort(); }
        memcpy(w->snap, w->slot, sizeof(w->snap));
        w->arrived = 0; w->gen++;
    } else {
        me->waiting = true; me->wait_gen = my;
        while (w->gen == my) yield_to_sched();
        me->waiting = false;
    }
    return w->snap;
}

// runs nwaves waves of `body` to completion, interleaved
static void run_waves(int nwaves, const std::function<void()> &body, size_t stack_bytes = 256 * 1024)
{
    std::vector<Wave *> waves;
    for (int i = 0; i < nwaves; ++i) { Wave *w = new Wave(); w->body = body; for (int l = 0; l < 64; ++l) fiber_init(w->f[l], w, l, stack_bytes); waves.push_back(w); }
    for (;;) {
        bool alive = false;
        for (Wave *w : waves) {
            if (w->finished == 64) continue;
            alive = true;
            bool progressed = false;
            for (int l = 0; l < 64; ++l) {
                Fiber &f = w->f[l];
                if (f.done || (f.waiting && w->gen == f.wait_gen)) continue;
                g_cur = &f;
                wvemu_switch(&g_sched_sp, f.sp);
                progressed = true;
            }
            if (!progressed) { fprintf(stderr, "wave_emu: deadlock (%d lanes left the kernel, %d wait in a collective)\n", w->finished, w->arrived); abort(); }
        }
        if (!alive) break;
    }
    for (Wave *w : waves) delete w;
}

} // namespace wvemu

// ---- the wv:: interface of bf_wave.h on top of the simulator
namespace wv {
// the call site of a collective = the source line of the call (default argument evaluated at the caller)
#define WV_SITE site
#define WV_SITE_ARG , long site = __builtin_LINE()
__attribute__((noinline)) static int lane() { return wvemu::g_cur->lane; }
__attribute__((noinline)) static unsigned long long ballot(bool b WV_SITE_ARG)
{
    const uint64_t *s = wvemu::collective(b ? 1 : 0, WV_SITE);
    unsigned long long m = 0; for (int i = 0; i < 64; ++i) if (s[i]) m |= 1ull << i;
    return m;
}
__attribute__((noinline)) static bool any(bool b WV_SITE_ARG)
{
    const uint64_t *s = wvemu::collective(b ? 1 : 0, WV_SITE);
    for (int i = 0; i < 64; ++i) if (s[i]) return true;
    return false;
}
__attribute__((noinline)) static void sync(long site = __builtin_LINE()) { (void)wvemu::collective(0, WV_SITE); }

template <class T> static inline uint64_t to_u64(T v) { uint64_t u = 0; memcpy(&u, &v, sizeof(T)); return u; }
template <class T> static inline T from_u64(uint64_t u) { T v; memcpy(&v, &u, sizeof(T)); return v; }

// src may differ per lane; the operand travels with the source lane index so that one rendezvous serves both
template <class T> __attribute__((noinline)) static T shfl(T v, int src WV_SITE_ARG)
{
    static_assert(sizeof(T) <= 8, "shfl operand");
    const uint64_t *s = wvemu::collective(to_u64(v), WV_SITE);
    return from_u64<T>(s[src & 63]);
}
template <class T> __attribute__((noinline)) static T bcast(T v, int src WV_SITE_ARG)
{
    const uint64_t *s = wvemu::collective(to_u64(v), WV_SITE);
    const uint64_t r = s[src & 63];                     // taken before the next rendezvous reuses the snapshot
    // src must be wave-uniform: checked with a second rendezvous
    const uint64_t *q = wvemu::collective((uint64_t)(unsigned)src, WV_SITE);
    for (int i = 1; i < 64; ++i) if (q[i] != q[0]) { fprintf(stderr, "wave_emu: bcast with a non-uniform source lane\n"); abort(); }
    return from_u64<T>(r);
}
template <class T> __attribute__((noinline)) static T shfl_up(T v, int delta WV_SITE_ARG)
{
    const int me = wvemu::g_cur->lane;
    const uint64_t *s = wvemu::collective(to_u64(v), WV_SITE);
    return from_u64<T>(me >= delta ? s[me - delta] : s[me]);
}
template <class T> __attribute__((noinline)) static T shfl_down(T v, int delta WV_SITE_ARG)
{
    const int me = wvemu::g_cur->lane;
    const uint64_t *s = wvemu::collective(to_u64(v), WV_SITE);
    return from_u64<T>(me + delta < 64 ? s[me + delta] : s[me]);
}
__attribute__((noinline)) static int incl_scan(int v WV_SITE_ARG)
{
    const int me = wvemu::g_cur->lane;
    const uint64_t *s = wvemu::collective((uint64_t)(int64_t)v, WV_SITE);
    int64_t a = 0; for (int i = 0; i <= me; ++i) a += (int64_t)s[i];
    return (int)a;
}
// the smallest value any lane holds
__attribute__((noinline)) static uint32_t min_all(uint32_t v WV_SITE_ARG)
{
    const uint64_t *s = wvemu::collective((uint64_t)v, WV_SITE);
    uint32_t m = 0xFFFFFFFFu; for (int i = 0; i < 64; ++i) if ((uint32_t)s[i] < m) m = (uint32_t)s[i];
    return m;
}
// a value every lane holds alike (the device moves it to a scalar register): checked
template <class T> __attribute__((noinline)) static T uni(T v WV_SITE_ARG)
{
    const uint64_t *s = wvemu::collective(to_u64(v), WV_SITE);
    for (int i = 1; i < 64; ++i) if (s[i] != s[0]) { fprintf(stderr, "wave_emu: uni() of a value that differs between lanes (line %ld)\n", site); abort(); }
    return v;
}
static inline int own(int v) { return v; }
static inline void arrived(int32_t &, int32_t &, int32_t &, int32_t &) {}      // device: the four loaded values are in their registers from here on
static inline uint32_t mbcnt(unsigned long long m) { return (uint32_t)__builtin_popcountll(m & ((1ull << wvemu::g_cur->lane) - 1ull)); }
static inline unsigned long long atomic_add(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
static inline void atomic_or(int *p, int v) { *p |= v; }
static inline void atomic_add_i32(int32_t *p, int32_t v) { *p += v; }
static inline uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
    const unsigned long long w = ((unsigned long long)hi << 32) | lo; uint32_t r = 0;
    for (int k = 0; k < 4; ++k) { const uint32_t s = (sel >> (8 * k)) & 0xFFu; r |= (s < 8u ? (uint32_t)((w >> (8 * s)) & 0xFFu) : 0u) << (8 * k); }
    return r;
}
static inline uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((unsigned long long)hi) << 32) | lo) >> (sh & 31u)); }
static inline void lds_or(uint32_t *p, uint32_t v) { *p |= v; }
static inline unsigned long long clock() { return 0; }
static inline uint32_t load_l2(const uint32_t *p) { return *p; }
static inline void atomic_or_u32(uint32_t *p, uint32_t v) { *p |= v; }
static inline void atomic_max_u32(uint32_t *p, uint32_t v) { if (v > *p) *p = v; }
static inline void lds_add(uint32_t *p, uint32_t v) { *p += v; }
} // namespace wv
