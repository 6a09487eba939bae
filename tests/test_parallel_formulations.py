"""The restatements the GPU kernels rely on, checked on the CPU against the sequential reference semantics.

* BPE (k_bpe_fused): the arc list splits at CUTS (a start position no earlier arc reaches) into segments that
  can be sorted and applied independently and in position order; inside a segment the surviving tokens are exactly the arcs
  [s, e] with s and e + 1 non-interior and s + 1 .. e interior.  Checked on the arc lists the host emulation collects
  (tests/hosttest), against the emulation's own ids (which tests/test_tables_and_emu.py ties to the oracle).
* NormalizeSpaces (k_normsp): a white-space character is written iff the character before it exists, is not white space and is not
  uSpace.  * IdsToText (k_i2t_*): the leading-space rule only touches the tokens up to the first "solid" one.
"""
import ctypes
import random
import struct

import pytest

import bfutil

BPE_MODELS = [m for m in ("gpt2.bin", "roberta.bin", "bpe_example.bin", "bpe_example2.bin") if bfutil.have_model(m)]


def _f32(bits):
    return struct.unpack("<f", struct.pack("<I", bits & 0xFFFFFFFF))[0]


def _key(a, merges):
    s, e, i, r = a
    return ((-_f32(r), i, s) if merges else (i, s))          # ..._with_merges_t.h:242-262 / ..._bpe_t.h:238-255


def _apply(arcs, lo, hi, L):
    """the reference's apply loop (..._bpe_t.h:274-296) on positions lo..hi: interior marks + the arc applied last from each start"""
    inter = set(); owner = {}
    for s, e, i, r in arcs:
        if s not in inter and (e + 1 == L or (e + 1) not in inter):
            owner[s] = (e, i)
            inter.update(range(s + 1, e + 1))
    return inter, owner


def _full(arcs, L, merges, unk):
    inter, owner = _apply(sorted(arcs, key=lambda a: _key(a, merges)), 0, L - 1, L)
    out = []; p = 0
    while p < L:                                              # ..._bpe_t.h:299-313
        e, i = owner.get(p, (0, unk))
        if e < p:
            return None                                       # the reference would walk backwards
        out.append(i); p = e + 1
    return out


def _segments(arcs, L, merges, unk):
    out = []; k = 0; n = len(arcs)
    while k < n:
        j = k; maxend = -1
        while j < n and (j == k or arcs[j][0] <= maxend):     # a cut: the next arc starts beyond everything seen so far
            maxend = max(maxend, arcs[j][1]); j += 1
        seg = arcs[k:j]
        if len(seg) == 1:
            out.append(seg[0][2])
        else:
            inter, _ = _apply(sorted(seg, key=lambda a: _key(a, merges)), seg[0][0], maxend, L)
            bounds = [q for q in range(seg[0][0], maxend + 1) if q not in inter]
            toks = {}
            for s, e, i, r in seg:                            # the pattern rule of the lane-local solve
                if s not in inter and (e + 1) not in inter and all(q in inter for q in range(s + 1, e + 1)):
                    toks[s] = i
            if sorted(toks) != bounds:
                return None                                   # a token start without an applied arc: the kernel hands the document to the full path
            out.extend(toks[q] for q in bounds)
        k = j
    return out


@pytest.mark.parametrize("model", BPE_MODELS)
def test_bpe_segments_equal_the_global_sort(model):
    L_ = ctypes.CDLL(bfutil.HOSTTEST_LIB)
    L_.bft_load.restype = ctypes.c_void_p
    L_.bft_load.argtypes = [ctypes.c_char_p]
    f = L_.bft_emu_bpe_arcs
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    ora = bfutil.oracle()
    ho = ora.load(bfutil.model_path(model))
    ora.lib.bfo_model_id_offset.argtypes = [ctypes.c_void_p]
    id_offset = ora.lib.bfo_model_id_offset(ho)
    ora.free(ho)
    h = L_.bft_load(bfutil.model_path(model).encode())
    docs = list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(1200, seed=53)
    checked = complex_segments = 0
    for b in docs:
        for unk in (0, 262):
            ids = (ctypes.c_int32 * 8192)(); buf = (ctypes.c_int32 * 200000)(); ni = ctypes.c_int(0)
            r = f(h, b, len(b), ids, 8192, unk, buf, 200000, ctypes.byref(ni))
            if ni.value < 3:
                continue
            L, na, kind = buf[0], buf[1], buf[2]
            arcs = [tuple(buf[3 + 4 * k: 7 + 4 * k]) for k in range(na)]
            assert all(arcs[k][0] <= arcs[k + 1][0] for k in range(na - 1))           # collected in ascending start order
            merges = kind == 4
            full = _full(arcs, L, merges, unk)
            seg = _segments(arcs, L, merges, unk)
            if r == -2:
                assert full is None
                continue
            want = [i - id_offset for i in ids[:r]]
            assert full == want, (model, b[:40], unk)
            if seg is not None:
                assert seg == want, (model, b[:40], unk)
                checked += 1
            complex_segments += 1
    assert checked > 1000


def _is_ws(c):
    return c <= 0x20 or c == 0xa0 or 0x2000 <= c <= 0x200f or c in (0x202f, 0x205f, 0x2060, 0x2420, 0x2424, 0x3000, 0xfeff)


def test_normalize_spaces_local_rule():
    L = bfutil.oracle().lib
    f = L.bfo_normalize_spaces
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    rng = random.Random(3)
    alphabet = [ord("a"), ord("_"), 0x20, 0x09, 0x2581, 0x3000, 0xe9, 0x1F600]
    for _ in range(4000):
        cps = [rng.choice(alphabet) for _ in range(rng.randrange(1, 24))]
        usp = rng.choice([0x2581, 0x20, ord("_"), 0x3000])
        out = []
        for k, c in enumerate(cps):                           # the kernel's rule: no state but the previous INPUT character
            if not _is_ws(c):
                out.append(c)
            elif k > 0 and not _is_ws(cps[k - 1]) and cps[k - 1] != usp:
                out.append(usp)
        if len(out) > 1 and out[-1] == usp:
            out.pop()
        b = "".join(map(chr, cps)).encode()
        o = ctypes.create_string_buffer(256)
        r = f(b, len(b), o, 256, usp)
        assert r == len("".join(map(chr, out)).encode()) and o.raw[:r] == "".join(map(chr, out)).encode(), (cps, usp)


def test_ids_to_text_leading_space_rule():
    ora = bfutil.oracle()
    f = ora.lib.bfo_ids_to_text
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    h = ora.load(bfutil.model_path("gpt2.i2w"))
    rng = random.Random(4)
    def tok(i):
        o = ctypes.create_string_buffer(64); a = (ctypes.c_int32 * 2)(1, i)      # behind a non-space token nothing is stripped
        r = f(ctypes.c_void_p(h), a, 2, o, 64, 0)
        one = ctypes.create_string_buffer(64); f(ctypes.c_void_p(h), (ctypes.c_int32 * 1)(1), 1, one, 64, 0)
        return o.raw[len(one.value):r - 1]
    pool = [220, 220, 1, 2, 15496, 2159, 995, 318, 262]      # 220 = " " in gpt2
    texts = {i: tok(i) for i in set(pool)}
    for _ in range(3000):
        ids = [rng.choice(pool) for _ in range(rng.randrange(1, 9))]
        toks = [texts[i] for i in ids]
        solid = [k for k, t in enumerate(toks) if len(t) > 0 and t != b" "]
        if solid:                                             # the kernel's rule: everything before the first solid token vanishes, it loses one leading space
            p = solid[0]
            want = (toks[p][1:] if toks[p][:1] == b" " else toks[p]) + b"".join(toks[p + 1:])
        else:
            want = b""
        o = ctypes.create_string_buffer(256)
        r = f(ctypes.c_void_p(h), (ctypes.c_int32 * len(ids))(*ids), len(ids), o, 256, 0)
        assert o.raw[:r - 1] == want, (ids, want, o.raw[:r])
    ora.free(h)
