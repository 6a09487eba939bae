"""NormalizeSpaces (reference tokdll:629-679) and TextToHashes (tokdll:683-815): model-free entry points the reference's
any_test resolves eagerly (any_test.cpp:82-115).  Oracle restatements pinned to the compiled reference; the GPU tests call the
product's single and batch forms."""
import ctypes

import numpy as np
import pytest

import bfutil

DOCS = list(bfutil.ADVERSARIAL) + [b"  a  b   c  ", b"_ a _  b_ ", "▁▁a ▁ b".encode(), b"   ", b" ", b"a", b"This is ok .", b"a  b", b" x ", b"x" * 700 + b" y"]
USPACES = (0x2581, 0x20, ord("_"), 0x3000, 0x1F600, 0xD800)


def _ns(fn, b, mx, usp):
    o = ctypes.create_string_buffer(b"\x7f" * (4 * len(b) + 16))
    r = fn(b, len(b), o, mx, usp)
    return r, (o.raw[:min(r + 1, mx)] if r >= 0 else b"")


def _th(fn, b, mx, ng, bucket):
    a = (ctypes.c_int32 * (8 * len(b) + 64))()
    r = fn(b, len(b), a, mx, ng, bucket)
    return r, (list(a[:r]) if 0 < r <= mx else [])


def _oracle_fns():
    L = bfutil.oracle().lib
    f, g = L.bfo_normalize_spaces, L.bfo_text_to_hashes
    f.restype = g.restype = ctypes.c_int
    f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    g.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return f, g


def _set(L):
    L.NormalizeSpaces.restype = L.TextToHashes.restype = ctypes.c_int
    L.NormalizeSpaces.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.TextToHashes.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return L.NormalizeSpaces, L.TextToHashes


def test_known_answers():
    f, g = _oracle_fns()
    assert _ns(f, "  Hello   wörld \t\n x ".encode(), 100, 0x20) == (14, "Hello wörld x\x00".encode())
    r, h = _th(g, b"This is ok .", 64, 2, 2000000)       # tokdll:777-779: 4 unigrams + 4 bigrams ("." + EOS included)
    assert r == 8 and len(set(h[:4])) == 4 and all(0 <= x < 2000000 for x in h[4:])


@pytest.mark.skipif(not bfutil.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_vs_live_reference():
    f, g = _oracle_fns()
    rf, rg = _set(bfutil.reference().lib)
    docs = DOCS + bfutil.fuzz_docs(1500, seed=31)
    for b in docs:
        for usp in USPACES:
            for mx in (4 * len(b) + 8, 3, len(b)):
                assert _ns(f, b, mx, usp) == _ns(rf, b, mx, usp), ("normalize", b[:40], usp, mx)
        for ng in (1, 2, 3):
            for bucket in (2000000, 7, -3):
                for mx in (8 * len(b) + 64, 4):
                    assert _th(g, b, mx, ng, bucket) == _th(rg, b, mx, ng, bucket), ("hashes", b[:40], ng, bucket, mx)


@pytest.mark.gpu
def test_gpu_normalize_spaces_and_hashes():
    import blingfire_amd as bf
    f, g = _oracle_fns()
    pf, pg = _set(bf.lib())
    docs = DOCS + bfutil.fuzz_docs(400, seed=37)
    for b in docs:
        for usp in USPACES:
            for mx in (4 * len(b) + 8, 3, len(b)):
                assert _ns(pf, b, mx, usp) == _ns(f, b, mx, usp), ("normalize", b[:40], usp, mx)
        for ng, bucket, mx in ((1, 2000000, 8 * len(b) + 64), (2, 2000000, 8 * len(b) + 64), (3, 7, 8 * len(b) + 64), (2, -3, 8 * len(b) + 64), (2, 2000000, 4)):
            assert _th(pg, b, mx, ng, bucket) == _th(g, b, mx, ng, bucket), ("hashes", b[:40], ng, bucket, mx)
    # batch forms = the single calls, concatenated
    text, off = bf.pack_docs(docs)
    L = bf.lib()
    L.NormalizeSpacesBatch.restype = L.TextToHashesBatch.restype = ctypes.c_int64
    L.NormalizeSpacesBatch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]
    L.TextToHashesBatch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    nd = len(docs)
    out = np.empty(4 * len(text) + 64, dtype=np.uint8); t_off = np.zeros(nd + 1, dtype=np.int64)
    n = L.NormalizeSpacesBatch(text.ctypes.data, off.ctypes.data, nd, out.ctypes.data, len(out), t_off.ctypes.data, 0x2581)
    assert n == t_off[-1] >= 0
    hs = np.empty(8 * len(text) + 64 * nd, dtype=np.int32); h_off = np.zeros(nd + 1, dtype=np.int64)
    m = L.TextToHashesBatch(text.ctypes.data, off.ctypes.data, nd, hs.ctypes.data, len(hs), h_off.ctypes.data, 2, 2000000)
    assert m == h_off[-1] > 0
    for d, b in enumerate(docs):
        r, o = _ns(f, b, 4 * len(b) + 8, 0x2581)
        assert out[t_off[d]:t_off[d + 1]].tobytes() == (o[:r] if r > 0 else b""), ("normalize batch", d, b[:40])
        r, h = _th(g, b, 8 * len(b) + 64, 2, 2000000)
        assert list(hs[h_off[d]:h_off[d + 1]]) == h, ("hashes batch", d, b[:40])
