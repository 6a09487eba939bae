"""Committed fixtures of the non-TextToIds entry points (tests/golden/api/fixtures.json, generated from the compiled reference
by tests/golden/make_golden.py): they pin the oracle where /root/reference is absent (the GPU box) and, on a GPU, the product."""
import ctypes
import json
import os

import pytest

import bfutil

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "api", "fixtures.json")))


def _impls(gpu):
    """(offsets fn, text fn factory, ids_to_text fn, loader, freer) of the oracle or of the product"""
    if not gpu:
        ora = bfutil.oracle()
        def offsets(h, b, mx, unk):
            return ora.with_offsets(h, b, mx, unk, "bfo_text_to_ids_with_offsets")
        def text(kind, h, b, mx):
            f = getattr(ora.lib, "bfo_text_to_%s_with_offsets" % kind)
            f.restype = ctypes.c_int
            f.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
            o = ctypes.create_string_buffer(max(mx, 1) + 4); s = (ctypes.c_int32 * max(mx, 1))(); e = (ctypes.c_int32 * max(mx, 1))()
            return f(ctypes.c_void_p(h), b, len(b), o, s, e, mx), o, s, e
        def i2t(h, ids, skip):
            f = ora.lib.bfo_ids_to_text
            f.restype = ctypes.c_int
            f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
            arr = (ctypes.c_int32 * len(ids))(*ids); o = ctypes.create_string_buffer(4096)
            return f(ctypes.c_void_p(h), arr, len(ids), o, 4096, skip), o
        return offsets, text, i2t, ora.load, ora.free, {"words": "wbd.bin", "sentences": "sbd.bin"}
    import blingfire_amd as bf
    L = bf.lib()
    def offsets(h, b, mx, unk):
        i = (ctypes.c_int32 * mx)(); s = (ctypes.c_int32 * mx)(); e = (ctypes.c_int32 * mx)()
        c = max(L.TextToIdsWithOffsets(ctypes.c_void_p(h), b, len(b), i, s, e, mx, unk), 0)
        return c, list(i)[:c], list(s)[:c], list(e)[:c]
    def text(kind, h, b, mx):
        f = L.TextToWordsWithOffsetsWithModel if kind == "words" else L.TextToSentencesWithOffsetsWithModel
        o = ctypes.create_string_buffer(max(mx, 1) + 4); s = (ctypes.c_int32 * max(mx, 1))(); e = (ctypes.c_int32 * max(mx, 1))()
        return f(b, len(b), o, s, e, mx, ctypes.c_void_p(h) if h else None), o, s, e
    def i2t(h, ids, skip):
        arr = (ctypes.c_int32 * len(ids))(*ids); o = ctypes.create_string_buffer(4096)
        return L.IdsToText(ctypes.c_void_p(h), arr, len(ids), o, 4096, bool(skip)), o
    return offsets, text, i2t, bf.load_model, bf.free_model, {"words": None, "sentences": None}


def _check(gpu):
    offsets, text, i2t, load, free, default = _impls(gpu)
    handles = {}
    def H(model):
        if model not in handles:
            handles[model] = load(bfutil.model_path(model))
        return handles[model]
    try:
        for r in FIX["offsets"]:
            b = bytes.fromhex(r["hex"])
            c, ids, st, en = offsets(H(r["model"]), b, r["max"], r["unk"])
            st_want, en_want = list(r["starts"]), list(r["ends"])
            if gpu:   # documented deviation: a token made of the dummy prefix alone reports end -1 (the reference reads the byte before the string)
                en_want = [(-1 if s == -1 and e <= 0 else e) for s, e in zip(st_want, en_want)]
                en = [(-1 if s == -1 and e <= 0 else e) for s, e in zip(st, en)]
            assert (c, ids, st, en) == (r["count"], r["ids"], st_want, en_want), ("offsets", r["model"], b[:40])
        for kind in ("words", "sentences"):
            for r in FIX[kind]:
                b = bytes.fromhex(r["hex"]); mx = 4 * len(b) + 8
                model = r["model"] if r["model"] else default[kind]      # the oracle has no built-in model: it gets the same file explicitly
                h = H(model) if model else None
                ret, o, s, e = text(kind, h, b, mx)
                k = len(r["starts"])
                assert ret == r["ret"] and (not (0 < ret <= mx) or o.raw[:ret].hex() == r["out_hex"]) and list(s[:k]) == r["starts"] and list(e[:k]) == r["ends"], (kind, r["model"], b[:40])
        for r in FIX["ids_to_text"]:
            ret, o = i2t(H(r["model"]), r["ids"], r["skip"])
            assert ret == r["ret"] and o.raw[:max(ret, 0)].hex() == r["out_hex"], ("ids_to_text", r["model"], r["ids"][:6], r["skip"])
    finally:
        for h in handles.values():
            free(h)


def test_oracle_matches_api_fixtures():
    _check(False)


@pytest.mark.gpu
def test_gpu_matches_api_fixtures():
    _check(True)
