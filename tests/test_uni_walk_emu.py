"""CPU: the two-stage Unigram path -- the trie walks of all start positions by wave (blingfire_amd/csrc/bf_uni_walk_body.h, in the 64-fibre
wave simulator) and the relaxations from their arc records by lane (bf_seg.h UniArcLane, driven sequentially) -- against the oracle:
adversarial input, fuzz, long unknown runs, the multilingual corpora of configs 4 / 5, a pool that is too small (documents flagged and
redone by the lane-per-document restatement, as on the device)."""
import ctypes

import numpy as np
import pytest

import bfutil
import blingfire_amd as bf

MODELS = ["xlm_roberta_base.bin", "laser500k.bin", "laser100k.bin", "xlnet.bin", "xlnet_nonorm.bin", "laser50k.bin"]


@pytest.fixture(scope="module")
def ht():
    L = ctypes.CDLL(bfutil.HOSTTEST_LIB)
    L.bft_load.restype = ctypes.c_void_p
    L.bft_load.argtypes = [ctypes.c_char_p]
    L.bft_free.argtypes = [ctypes.c_void_p]
    L.bft_emu_uni_walk_batch.restype = ctypes.c_long
    L.bft_emu_uni_walk_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return L


def run(ht, h, text, off, mx, unk, nw=2, pool=1 << 22, rows=32):
    nd = len(off) - 1
    cap = 2 * len(text) + 2 * nd + 16
    ids = np.full(cap, -9, dtype=np.int32)
    ido = np.zeros(nd + 1, dtype=np.int64)
    fl = np.zeros(nd + 1, dtype=np.int32)
    st = np.zeros(16, dtype=np.uint64)
    r = ht.bft_emu_uni_walk_batch(h, text.ctypes.data, off.ctypes.data, nd, mx, unk, nw, pool, rows, ids.ctypes.data, cap, ido.ctypes.data, fl.ctypes.data, st.ctypes.data)
    return r, ids[:max(r, 0)], ido, fl[:nd], st


def check(ht, model, docs, confs, nw=2, pool=1 << 22, max_back=0):
    mp = bfutil.model_path(model)
    h = ht.bft_load(mp.encode())
    ora = bfutil.oracle()
    ho = ora.load(mp)
    text, off = docs if isinstance(docs, tuple) else bf.pack_docs(docs)
    for (mx, unk, rows) in confs:
        r, ids, ido, fl, st = run(ht, h, text, off, mx, unk, nw, pool, rows)
        if r == -1 and rows == 16:
            continue                                    # entries longer than 16 symbols: the 32-row instance is this model's
        gids, goff = ora.batch(ho, text, off, mx, unk)
        assert r >= 0, (model, r)
        if not (np.array_equal(ido, goff) and np.array_equal(ids, gids)):
            for d in range(len(off) - 1):
                a, b = ids[ido[d]:ido[d + 1]], gids[goff[d]:goff[d + 1]]
                assert np.array_equal(a, b), (model, (mx, unk, rows), d, int(fl[d]), bytes(text[off[d]:off[d + 1]])[:80], a.tolist()[:24], b.tolist()[:24])
        if max_back is not None:
            assert int(fl.sum()) <= max_back, (model, "documents flagged", int(fl.sum()))
    ora.free(ho)
    ht.bft_free(h)


@pytest.mark.parametrize("model", MODELS)
def test_adversarial_and_fuzz(ht, model):
    if not bfutil.have_model(model):
        pytest.skip(model)
    docs = list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(300, seed=29) + [b"", b"a"]
    check(ht, model, docs, [(2048, 3, 32), (3, 0, 16), (1, 1, 216), (2048, -5, 408)], max_back=None)
    check(ht, model, docs[:120], [(2048, 3, 302)], max_back=None)          # a stage of two entries per start: many documents are flagged and redone


@pytest.mark.parametrize("model", ["xlm_roberta_base.bin", "laser500k.bin"])
def test_long_unknown_runs_and_long_documents(ht, model):
    if not bfutil.have_model(model):
        pytest.skip(model)
    docs = ["\U000F0000".encode() * k for k in (1, 2, 63, 64, 65, 4094, 4095, 4096, 4097, 9000)]
    docs += [("a" * k + " \U000F0000" * (k % 5) + " the end").encode("utf-8") for k in range(1, 40)]
    docs += [("word " * 3000).encode(), ("中文" * 2000).encode()]
    check(ht, model, docs, [(1 << 20, 3, 32), (5, 3, 216), (1 << 20, 3, 408)], nw=3, max_back=None)


@pytest.mark.parametrize("wl", ["config4", "config5"])
def test_corpora(ht, wl):
    w = bfutil.WORKLOADS[wl]
    if not bfutil.have_model(w["model"]):
        pytest.skip(w["model"])
    check(ht, w["model"], bfutil.gen_workload(wl, 120), [(w["max_ids"], w["unk"], 216), (w["max_ids"], w["unk"], 32)], nw=4)
    check(ht, w["model"], bfutil.gen_workload(wl, 120), [(w["max_ids"], w["unk"], 408)], nw=4, max_back=None)


def test_a_pool_that_is_too_small_flags_documents(ht):
    model = "xlm_roberta_base.bin"
    if not bfutil.have_model(model):
        pytest.skip(model)
    text, off = bfutil.gen_workload("config4", 60)
    check(ht, model, (text, off), [(1024, 3, 16)], nw=2, pool=3 * 8192, max_back=None)
    h = ht.bft_load(bfutil.model_path(model).encode())
    r, ids, ido, fl, st = run(ht, h, text, off, 1024, 3, 2, 3 * 8192, 16)
    assert 0 < int(fl.sum()) < 60            # some documents fit, some were flagged
    ht.bft_free(h)
