"""CPU: the one-wave-per-document BPE program (blingfire_amd/csrc/bf_bpe_seg_body.h -- the path of documents whose arcs exceed what the
lane-per-document kernels reserve) in the 64-fibre wave simulator against the oracle, all three BPE flavours: adversarial input, fuzz,
unknown symbols with UnkId values that are real ids, long runs of one character (the inputs that used to fail the whole batch), pool
exhaustion as a per-document failure."""
import ctypes
import random

import numpy as np
import pytest

import bfutil
import blingfire_amd as bf

MODELS = ["gpt2.bin", "roberta.bin", "bpe_example.bin", "bpe_example2.bin"]


@pytest.fixture(scope="module")
def ht():
    L = ctypes.CDLL(bfutil.HOSTTEST_LIB)
    L.bft_load.restype = ctypes.c_void_p
    L.bft_load.argtypes = [ctypes.c_char_p]
    L.bft_free.argtypes = [ctypes.c_void_p]
    L.bft_bpe_seg_ok.argtypes = [ctypes.c_void_p]
    L.bft_emu_bpe_seg_batch.restype = ctypes.c_long
    L.bft_emu_bpe_seg_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long,
                                        ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return L


def run(ht, h, text, off, mx, unk, nw=2, pool=64 << 20):
    nd = len(off) - 1
    cap = 2 * len(text) + 2 * nd + 16
    ids = np.full(cap, -9, dtype=np.int32)
    ido = np.zeros(nd + 1, dtype=np.int64)
    status = ctypes.c_int(0); used = ctypes.c_ulonglong(0)
    st = np.zeros(16, dtype=np.uint64)
    r = ht.bft_emu_bpe_seg_batch(h, text.ctypes.data, off.ctypes.data, nd, mx, unk, nw, pool, ids.ctypes.data, cap, ido.ctypes.data, None,
                                 ctypes.byref(status), ctypes.byref(used), st.ctypes.data)
    return r, ids[:max(r, 0)], ido, status.value, used.value, st


def check(ht, model, docs, confs, nw=2):
    mp = bfutil.model_path(model)
    h = ht.bft_load(mp.encode())
    assert ht.bft_bpe_seg_ok(h) == 1
    ora = bfutil.oracle()
    ho = ora.load(mp)
    text, off = docs if isinstance(docs, tuple) else bf.pack_docs(docs)
    for (mx, unk) in confs:
        r, ids, ido, status, _, _ = run(ht, h, text, off, mx, unk, nw)
        gids, goff = ora.batch(ho, text, off, mx, unk)
        assert r >= 0 and status == 0, (model, r, status)
        if not (np.array_equal(ido, goff) and np.array_equal(ids, gids)):
            for d in range(len(off) - 1):
                a, b = ids[ido[d]:ido[d + 1]], gids[goff[d]:goff[d + 1]]
                assert np.array_equal(a, b), (model, (mx, unk), d, bytes(text[off[d]:off[d + 1]])[:80], a.tolist()[:24], b.tolist()[:24])
    ora.free(ho)
    ht.bft_free(h)


@pytest.mark.parametrize("model", MODELS)
def test_adversarial_and_fuzz(ht, model):
    if not bfutil.have_model(model):
        pytest.skip(model)
    docs = list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(300, seed=23)
    check(ht, model, docs, [(2048, 0), (3, 0), (64, 3), (1, 1), (2048, 262), (2048, -7), (2048, 1 << 21)])


@pytest.mark.parametrize("model", MODELS)
def test_runs_of_one_character(ht, model):
    """what the lane kernels' 6 * L + 32 arcs cannot hold: the reference collects them all (..._bpe_t.h:143-144,197)"""
    if not bfutil.have_model(model):
        pytest.skip(model)
    docs = []
    for ch in "-.=_#*a ~":
        for k in (1, 2, 15, 16, 19, 63, 64, 65, 130, 500, 1500):
            docs.append((ch * k).encode())
            docs.append(("see " + ch * k + " end").encode())
    docs.append(("ab" * 700).encode())
    docs.append(("-" * 300 + " " + "=" * 300 + "\n" + "." * 300).encode())
    check(ht, model, docs, [(1 << 20, 0), (5, 3)])


def test_a_long_run_inside_a_batch(ht):
    if not bfutil.have_model("gpt2.bin"):
        pytest.skip("gpt2.bin")
    docs = bfutil.fuzz_docs(40, seed=3) + [b"-" * 10000] + bfutil.fuzz_docs(40, seed=4)
    check(ht, "gpt2.bin", docs, [(1 << 20, 0)], nw=3)


def test_config3_corpus(ht):
    text, off = bfutil.gen_workload("config3", 150)
    check(ht, "gpt2.bin", (text, off), [(2048, 0)])
    if bfutil.have_model("roberta.bin"):
        check(ht, "roberta.bin", (text, off), [(2048, 3)])


def test_pool_exhaustion_is_per_document(ht):
    """a pool that holds the small documents but not the big one: the big one gets count 0 and BF_STATUS_POOL, the others their ids;
    the claim counter says how large the pool has to be, and with that pool every document is answered"""
    mp = bfutil.model_path("gpt2.bin")
    h = ht.bft_load(mp.encode())
    ora = bfutil.oracle(); ho = ora.load(mp)
    docs = [b"hello world", b"=" * 4000, b"the quick brown fox", b"#" * 50]
    text, off = bf.pack_docs(docs)
    gids, goff = ora.batch(ho, text, off, 1 << 20, 0)
    r, ids, ido, status, used, _ = run(ht, h, text, off, 1 << 20, 0, nw=1, pool=64 << 10)
    assert r >= 0 and status == 64 and used > (64 << 10)
    cnt, gcnt = np.diff(ido), np.diff(goff)
    assert cnt[1] == 0 and cnt[0] == gcnt[0] and cnt[2] == gcnt[2] and cnt[3] == gcnt[3]
    r, ids, ido, status, used2, _ = run(ht, h, text, off, 1 << 20, 0, nw=1, pool=used)
    assert status == 0 and np.array_equal(ido, goff) and np.array_equal(ids, gids)
    ora.free(ho); ht.bft_free(h)


def _bpe_model_with_a_large_id(tmp_path):
    """a copy of bpe_example.bin whose first I2Info row holds the id 2^20 (the multi-map dump of [pos-dict] patched in place, the CRC dump written anew)"""
    import struct
    import ldbedit
    src = bfutil.model_path("bpe_example.bin")
    dumps = ldbedit.read_ldb(src)
    conf = ldbedit.decode_conf(dumps[0])
    mm = dict((p, v) for p, v in ldbedit.params(conf[ldbedit.FUNC_POS_DICT]) if v is not None)[25]       # PARAM_MULTI_MAP: the dump of the I2Info rows
    d = bytearray(dumps[mm])
    size_of_value, max_count = struct.unpack_from("<Ii", d, 0)
    assert size_of_value == 4 and max_count >= 1
    struct.pack_into("<i", d, 16 + 4, 1 << 20)                         # row 0: [count][id][score]
    dumps[mm] = bytes(d)
    return ldbedit.write_ldb(str(tmp_path / "bpe_large_id.bin"), dumps, conf)


def test_model_outside_the_key_format_is_marked(ht, tmp_path):
    """k_bpe_seg keeps an id in 20 bits of its keys: a model with a larger id is not `bpe_seg_ok` (bf_model.cpp), and the library refuses to load it (GPU test below)"""
    if not bfutil.have_model("bpe_example.bin"):
        pytest.skip("model not present")
    p = _bpe_model_with_a_large_id(tmp_path)
    h = ht.bft_load(p.encode())
    assert h and ht.bft_bpe_seg_ok(h) == 0
    ht.bft_free(h)
    ho = bfutil.oracle().load(p)                                          # (the oracle, which has no such limit, loads it)
    assert ho
    bfutil.oracle().free(ho)


@pytest.mark.gpu
def test_model_outside_the_key_format_is_refused_at_load(tmp_path):
    import blingfire_amd as bf
    if not bfutil.have_model("bpe_example.bin"):
        pytest.skip("model not present")
    p = _bpe_model_with_a_large_id(tmp_path)
    bf.lib().LoadModel.restype = ctypes.c_void_p
    h = bf.lib().LoadModel(p.encode())
    assert not h
    bf.lib().BfLastError.restype = ctypes.c_char_p
    assert b"BPE" in (bf.lib().BfLastError() or b"") or b"bpe" in (bf.lib().BfLastError() or b"")
