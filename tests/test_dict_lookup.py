"""Dictionary key -> info lookup (reference FADictInterpreter_t<int>::GetInfo,
blingfireclient.library/inc/FADictInterpreter_t.h:334-390).

CPU: the oracle restatement (bfo_dict_get_info*) is pinned against the reference's own interpreter (oracle/_ref/libdictref.so =
oracle/ref_dict_glue.cpp + the reference sources, built by oracle/Makefile) and against a known answer of the reference's docs
(ldbsrc/gpt2/README.TXT:40-47: `pedia` -> id 50236); the device program (bf_seg.h dict_info_id) runs on the host against the oracle.
GPU: DictGetInfoBatch through the C-ABI against the oracle."""
import ctypes
import gzip
import os
import random

import numpy as np
import pytest

import bfutil

MODELS = ["gpt2.bin", "roberta.bin", "xlm_roberta_base.bin", "laser500k.bin", "xlnet.bin", "bpe_example.bin", "uri100k.bin", "laser100k.bin"]
DICTREF = os.path.join(bfutil.ROOT, "oracle", "_ref", "libdictref.so")


def keys_for(model, n_random=3000, seed=1, negative=True):
    ws = []
    name = "xlmr" if "xlm" in model else "laser500k" if "laser500" in model else None
    if name:
        for line in gzip.open(os.path.join(bfutil.ROOT, "tests", "data", "pieces_%s.tsv.gz" % name), "rb"):
            b, p = line.rstrip(b"\n").split(b"\t", 1)
            if b != b"charmap":
                ws.append(p.decode())
        ws = ws[::5]
    rnd = random.Random(seed)
    words = open(bfutil.WORDS_EN).read().split()
    ws += ["pedia", "the", "Ġthe", "▁the", "▁", "a", "", "zzzzqqq", "▁Hello", "hello world", "x" * 301, "x" * 300, "ª", "ﬁ", "㍿", "\U0001F600", "▁\U00010000"]
    ws += rnd.sample(words, 1500) + ["▁" + w for w in rnd.sample(words, 1500)] + ["Ġ" + w for w in rnd.sample(words, 300)]
    ws += ["".join(rnd.choice("abcdefghijklmnop ▁Ġ") for _ in range(rnd.randint(1, 8))) for _ in range(n_random)]
    keys = [[ord(c) for c in w] for w in ws]
    keys += [[0x110000, 97], [97, 0x7FFFFFFF], list(range(97, 97 + 26))]      # symbols outside the code point range never match
    if negative:
        keys += [[-5], [97, -1]]      # a negative symbol makes the reference itself read out of bounds (it segfaults): oracle / product answer -1
    return keys


def oracle_lookup(ora_lib, ho, key, max_out=8):
    arr = (ctypes.c_int32 * max(len(key), 1))(*key)
    out = (ctypes.c_int32 * max_out)(*([-7] * max_out))
    r = ora_lib.bfo_dict_get_info(ctypes.c_void_p(ho), arr, len(key), out, max_out)
    i = ora_lib.bfo_dict_get_info_id(ctypes.c_void_p(ho), arr, len(key))
    return r, i, list(out)


@pytest.fixture(scope="module")
def ora_lib():
    L = ctypes.CDLL(bfutil.ORACLE_LIB)
    L.bfo_load_model.restype = ctypes.c_void_p
    L.bfo_load_model.argtypes = [ctypes.c_char_p]
    L.bfo_free_model.argtypes = [ctypes.c_void_p]
    L.bfo_dict_get_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L.bfo_dict_get_info_id.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return L


def test_known_answer_gpt2_pedia(ora_lib):
    """ldbsrc/gpt2/README.TXT:40-47: the pos-dict entry `pedia` carries token id 50236"""
    ho = ora_lib.bfo_load_model(bfutil.model_path("gpt2.bin").encode())
    r, i, out = oracle_lookup(ora_lib, ho, [ord(c) for c in "pedia"])
    assert r >= 1 and out[0] == 50236
    assert oracle_lookup(ora_lib, ho, [ord(c) for c in "zzzzqqq"])[0] == -1
    ora_lib.bfo_free_model(ctypes.c_void_p(ho))


@pytest.mark.parametrize("model", MODELS)
def test_oracle_matches_the_reference_interpreter(ora_lib, model):
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)
    if not os.path.exists(DICTREF):
        pytest.skip("oracle/_ref/libdictref.so not built (needs the reference checkout)")
    R = ctypes.CDLL(DICTREF)
    R.refdict_load.restype = ctypes.c_void_p
    R.refdict_load.argtypes = [ctypes.c_char_p]
    R.refdict_free.argtypes = [ctypes.c_void_p]
    R.refdict_get_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    R.refdict_get_info_id.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    mp = bfutil.model_path(model).encode()
    hr, ho = R.refdict_load(mp), ora_lib.bfo_load_model(mp)
    assert hr and ho
    found = 0
    for key in keys_for(model, n_random=1500, negative=False):
        arr = (ctypes.c_int32 * max(len(key), 1))(*key)
        for max_out in (8, 1, 0):
            out = (ctypes.c_int32 * 8)(*([-7] * 8))
            rr = R.refdict_get_info(ctypes.c_void_p(hr), arr, len(key), out, max_out)
            r, i, got = oracle_lookup(ora_lib, ho, key) if max_out == 8 else (None, None, None)
            if max_out == 8:
                assert rr == r and list(out) == got, (model, key[:12])
                assert R.refdict_get_info_id(ctypes.c_void_p(hr), arr, len(key)) == i
                found += i != -1
            else:
                o2 = (ctypes.c_int32 * 8)(*([-7] * 8))
                assert ora_lib.bfo_dict_get_info(ctypes.c_void_p(ho), arr, len(key), o2, max_out) == rr and list(o2) == list(out)
    assert found > 100
    R.refdict_free(ctypes.c_void_p(hr))
    ora_lib.bfo_free_model(ctypes.c_void_p(ho))


@pytest.mark.parametrize("model", MODELS)
def test_device_program_on_host_matches_oracle(ora_lib, model):
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)
    H = ctypes.CDLL(bfutil.HOSTTEST_LIB)
    H.bft_load.restype = ctypes.c_void_p
    H.bft_load.argtypes = [ctypes.c_char_p]
    H.bft_free.argtypes = [ctypes.c_void_p]
    H.bft_emu_dict_get_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    mp = bfutil.model_path(model).encode()
    hh, ho = H.bft_load(mp), ora_lib.bfo_load_model(mp)
    for key in keys_for(model, seed=2):
        arr = (ctypes.c_int32 * max(len(key), 1))(*key)
        iid = ctypes.c_int32(0)
        vals = (ctypes.c_int32 * 8)(*([-7] * 8))
        r = H.bft_emu_dict_get_info(ctypes.c_void_p(hh), arr, len(key), ctypes.byref(iid), vals, 8)
        gr, gi, gout = oracle_lookup(ora_lib, ho, key)
        assert (r, iid.value) == (gr, gi) and list(vals) == gout, (model, key[:12])
    H.bft_free(ctypes.c_void_p(hh))
    ora_lib.bfo_free_model(ctypes.c_void_p(ho))


@pytest.mark.gpu
@pytest.mark.parametrize("model", MODELS)
def test_gpu_batch_lookup_matches_oracle(ora_lib, model):
    import blingfire_amd as bf
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)
    keys = keys_for(model, n_random=20000, seed=3)
    h = bf.load_model(bfutil.model_path(model))
    ho = ora_lib.bfo_load_model(bfutil.model_path(model).encode())
    try:
        ret, ids, vals, v_off = bf.dict_get_info_batch(h, keys)
        for k, key in enumerate(keys):
            gr, gi, gout = oracle_lookup(ora_lib, ho, key)
            assert (int(ret[k]), int(ids[k])) == (gr, gi), (model, key[:12])
            n = max(gr, 0)
            assert v_off[k + 1] - v_off[k] == n and list(vals[v_off[k]:v_off[k + 1]]) == gout[:n]
        assert bf.dict_get_info_batch(h, [])[3].tolist() == [0]
    finally:
        bf.free_model(h)
        ora_lib.bfo_free_model(ctypes.c_void_p(ho))


@pytest.mark.gpu
def test_lexer_model_has_no_dictionary():
    import blingfire_amd as bf
    h = bf.load_model(bfutil.model_path(bfutil.bert_model_name()))
    try:
        with pytest.raises(RuntimeError):
            bf.dict_get_info_batch(h, ["hello"])
    finally:
        bf.free_model(h)
