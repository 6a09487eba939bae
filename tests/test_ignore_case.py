"""ignore-case lexers and dictionaries, and moore-multi-dfa [wbd] sections: any reference .bin loads.

Reference: FALexTools_t.h:262-264 (every letter is folded by FAUtf32ToLower before GetDest), FADictInterpreter_t.h:203-205, 231-247 (keys
of an ignore-case dictionary are folded, then charmapped, in either direction), FAUtf32Utils.cpp:45-81 (the fold), FAWbdConfKeeper.cpp:
219-224 + FALexTools_t.h:134, 412-414 (a moore-multi-dfa [wbd] has no State2Ow map: the lexer answers -1, TextToIds 0 ids, TextToWords -1).

No shipped model sets these, so the test models are variants of shipped ones with a rewritten configuration dump (tests/ldbedit.py: the
automata stay byte for byte).  The compiled reference loads them and is the checker: oracle == reference (CPU tier, needs oracle/_ref),
host programs == oracle, and on the GPU the product == reference (oracle/_ref travels) or == oracle."""
import ctypes
import json
import os
import random

import numpy as np
import pytest

import bfutil
import ldbedit

GOLDEN = os.path.join(bfutil.ROOT, "tests", "golden", "tolower_pairs.json")
DICTREF = os.path.join(bfutil.ROOT, "oracle", "_ref", "libdictref.so")
LEXERS = ["bert_base_cased_tok.bin", "bert_base_tok.bin", "bert_chinese.bin", "wbd.bin"]       # with / without a charmap, unit form / general lexer
DICTS = [("gpt2.bin", None), ("xlm_roberta_base.bin", None), ("xlm_roberta_base.bin", 1), ("bpe_example.bin", 1)]      # (model, direction override): l2r without / with a charmap, r2l


def variant(tmp_path_factory, model, section, tag, **kw):
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)
    d = tmp_path_factory.getbasetemp() / ("%s_%s.bin" % (model.replace(".bin", ""), tag))
    if not d.exists():
        ldbedit.make_variant(bfutil.model_path(model), str(d), section, **kw)
    return str(d)


def case_docs(n, seed):
    """mixed-case text: English words in three cases, every code point the fold touches, and the usual adversarial / fuzz documents"""
    rnd = random.Random(seed)
    words = open(bfutil.WORDS_EN).read().split()
    pairs = json.load(open(GOLDEN))["pairs"]
    docs = list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(n // 2, seed=seed)
    for _ in range(n):
        ws = []
        for w in rnd.sample(words, rnd.randint(1, 12)):
            k = rnd.random()
            ws.append(w.upper() if k < 0.3 else w.capitalize() if k < 0.6 else "".join(c.upper() if rnd.random() < 0.5 else c for c in w) if k < 0.8 else w)
            if rnd.random() < 0.15:
                a, b = rnd.choice(pairs)
                ws.append("".join(chr(x) for x in (a, b, a)))
        docs.append(" ".join(ws).encode("utf-8", "ignore"))
    for i in range(0, len(pairs), 40):        # every folded code point once, in runs and alone
        docs.append(" ".join(chr(a) for a, _ in pairs[i:i + 40] if not 0xD800 <= a < 0xE000).encode())
        docs.append("".join(chr(a) for a, _ in pairs[i:i + 40] if not 0xD800 <= a < 0xE000).encode())
    return docs


@pytest.fixture(scope="module")
def ht():
    L = ctypes.CDLL(bfutil.HOSTTEST_LIB)
    L.bft_load.restype = ctypes.c_void_p
    L.bft_load.argtypes = [ctypes.c_char_p]
    L.bft_free.argtypes = [ctypes.c_void_p]
    L.bft_error.restype = ctypes.c_char_p
    L.bft_error.argtypes = [ctypes.c_void_p]
    L.bft_emu_text_to_ids.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.bft_wave_ok.argtypes = [ctypes.c_void_p]
    L.bft_lexer_void.argtypes = [ctypes.c_void_p]
    L.bft_emu_wave_batch.restype = ctypes.c_long
    L.bft_emu_wave_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
    L.bft_emu_dict_get_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return L


# ------------------------------------------------------------------------------------------------------------------------------
# the fold itself
# ------------------------------------------------------------------------------------------------------------------------------
def test_fold_of_product_and_oracle_match_the_fixture(ht):
    g = json.load(open(GOLDEN))
    m = dict((a, b) for a, b in g["pairs"])
    ora = ctypes.CDLL(bfutil.ORACLE_LIB)
    assert len(m) > 800 and m[0x41] == 0x61 and m[0xD7] == 0xF7 and m[0x17F] == 0x73      # the reference's quirks are part of the data
    for cp in list(range(-3, g["limit"] + 16)) + [0x10FFFF, 0x110000, 0x7FFFFFFF, -2147483648]:
        want = m.get(cp, cp)
        assert ht.bft_tolower(cp) == want and ora.bfo_tolower_sym(cp) == want, hex(cp)


@pytest.mark.skipif(not bfutil.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_fixture_is_the_reference_fold():
    R = ctypes.CDLL(bfutil.REF_LIB)
    f = R._ZN9BlingFire14FAUtf32ToLowerEi
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int]
    g = json.load(open(GOLDEN))
    m = dict((a, b) for a, b in g["pairs"])
    for cp in list(range(-3, g["limit"] + 16)) + [0x10FFFF, 0x110000, 0x7FFFFFFF]:
        assert f(cp) == m.get(cp, cp), hex(cp)


@pytest.mark.skipif(not bfutil.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_committed_fold_tables_are_what_the_generator_writes():
    """blingfire_amd/csrc/bf_tolower.h and oracle/bf_tolower_tab.h hold exactly the runs tools/make_tolower.py derives from the compiled reference"""
    import re
    import sys
    sys.path.insert(0, os.path.join(bfutil.ROOT, "tools"))
    import make_tolower
    want = make_tolower.runs_of(make_tolower.observe())
    for path in (os.path.join(bfutil.ROOT, "blingfire_amd", "csrc", "bf_tolower.h"), os.path.join(bfutil.ROOT, "oracle", "bf_tolower_tab.h")):
        got = [(int(a, 16), int(b), int(c), int(d)) for a, b, c, d in re.findall(r"\{0x([0-9A-F]+), (\d+), (\d+), (-?\d+)\}", open(path).read())]
        assert got == want, path


# ------------------------------------------------------------------------------------------------------------------------------
# ignore-case lexers
# ------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.skipif(not bfutil.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("model", LEXERS)
def test_oracle_matches_reference_on_ignore_case_lexer(tmp_path_factory, model):
    p = variant(tmp_path_factory, model, ldbedit.FUNC_WBD, "ic", add_boolean=ldbedit.PARAM_IGNORE_CASE)
    ref, ora = bfutil.reference(), bfutil.oracle()
    hr, ho, hp = ref.load(p), ora.load(p), ref.load(bfutil.model_path(model))
    differs = 0
    for b in case_docs(600, 11):
        a = ref.text_to_ids(hr, b, 96, 100)
        assert a == ora.text_to_ids(ho, b, 96, 100), (model, b[:80])
        differs += a != ref.text_to_ids(hp, b, 96, 100)
    # the flag changes what the model answers: the variant is not the plain model (the uncased models lower-case in their charmap already:
    # there only the letters the charmap leaves alone differ)
    assert differs > (100 if model == "bert_base_cased_tok.bin" else 5)      # wbd.bin: tags, not vocabulary ids -- few tokens change
    # TextToWords through the same lexer (no charmap on that path)
    from test_words import _call, _oracle_fn
    _, f = _oracle_fn()
    g = ref.lib.TextToWordsWithOffsetsWithModel
    g.restype = ctypes.c_int
    g.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    for k, b in enumerate(case_docs(200, 12)):
        mx = (4 * len(b) + 8, 5, 0)[k % 3]
        assert _call(f, (ctypes.c_void_p(ho),), b, mx) == _call(g, (), b, mx, (ctypes.c_void_p(hr),)), (model, b[:60], mx)
    for h in (hr, hp):
        ref.free(h)
    ora.free(ho)


@pytest.mark.parametrize("model", LEXERS)
def test_host_programs_match_oracle_on_ignore_case_lexer(tmp_path_factory, ht, model):
    p = variant(tmp_path_factory, model, ldbedit.FUNC_WBD, "ic", add_boolean=ldbedit.PARAM_IGNORE_CASE)
    ora = bfutil.oracle()
    ho = ora.load(p)
    h = ht.bft_load(p.encode())
    assert ht.bft_error(h) == b"", ht.bft_error(h)
    docs = case_docs(500, 13)
    for b in docs:                                                          # the lane-per-document program
        arr = (ctypes.c_int32 * 96)(*([-7] * 96))
        c = ht.bft_emu_text_to_ids(h, b, len(b), arr, 96, 100)
        gc, gids = ora.text_to_ids(ho, b, 96, 100)        # (the host harness uses its array as the staging slot: what lies behind the count is not output)
        assert (c, list(arr)[:c]) == (gc, gids[:gc]), (model, b[:80])
    if ht.bft_wave_ok(h) == 1:                                              # the wave program (unit-form lexers)
        import blingfire_amd as bf
        text, off = bf.pack_docs(docs)
        cap = len(text) + 16
        ids = np.full(cap, -9, dtype=np.int32)
        ido = np.zeros(len(docs) + 1, dtype=np.int64)
        st = np.zeros(16, dtype=np.uint64)
        r = ht.bft_emu_wave_batch(h, text.ctypes.data, len(text), off.ctypes.data, len(docs), 96, 100, 2, 4, 0, ids.ctypes.data, cap, ido.ctypes.data, st.ctypes.data)
        gids, goff = ora.batch(ho, text, off, 96, 100)
        assert r >= 0 and np.array_equal(ido, goff) and np.array_equal(ids[:r], gids), model
    else:
        assert model == "wbd.bin"
    ht.bft_free(h)
    ora.free(ho)


# ------------------------------------------------------------------------------------------------------------------------------
# moore-multi-dfa [wbd]
# ------------------------------------------------------------------------------------------------------------------------------
def test_moore_multi_dfa_lexer_answers_nothing(tmp_path_factory, ht):
    p = variant(tmp_path_factory, "bert_base_cased_tok.bin", ldbedit.FUNC_WBD, "multi", set_param=(ldbedit.PARAM_FSM_TYPE, ldbedit.TYPE_MOORE_MULTI_DFA))
    ora = bfutil.oracle()
    ho = ora.load(p)
    h = ht.bft_load(p.encode())
    assert ht.bft_error(h) == b"" and ht.bft_lexer_void(h) == 1
    ref = bfutil.reference() if bfutil.have_ref() else None
    hr = ref.load(p) if ref else None
    from test_words import _call, _oracle_fn
    _, f = _oracle_fn()
    for b in case_docs(40, 14)[:120]:
        arr = (ctypes.c_int32 * 16)(*([-7] * 16))
        assert ht.bft_emu_text_to_ids(h, b, len(b), arr, 16, 100) == 0
        assert ora.text_to_ids(ho, b, 16, 100) == (0, [-7] * 16)
        w = _call(f, (ctypes.c_void_p(ho),), b, 64)
        if ref:
            assert ref.text_to_ids(hr, b, 16, 100) == (0, [-7] * 16)
            g = ref.lib.TextToWordsWithOffsetsWithModel
            g.restype = ctypes.c_int
            g.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
            assert w == _call(g, (), b, 64, (ctypes.c_void_p(hr),)), b[:40]
    ht.bft_free(h)
    ora.free(ho)


# ------------------------------------------------------------------------------------------------------------------------------
# ignore-case dictionaries
# ------------------------------------------------------------------------------------------------------------------------------
def dict_keys(model, seed):
    from test_dict_lookup import keys_for
    rnd = random.Random(seed)
    keys = keys_for(model, n_random=600, seed=seed, negative=False)
    out = []
    for k in keys:
        out.append(k)
        s = "".join(chr(c) for c in k if 0 <= c < 0x110000 and not 0xD800 <= c < 0xE000)
        if s and rnd.random() < 0.7:
            t = s.upper() if rnd.random() < 0.5 else "".join(c.upper() if rnd.random() < 0.5 else c for c in s)
            out.append([ord(c) for c in t][:300])
            out.append(out[-1][::-1])                 # found by right-to-left dictionaries
    return out


def dict_variant(tmp_path_factory, model, direction):
    kw = dict(add_boolean=ldbedit.PARAM_IGNORE_CASE)
    if direction is not None:
        kw["set_param"] = (11, direction)             # PARAM_DIRECTION, FAFsmConst.h DIR_R2L = 1
    return variant(tmp_path_factory, model, ldbedit.FUNC_POS_DICT, "dic%s" % ("" if direction is None else direction), **kw)


def ora_dict_lib():
    L = ctypes.CDLL(bfutil.ORACLE_LIB)
    L.bfo_load_model.restype = ctypes.c_void_p
    L.bfo_load_model.argtypes = [ctypes.c_char_p]
    L.bfo_free_model.argtypes = [ctypes.c_void_p]
    L.bfo_dict_get_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L.bfo_dict_get_info_id.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return L


@pytest.mark.skipif(not os.path.exists(DICTREF), reason="oracle/_ref/libdictref.so not built (needs the reference checkout)")
@pytest.mark.parametrize("model,direction", DICTS)
def test_oracle_matches_reference_interpreter_on_ignore_case_dictionary(tmp_path_factory, model, direction):
    from test_dict_lookup import oracle_lookup
    p = dict_variant(tmp_path_factory, model, direction)
    R = ctypes.CDLL(DICTREF)
    R.refdict_load.restype = ctypes.c_void_p
    R.refdict_load.argtypes = [ctypes.c_char_p]
    R.refdict_free.argtypes = [ctypes.c_void_p]
    R.refdict_get_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    R.refdict_get_info_id.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L = ora_dict_lib()
    hr, ho, hp = R.refdict_load(p.encode()), L.bfo_load_model(p.encode()), L.bfo_load_model(bfutil.model_path(model).encode())
    assert hr and ho
    found = extra = 0
    for key in dict_keys(model, 21):
        arr = (ctypes.c_int32 * max(len(key), 1))(*key)
        out = (ctypes.c_int32 * 8)(*([-7] * 8))
        rr = R.refdict_get_info(ctypes.c_void_p(hr), arr, len(key), out, 8)
        r, i, got = oracle_lookup(L, ho, key)
        assert (rr, list(out)) == (r, got) and R.refdict_get_info_id(ctypes.c_void_p(hr), arr, len(key)) == i, (model, direction, key[:12])
        found += i != -1
        extra += i != -1 and oracle_lookup(L, hp, key)[1] == -1
    assert found > 50 and (extra > 10 or direction == 1)       # upper-cased keys are found only because of the flag
    R.refdict_free(ctypes.c_void_p(hr))
    for h in (ho, hp):
        L.bfo_free_model(ctypes.c_void_p(h))


@pytest.mark.parametrize("model,direction", DICTS)
def test_device_dictionary_program_on_host_matches_oracle_ignore_case(tmp_path_factory, ht, model, direction):
    from test_dict_lookup import oracle_lookup
    p = dict_variant(tmp_path_factory, model, direction)
    L = ora_dict_lib()
    ho = L.bfo_load_model(p.encode())
    hh = ht.bft_load(p.encode())
    assert ht.bft_error(hh) == b"", ht.bft_error(hh)
    for key in dict_keys(model, 22):
        arr = (ctypes.c_int32 * max(len(key), 1))(*key)
        iid = ctypes.c_int32(0)
        vals = (ctypes.c_int32 * 8)(*([-7] * 8))
        r = ht.bft_emu_dict_get_info(ctypes.c_void_p(hh), arr, len(key), ctypes.byref(iid), vals, 8)
        gr, gi, gout = oracle_lookup(L, ho, key)
        assert (r, iid.value) == (gr, gi) and list(vals) == gout, (model, direction, key[:12])
    ht.bft_free(hh)
    L.bfo_free_model(ctypes.c_void_p(ho))


# ------------------------------------------------------------------------------------------------------------------------------
# GPU: the product through the C-ABI
# ------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("model", LEXERS)
def test_gpu_ignore_case_lexer(tmp_path_factory, model):
    import blingfire_amd as bf
    p = variant(tmp_path_factory, model, ldbedit.FUNC_WBD, "ic", add_boolean=ldbedit.PARAM_IGNORE_CASE)
    ck = bfutil.reference() if bfutil.have_ref() else bfutil.oracle()       # the compiled reference travels to the GPU box
    hc = ck.load(p)
    docs = case_docs(1500, 15)
    text, off = bf.pack_docs(docs)
    h = bf.load_model(p)
    try:
        for variant_bits in (0, 2):                                          # the wave program (unit-form lexers) and the lane kernels
            bf.lib().BfSetVariant(h, variant_bits)
            ids, id_off = bf.text_to_ids_batch(h, (text, off), 96, 100)
            for d, b in enumerate(docs):
                c, want = ck.text_to_ids(hc, b, 96, 100)
                assert ids[id_off[d]:id_off[d + 1]].tolist() == want[:c], (model, variant_bits, b[:80])
        from test_words import _call, _oracle_fn
        L = bf.lib()
        g = L.TextToWordsWithOffsetsWithModel
        g.restype = ctypes.c_int
        g.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        ora, f = _oracle_fn()
        ho = ora.load(p)
        for k, b in enumerate(docs[:300]):
            mx = (4 * len(b) + 8, 5, 0)[k % 3]
            assert _call(g, (), b, mx, (ctypes.c_void_p(h),)) == _call(f, (ctypes.c_void_p(ho),), b, mx), (model, b[:60], mx)
        ora.free(ho)
    finally:
        bf.free_model(h)
        ck.free(hc)


@pytest.mark.gpu
def test_gpu_moore_multi_dfa_lexer(tmp_path_factory):
    import blingfire_amd as bf
    p = variant(tmp_path_factory, "bert_base_cased_tok.bin", ldbedit.FUNC_WBD, "multi", set_param=(ldbedit.PARAM_FSM_TYPE, ldbedit.TYPE_MOORE_MULTI_DFA))
    docs = case_docs(100, 16)
    h = bf.load_model(p)
    try:
        ids, id_off = bf.text_to_ids_batch(h, docs, 64, 100)
        assert len(ids) == 0 and not id_off.any()
        L = bf.lib()
        L.TextToWordsWithModel.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        o = ctypes.create_string_buffer(64)
        assert L.TextToWordsWithModel(b"Hello world", 11, o, 64, ctypes.c_void_p(h)) == -1
        assert L.TextToWordsWithModel(b"", 0, o, 64, ctypes.c_void_p(h)) == 0
    finally:
        bf.free_model(h)


@pytest.mark.gpu
@pytest.mark.parametrize("model,direction", DICTS)
def test_gpu_ignore_case_dictionary(tmp_path_factory, model, direction):
    import blingfire_amd as bf
    from test_dict_lookup import oracle_lookup
    p = dict_variant(tmp_path_factory, model, direction)
    L = ora_dict_lib()
    ho = L.bfo_load_model(p.encode())
    keys = dict_keys(model, 23)
    h = bf.load_model(p)
    try:
        ret, ids, vals, v_off = bf.dict_get_info_batch(h, keys)
        for k, key in enumerate(keys):
            gr, gi, gout = oracle_lookup(L, ho, key)
            assert (int(ret[k]), int(ids[k])) == (gr, gi), (model, direction, key[:12])
            n = max(gr, 0)
            assert v_off[k + 1] - v_off[k] == n and list(vals[v_off[k]:v_off[k + 1]]) == gout[:n]
    finally:
        bf.free_model(h)
        L.bfo_free_model(ctypes.c_void_p(ho))
