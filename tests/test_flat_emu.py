"""CPU: the flat program of the WordPiece path (blingfire_amd/csrc/bf_flat_body.h -- the sources the GPU kernels k_wp_flat, k_wp_units,
k_wp_count and k_wp_merge run) executed inside the 64-fibre wave simulator of tests/hosttest/wave_emu.h, with k_wp_pre, the list of
the documents it hands back, the wave program's LIST instance on those and the scan restated around it (tests/hosttest/bf_wavetest.cpp
bft_emu_flat_batch), against the oracle.  What a per-lane emulation cannot cover: ranges that ignore document boundaries, the token list
and its look-up trips, the word lists of a range, the streaming count / merge over contiguous pieces, and every way a document is handed
back (a run of more than 48 bytes, an element the automaton itself decides, a batch that is not fit)."""
import ctypes
import random

import numpy as np
import pytest

import bfutil
import blingfire_amd as bf

WP_MODELS = ["bert_base_tok.bin", "bert_base_cased_tok.bin", "bert_chinese.bin"]
# (max_ids, unk, waves, ranges): ranges 0 = four per wave (as many as documents allow)
CONFS = [(512, 100, 1, 0), (512, 100, 2, 3), (16, 7, 1, 1), (64, 5, 3, 7), (0, 100, 2, 1), (512, 100, 2, 1)]


@pytest.fixture(scope="module")
def ht():
    L = ctypes.CDLL(bfutil.HOSTTEST_LIB)
    L.bft_load.restype = ctypes.c_void_p
    L.bft_load.argtypes = [ctypes.c_char_p]
    L.bft_free.argtypes = [ctypes.c_void_p]
    L.bft_flat_ok.argtypes = [ctypes.c_void_p]
    L.bft_flat_words.argtypes = [ctypes.c_void_p]
    L.bft_emu_flat_batch.restype = ctypes.c_long
    L.bft_emu_flat_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    L.bft_emu_flat_batch_offsets.restype = ctypes.c_long
    L.bft_emu_flat_batch_offsets.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
    return L


def flat_batch_offsets(ht, h, text, off, max_ids, unk, nwaves, nranges):
    nd = len(off) - 1
    cap = len(text) + 16
    ids, sts, ens = (np.full(cap, -9, dtype=np.int32) for _ in range(3))
    ido = np.zeros(nd + 1, dtype=np.int64)
    st = np.zeros(16, dtype=np.uint64)
    text = np.ascontiguousarray(text)
    r = ht.bft_emu_flat_batch_offsets(h, text.ctypes.data, len(text), off.ctypes.data, nd, max_ids, unk, nwaves, nranges, ids.ctypes.data, sts.ctypes.data, ens.ctypes.data,
                                      cap, ido.ctypes.data, st.ctypes.data)
    n = max(r, 0)
    return r, ids[:n], sts[:n], ens[:n], ido, st


def flat_batch(ht, h, text, off, max_ids, unk, nwaves, nranges):
    nd = len(off) - 1
    cap = len(text) + 16
    ids = np.full(cap, -9, dtype=np.int32)
    ido = np.zeros(nd + 1, dtype=np.int64)
    st = np.zeros(16, dtype=np.uint64)
    text = np.ascontiguousarray(text)
    r = ht.bft_emu_flat_batch(h, text.ctypes.data, len(text), off.ctypes.data, nd, max_ids, unk, nwaves, nranges, ids.ctypes.data, cap, ido.ctypes.data, st.ctypes.data, 0)
    return r, ids[:max(r, 0)], ido, st


offsets_oracle = {}


def check(ht, model, docs, confs, want=None):
    offsets_oracle.clear()
    mp = bfutil.model_path(model)
    h = ht.bft_load(mp.encode())
    assert ht.bft_flat_ok(h) == 1
    ora = bfutil.oracle()
    ho = ora.load(mp)
    text, off = docs if isinstance(docs, tuple) else bf.pack_docs(docs)
    last = None
    for (mx, unk, nw, nr) in confs:
        r, ids, ido, st = flat_batch(ht, h, text, off, mx, unk, nw, nr)
        gids, goff = ora.batch(ho, text, off, mx, unk)
        assert r >= 0, (model, r)
        if not (np.array_equal(ido, goff) and np.array_equal(ids, gids)):
            for d in range(len(off) - 1):
                a, b = ids[ido[d]:ido[d + 1]], gids[goff[d]:goff[d + 1]]
                assert np.array_equal(a, b), (model, (mx, unk, nw, nr), d, bytes(text[off[d]:off[d + 1]])[:80], a.tolist()[:20], b.tolist()[:20])
        if want is not None:
            want(st)
        last = st
        # the offsets API through the same program: the same ids, and the byte offsets of every id (oracle: per document)
        r2, ids2, sts, ens, ido2, st2 = flat_batch_offsets(ht, h, text, off, mx, unk, nw, nr)
        assert r2 == r and np.array_equal(ids2, ids) and np.array_equal(ido2, ido), (model, r2, r)
        if offsets_oracle.get((model, id(docs), mx, unk)) is None:
            raw = text.tobytes()
            ws, we = [], []
            for d in range(len(off) - 1):
                c, _, s_, e_ = ora.with_offsets(ho, raw[off[d]:off[d + 1]], mx, unk, "bfo_text_to_ids_with_offsets")
                ws += s_[:min(c, mx)]; we += e_[:min(c, mx)]
            offsets_oracle[(model, id(docs), mx, unk)] = (np.array(ws, dtype=np.int32), np.array(we, dtype=np.int32))
        ws, we = offsets_oracle[(model, id(docs), mx, unk)]
        if not (np.array_equal(sts, ws) and np.array_equal(ens, we)):
            for d in range(len(off) - 1):
                a, b = slice(ido[d], ido[d + 1]), slice(goff[d], goff[d + 1])
                assert np.array_equal(sts[a], ws[b]) and np.array_equal(ens[a], we[b]), (model, (mx, unk, nw, nr), d, bytes(text[off[d]:off[d + 1]])[:80],
                                                                                        sts[a].tolist()[:24], ws[b].tolist()[:24], ens[a].tolist()[:24], we[b].tolist()[:24])
        if want is not None:
            want(st2)
    ora.free(ho)
    ht.bft_free(h)
    return last


def test_flat_form_is_proven_for_the_bert_lexers_only(ht):
    for model, want in [(m, 1) for m in WP_MODELS] + [("wbd.bin", 0), ("sbd.bin", 0), ("wbd_chuni.bin", 0), ("xlnet.bin", 0)]:
        if not bfutil.have_model(model):
            continue
        h = ht.bft_load(bfutil.model_path(model).encode())
        assert ht.bft_flat_ok(h) == want, model
        if want:
            assert ht.bft_flat_words(h) > 5000        # the one-piece words of <= 12 characters of the vocabulary
        ht.bft_free(h)


@pytest.mark.parametrize("model", WP_MODELS)
def test_adversarial_and_fuzz(ht, model):
    if not bfutil.have_model(model):
        pytest.skip("model not present")
    check(ht, model, list(bfutil.ADVERSARIAL), CONFS)
    check(ht, model, bfutil.fuzz_docs(300, seed=3), CONFS[:4])
    check(ht, model, bfutil.fuzz_docs(200, seed=17, maxwords=200), [(512, 100, 2, 0), (512, 100, 1, 1)])


@pytest.mark.parametrize("model", WP_MODELS)
def test_corpora(ht, model):
    if not bfutil.have_model(model):
        pytest.skip("model not present")
    # the metric's documents (most blocks of 64 documents stream; several ranges per wave and one range for all)
    st = check(ht, model, bfutil.gen_workload("headline512", 300), [(512, 100, 2, 0), (512, 100, 2, 1), (40, 100, 1, 2)])
    assert st[3] > 0.6 * st[2]                        # most tokens are answered by the table
    check(ht, model, bfutil.gen_workload("config2", 600), [(512, 100, 3, 0), (512, 100, 2, 1), (30, 100, 2, 1)])


def test_document_shapes(ht):
    model = bfutil.bert_model_name()
    rnd = random.Random(5)
    words = [b"the", b"of", b"unaffable", b"qzxjkvw", "café".encode(), "naïve".encode(), b"e-mail", b"3,000.50", "日本語".encode(), b"internationalization", b"x"]
    # thousands of one- and two-byte documents, empty documents in runs, documents that end inside a chunk, at its end, one byte behind it
    tiny = [rnd.choice([b"a", b"", b"", b"to", b".", b" ", b"\xc3\xa9", b"\xff", b"ab"]) for _ in range(3000)]
    check(ht, model, tiny, [(512, 100, 2, 0), (512, 100, 1, 1), (1, 100, 3, 5)])
    edges = []
    for n in (1, 7, 8, 9, 63, 64, 65, 511, 512, 513, 1023, 1024, 1025, 1536):
        for fill in (b"a ", b"ab, ", "é ".encode(), b"word "):
            edges.append((fill * (n // len(fill) + 1))[:n])
    check(ht, model, edges, [(512, 100, 1, 1), (2000, 100, 2, 0)])
    mixed = [b" ".join(rnd.choice(words) for _ in range(rnd.randint(0, 120))) for _ in range(200)]
    check(ht, model, mixed, [(512, 100, 2, 0), (512, 100, 3, 1), (5, 100, 1, 2)])
    # a character the vocabulary does not hold as the last bytes of the batch (its word is read from the text, character by character)
    for tail in ("tail \U00020000", "tail \u0e5b", "x\u4e00"):
        check(ht, model, [b"plain words", tail.encode()], [(512, 100, 1, 1)])
    # a chunk of 512 one-byte tokens behind a run that ends with the chunk before it: 513 tokens on the list of one chunk (the list once had 512
    # places, the 513th token went to the first record that waited: found by tools/stress_flat_emu.py).  The document before ends at a chunk boundary
    # with a word the table does not hold, so that a record waits while the next chunk is listed
    for tail in (b"zqxjkvw", b"caf\xc3\xa9s", b"word"):
        first = (b"zqxjkvw " + b"some words and " * 40)[:512 - len(tail) - 1] + b" " + tail          # (its first word leaves a record that waits)
        assert len(first) == 512
        check(ht, model, [first, b"." * 600, b"after"], [(4096, 100, 1, 1), (4096, 100, 2, 0)])
        check(ht, model, [first + b"." * 1100 + b" " + tail], [(4096, 100, 1, 1)])
    # more words the table does not answer than a range's list holds (a record per four bytes of the range): the documents of the words that do not
    # fit are handed back; characters the vocabulary lacks, with and without blanks between them, in one range and in several
    for body in ("͸ " * 2000, "͸" * 3000, "\U00020000" * 1500, "͸a͹b " * 1200):
        check(ht, model, [b"plain first", body.encode(), b"plain last"], [(4096, 100, 2, 0), (4096, 100, 2, 1)])
    # words of many pieces one after the other (more ids per trip of the merge than its buffer holds), among documents of plain words
    many = [b" ".join(bytes(rnd.choice(b"qzxjkvw") for _ in range(rnd.randint(6, 14))) for _ in range(rnd.randint(1, 400))) if i % 3 else b"plain words only , here"
            for i in range(150)]
    check(ht, model, many, [(4096, 100, 2, 0), (4096, 100, 1, 1), (100, 100, 2, 1)])


def test_handed_back(ht):
    model = bfutil.bert_model_name()
    # runs of more than 48 bytes (one that crosses a chunk, one that fills several chunks), an element the automaton itself decides ('['),
    # next to documents the flat program keeps: only the former go to the wave program
    docs = [b"plain words only", b"a" * 49, b"fine again", b"b" * 700 + b" tail", b"with [UNK] inside", b"x" * 48, b"ok " * 100, (b"q" * 300 + b" ") * 5, b"[", b"end"]

    def want(st):
        assert st[8] == 5 and st[9] == 0, st.tolist()          # five documents on the list; the batch was fit

    check(ht, model, docs, [(512, 100, 1, 1), (512, 100, 2, 0), (3, 100, 2, 3)], want)
    # words of 17 .. 48 bytes and words with characters outside ASCII: the second list of a range
    longw = [("über" + "x" * k).encode() + b" " + b"y" * (17 + k) + b" fin" for k in range(0, 31)]
    check(ht, model, longw, [(512, 100, 1, 1), (512, 100, 2, 0)])


def test_batch_not_fit(ht):
    model = bfutil.bert_model_name()
    # a document of more than 4 MiB: the whole batch is handed to the wave program
    big = (b"word " * 900000)[:(1 << 22) + 5]
    text, off = bf.pack_docs([b"small one", big, b"small two"])

    def want(st):
        assert st[9] == 1 and st[8] == 3, st.tolist()

    check(ht, model, (text, off), [(64, 100, 2, 0)], want)


def test_invalid_utf8_everywhere(ht):
    model = bfutil.bert_model_name()
    rnd = random.Random(9)
    bad = [b"\xff", b"\xc3", b"\xe2\x82", b"\xf0\x9f\x98", b"\x80", b"\xed\xa0\x80", b"\xc0\xaf", b"\xf4\x90\x80\x80"]
    docs = []
    for i in range(400):
        n = rnd.randint(0, 1100)
        body = bytearray((b"some words, and more " * 60)[:n])
        if i % 3 == 0 and n > 0:
            at = rnd.randint(0, n)
            body[at:at] = rnd.choice(bad)              # at any place: inside a chunk, across a chunk boundary, at the very end
        if i % 7 == 0:
            body = bytearray(b"\xef\xbb\xbf") + body   # a BOM at the start is skipped, elsewhere it is a character
        docs.append(bytes(body))
    check(ht, model, docs, [(512, 100, 2, 0), (512, 100, 1, 1), (512, 100, 3, 9)])


def test_any_batch_hypothesis(ht):
    hyp = pytest.importorskip("hypothesis")
    st = hyp.strategies
    model = bfutil.bert_model_name()
    mp = bfutil.model_path(model)
    h = ht.bft_load(mp.encode())
    ora = bfutil.oracle()
    ho = ora.load(mp)
    alphabet = ["a", "b", "the", " ", " ", ",", ".", "é", "日", "[", " ", "ß", "İ", "x" * 50, "﻿", "un", "##ing", "1", "-"]
    doc = st.lists(st.sampled_from(alphabet), min_size=0, max_size=150).map(lambda xs: "".join(xs).encode())
    raw = st.binary(min_size=0, max_size=40)

    @hyp.settings(max_examples=60, deadline=None)
    @hyp.given(st.lists(st.one_of(doc, doc, raw), min_size=1, max_size=90), st.integers(0, 40), st.integers(1, 3), st.integers(0, 6))
    def run(docs, mx, nw, nr):
        text, off = bf.pack_docs(docs)
        r, ids, ido, _ = flat_batch(ht, h, text, off, mx, 100, nw, nr)
        gids, goff = ora.batch(ho, text, off, mx, 100)
        assert r >= 0 and np.array_equal(ido, goff) and np.array_equal(ids, gids)

    run()
    ora.free(ho)
    ht.bft_free(h)
