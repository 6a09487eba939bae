"""TextToWords family (reference tokdll:415-614; BASELINE.json configs[0] is its CPU-only case).  CPU tests pin the oracle's
restatement to the compiled reference and run the words-mode lane program on the host; the GPU test calls the product's
TextToWords* exports (lexer on the GPU, built-in wbd.bin embedded like the reference's)."""
import ctypes

import pytest

import bfutil

WORD_MODELS = ["wbd.bin", "wbd_chuni.bin", "bert_base_cased_tok.bin", "bert_chinese.bin"]


def _docs(n, seed):
    return list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(n, seed=seed) + [b"Hello world . This is a test ."]


def _call(fn, args_before, b, mx, args_after=()):
    o = ctypes.create_string_buffer(b"\x7f" * (max(mx, 1) + 4))
    s = (ctypes.c_int32 * max(mx, 1))(*([-7] * max(mx, 1)))
    e = (ctypes.c_int32 * max(mx, 1))(*([-7] * max(mx, 1)))
    r = fn(*args_before, b, len(b), o, s, e, mx, *args_after)
    return r, o.raw, list(s), list(e)


def _oracle_fn():
    ora = bfutil.oracle()
    f = ora.lib.bfo_text_to_words_with_offsets
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return ora, f


def test_readme_known_answer():
    """reference README.md:64-75: text_to_words on the built-in model"""
    ora, f = _oracle_fn()
    h = ora.load(bfutil.model_path("wbd.bin"))
    s = "After reading this post, you will know: What \"natural language\" is and how it is different from other types of data.".encode()
    r, out, _, _ = _call(f, (ctypes.c_void_p(h),), s, 4 * len(s))
    assert out[:r - 1].decode() == ("After reading this post , you will know : What \" natural language \" is and how it is different from other "
                                    "types of data .")
    ora.free(h)


@pytest.mark.skipif(not bfutil.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("model", WORD_MODELS + [None])
def test_oracle_words_vs_live_reference(model):
    ora, f = _oracle_fn()
    ref = bfutil.reference()
    g = ref.lib.TextToWordsWithOffsetsWithModel
    g.restype = ctypes.c_int
    g.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    ho = ora.load(bfutil.model_path(model or "wbd.bin"))
    hr = ref.load(bfutil.model_path(model)) if model else None      # None = the reference's built-in model
    for k, b in enumerate(_docs(1500, 61)):
        mx = (4 * len(b) + 8, 5, 0)[k % 3]
        assert _call(f, (ctypes.c_void_p(ho),), b, mx) == _call(g, (), b, mx, (ctypes.c_void_p(hr) if hr else None,)), (model, b[:60], mx)
    ora.free(ho)


@pytest.mark.parametrize("model", WORD_MODELS)
def test_words_lane_program_on_host_matches_oracle(model):
    ora, f = _oracle_fn()
    L = ctypes.CDLL(bfutil.HOSTTEST_LIB)
    L.bft_load.restype = ctypes.c_void_p
    L.bft_load.argtypes = [ctypes.c_char_p]
    g = L.bft_emu_text_to_words
    g.restype = ctypes.c_int
    g.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    h = L.bft_load(bfutil.model_path(model).encode())
    ho = ora.load(bfutil.model_path(model))
    for k, b in enumerate(_docs(3000, 67)):
        mx = (4 * len(b) + 8, 5, 0)[k % 3]
        assert _call(g, (ctypes.c_void_p(h),), b, mx) == _call(f, (ctypes.c_void_p(ho),), b, mx), (model, b[:60], mx)
    ora.free(ho)


def test_config1_reference_cpu_case():
    """BASELINE.json configs[0]: default pattern tokenizer on 10k short English lines, CPU reference path only -- the oracle
    (and the compiled reference when present) agree on the 10,000 lines SURVEY.md section 8(d) names: the first non-empty lines of the
    reference's own ldbsrc/bert_multi_cased/test.legacy.txt.zip (tests/data/config1_lines.txt.gz, 427,735 bytes)."""
    ora, f = _oracle_fn()
    ho = ora.load(bfutil.model_path("wbd.bin"))
    text, off = bfutil.gen_workload("config1", 10000)
    assert len(text) == 427735 and len(off) == 10001
    raw = text.tobytes()
    ref = bfutil.reference() if bfutil.have_ref() else None
    if ref:
        g = ref.lib.TextToWordsWithOffsetsWithModel
        g.restype = ctypes.c_int
        g.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    nwords = 0
    for d in range(10000):
        b = raw[off[d]:off[d + 1]]
        a = _call(f, (ctypes.c_void_p(ho),), b, 4 * len(b) + 8)
        nwords += a[1][:a[0]].count(b" ") + 1
        if ref:
            assert a == _call(g, (), b, 4 * len(b) + 8, (None,))
    assert nwords > 50000
    ora.free(ho)


@pytest.mark.gpu
@pytest.mark.parametrize("model", WORD_MODELS + [None])
def test_gpu_text_to_words(model):
    import blingfire_amd as bf
    L = bf.lib()
    g = L.TextToWordsWithOffsetsWithModel
    g.restype = ctypes.c_int
    g.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    ora, f = _oracle_fn()
    ho = ora.load(bfutil.model_path(model or "wbd.bin"))
    h = bf.load_model(bfutil.model_path(model)) if model else None
    try:
        for k, b in enumerate(_docs(400, 71)):
            mx = (4 * len(b) + 8, 5, 0)[k % 3]
            assert _call(g, (), b, mx, (ctypes.c_void_p(h) if h else None,)) == _call(f, (ctypes.c_void_p(ho),), b, mx), (model, b[:60], mx)
        if model is None:
            L.TextToWords.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
            o = ctypes.create_string_buffer(256)
            s = b"Hello, world! It's 3.14."
            r = L.TextToWords(s, len(s), o, 256)
            assert o.raw[:r - 1] == b"Hello , world ! It 's 3.14 ."
    finally:
        if h:
            bf.free_model(h)
        ora.free(ho)


@pytest.mark.gpu
def test_gpu_python_mirror_words():
    """reference README.md:64-75 and dist-pypi/blingfire/__init__.py:222 semantics through the Python mirror"""
    import blingfire_amd as bf
    s = "Hello, wörld! It's 3.14."
    assert bf.text_to_words(s) == "Hello , wörld ! It 's 3.14 ."
    words, spans = bf.text_to_words_with_offsets(s)
    assert [s[b:e] for b, e in spans] == words.split(" ")
    h = bf.load_model(bfutil.model_path("bert_base_cased_tok.bin"))
    try:
        assert bf.text_to_words_with_model(h, "unaffable!") == "unaffable un af fa ble ! !"   # word and sub-word tokens both reported
    finally:
        bf.free_model(h)


# ---- TextToSentences family (reference tokdll:163-402): same machinery, every token ends a sentence

SENT_MODELS = ["sbd.bin", "wbd.bin"]
SENT_DOCS = [b"Hello world. This is a test! Is it? Yes.\nNew line here. And Mr. Smith went to Washington D.C. yesterday.",
             b"  leading. trailing  ", "Привет мир. Как дела? 好的。谢谢".encode(), b"No terminator", b"One. Two.\n\nThree.\x00Four."]


def _oracle_sent_fn():
    ora = bfutil.oracle()
    f = ora.lib.bfo_text_to_sentences_with_offsets
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return ora, f


def test_sentences_readme_known_answer():
    """reference README.md:60-75: text_to_sentences on the built-in model"""
    ora, f = _oracle_sent_fn()
    h = ora.load(bfutil.model_path("sbd.bin"))
    s = ("After reading this post, you will know: What \"natural language\" is and how it is different from other types of data. "
         "What makes working with natural language so challenging. [1]").encode()
    r, out, _, _ = _call(f, (ctypes.c_void_p(h),), s, 4 * len(s))
    assert out[:r - 1].decode() == ("After reading this post, you will know: What \"natural language\" is and how it is different from other types of data.\n"
                                    "What makes working with natural language so challenging. [1]")
    ora.free(h)


@pytest.mark.skipif(not bfutil.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("model", SENT_MODELS + [None])
def test_oracle_sentences_vs_live_reference(model):
    ora, f = _oracle_sent_fn()
    ref = bfutil.reference()
    g = ref.lib.TextToSentencesWithOffsetsWithModel
    g.restype = ctypes.c_int
    g.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    ho = ora.load(bfutil.model_path(model or "sbd.bin"))
    hr = ref.load(bfutil.model_path(model)) if model else None      # None = the reference's built-in model
    for k, b in enumerate(_docs(1500, 83) + SENT_DOCS):
        mx = (4 * len(b) + 8, 5, 0)[k % 3]
        assert _call(f, (ctypes.c_void_p(ho),), b, mx) == _call(g, (), b, mx, (ctypes.c_void_p(hr) if hr else None,)), (model, b[:60], mx)
    ora.free(ho)


@pytest.mark.gpu
@pytest.mark.parametrize("model", SENT_MODELS + [None])
def test_gpu_text_to_sentences(model):
    import blingfire_amd as bf
    L = bf.lib()
    g = L.TextToSentencesWithOffsetsWithModel
    ora, f = _oracle_sent_fn()
    ho = ora.load(bfutil.model_path(model or "sbd.bin"))
    h = bf.load_model(bfutil.model_path(model)) if model else None
    try:
        for k, b in enumerate(_docs(400, 89) + SENT_DOCS):
            mx = (4 * len(b) + 8, 5, 0)[k % 3]
            assert _call(g, (), b, mx, (ctypes.c_void_p(h) if h else None,)) == _call(f, (ctypes.c_void_p(ho),), b, mx), (model, b[:60], mx)
        if model is None:
            s = "Hello wörld. This is a test! Is it?"
            assert bf.text_to_sentences(s) == "Hello wörld.\nThis is a test!\nIs it?"
            sents, spans = bf.text_to_sentences_and_offsets(s)
            assert [s[b:e] for b, e in spans] == sents.split("\n")
    finally:
        if h:
            bf.free_model(h)
        ora.free(ho)


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["wbd.bin", "bert_base_cased_tok.bin", None])
def test_gpu_text_to_words_batch(model):
    """additive batch form: per document exactly the string TextToWordsWithModel produces (checked against the oracle)"""
    import blingfire_amd as bf
    ora, f = _oracle_fn()
    ho = ora.load(bfutil.model_path(model or "wbd.bin"))
    h = bf.load_model(bfutil.model_path(model)) if model else None
    try:
        docs = _docs(1500, 97) + [b"", b"Hello world . This is a test ."] * 3
        text, off = bfutil.gen_corpus(3000, seed=5, mean=43, sd=12, minlen=8, maxlen=120)
        raw = text.tobytes()
        docs += [raw[off[d]:off[d + 1]] for d in range(3000)]
        out, t_off = bf.text_to_words_batch(docs, h)
        for d, b in enumerate(docs):
            r, o, _, _ = _call(f, (ctypes.c_void_p(ho),), b, 4 * len(b) + 8)
            want = o[:r - 1] if r > 0 else b""
            assert out[t_off[d]:t_off[d + 1]].tobytes() == want, (model, d, b[:60])
    finally:
        if h:
            bf.free_model(h)
        ora.free(ho)


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["sbd.bin", None])
def test_gpu_text_to_sentences_batch(model):
    """additive batch form: per document exactly the string TextToSentencesWithModel produces (checked against the oracle)"""
    import blingfire_amd as bf
    ora, f = _oracle_sent_fn()
    ho = ora.load(bfutil.model_path(model or "sbd.bin"))
    h = bf.load_model(bfutil.model_path(model)) if model else None
    try:
        docs = _docs(1500, 101) + SENT_DOCS * 3 + [b""]
        text, off = bfutil.gen_corpus(2000, seed=6, mean=300, sd=80, minlen=20, maxlen=900)
        raw = text.tobytes()
        docs += [raw[off[d]:off[d + 1]].replace(b" the ", b". The ") for d in range(2000)]
        out, t_off = bf.text_to_sentences_batch(docs, h)
        for d, b in enumerate(docs):
            r, o, _, _ = _call(f, (ctypes.c_void_p(ho),), b, 4 * len(b) + 8)
            want = o[:r - 1] if r > 0 else b""
            assert out[t_off[d]:t_off[d + 1]].tobytes() == want, (model, d, b[:60])
    finally:
        if h:
            bf.free_model(h)
        ora.free(ho)


# ---- long documents of the words modes (bf_lex.h lex_one_start / lex_chain_visit, bf_kernels.hip k_lex_long*): every start position of
#      a document on its own, the chain of the positions the reference's loop visits (FALexTools_t.h:229-393), those again with output

def _long_docs(seed):
    import random
    rnd = random.Random(seed)
    text, off = bfutil.gen_workload("config1", 10000)
    raw = text.tobytes()
    lines = [raw[off[d]:off[d + 1]] for d in range(10000)]
    docs = sorted(lines, key=len)[-25:] + rnd.sample(lines, 200)
    docs += [b" ".join(rnd.sample(lines, 40)), b". ".join(rnd.sample(lines, 25)), b"x" * 700, b" " * 400, ("д" * 301 + " 好的。" * 50).encode()]
    docs += [b for b in bfutil.fuzz_docs(150, seed=seed) if len(b) > 40]
    # more than 2048 bytes / 1024 tokens: the sixteen-wave forms of the decode and of the string assembly (k_prep_wp_long, k_w2t_copy_long) --
    # multi-byte characters across the 512-byte pieces, a byte order mark, invalid and truncated UTF-8 (the document yields nothing)
    mixed = ("д好x. Ünï çødé\u3000text! " * 900).encode()
    docs += [mixed, b"\xef\xbb\xbf" + mixed[:30011], b"abc d. " * 700 + b"\xff" + b"def g. " * 700, mixed[:8191], mixed[1:9000], b"a b. " * 5000,
             (b"One sentence here. " * 40 + b"\n\n") * 60, b" \n" * 3000 + b"end."]
    return docs


@pytest.mark.parametrize("model,mode", [("wbd.bin", 1), ("sbd.bin", 2), ("wbd_chuni.bin", 1), ("bert_base_cased_tok.bin", 1), ("wbd.bin", 2), ("sbd.bin", 1)])
def test_long_form_on_host_equals_the_sequential_program(model, mode):
    """the three steps of the long-document form, run on the host over the same tables, give the tokens of the sequential lane program
    (which test_words_lane_program_on_host_matches_oracle pins to the oracle) -- also when the triple buffer fills in the middle"""
    import numpy as np
    L = ctypes.CDLL(bfutil.HOSTTEST_LIB)
    L.bft_load.restype = ctypes.c_void_p
    L.bft_load.argtypes = [ctypes.c_char_p]
    g = L.bft_emu_lex_tokens
    g.restype = ctypes.c_int
    g.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    h = L.bft_load(bfutil.model_path(model).encode())

    def toks(b, long_form, cap):
        n = len(b)
        tags, sp, vis = np.zeros(n + 2, np.int32), np.zeros(2 * n + 4, np.int32), np.zeros(n + 4, np.int32)
        w = g(h, b, n, mode, long_form, cap, tags.ctypes.data, sp.ctypes.data, n + 1, vis.ctypes.data)
        return w, tags[:max(w, 0)].tolist(), sp[:2 * max(w, 0)].tolist()
    ntok = 0
    for b in _long_docs(3):
        for cap in (0, 3, max(1, len(b) // 9)):
            a = toks(b, 0, cap)
            assert a == toks(b, 1, cap), (model, mode, cap, b[:60])
            ntok += max(a[0], 0)
    assert ntok > 200
    L.bft_free(ctypes.c_void_p(h))


@pytest.mark.gpu
@pytest.mark.parametrize("model,mode", [("wbd.bin", 1), (None, 1), ("sbd.bin", 2), (None, 2), ("bert_base_cased_tok.bin", 1), ("wbd.bin", 2)])
def test_gpu_long_documents(model, mode):
    """batches that mix short lines with long ones (up to 1 MB in one document): per document the string of the single-document reference call,
    with the long-document path at its default threshold, at 16 characters, and switched off"""
    import random
    import blingfire_amd as bf
    ora, f = _oracle_fn() if mode == 1 else _oracle_sent_fn()
    default = "wbd.bin" if mode == 1 else "sbd.bin"
    ho = ora.load(bfutil.model_path(model or default))
    h = bf.load_model(bfutil.model_path(model)) if model else None
    fn = bf.text_to_words_batch if mode == 1 else bf.text_to_sentences_batch
    try:
        docs = _long_docs(5)
        rnd = random.Random(9)
        text, off = bfutil.gen_workload("config1", 10000)
        raw = text.tobytes()
        lines = [raw[off[d]:off[d + 1]] for d in range(10000)]
        big = (b" " if mode == 1 else b". ").join(rnd.choices(lines, k=24000))[:1 << 20]
        docs += [big, b"", lines[0], big[:200000]]
        want = []
        for b in docs:
            r, o, _, _ = _call(f, (ctypes.c_void_p(ho),), b, 4 * len(b) + 8)
            want.append(o[:r - 1] if r > 0 else b"")
        # (0x10000000: a workspace of 40 chunks -- most of the long documents do not fit and stay with the lane kernel)
        # the single-document calls (reference signatures, byte offsets of every token) on long documents: the same kernels behind run_host
        L = bf.lib()
        g1 = L.TextToWordsWithOffsetsWithModel if mode == 1 else L.TextToSentencesWithOffsetsWithModel
        g1.restype = ctypes.c_int
        g1.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        for b in [docs[0], docs[24], big[:50000], docs[-12]]:
            mx = 4 * len(b) + 8
            assert _call(g1, (), b, mx, (ctypes.c_void_p(h) if h else None,)) == _call(f, (ctypes.c_void_p(ho),), b, mx), (model, mode, len(b), b[:60])
        # (0x08000000: the two-level chain of very long documents from 256 cells on instead of 512 K)
        variants = [0] if h is None else [0, 1 << 12, 0x40000000, 0x10000000 | (1 << 12), 0x10000000 | (3 << 12), 0x08000000, 0x08000000 | (1 << 12)]
        for v in variants:
            if h is not None:
                assert bf.lib().BfSetVariant(ctypes.c_void_p(h), v) >= 0
            out, t_off = fn(docs, h)
            for d, b in enumerate(docs):
                assert out[t_off[d]:t_off[d + 1]].tobytes() == want[d], (model, mode, hex(v), d, len(b), b[:60])
        if h is not None:
            # the position at which the reference's triple buffer fills (FALexTools_t.h:337-340; n triples in these modes, which no text
            # reaches): with the test knob 0x20000000 the buffer holds n / 8, and the long-document path ends every document where the
            # lane kernel does
            res = []
            for v in (0x20000000 | 0x40000000, 0x20000000 | (1 << 12), 0x20000000, 0x28000000 | (1 << 12)):
                assert bf.lib().BfSetVariant(ctypes.c_void_p(h), v) >= 0
                res.append(fn(docs, h))
            cut = 0
            for d, b in enumerate(docs):
                a = res[0][0][res[0][1][d]:res[0][1][d + 1]].tobytes()
                cut += a != want[d]
                for k in (1, 2, 3):
                    assert res[k][0][res[k][1][d]:res[k][1][d + 1]].tobytes() == a, (model, mode, k, d, len(b), b[:60])
            assert mode == 2 or cut > len(docs) // 4
    finally:
        if h:
            bf.free_model(h)
        ora.free(ho)
