"""GPU parity, WordPiece path: the HIP pipeline (through the C-ABI) vs the CPU checker on the same inputs.
Bar: bit-exact ids and counts (integer work)."""
import ctypes
import numpy as np
import pytest

import bfutil
import blingfire_amd as bf

pytestmark = pytest.mark.gpu

WP_MODELS = [m for m in ("bert_base_tok.bin", "bert_base_cased_tok.bin", "bert_chinese.bin", "wbd.bin", "sbd.bin") if bfutil.have_model(m)]


@pytest.fixture(scope="module")
def checker():
    return bfutil.reference() if bfutil.have_ref() else bfutil.oracle()


def _compare(h, ck, hck, docs, max_ids, unk):
    text, off = bf.pack_docs(docs)
    ids, id_off = bf.text_to_ids_batch(h, (text, off), max_ids, unk)
    gids, goff = ck.batch(hck, text, off, max_ids, unk)
    if not np.array_equal(id_off, goff) or not np.array_equal(ids, gids):
        for d in range(len(docs)):
            a = ids[id_off[d]:id_off[d + 1]]
            b = gids[goff[d]:goff[d + 1]]
            if not np.array_equal(a, b):
                raise AssertionError("doc %d %r (max %d unk %d): gpu %s != ref %s" % (d, docs[d][:80], max_ids, unk, a[:40], b[:40]))
        raise AssertionError("offset arrays differ")


@pytest.mark.parametrize("model", WP_MODELS)
def test_adversarial_and_fuzz(model, checker):
    h = bf.load_model(bfutil.model_path(model))
    hck = checker.load(bfutil.model_path(model))
    try:
        docs = list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(3000, seed=11)
        for max_ids, unk in ((512, 100), (3, 100), (1, 0), (64, 7)):
            _compare(h, checker, hck, docs, max_ids, unk)
    finally:
        bf.free_model(h)
        checker.free(hck)


def test_single_doc_api_untouched_tail(checker):
    """TextToIds writes only `count` ids: the rest of the caller's array is untouched (tokdll:1098-1101)."""
    model = bfutil.bert_model_name()
    h = bf.load_model(bfutil.model_path(model))
    hck = checker.load(bfutil.model_path(model))
    try:
        import ctypes
        for b in bfutil.ADVERSARIAL:
            for max_ids in (0, 3, 64):
                n = max(max_ids, 1)
                arr = (ctypes.c_int32 * n)(*([-7] * n))
                c = bf.lib().TextToIds(ctypes.c_void_p(h), b, len(b), arr, max_ids, 100)
                gc, gbuf = checker.text_to_ids(hck, b, max_ids, 100)
                assert (c, list(arr)) == (gc, gbuf), (b, max_ids)
        assert bf.lib().TextToIds(None, b"abc", 3, None, 4, 0) == 0
        assert bf.text_to_ids(h, "", 8).tolist() == [0] * 8
    finally:
        bf.free_model(h)
        checker.free(hck)


def test_config2_corpus_bit_exact(checker):
    """BASELINE.json configs[1] at a size the CPU checker finishes in seconds; golden from the compiled reference."""
    model = bfutil.bert_model_name()
    text, off = bfutil.gen_corpus(20000, **bfutil.WORKLOADS["config2"]["gen"])
    lib_path, _ = bfutil.checker_lib_path()
    _, _, gids, goff = bfutil.cpu_text_to_ids_batch(lib_path, bfutil.model_path(model), text, off, 512, 100)
    h = bf.load_model(bfutil.model_path(model))
    try:
        ids, id_off = bf.text_to_ids_batch(h, (text, off), 512, 100)
        assert np.array_equal(id_off, goff)
        assert np.array_equal(ids, gids)
    finally:
        bf.free_model(h)


def test_headline_corpus_properties():
    """Full-size-independent properties on the 512-byte north-star documents: determinism, shard invariance
    (tokenising any contiguous shard == the slice of the whole), truncation prefix property."""
    model = bfutil.bert_model_name()
    text, off = bfutil.gen_corpus(30000, **bfutil.WORKLOADS["headline512"]["gen"])
    h = bf.load_model(bfutil.model_path(model))
    try:
        ids, id_off = bf.text_to_ids_batch(h, (text, off), 512, 100)
        ids2, id_off2 = bf.text_to_ids_batch(h, (text, off), 512, 100)
        assert np.array_equal(ids, ids2) and np.array_equal(id_off, id_off2)
        lo, hi = 7000, 19000
        sids, soff = bf.text_to_ids_batch(h, (text, off[lo:hi + 1]), 512, 100)
        assert np.array_equal(sids, ids[id_off[lo]:id_off[hi]])
        assert np.array_equal(soff, id_off[lo:hi + 1] - id_off[lo])
        tids, toff = bf.text_to_ids_batch(h, (text, off), 16, 100)
        cnt = np.minimum(np.diff(id_off), 16)
        assert np.array_equal(np.diff(toff), cnt)
        for d in range(0, 30000, 997):
            assert np.array_equal(tids[toff[d]:toff[d + 1]], ids[id_off[d]:id_off[d] + cnt[d]])
    finally:
        bf.free_model(h)


@pytest.mark.parametrize("model", [m for m in ("bert_base_tok.bin", "bert_base_cased_tok.bin", "bert_chinese.bin") if bfutil.have_model(m)])
def test_lane_per_document_kernels_on_unit_form_models(model, checker):
    """the BERT lexers take the wave program (bf_wave.h) for plain ids; variant 2 keeps the lane-per-document kernels (bf_lex.h: what the
    offsets API, TextToWords and lexers outside the unit form run) under the same parity bar on the same inputs"""
    h = bf.load_model(bfutil.model_path(model))
    hck = checker.load(bfutil.model_path(model))
    try:
        bf.lib().BfSetVariant(h, 2)
        docs = list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(2000, seed=13)
        for max_ids, unk in ((512, 100), (3, 100)):
            _compare(h, checker, hck, docs, max_ids, unk)
    finally:
        bf.free_model(h)
        checker.free(hck)


def test_wave_program_long_words_and_documents(checker):
    """what the wave program treats specially: words of 64 .. 2000 letters (window / chunk edges, the max-length cut at 300), long runs of
    one-character tokens, a 200 KB document among short ones, documents of one byte"""
    import random
    rnd = random.Random(7)
    alpha = "abcdefghijklmnopqrstuvwxyz"
    docs = []
    for L in [63, 64, 65, 127, 128, 129, 299, 300, 301, 511, 512, 513, 600, 1023, 1024, 1025, 2000]:
        docs += [("a" * L).encode(), (" " + "b" * L + " c").encode(), ("x y " + "é" * L).encode(), "".join(rnd.choice(alpha) for _ in range(L)).encode(),
                 ("好" * L).encode(), ("." * L).encode(), (" " * L).encode(), ("[UNK]" * L).encode(), ("[UN" * L).encode()]
    docs.append(" ".join("".join(rnd.choice(alpha) for _ in range(rnd.randint(1, 12))) for _ in range(30000)).encode())
    docs += [bytes([rnd.randrange(32, 127)]) for _ in range(300)]
    for model in [m for m in ("bert_base_tok.bin", "bert_chinese.bin") if bfutil.have_model(m)]:
        h = bf.load_model(bfutil.model_path(model))
        hck = checker.load(bfutil.model_path(model))
        try:
            for max_ids, unk in ((1 << 20, 100), (64, 100)):
                _compare(h, checker, hck, docs, max_ids, unk)
        finally:
            bf.free_model(h)
            checker.free(hck)


def test_flat_wave_and_lane_programs_agree(checker):
    """the three ways a WordPiece batch can take -- the flat program (bf_flat.h: what large batches take; BfSetVariant 4 = every batch), the wave
    program (5 = never the flat one: what small batches and the documents the flat program hands back take) and the lane-per-document kernels
    (2) -- give the checker's ids on the same inputs; 3 = the default choice by batch size"""
    import random
    rnd = random.Random(23)
    for model in [m for m in ("bert_base_tok.bin", "bert_base_cased_tok.bin", "bert_chinese.bin") if bfutil.have_model(m)]:
        h = bf.load_model(bfutil.model_path(model))
        hck = checker.load(bfutil.model_path(model))
        try:
            docs = list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(1500, seed=19) + [("a" * 700 + " b").encode(), ("x " * 600).encode()]
            text, off = bfutil.gen_workload("headline512", 3000)
            raw = text.tobytes()
            docs += [raw[off[d]:off[d + 1]] for d in range(3000)]
            # what the flat program hands back or treats apart: runs of more than 48 bytes, '[', words of 17 .. 48 bytes, words with characters outside
            # ASCII, thousands of tiny and empty documents, invalid UTF-8 at any place
            docs += [b"a" * 49, b"b" * 700 + b" tail", b"with [UNK] inside", b"x" * 48, (b"q" * 300 + b" ") * 5, b"["]
            docs += [("über" + "x" * k).encode() + b" " + b"y" * (17 + k) + b" fin" for k in range(0, 31)]
            docs += [rnd.choice([b"a", b"", b"", b"to", b".", b" ", b"\xc3\xa9", b"\xff", b"ab"]) for _ in range(4000)]
            for i in range(300):
                body = bytearray((b"some words, and more " * 60)[:rnd.randint(0, 1100)])
                if i % 2 == 0 and body:
                    at = rnd.randint(0, len(body))
                    body[at:at] = rnd.choice([b"\xff", b"\xc3", b"\xe2\x82", b"\xf0\x9f\x98", b"\x80", b"\xed\xa0\x80"])
                docs.append(bytes(body))
            for variant in (4, 5, 3, 2):
                bf.lib().BfSetVariant(h, variant)
                for max_ids, unk in ((512, 100), (5, 7)):
                    _compare(h, checker, hck, docs, max_ids, unk)
        finally:
            bf.free_model(h)
            checker.free(hck)


def test_flat_program_batch_not_fit(checker):
    """a batch with a document of more than 4 MiB is not taken by the flat program (k_wp_pre): every document goes to the wave program, the answer is the same"""
    model = bfutil.bert_model_name()
    h = bf.load_model(bfutil.model_path(model))
    hck = checker.load(bfutil.model_path(model))
    try:
        big = (b"word " * 900000)[:(1 << 22) + 5]
        docs = [b"small one", big, b"small two"] + bfutil.fuzz_docs(100, seed=2)
        bf.lib().BfSetVariant(h, 4)
        _compare(h, checker, hck, docs, 1 << 22, 100)
    finally:
        bf.free_model(h)
        checker.free(hck)


@pytest.mark.parametrize("workload", ["headline512", "config3"])
def test_bench_generators_pinned_to_the_reference(workload):
    """50 k documents of the generators bench.py times -- the metric's 512-byte corpus and config 3's 32-2048-byte mix -- through the
    C-ABI: every id equals the compiled reference's (the oracle port when the reference binary is absent).  bench.py's verify gate
    makes the same comparison on the whole shard; this pins it in the GPU tier as well."""
    wl = bfutil.WORKLOADS[workload]
    model = wl["model"] or bfutil.bert_model_name()
    if not bfutil.have_model(model):
        pytest.skip(model + " not present")
    text, off = bfutil.gen_workload(workload, 50000)
    lib_path, _ = bfutil.checker_lib_path()
    _, gids, goff = bfutil.cpu_ids_compact(lib_path, bfutil.model_path(model), text, off, wl["max_ids"], wl["unk"])
    h = bf.load_model(bfutil.model_path(model))
    try:
        ids, id_off = bf.text_to_ids_batch(h, (text, off), wl["max_ids"], wl["unk"])
        assert np.array_equal(id_off, goff)
        assert np.array_equal(ids, gids)
    finally:
        bf.free_model(h)


@pytest.mark.gpu
def test_flat_program_output_smaller_than_the_ids():
    """ids_cap too small on the device call through the flat program: BfLastStatus bit 0, the id offsets are complete (they say what was needed), the
    documents that lie wholly inside the buffer have their ids, nothing is written behind it; the host call answers BF_E_CAPACITY"""
    import torch
    model = bfutil.bert_model_name()
    h = bf.load_model(bfutil.model_path(model))
    ck = bfutil.reference() if bfutil.have_ref() else bfutil.oracle()
    hck = ck.load(bfutil.model_path(model))
    try:
        text, off = bfutil.gen_workload("config2", 3000)
        want_ids, want_off = ck.batch(hck, text, off, 512, 100)
        bf.lib().BfSetVariant(h, 4)
        dev = torch.device("cuda", 0)
        dt, do = torch.from_numpy(text).to(dev), torch.from_numpy(off).to(dev)
        for cap in (int(want_off[-1]) - 1, int(want_off[-1]) // 2, 7):
            out = torch.full((cap + 64,), -7, dtype=torch.int32, device=dev)
            ido = torch.empty(len(off), dtype=torch.int64, device=dev)
            r = bf.lib().TextToIdsBatchDevice(ctypes.c_void_p(h), dt.data_ptr(), do.data_ptr(), len(off) - 1, len(text), out.data_ptr(), cap, ido.data_ptr(), 512, 100,
                                              ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            assert r == 0
            torch.cuda.synchronize(dev)
            assert bf.lib().BfLastStatus(ctypes.c_void_p(h)) & 1
            g_off, g = ido.cpu().numpy(), out.cpu().numpy()
            assert np.array_equal(g_off, want_off)
            nfit = int(np.searchsorted(want_off, cap, side="right")) - 1          # documents [0, nfit) end at or before cap
            assert np.array_equal(g[:want_off[nfit]], want_ids[:want_off[nfit]])
            assert (g[cap:] == -7).all()
        ids = np.zeros(7, dtype=np.int32); id_off = np.zeros(len(off), dtype=np.int64)
        r = bf.lib().TextToIdsBatch(ctypes.c_void_p(h), text.ctypes.data, off.ctypes.data, len(off) - 1, ids.ctypes.data, 7, id_off.ctypes.data, 512, 100)
        assert r == -3
    finally:
        bf.free_model(h)
        ck.free(hck)


@pytest.mark.gpu
def test_flat_program_chunk_of_513_tokens():
    """a chunk of 512 one-byte tokens behind a run that ends with the chunk before it (513 tokens on one chunk's list) while a record of an earlier
    word waits: the case tools/stress_flat_emu.py found (tests/test_flat_emu.py has it for the simulator)"""
    model = bfutil.bert_model_name()
    h = bf.load_model(bfutil.model_path(model))
    ck = bfutil.reference() if bfutil.have_ref() else bfutil.oracle()
    hck = ck.load(bfutil.model_path(model))
    try:
        docs = []
        for tail in (b"zqxjkvw", b"caf\xc3\xa9s", b"word"):
            first = (b"zqxjkvw " + b"some words and " * 40)[:512 - len(tail) - 1] + b" " + tail
            docs += [first, b"." * 600, b"after", first + b"." * 1100 + b" " + tail]
        docs = docs * 40                                   # (several ranges, every alignment of the pattern to a range)
        text, off = bf.pack_docs(docs)
        want_ids, want_off = ck.batch(hck, text, off, 4096, 100)
        for variant in (4, 5):
            bf.lib().BfSetVariant(h, variant)
            ids, id_off = bf.text_to_ids_batch(h, (text, off), 4096, 100)
            assert np.array_equal(id_off, want_off) and np.array_equal(ids, want_ids), variant
    finally:
        bf.free_model(h)
        ck.free(hck)
