"""GPU parity, SentencePiece-style branch (Unigram-LM / BPE / BPE with merge ranks): the HIP pipeline through the
C-ABI vs the CPU checker.  Bar: bit-exact ids and counts (the Unigram scores are float32 bit patterns accumulated
in float64 in the reference's exact order, so even the floating-point path must agree exactly)."""
import numpy as np
import pytest

import bfutil
import blingfire_amd as bf

pytestmark = pytest.mark.gpu

SP_MODELS = [m for m in ("gpt2.bin", "roberta.bin", "bpe_example.bin", "xlnet.bin", "xlnet_nonorm.bin", "laser100k.bin",
                         "xlm_roberta_base.bin", "laser500k.bin", "uri100k.bin", "uri100kint.bin", "laser50k.bin", "bpe_example2.bin") if bfutil.have_model(m)]


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


@pytest.mark.parametrize("model", SP_MODELS)
def test_adversarial_and_fuzz(model, checker):
    h = bf.load_model(bfutil.model_path(model))
    hck = checker.load(bfutil.model_path(model))
    try:
        docs = list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(2500, seed=13)
        # unk values: ordinary, equal to real token ids (the merge quirk of ..._bpe_t.h:217-225), negative and beyond 2^20
        # (both switch the BPE kernel's lane-local window off)
        for max_ids, unk in ((2048, 0), (3, 0), (64, 3), (1, 1), (2048, 262), (2048, -7), (2048, 3000000)):
            _compare(h, checker, hck, docs, max_ids, unk)
    finally:
        bf.free_model(h)
        checker.free(hck)


@pytest.mark.parametrize("model,workload,ndocs", [("gpt2.bin", "config3", 6000), ("xlm_roberta_base.bin", "config4", 100000),
                                                  ("laser500k.bin", "config5", 100000)])
def test_corpus_bit_exact(model, workload, ndocs, checker):
    """the configuration's own corpus (configs 4 / 5: the multilingual generator of SURVEY.md section 8d, 100 k documents of all
    seven script buckets): id counts and EVERY id as arrays against the CPU checker (round 6: no longer through the per-document hash)"""
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)
    wl = bfutil.WORKLOADS[workload]
    text, off = bfutil.gen_workload(workload, ndocs)
    max_ids, unk = wl["max_ids"], wl["unk"]
    lib_path, _ = bfutil.checker_lib_path()
    _, want_ids, want_off = bfutil.cpu_ids_compact(lib_path, bfutil.model_path(model), text, off, max_ids, unk)
    h = bf.load_model(bfutil.model_path(model))
    try:
        for variant in ((3, 6, 3 | 0x20) if workload != "config3" else (3,)):      # Unigram: the cut form, the round-4 kernels, the cut form with the short way out
            bf.lib().BfSetVariant(h, variant)
            ids, id_off = bf.text_to_ids_batch(h, (text, off), max_ids, unk)
            assert np.array_equal(id_off, want_off), (model, variant)
            bad = np.nonzero(ids != want_ids)[0]
            assert len(bad) == 0, "variant %d: first differing id at %d (document %d)" % (variant, bad[0], int(np.searchsorted(want_off, bad[0], side="right") - 1))
    finally:
        bf.free_model(h)


@pytest.mark.parametrize("model", [m for m in ("xlm_roberta_base.bin", "laser500k.bin", "xlnet.bin", "laser100k.bin") if bfutil.have_model(m)])
def test_unigram_cut_form_rings_and_periods(model, checker):
    """round 6, the Unigram cut form (k_uni_cut + k_uni_ids): words longer than the record ring (entries of 15 and 16 symbols back to back, Thai
    and CJK runs without a blank, unknown runs of 20 .. 200 symbols: the spill path), documents of one symbol, every emission period from 1 to
    64 trips, the short way out on and off, 10 and 13 waves per CU -- all against the CPU checker; and the offsets API, which still takes the
    forward / backward kernels of round 4"""
    docs = list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(1500, seed=71)
    docs += [("x" * n + " " + "\u0e01\u0e32\u0e23" * n + " internationalization" * (n % 7)).encode("utf-8") for n in range(1, 120, 7)]
    docs += [("\U000F0000" * n + " a " + "\u4e2d\u6587" * n).encode("utf-8") for n in (1, 20, 33, 70, 200)]
    docs += [b"a", b" ", "\u2581".encode("utf-8"), ("pneumonoultramicroscopicsilicovolcanoconiosis " * 12).encode("utf-8")]
    h = bf.load_model(bfutil.model_path(model))
    hck = checker.load(bfutil.model_path(model))
    try:
        for variant in (3, 3 | (1 << 8), 3 | (2 << 8), 3 | (64 << 8), 3 | 0x20, 3 | 0x20 | (1 << 8), 3 | 0x10, 3 | (10 << 16), 3 | (13 << 16) | 0x20, 6):
            bf.lib().BfSetVariant(h, variant)
            _compare(h, checker, hck, docs, 1024, 3)
            _compare(h, checker, hck, docs, 4, 0)
    finally:
        bf.free_model(h)
        checker.free(hck)


@pytest.mark.parametrize("model,workload,ndocs", [("gpt2.bin", "config3", 6000)])
def test_corpus_hash_bit_exact(model, workload, ndocs, checker):
    """the per-document 64-bit hash bench.py used before round 3 stays pinned on one corpus"""
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)
    wl = bfutil.WORKLOADS[workload]
    text, off = bfutil.gen_workload(workload, ndocs)
    max_ids, unk = wl["max_ids"], wl["unk"]
    lib_path, _ = bfutil.checker_lib_path()
    _, c_counts, c_hash = bfutil.cpu_doc_hashes(lib_path, bfutil.model_path(model), text, off, max_ids, unk)
    h = bf.load_model(bfutil.model_path(model))
    try:
        ids, id_off = bf.text_to_ids_batch(h, (text, off), max_ids, unk)
        assert np.array_equal(np.diff(id_off), c_counts)
        g_hash = bfutil.ids_hash_np(ids, id_off)
        bad = np.nonzero(g_hash != c_hash)[0]
        assert len(bad) == 0, "first differing document %d: %r" % (bad[0], bytes(text[off[bad[0]]:off[bad[0] + 1]]))
    finally:
        bf.free_model(h)


def test_no_dummy_prefix_switch(checker):
    """SetNoDummyPrefix (tokdll:1669-1679) changes the prologue at run time"""
    import ctypes
    model = "xlnet.bin"
    h = bf.load_model(bfutil.model_path(model))
    hck = checker.load(bfutil.model_path(model))
    try:
        setter = getattr(checker.lib, "SetNoDummyPrefix", None) or getattr(checker.lib, "bfo_set_no_dummy_prefix")
        setter.argtypes = [ctypes.c_void_p, ctypes.c_int]
        docs = list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(300, seed=17)
        for flag in (1, 0):
            bf.change_settings_dummy_prefix(h, not flag)
            setter(ctypes.c_void_p(hck), flag)
            _compare(h, checker, hck, docs, 256, 0)
    finally:
        bf.free_model(h)
        checker.free(hck)


@pytest.mark.parametrize("model", ["xlnet.bin", "xlm_roberta_base.bin", "laser500k.bin"])
def test_long_unknown_runs(model, checker):
    """unknown runs around and beyond the 4095-position limit of the packed Viterbi record (bf_seg.h uni_rec): escape hops in the
    backward pass; also documents whose slots start at every alignment of the 16-byte record groups"""
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)
    docs = []
    for n in (4094, 4095, 4096, 4097, 8191, 12290):
        docs.append(("hello " + "\U000F0000" * n + " world " + "\U000F0000" * 3 + "x").encode("utf-8"))
    for k in range(1, 40):
        docs.append(("a" * k + " \U000F0000" * (k % 5) + " the end").encode("utf-8"))
    h = bf.load_model(bfutil.model_path(model))
    hck = checker.load(bfutil.model_path(model))
    try:
        _compare(h, checker, hck, docs, 8192, 3)
        _compare(h, checker, hck, docs, 5, 0)
    finally:
        bf.free_model(h)
        checker.free(hck)


@pytest.mark.parametrize("model", [m for m in ("gpt2.bin", "bpe_example.bin", "bpe_example2.bin") if bfutil.have_model(m)])
def test_bpe_wave_program_variant(model, checker):
    """The BPE wave program (bf_bpe_wave_body.h) is the default path of the models its load-time analysis admits; BfSetVariant bit 0x40
    switches it off (the lane-per-document kernels alone).  Both give the ids of the CPU checker on adversarial input, fuzz and the
    config-3 corpus"""
    h = bf.load_model(bfutil.model_path(model))
    hck = checker.load(bfutil.model_path(model))
    try:
        docs = list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(2500, seed=17)
        text, off = bfutil.gen_workload("config3", 20000)
        gids, goff = checker.batch(hck, text, off, 2048, 0)
        # the wave program as shipped (HOME form + word table, round 6), the lane kernels alone, without the table, the in-order form of rounds 4..5 with and without it
        for variant in (3, 3 | 0x40, 3 | 0x100000, 3 | (8 << 8), 3 | (8 << 8) | 0x100000, 3 | (9 << 8)):
            bf.lib().BfSetVariant(h, variant)
            for max_ids, unk in ((2048, 0), (3, 0), (64, 3), (1, 1), (2048, 262)):
                _compare(h, checker, hck, docs, max_ids, unk)
            ids, id_off = bf.text_to_ids_batch(h, (text, off), 2048, 0)
            assert np.array_equal(id_off, goff) and np.array_equal(ids, gids)
    finally:
        bf.free_model(h)
        checker.free(hck)


@pytest.mark.parametrize("model", [m for m in ("xlm_roberta_base.bin", "gpt2.bin", "laser100k.bin") if bfutil.have_model(m)])
def test_prologue_instances_agree(model, checker):
    """the _sp prologue as shipped (eight bytes per lane, compiled for eight waves per SIMD), compiled for six / seven waves (BfSetVariant
    bits 24..27) and in its byte-per-lane form (bit 0x80): the same ids"""
    h = bf.load_model(bfutil.model_path(model))
    hck = checker.load(bfutil.model_path(model))
    try:
        docs = list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(1200, seed=23)
        for variant in (3, 3 | (6 << 24), 3 | (7 << 24), 3 | 0x80):
            if bf.lib().BfSetVariant(h, variant) == -5:           # BF_E_UNSUPPORTED: a measurement instance, compiled with BF_EXPERIMENTS only
                assert variant >> 24
                continue
            _compare(h, checker, hck, docs, 512, 3)
    finally:
        bf.free_model(h)
        checker.free(hck)
