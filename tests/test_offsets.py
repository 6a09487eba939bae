"""TextToIdsWithOffsets (reference tokdll:1562-1609): CPU tests pin the oracle's offsets restatement to the compiled
reference and run the lane programs' span reporting on the host; the GPU test compares the HIP pipeline with the checker."""
import ctypes

import numpy as np
import pytest

import bfutil

MODELS = ["bert_base_tok.bin", "bert_base_cased_tok.bin", "bert_chinese.bin", "wbd.bin", "gpt2.bin", "roberta.bin", "bpe_example.bin",
          "xlnet.bin", "xlnet_nonorm.bin", "laser100k.bin", "xlm_roberta_base.bin", "laser500k.bin"]


def _docs(n, seed):
    return list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(n, seed=seed)


def test_readme_known_answer_offsets():
    """reference README.md:232-268: the 49 tokens' surface strings recovered from the offsets"""
    if not bfutil.have_model("xlm_roberta_base.bin"):
        pytest.skip("model not present")
    s = ("Autophobia, also called monophobia, isolophobia, or eremophobia, is the specific phobia of isolation. I saw a girl with a "
         "telescope. Я увидел девушку с телескопом.").encode("utf-8")
    ora = bfutil.oracle()
    h = ora.load(bfutil.model_path("xlm_roberta_base.bin"))
    c, ids, st, en = ora.with_offsets(h, s, 128, 0, "bfo_text_to_ids_with_offsets")
    ora.free(h)
    assert c == 49 and st[0] == -1                     # the first token starts with the dummy prefix, whose offset is -1 (tokdll:1387)
    toks = [s[max(a, 0):b + 1].decode("utf-8") for a, b in zip(st, en)]
    assert toks[:6] == ["Auto", "pho", "bia", ",", " also", " called"]
    assert toks[-8:] == [" дев", "у", "шку", " с", " теле", "скоп", "ом", "."]


@pytest.mark.skipif(not bfutil.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("model", MODELS)
def test_oracle_offsets_vs_live_reference(model):
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)
    ora, ref = bfutil.oracle(), bfutil.reference()
    ho, hr = ora.load(bfutil.model_path(model)), ref.load(bfutil.model_path(model))
    for k, b in enumerate(_docs(800, 91)):
        mx = (64, 3, 512)[k % 3]
        unk = (0, 100)[k % 2]
        assert ora.with_offsets(ho, b, mx, unk, "bfo_text_to_ids_with_offsets") == ref.with_offsets(hr, b, mx, unk, "TextToIdsWithOffsets"), (model, b[:60])
    ora.free(ho)
    ref.free(hr)


@pytest.mark.parametrize("model", MODELS)
def test_lane_program_spans_on_host_match_oracle(model):
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)
    L = ctypes.CDLL(bfutil.HOSTTEST_LIB)
    L.bft_load.restype = ctypes.c_void_p
    L.bft_load.argtypes = [ctypes.c_char_p]
    L.bft_free.argtypes = [ctypes.c_void_p]
    f = L.bft_emu_text_to_ids_with_offsets
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    h = L.bft_load(bfutil.model_path(model).encode())
    ora = bfutil.oracle()
    ho = ora.load(bfutil.model_path(model))
    for k, b in enumerate(_docs(2500, 37)):
        mx = (512, 3, 64)[k % 3]
        unk = (100, 0)[k % 2]
        n = max(mx, 1)
        i, s, e = (ctypes.c_int32 * n)(), (ctypes.c_int32 * n)(), (ctypes.c_int32 * n)()
        c = f(h, b, len(b), i, s, e, mx, unk)
        assert (c, list(i)[:c], list(s)[:c], list(e)[:c]) == ora.with_offsets(ho, b, mx, unk, "bfo_text_to_ids_with_offsets"), (model, b[:60], mx)
    ora.free(ho)
    L.bft_free(h)


@pytest.mark.gpu
@pytest.mark.parametrize("model", MODELS)
def test_gpu_offsets_match_checker(model):
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)
    import blingfire_amd as bf
    ck = bfutil.reference() if bfutil.have_ref() else bfutil.oracle()
    name = "TextToIdsWithOffsets" if bfutil.have_ref() else "bfo_text_to_ids_with_offsets"
    h = bf.load_model(bfutil.model_path(model))
    hck = ck.load(bfutil.model_path(model))
    try:
        docs = _docs(1200, 53)
        # WordPiece models in unit form: the wave program's offsets instance (what a batch this small takes by default: 5); 4: the flat program
        # (every larger batch); variant 2: the lane-per-document kernels
        for mx, unk, variant in ((256, 100, 3), (3, 0, 3), (256, 100, 4), (3, 0, 4), (256, 100, 2)):
            if variant != 3 and bf.lib().BfModelKind(h) != 0:
                continue
            bf.lib().BfSetVariant(h, variant)                     # 3 = the default of a fresh handle
            ids, st, en, off = bf.text_to_ids_with_offsets_batch(h, docs, mx, unk)
            for d, b in enumerate(docs):
                want = ck.with_offsets(hck, b, mx, unk, name)
                got = (int(off[d + 1] - off[d]), ids[off[d]:off[d + 1]].tolist(), st[off[d]:off[d + 1]].tolist(), en[off[d]:off[d + 1]].tolist())
                assert got == want, (model, d, b[:60], mx)
        # the reference wrapper's single-document form (zero padded)
        s = "Hello unaffable world ▁ ünï".encode("utf-8")
        i1, s1, e1 = bf.utf8text_to_ids_with_offsets(h, s, 32, 100, no_padding=True)
        c, wi, ws, we = ck.with_offsets(hck, s, 32, 100, name)
        assert (i1.astype(np.int64).tolist(), s1.tolist(), e1.tolist()) == ([x & 0xffffffff for x in wi], ws, we)
    finally:
        bf.free_model(h)
        ck.free(hck)


@pytest.mark.gpu
@pytest.mark.parametrize("workload,ndocs", [("headline512", 50000), ("config2", 20000)])
def test_gpu_flat_offsets_on_the_metric_corpus(workload, ndocs):
    """The flat program's OFFS instance on the corpora it is timed on: TextToIdsWithOffsetsBatch with the default variant
    -- batches of this size take the flat program, several ranges -- against the compiled reference's TextToIdsWithOffsets per document: ids,
    first bytes and last bytes, at max_ids 512 and 16 (truncation, tokdll:1263-1297,1308-1310)"""
    if not bfutil.have_ref():
        pytest.skip("oracle/_ref not built")
    import blingfire_amd as bf
    model = bfutil.bert_model_name()
    text, off = bfutil.gen_workload(workload, ndocs)
    raw = text.tobytes()
    ref = bfutil.reference()
    h = bf.load_model(bfutil.model_path(model))
    hr = ref.load(bfutil.model_path(model))
    try:
        for mx in (512, 16):
            ids, st, en, id_off = bf.text_to_ids_with_offsets_batch(h, (text, off), mx, 100)
            bf.lib().BfTokeniseKernel.restype = ctypes.c_char_p
            assert bf.lib().BfTokeniseKernel(ctypes.c_void_p(h)) == b"k_wp_flat"
            step = 1 if mx == 512 else 7                      # every document at 512; every seventh at 16 (the reference call is the slow part)
            for d in range(0, ndocs, step):
                b = raw[off[d]:off[d + 1]]
                c, wi, ws, we = ref.with_offsets(hr, b, mx, 100, "TextToIdsWithOffsets")
                a, z = int(id_off[d]), int(id_off[d + 1])
                assert z - a == c and ids[a:z].tolist() == wi[:c] and st[a:z].tolist() == ws[:c] and en[a:z].tolist() == we[:c], (workload, d, mx, b[:60])
    finally:
        bf.free_model(h)
        ref.free(hr)
