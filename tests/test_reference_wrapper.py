"""GPU: the reference's OWN Python wrapper (dist-pypi/blingfire/__init__.py, staged unchanged under oracle/_ref/ by
oracle/Makefile -- never committed) runs on top of the product library: the file is placed in a scratch package directory next
to a copy of blingfire_amd/libblingfiretokdll.so, exactly the "replace the .so" deployment INTEGRATION.md describes, and the same
wrapper on top of the compiled reference (oracle/_ref) is the expected answer for every call."""
import importlib.util
import os
import shutil

import numpy as np
import pytest

import bfutil

WRAPPER = os.path.join(bfutil.ROOT, "oracle", "_ref", "blingfire", "__init__.py")

pytestmark = pytest.mark.gpu


def _load(tmp, name, so_path):
    d = os.path.join(tmp, name, "blingfire")
    os.makedirs(d)
    shutil.copy(WRAPPER, os.path.join(d, "__init__.py"))
    shutil.copy(so_path, os.path.join(d, "libblingfiretokdll.so"))      # the wrapper loads this file name from its own directory
    spec = importlib.util.spec_from_file_location("blingfire_" + name, os.path.join(d, "__init__.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def wrappers(tmp_path_factory):
    if not os.path.exists(WRAPPER) or not bfutil.have_ref():
        pytest.skip("oracle/_ref (compiled reference + its wrapper) is not staged")
    import blingfire_amd as bf
    tmp = str(tmp_path_factory.mktemp("refwrap"))
    return _load(tmp, "prod", bf.LIB_PATH), _load(tmp, "ref", bfutil.REF_LIB)


TEXTS = ["Hello, world! This is a test of unaffable tokenisation.", "I saw a girl with a telescope. Я увидел девушку с телескопом.",
         "Autophobia, also called monophobia, isolophobia, or eremophobia, is the specific phobia of isolation.", "a", " ", "don't U.S.A. e-mail 3,000.50",
         "好好好 ok", "Sergei Alonichau I saw a girl with a \ttelescope."]


def test_model_free_calls(wrappers):
    prod, ref = wrappers
    assert prod.get_blingfiretok_version() == ref.get_blingfiretok_version()
    for t in TEXTS:
        assert prod.text_to_words(t) == ref.text_to_words(t)
        assert prod.text_to_sentences(t) == ref.text_to_sentences(t)
        assert prod.text_to_words_with_offsets(t) == ref.text_to_words_with_offsets(t)
        assert prod.text_to_sentences_and_offsets(t) == ref.text_to_sentences_and_offsets(t)
        if t.strip():
            assert prod.normalize_spaces(t) == ref.normalize_spaces(t)
            w = ref.text_to_words(t)
            assert np.array_equal(prod.text_to_hashes(w, 2, 2000000), ref.text_to_hashes(w, 2, 2000000))


@pytest.mark.parametrize("model,unk", [("bert_base_tok.bin", 100), ("bert_base_cased_tok.bin", 100), ("xlm_roberta_base.bin", 3), ("gpt2.bin", 0), ("xlnet.bin", 0)])
def test_text_to_ids_through_the_reference_wrapper(wrappers, model, unk):
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)
    prod, ref = wrappers
    hp, hr = prod.load_model(bfutil.model_path(model)), ref.load_model(bfutil.model_path(model))
    try:
        for t in TEXTS:
            for max_len, no_padding in ((128, False), (128, True), (5, False)):
                assert np.array_equal(prod.text_to_ids(hp, t, max_len, unk, no_padding), ref.text_to_ids(hr, t, max_len, unk, no_padding))
            a, b = prod.utf8text_to_ids_with_offsets(hp, t.encode("utf-8"), 64, unk), ref.utf8text_to_ids_with_offsets(hr, t.encode("utf-8"), 64, unk)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            # documented deviation: for a token made of the dummy prefix alone the reference adds the UTF-8 size of the byte
            # BEFORE the caller's string to -1 (tokdll:1527) -- here the last byte of the bytes object's header, its cached hash: -1 when it is
            # 0xFF (not hashed yet), 0 .. 2 otherwise; the product reports -1
            lone = (a[1] == -1) & (a[2] == -1)
            assert np.array_equal(a[2][~lone], b[2][~lone]) and (b[2][lone] <= 2).all()
            assert prod.text_to_words_with_model(None, t) == ref.text_to_words_with_model(None, t)
    finally:
        prod.free_model(hp)
        ref.free_model(hr)


def test_ids_to_text_through_the_reference_wrapper(wrappers):
    prod, ref = wrappers
    i2w = bfutil.model_path("bert_base_cased_tok.i2w")
    tok = bfutil.model_path("bert_base_cased_tok.bin")
    if not (os.path.exists(i2w) and os.path.exists(tok)):
        pytest.skip("bert_base_cased_tok model pair not present")
    hp, hr = prod.load_model(tok), ref.load_model(tok)
    ip, ir = prod.load_model(i2w), ref.load_model(i2w)
    try:
        for t in TEXTS:
            ids = ref.text_to_ids(hr, t, 64, 100, True)
            assert np.array_equal(prod.text_to_ids(hp, t, 64, 100, True), ids)
            assert prod.ids_to_text(ip, ids) == ref.ids_to_text(ir, ids)
            assert prod.ids_to_text(ip, ids, skip_special_tokens=False) == ref.ids_to_text(ir, ids, skip_special_tokens=False)
    finally:
        for m, h in ((prod, hp), (prod, ip), (ref, hr), (ref, ir)):
            m.free_model(h)
