"""Maximum-size style inputs: single large documents (every lane-per-document kernel walks them sequentially), a run of
100,000 letters (MaxTokenLength splits it), large random byte strings (BPE: segments far beyond the in-kernel limits -> the
full path; more than 4096 arcs -> the slow sort path), all bit-exact against the CPU checker."""
import numpy as np
import pytest

import bfutil

bf = pytest.importorskip("blingfire_amd")
pytestmark = pytest.mark.gpu


def _docs():
    rng = np.random.default_rng(7)
    text, off = bfutil.gen_corpus(1, seed=99, mean=200000, sd=10, minlen=150000, maxlen=250000)
    big_english = text.tobytes()
    rnd = rng.integers(1, 256, size=20000, dtype=np.uint8).tobytes()
    multibyte = ("日本語のテキスト。Ünïcödé wörds ümlaut. " * 1500).encode()
    return [big_english, b"a" * 100000, rnd, multibyte, b"x", (b"word " * 20000)]


@pytest.mark.parametrize("model", ["bert_base_cased_tok.bin", "gpt2.bin", "xlm_roberta_base.bin", "roberta.bin"])
def test_large_documents(model):
    lib_path, _ = bfutil.checker_lib_path()
    docs = _docs()
    text, off = bf.pack_docs(docs)
    h = bf.load_model(bfutil.model_path(model))
    try:
        for max_ids in (1 << 20, 1000):
            ids, id_off = bf.text_to_ids_batch(h, (text, off), max_ids, 3)
            _, _, gids, goff = bfutil.cpu_text_to_ids_batch(lib_path, bfutil.model_path(model), text, off, max_ids, 3)
            assert np.array_equal(id_off, goff), (model, max_ids, id_off, goff)
            assert np.array_equal(ids, gids), (model, max_ids)
    finally:
        bf.free_model(h)
