"""IdsToText (reference tokdll:1689-1745): the oracle's restatement is pinned to the compiled
reference on the reference's own .i2w files; the GPU tests call the product's IdsToText / IdsToTextBatch."""
import ctypes
import random

import numpy as np
import pytest

import bfutil

I2W_MODELS = ["gpt2.i2w", "bert_base_cased_tok.i2w", "xlnet.i2w", "laser100k.i2w", "roberta.i2w"]


def _oracle_fn():
    ora = bfutil.oracle()
    f = ora.lib.bfo_ids_to_text
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    ora.lib.bfo_i2w_count.argtypes = [ctypes.c_void_p]
    return ora, f


def _cases(cnt, n_cases, seed):
    rng = random.Random(seed)
    for t in range(n_cases):
        n = rng.choice([0, 1, 2, 5, 20, 100, 700])
        ids = [rng.randrange(0, cnt) for _ in range(n)]
        if t % 7 == 0 and n:
            ids[rng.randrange(n)] = rng.choice([-1, cnt, cnt + 5, 0, 1, 2, 3])
        if t % 11 == 0 and n > 2:
            ids[0] = ids[1] = 220 if cnt > 50000 else ids[0]            # gpt2 / roberta: id 220 is " " -- the leading-space rule
        yield ids, rng.choice([8192, 8, 0, 64]), t % 2


def _run(fn, h, ids, mx, skip, as_bool=False):
    arr = (ctypes.c_int32 * max(len(ids), 1))(*ids)
    o = ctypes.create_string_buffer(b"\x7f" * 8200)
    r = fn(h, arr, len(ids), o, mx, bool(skip) if as_bool else skip)
    return r, (o.raw[:r] if 0 < r <= mx else b"")


def test_known_answer_gpt2():
    """reference README.md:232-268 round trip: the ids of 'Hello World!' (gpt2, without the +1 of the .bin) decode to the text"""
    ora, f = _oracle_fn()
    h = ora.load(bfutil.model_path("gpt2.i2w"))
    r, out = _run(f, ctypes.c_void_p(h), [15496, 2159, 0], 64, 0)
    assert out == b"Hello World!\x00"
    ora.free(h)


@pytest.mark.skipif(not bfutil.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("model", I2W_MODELS)
def test_oracle_vs_live_reference(model):
    ora, f = _oracle_fn()
    ref = bfutil.reference()
    g = ref.lib.IdsToText
    g.restype = ctypes.c_int
    g.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_bool]
    ho, hr = ora.load(bfutil.model_path(model)), ref.load(bfutil.model_path(model))
    cnt = ora.lib.bfo_i2w_count(ho)
    for ids, mx, skip in _cases(cnt, 1500, 5):
        assert _run(f, ctypes.c_void_p(ho), ids, mx, skip) == _run(g, ctypes.c_void_p(hr), ids, mx, skip, True), (model, ids[:8], mx, skip)
    ora.free(ho)


@pytest.mark.gpu
@pytest.mark.parametrize("model", I2W_MODELS)
def test_gpu_ids_to_text(model):
    import blingfire_amd as bf
    ora, f = _oracle_fn()
    ho = ora.load(bfutil.model_path(model))
    cnt = ora.lib.bfo_i2w_count(ho)
    h = bf.load_model(bfutil.model_path(model))
    L = bf.lib()
    try:
        cases = list(_cases(cnt, 300, 9))
        for ids, mx, skip in cases:
            assert _run(L.IdsToText, ctypes.c_void_p(h), ids, mx, skip, True) == _run(f, ctypes.c_void_p(ho), ids, mx, skip), (model, ids[:8], mx, skip)
        # the batch form = the single calls, concatenated without terminators; a sequence with an unknown id yields nothing
        for skip in (0, 1):
            flat = np.array([i for ids, _, _ in cases for i in ids], dtype=np.int32)
            off = np.zeros(len(cases) + 1, dtype=np.int64)
            np.cumsum([len(ids) for ids, _, _ in cases], out=off[1:])
            text, t_off = bf.ids_to_text_batch(h, flat, off, bool(skip))
            for d, (ids, _, _) in enumerate(cases):
                r, out = _run(f, ctypes.c_void_p(ho), ids, 8192, skip)
                want = out[:-1] if r > 0 else b""
                assert text[t_off[d]:t_off[d + 1]].tobytes() == want, (model, d, ids[:8], skip)
        assert bf.text_to_ids(h, "hello", 8).sum() == 0        # an [i2w]-only model has no tokenizer: TextToIds returns 0 ids
    finally:
        bf.free_model(h)
        ora.free(ho)


@pytest.mark.gpu
def test_gpu_round_trip_gpt2():
    """tokenise a corpus with gpt2.bin, detokenise with gpt2.i2w (ids of the .bin carry the reference's +1, ldbsrc/gpt2/README.TXT)"""
    import blingfire_amd as bf
    ht, hd = bf.load_model(bfutil.model_path("gpt2.bin")), bf.load_model(bfutil.model_path("gpt2.i2w"))
    try:
        text, off = bfutil.gen_corpus(3000, seed=11, minlen=32, maxlen=600, loguniform=True)
        ids, id_off = bf.text_to_ids_batch(ht, (text, off), 4096, 0)
        out, t_off = bf.ids_to_text_batch(hd, ids, id_off, False)
        raw = text.tobytes()
        for d in range(3000):
            src = raw[off[d]:off[d + 1]].decode("utf-8", "replace")
            got = out[t_off[d]:t_off[d + 1]].tobytes().decode("utf-8", "replace")
            assert " ".join(got.split()) == " ".join(src.split()), (d, src[:60], got[:60])
    finally:
        bf.free_model(ht)
        bf.free_model(hd)
