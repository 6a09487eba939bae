"""CPU: property-based fuzz of the per-lane device programs (compiled for the host by tests/hosttest) against the oracle -- inputs
drawn by hypothesis instead of the seeded generators of bfutil: arbitrary Unicode text, arbitrary bytes (mostly invalid UTF-8),
and text with long runs of one character.  The property is the parity bar itself: same count, same ids, for any max_ids / unk."""
import ctypes

import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import bfutil

MODELS = ["bert_base_tok.bin", "bert_chinese.bin", "xlnet.bin", "xlm_roberta_base.bin", "gpt2.bin", "roberta.bin", "wbd.bin"]

_state = {}


def _handles(model):
    if model not in _state:
        L = ctypes.CDLL(bfutil.HOSTTEST_LIB)
        L.bft_load.restype = ctypes.c_void_p
        L.bft_load.argtypes = [ctypes.c_char_p]
        L.bft_emu_text_to_ids.restype = ctypes.c_int
        L.bft_emu_text_to_ids.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        ora = bfutil.oracle()
        _state[model] = (L, L.bft_load(bfutil.model_path(model).encode()), ora, ora.load(bfutil.model_path(model)))
    return _state[model]


def _check(model, b, mx, unk):
    L, h, ora, ho = _handles(model)
    arr = (ctypes.c_int32 * max(mx, 1))()
    c = L.bft_emu_text_to_ids(h, b, len(b), arr, mx, unk)
    gc, gbuf = ora.text_to_ids(ho, b, mx, unk)
    assert c == gc and list(arr)[:c] == gbuf[:gc], (model, b[:80], mx, unk)


text_st = st.text(alphabet=st.characters(blacklist_categories=("Cs",)), max_size=200)
runs_st = st.lists(st.tuples(st.sampled_from(["a", " ", ".", "é", "中", "▁", "\U0001F600", "##", "ing", " ", "​", "-"]),
                             st.integers(min_value=1, max_value=70)), max_size=12).map(lambda xs: "".join(c * n for c, n in xs))
mx_st = st.sampled_from([0, 1, 2, 7, 64, 512])
unk_st = st.sampled_from([0, 1, 100, 3, 50256])


@pytest.mark.parametrize("model", MODELS)
def test_lane_programs_any_text(model):
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)

    @settings(max_examples=250, deadline=None, suppress_health_check=list(HealthCheck))
    @given(t=st.one_of(text_st, runs_st), mx=mx_st, unk=unk_st)
    def run(t, mx, unk):
        _check(model, t.encode("utf-8"), mx, unk)

    run()


@pytest.mark.parametrize("model", MODELS)
def test_lane_programs_any_bytes(model):
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)

    @settings(max_examples=200, deadline=None, suppress_health_check=list(HealthCheck))
    @given(b=st.binary(max_size=120), mx=mx_st, unk=unk_st)
    def run(b, mx, unk):
        _check(model, b, mx, unk)

    run()


@pytest.mark.parametrize("model", ["gpt2.bin", "roberta.bin"])
def test_bpe_documents_beyond_the_arc_reserve(model):
    """The BPE lane programs reserve 6 * L + 32 arcs per document; a document that is mostly one long run of a character whose run-length
    tokens are in the vocabulary needs more ('-' * 15, '.' * 19, '=' * 36, '#' * 93 with gpt2.bin -- found by the property tests above).
    Such documents take the pool path (bf_seg.h seg_bpe_doc_big; on the device k_bpe_big) and must give the reference's ids; a pool
    that is exhausted is a loud error (-2 here, BF_E_INTERNAL in the product), never wrong ids."""
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)
    L, h, ora, ho = _handles(model)
    L.bft_set_big_pool.argtypes = [ctypes.c_long]
    arr = (ctypes.c_int32 * 4096)()
    for ch in (b"-", b".", b"=", b"*", b"_", b"#", b"-=", b". ", "—".encode("utf-8")):
        for n in list(range(1, 130, 3)) + [200, 513, 2000]:
            b = ch * n
            c = L.bft_emu_text_to_ids(h, b, len(b), arr, 4096, 0)
            gc, gbuf = ora.text_to_ids(ho, b, 4096, 0)
            assert c == gc and list(arr)[:c] == gbuf[:gc], (model, ch, n)
    try:
        L.bft_set_big_pool(1000)
        b = b"-" * 300
        assert L.bft_emu_text_to_ids(h, b, len(b), arr, 4096, 0) == -2
    finally:
        L.bft_set_big_pool(64 << 20)


@pytest.mark.parametrize("model", ["bert_base_tok.bin", "bert_chinese.bin", "xlnet.bin", "xlm_roberta_base.bin", "gpt2.bin"])
def test_lane_programs_offsets_any_text(model):
    """TextToIdsWithOffsets: ids AND the inclusive byte spans of every id, any text / any bytes"""
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)
    L, h, ora, ho = _handles(model)
    f = L.bft_emu_text_to_ids_with_offsets
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int]

    @settings(max_examples=200, deadline=None, suppress_health_check=list(HealthCheck))
    @given(b=st.one_of(text_st.map(lambda t: t.encode("utf-8")), runs_st.map(lambda t: t.encode("utf-8")), st.binary(max_size=80)),
           mx=st.sampled_from([1, 3, 64, 512]), unk=unk_st)
    def run(b, mx, unk):
        i = (ctypes.c_int32 * mx)(); s = (ctypes.c_int32 * mx)(); e = (ctypes.c_int32 * mx)()
        c = f(h, b, len(b), i, s, e, mx, unk)
        assert (c, list(i)[:c], list(s)[:c], list(e)[:c]) == ora.with_offsets(ho, b, mx, unk, "bfo_text_to_ids_with_offsets"), (model, b[:60], mx)

    run()
