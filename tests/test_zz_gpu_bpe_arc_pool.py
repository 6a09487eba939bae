"""GPU parity, BPE documents beyond the per-document arc reserve (bf_kernels.hip k_bpe_big -> bf_seg.h seg_bpe_doc_big).  Written at the
very end of round 2, after the GPU budget was spent: the host form of the same code is tested against the oracle
(tests/test_hypothesis_emu.py::test_bpe_documents_beyond_the_arc_reserve); this file is the device-side check and sorts last so that
the rest of the GPU tier runs before it."""
import ctypes

import numpy as np
import pytest

import bfutil
import blingfire_amd as bf

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("model", [m for m in ("gpt2.bin", "roberta.bin") if bfutil.have_model(m)])
def test_bpe_documents_beyond_the_arc_reserve(model, checker):
    """documents that are mostly one long run of a character whose run-length tokens are in the vocabulary ('-' * 15, '.' * 19,
    '=' * 36, '#' * 93 ...) collect more arcs than the 6 * L + 32 reserved per document: they take the pool path (k_bpe_big) and must
    come out like the reference's, inside a batch of ordinary documents, with and without offsets"""
    h = bf.load_model(bfutil.model_path(model))
    hck = checker.load(bfutil.model_path(model))
    try:
        runs = [ch * n for ch in (b"-", b".", b"=", b"*", b"_", b"#", b"-=", "—".encode("utf-8")) for n in (14, 15, 16, 19, 36, 40, 93, 120, 300)] + [b"-" * 1500]
        docs = bfutil.fuzz_docs(300, seed=3)
        mixed = []
        for k, r in enumerate(runs):
            mixed.append(docs[k % len(docs)])
            mixed.append(r)
            mixed.append(b"Section " + r + b" end of the rule.")
        for max_ids, unk in ((2048, 0), (5, 0), (2048, 262)):
            _compare(h, checker, hck, mixed, max_ids, unk)
        assert bf.lib().BfLastStatus(ctypes.c_void_p(h)) & 2 == 0
        # the offsets API goes through the same pool path
        name = "TextToIdsWithOffsets" if bfutil.have_ref() else "bfo_text_to_ids_with_offsets"
        ids, st, en, off = bf.text_to_ids_with_offsets_batch(h, mixed, 2048, 0)
        for d, b in enumerate(mixed):
            c, gi, gs, ge = checker.with_offsets(hck, b, 2048, 0, name)
            a, z = int(off[d]), int(off[d + 1])
            assert (z - a, ids[a:z].tolist(), st[a:z].tolist(), en[a:z].tolist()) == (c, gi[:c], gs[:c], ge[:c]), (model, d, b[:40])
    finally:
        bf.free_model(h)
        checker.free(hck)
