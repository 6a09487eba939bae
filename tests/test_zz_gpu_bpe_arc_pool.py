"""GPU parity, BPE documents beyond the per-document arc reserve: bf_kernels.hip k_bpe_seg (bf_bpe_seg_body.h: one wave per document, arcs
from a pool that the host-buffer calls grow on demand).  The same source runs in the wave simulator against the oracle
(tests/test_bpe_seg_emu.py); this file is the device-side check against the compiled reference, up to 10^6 identical characters, and
sorts last so that the rest of the GPU tier runs before it."""
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
    '=' * 36, '#' * 93 ...) collect more arcs than the 6 * L + 32 reserved per document: they take the pool path (k_bpe_seg) and must
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


@pytest.mark.parametrize("model", [m for m in ("gpt2.bin", "roberta.bin") if bfutil.have_model(m)])
@pytest.mark.parametrize("n", [10000, 50000, 1000000])
def test_long_runs_of_one_character_inside_a_batch(model, n, checker):
    """10^4, 5 * 10^4 and 10^6 identical characters (the reference tokenises anything up to FALimits::MaxArrSize: its arc list is a
    std::vector, ..._bpe_t.h:143-144,197) inside a batch of ordinary documents: every id of every document equals the reference's.
    The 10^6 case needs ~350 MB of arcs: TextToIdsBatch grows the pool and runs the batch again."""
    h = bf.load_model(bfutil.model_path(model))
    hck = checker.load(bfutil.model_path(model))
    try:
        docs = bfutil.fuzz_docs(60, seed=11)
        runs = [b"-" * n] if n >= 1000000 else [b"-" * n, b"=" * n, b"see " + b"." * n + b" end", b"ab" * (n // 2)]
        mixed = docs[:30] + runs + docs[30:]
        _compare(h, checker, hck, mixed, 1 << 22, 0)
        _compare(h, checker, hck, mixed, 7, 0)
        assert bf.lib().BfLastStatus(ctypes.c_void_p(h)) & (2 | 16 | 32 | 64) == 0
    finally:
        bf.free_model(h)
        checker.free(hck)


def test_device_batch_reports_a_full_pool_per_document():
    """...BatchDevice cannot allocate: with a pool too small for one document that document gets 0 ids and BfLastStatus bit 64, every
    other document its ids; after BfSetBpePoolBytes the same batch is complete"""
    import torch
    model = "gpt2.bin"
    ck = bfutil.reference() if bfutil.have_ref() else bfutil.oracle()
    h = bf.load_model(bfutil.model_path(model))
    hck = ck.load(bfutil.model_path(model))
    try:
        docs = [b"hello world", b"=" * 200000, b"the quick brown fox", b"#" * 50]
        text, off = bf.pack_docs(docs)
        gids, goff = ck.batch(hck, text, off, 1 << 20, 0)
        assert bf.lib().BfSetBpePoolBytes(ctypes.c_void_p(h), 1 << 20) == 64 << 20
        d_text = torch.from_numpy(text).cuda(); d_off = torch.from_numpy(off).cuda()
        ids, id_off = bf.text_to_ids_batch_device(h, d_text, d_off, 1 << 20, 0); torch.cuda.synchronize()
        id_off = id_off.cpu().numpy(); cnt, gcnt = np.diff(id_off), np.diff(goff)
        assert bf.lib().BfLastStatus(ctypes.c_void_p(h)) & 64
        assert cnt[1] == 0 and cnt[0] == gcnt[0] and cnt[2] == gcnt[2] and cnt[3] == gcnt[3]
        bf.lib().BfSetBpePoolBytes(ctypes.c_void_p(h), 256 << 20)
        ids, id_off = bf.text_to_ids_batch_device(h, d_text, d_off, 1 << 20, 0); torch.cuda.synchronize()
        assert bf.lib().BfLastStatus(ctypes.c_void_p(h)) & (2 | 16 | 32 | 64) == 0
        assert np.array_equal(id_off.cpu().numpy(), goff) and np.array_equal(ids.cpu().numpy()[:len(gids)], gids)
    finally:
        bf.free_model(h)
        ck.free(hck)
