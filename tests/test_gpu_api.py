"""GPU: C-ABI behaviour around the hot path -- empty batches, empty documents inside a batch, SetModel from memory,
concurrent calls from several host threads (on one handle and on two handles), error codes."""
import ctypes
import threading

import numpy as np
import pytest

import bfutil
import blingfire_amd as bf

pytestmark = pytest.mark.gpu


def test_empty_batch_and_empty_documents():
    h = bf.load_model(bfutil.model_path(bfutil.bert_model_name()))
    try:
        ids, off = bf.text_to_ids_batch(h, [], 16, 100)
        assert len(ids) == 0 and off.tolist() == [0]
        docs = [b"", b"hello world", b"", b"", b"a", b""]
        ids, off = bf.text_to_ids_batch(h, docs, 16, 100)
        ora = bfutil.oracle()
        ho = ora.load(bfutil.model_path(bfutil.bert_model_name()))
        text, doff = bf.pack_docs(docs)
        gids, goff = ora.batch(ho, text, doff, 16, 100)
        ora.free(ho)
        assert np.array_equal(ids, gids) and np.array_equal(off, goff)
        assert off[1] == 0 and off[3] == off[2]
    finally:
        bf.free_model(h)


def test_set_model_from_memory_and_capacity_error():
    img = open(bfutil.model_path("gpt2.bin"), "rb").read()
    L = bf.lib()
    h = L.SetModel(img, len(img))
    assert h
    try:
        assert L.BfModelKind(ctypes.c_void_p(h)) == 3          # BPE-opt
        docs = [b"Hello, world! This is a test."] * 50
        ids, off = bf.text_to_ids_batch(h, docs, 64, 0)
        assert ids[:9].tolist() == [18435, 11, 995, 0, 770, 318, 257, 1332, 13]
        assert np.array_equal(np.diff(off), np.full(50, 9))
        # ids_cap too small -> BF_E_CAPACITY, never a silent truncation
        text, doff = bf.pack_docs(docs)
        small = np.zeros(10, dtype=np.int32)
        id_off = np.zeros(51, dtype=np.int64)
        r = L.TextToIdsBatch(ctypes.c_void_p(h), text.ctypes.data, doff.ctypes.data, 50, small.ctypes.data, 10, id_off.ctypes.data, 64, 0)
        assert r == -3
    finally:
        L.FreeModel(ctypes.c_void_p(h))
    assert L.SetModel(b"garbage-not-a-model", 19) is None or L.SetModel(b"garbage-not-a-model", 19) == 0
    assert L.LoadModel(b"/nonexistent/model.bin") in (None, 0)


def test_small_wordpiece_batch_capacity_error_writes_no_ids():
    """the mapped path of small WordPiece batches (<= 256 documents, 64 KB): ids_cap too small -> BF_E_CAPACITY with the offsets complete (they
    tell the size) and not one id written -- like the regular path (advisor finding of round 3)"""
    L = bf.lib()
    h = bf.load_model(bfutil.model_path(bfutil.bert_model_name()))
    try:
        docs = [b"Hello, world! This is a test."] * 40
        want, want_off = bf.text_to_ids_batch(h, docs, 64, 100)
        text, doff = bf.pack_docs(docs)
        small = np.full(len(want) - 1, -7, dtype=np.int32)
        id_off = np.full(41, -1, dtype=np.int64)
        r = L.TextToIdsBatch(ctypes.c_void_p(h), text.ctypes.data, doff.ctypes.data, 40, small.ctypes.data, len(small), id_off.ctypes.data, 64, 100)
        assert r == -3 and np.array_equal(id_off, want_off) and (small == -7).all()
        exact = np.full(len(want), -7, dtype=np.int32)
        r = L.TextToIdsBatch(ctypes.c_void_p(h), text.ctypes.data, doff.ctypes.data, 40, exact.ctypes.data, len(exact), id_off.ctypes.data, 64, 100)
        assert r == len(want) and np.array_equal(exact, want)
    finally:
        bf.free_model(h)


def test_concurrent_callers():
    """The reference is re-entrant on a loaded handle (README.md:105,215); calls on one handle serialise here,
    different handles run independently -- results must be the same as a serial run."""
    name = bfutil.bert_model_name()
    h1 = bf.load_model(bfutil.model_path(name))
    h2 = bf.load_model(bfutil.model_path(name))
    docs = bfutil.fuzz_docs(400, seed=23)
    want = bf.text_to_ids_batch(h1, docs, 64, 100)
    errs = []

    def work(h, k):
        try:
            for _ in range(5):
                ids, off = bf.text_to_ids_batch(h, docs, 64, 100)
                if not (np.array_equal(ids, want[0]) and np.array_equal(off, want[1])):
                    errs.append("thread %d: mismatch" % k)
                s = docs[k]
                assert bf.text_to_ids(h, s, 32, 100, no_padding=True).tolist() == want[0][want[1][k]:want[1][k + 1]][:32].tolist()
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(h1 if k % 2 == 0 else h2, k)) for k in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    bf.free_model(h1)
    bf.free_model(h2)
    assert not errs, errs


def test_device_api_on_torch_tensors():
    import torch
    name = "xlnet.bin"
    h = bf.load_model(bfutil.model_path(name))
    try:
        docs = bfutil.fuzz_docs(500, seed=29)
        text, off = bf.pack_docs(docs)
        d_text = torch.from_numpy(text.copy()).cuda()
        d_off = torch.from_numpy(off).cuda()
        ids, id_off = bf.text_to_ids_batch_device(h, d_text, d_off, 128, 0)
        torch.cuda.synchronize()
        n = int(id_off[-1].item())
        want = bf.text_to_ids_batch(h, (text, off), 128, 0)
        assert np.array_equal(ids[:n].cpu().numpy(), want[0]) and np.array_equal(id_off.cpu().numpy(), want[1])
        assert bf.lib().BfLastStatus(h) == 0
        ms = bf.last_kernel_ms(h)
        assert len(ms) == 6 and ms[4] > 0
    finally:
        bf.free_model(h)


@pytest.mark.parametrize("model,unk", [(None, 100), ("xlnet.bin", 0)])
def test_device_api_rejects_ranges_outside_the_buffer(model, unk):
    """...BatchDevice precondition (include/blingfiretokdll_amd.h): offsets are relative to d_text and end at total_bytes.  A
    document whose byte range leaves [0, total_bytes] must not be read: it yields no ids and BfLastStatus reports bit 3; the
    other documents are unaffected."""
    import torch
    name = model or bfutil.bert_model_name()
    h = bf.load_model(bfutil.model_path(name))
    try:
        docs = [b"hello world", b"unaffable telescope", b"I saw a girl"]
        text, off = bf.pack_docs(docs)
        good_ids, good_off = bf.text_to_ids_batch(h, (text, off), 32, unk)
        d_text = torch.from_numpy(text.copy()).cuda()
        bad = off.copy()
        bad[-1] += 1000                                    # the last document claims bytes past the end of the buffer
        out_ids, out_off = bf.text_to_ids_batch_device(h, d_text, torch.from_numpy(bad).cuda(), 32, unk)
        torch.cuda.synchronize()
        assert bf.lib().BfLastStatus(ctypes.c_void_p(h)) & 8
        o = out_off.cpu().numpy()
        assert np.array_equal(o[:3], good_off[:3]) and o[3] == o[2]      # the first two documents as usual, the bad one empty
        assert np.array_equal(out_ids[:int(o[2])].cpu().numpy(), good_ids[:int(good_off[2])])
        neg = off.copy() - 5                               # offsets that do not start at 0: the first document begins before the buffer
        out_ids, out_off = bf.text_to_ids_batch_device(h, d_text, torch.from_numpy(neg).cuda(), 32, unk)
        torch.cuda.synchronize()
        assert bf.lib().BfLastStatus(ctypes.c_void_p(h)) & 8
        assert int(out_off[1].item()) == 0
        out_ids, out_off = bf.text_to_ids_batch_device(h, d_text, torch.from_numpy(off).cuda(), 32, unk)     # and a clean call resets the status
        torch.cuda.synchronize()
        assert bf.lib().BfLastStatus(ctypes.c_void_p(h)) == 0
        assert np.array_equal(out_off.cpu().numpy(), good_off)
    finally:
        bf.free_model(h)


def test_reserve_then_no_growth():
    """BfReserve sizes the workspaces once; a later batch within those bounds gives the same ids"""
    h = bf.load_model(bfutil.model_path(bfutil.bert_model_name()))
    try:
        bf.reserve(h, 1000, 1 << 20)
        docs = [b"hello world", b"unaffable telescope"] * 100
        a = bf.text_to_ids_batch(h, docs, 32, 100)
        bf.reserve(h, 10, 100)                             # never shrinks
        b = bf.text_to_ids_batch(h, docs, 32, 100)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    finally:
        bf.free_model(h)


def test_single_document_calls_from_many_threads_are_combined_correctly():
    """text_to_ids_one (bf_capi.cpp): concurrent TextToIds / TextToIdsWithOffsets calls on ONE handle share launches; every caller
    must still get exactly its own answer -- different documents, max_ids and unk values in flight at the same time"""
    name = bfutil.bert_model_name()
    h = bf.load_model(bfutil.model_path(name))
    ora = bfutil.oracle()
    ho = ora.load(bfutil.model_path(name))
    docs = [d for d in bfutil.fuzz_docs(600, seed=41) if len(d) < 3000]
    params = [(64, 100), (8, 100), (64, 7), (300, 100)]
    want = {}
    for k, d in enumerate(docs):
        mx, unk = params[k % 4]
        c, buf = ora.text_to_ids(ho, d, mx, unk)
        want[k] = buf[:c]
    ora.free(ho)
    errs = []

    def work(t):
        try:
            for k in range(t, len(docs), 12):
                mx, unk = params[k % 4]
                got = bf.text_to_ids(h, docs[k], mx, unk, no_padding=True).view(np.int32).tolist()
                if got != want[k]:
                    errs.append("doc %d: %r != %r" % (k, got[:8], want[k][:8]))
                if k % 5 == 0:
                    i, s, e = bf.utf8text_to_ids_with_offsets(h, docs[k], mx, unk, no_padding=True)
                    if i.view(np.int32).tolist() != want[k]:
                        errs.append("doc %d (offsets form) differs" % k)
        except Exception as ex:   # noqa: BLE001
            errs.append(repr(ex))

    ts = [threading.Thread(target=work, args=(t,)) for t in range(12)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    bf.free_model(h)
    assert not errs, errs[:5]


@pytest.mark.parametrize("model", ["bert_base_tok.bin", "xlnet.bin", "gpt2.bin"])
def test_chunked_host_batch_equals_unchunked(model):
    """TextToIdsBatch on host buffers cuts large batches into chunks that flow through pinned staging (bf_capi.cpp run_host_chunked);
    with a tiny chunk size the boundaries fall everywhere -- ids and offsets must be those of the unchunked path and of the oracle,
    including empty documents at chunk edges, a document larger than a chunk, and the capacity error"""
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)
    L = bf.lib()
    L.BfSetHostChunkBytes.restype = ctypes.c_int64
    L.BfSetHostChunkBytes.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    h = bf.load_model(bfutil.model_path(model))
    try:
        docs = bfutil.fuzz_docs(3000, seed=5) + [b""] * 3 + [b"one big document " * 400] + [b"", b"tail"]
        text, off = bf.pack_docs(docs)
        old = L.BfSetHostChunkBytes(ctypes.c_void_p(h), 0)
        assert old == 128 << 20
        want_ids, want_off = bf.text_to_ids_batch(h, (text, off), 48, 7)
        ora = bfutil.oracle()
        ho = ora.load(bfutil.model_path(model))
        gids, goff = ora.batch(ho, text, off, 48, 7)
        ora.free(ho)
        assert np.array_equal(want_ids, gids) and np.array_equal(want_off, goff)
        for chunk in (1, 700, 4096, 50000):
            L.BfSetHostChunkBytes(ctypes.c_void_p(h), chunk)
            for _ in range(2):                                   # twice: staging buffers are reused
                ids, ioff = bf.text_to_ids_batch(h, (text, off), 48, 7)
                assert np.array_equal(ids, want_ids) and np.array_equal(ioff, want_off), (model, chunk)
        # a sub-range of a larger buffer (doc_offsets[0] != 0)
        sub = off[100:2001].copy()
        L.BfSetHostChunkBytes(ctypes.c_void_p(h), 2000)
        ids, ioff = bf.text_to_ids_batch(h, (text, sub), 48, 7)
        assert np.array_equal(ids, want_ids[want_off[100]:want_off[2000]]) and np.array_equal(ioff, want_off[100:2001] - want_off[100])
        # capacity error: offsets are still complete
        small = np.zeros(100, dtype=np.int32)
        ioff = np.zeros(len(off), dtype=np.int64)
        r = L.TextToIdsBatch(ctypes.c_void_p(h), text.ctypes.data, off.ctypes.data, len(off) - 1, small.ctypes.data, 100, ioff.ctypes.data, 48, 7)
        assert r == -3 and np.array_equal(ioff, want_off)
    finally:
        bf.free_model(h)


def test_single_document_calls_from_64_native_threads(tmp_path):
    """tools/single_calls.c: 1 / 4 / 16 / 64 native threads call TextToIds on ONE handle back to back (no interpreter lock between
    them: the lead of the combined launches changes hands thousands of times per second, callers spin and sleep on their own state
    words); every result is compared with what a quiet single-threaded pass returned for the same document"""
    import os
    import subprocess
    exe = os.path.join(bfutil.ROOT, "tools", "single_calls")
    if not os.path.exists(exe):
        pytest.skip("tools/single_calls not built")
    text, off = bfutil.gen_workload("config2", 1500)
    raw = text.tobytes()
    docs = tmp_path / "docs.txt"
    with open(docs, "wb") as f:
        for i in range(1500):
            d = raw[off[i]:off[i + 1]].replace(b"\n", b" ").replace(b"\r", b" ")
            if d.strip():
                f.write(d + b"\n")
    lib = os.path.join(bfutil.ROOT, "blingfire_amd", "libblingfiretokdll.so")
    r = subprocess.run([exe, lib, bfutil.model_path(bfutil.bert_model_name()), str(docs), "0.4"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "64 threads" in r.stdout and " 0 results that differ" in r.stdout.splitlines()[-1], r.stdout


def test_step_diagnostics_name_the_kernels_that_ran():
    """BfTokeniseKernel / BfStepKernels / BfLastKernelMs after a batch: the flat program for a large WordPiece batch, the wave program for a small one"""
    L = bf.lib()
    L.BfTokeniseKernel.restype = ctypes.c_char_p; L.BfTokeniseKernel.argtypes = [ctypes.c_void_p]
    L.BfStepKernels.restype = ctypes.c_char_p; L.BfStepKernels.argtypes = [ctypes.c_void_p]
    h = bf.load_model(bfutil.model_path(bfutil.bert_model_name()))
    try:
        text, off = bfutil.gen_workload("config2", 20000)                     # 2.5 MB: the flat program
        bf.text_to_ids_batch(h, (text, off), 128, 100)
        assert L.BfTokeniseKernel(ctypes.c_void_p(h)) == b"k_wp_flat"
        names = L.BfStepKernels(ctypes.c_void_p(h)).decode()
        for k in ("k_wp_pre", "k_wp_flat", "k_wp_units", "k_wp_count", "k_wp_merge"):
            assert k in names, names
        ms = bf.last_kernel_ms(h)
        assert len(ms) == 6 and 0 < ms[5] <= ms[1] <= ms[4]
        bf.text_to_ids_batch(h, [b"a small batch", b"of two documents"], 128, 100)   # the wave program
        assert L.BfTokeniseKernel(ctypes.c_void_p(h)) == b"k_wp_wave"
        assert "k_wp_flat" not in L.BfStepKernels(ctypes.c_void_p(h)).decode()
    finally:
        bf.free_model(h)
