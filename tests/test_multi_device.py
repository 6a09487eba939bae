"""Multi-GPU behind the API (SURVEY.md section 8b "SetDevices", 8e): BfSetDevices range-shards the host-buffer batch calls of a handle
over devices, no collective.  CPU: the range split (BfShardRanges is pure host arithmetic).  GPU: G logical shards on ONE device return
the bytes G = 1 returns (SURVEY.md 8e "Measurability": one device per box), through the per-range handles as well; two ranks of
bench.py sharing the device (BF_BENCH_SHARE_GPU=1) run the library under torch.distributed."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import bfutil

bf = pytest.importorskip("blingfire_amd")


def test_shard_ranges_are_contiguous_balanced_and_exhaustive():
    rng = np.random.default_rng(5)
    for trial in range(200):
        nd = int(rng.integers(0, 400))
        lens = rng.integers(0, 60, size=nd)
        if trial % 7 == 0 and nd:
            lens[rng.integers(0, nd)] = 50000                 # one huge document
        off = np.zeros(nd + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        off += int(rng.integers(0, 3)) * 17                  # a batch need not start at byte 0
        for G in (1, 2, 3, 8, 64):
            b = bf.shard_ranges(off, G)
            assert b[0] == 0 and b[-1] == nd and np.all(np.diff(b) >= 0), (off, G, b)
            total = int(off[-1] - off[0])
            for g in range(1, G):
                # no other document boundary is closer to g/G of the text
                target = int(off[0]) + total * g // G
                best = np.min(np.abs(off - target)) if nd >= 0 else 0
                assert abs(int(off[b[g]]) - target) == best or b[g] == b[g - 1], (off.tolist(), G, g, b.tolist())


def _batch(model, ndocs, seed):
    if model.startswith("bert"):
        return bfutil.gen_corpus(ndocs, seed=seed, mean=200, sd=80, minlen=1, maxlen=900)
    return bfutil.gen_corpus(ndocs, seed=seed, mean=150, sd=60, minlen=1, maxlen=700, multibyte=True)


@pytest.mark.gpu
@pytest.mark.parametrize("model,max_ids,unk", [(None, 128, 100), ("gpt2.bin", 512, 0), ("xlm_roberta_base.bin", 64, 3)])
def test_logical_shards_on_one_device_return_the_same_bytes(model, max_ids, unk):
    model = model or bfutil.bert_model_name()
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)
    text, off = _batch(model, 5000, 77)
    h = bf.load_model(bfutil.model_path(model))
    try:
        ids1, off1 = bf.text_to_ids_batch(h, (text, off), max_ids, unk)
        for G in (2, 3):
            bf.set_devices(h, [0] * G)
            idsg, offg = bf.text_to_ids_batch(h, (text, off), max_ids, unk)
            assert np.array_equal(offg, off1) and np.array_equal(idsg, ids1), (model, G)
            a = bf.text_to_ids_with_offsets_batch(h, (text, off), max_ids, unk)
            # the per-range handles, driven by the caller (what bench.py --inproc does with device-resident shards)
            b = bf.shard_ranges(off, G)
            parts = []
            for g in range(G):
                hg = bf.shard_handle(h, g)
                assert hg is not None
                lo, hi = int(b[g]), int(b[g + 1])
                pi, po = bf.text_to_ids_batch(hg, (text[off[lo]:off[hi]], off[lo:hi + 1] - off[lo]), max_ids, unk)
                parts.append(pi)
            assert bf.shard_handle(h, G) is None
            assert np.array_equal(np.concatenate(parts), ids1)
            bf.set_devices(h, [0])
            s = bf.text_to_ids_with_offsets_batch(h, (text, off), max_ids, unk)
            assert all(np.array_equal(x, y) for x, y in zip(a, s)), (model, G, "offsets form")
        # an empty batch and a batch of fewer documents than shards
        bf.set_devices(h, [0, 0, 0, 0])
        e_ids, e_off = bf.text_to_ids_batch(h, (text[:0], off[:1]), max_ids, unk)
        assert len(e_ids) == 0 and e_off.tolist() == [0]
        t_ids, t_off = bf.text_to_ids_batch(h, (text[:off[2]], off[:3]), max_ids, unk)
        assert np.array_equal(t_ids, ids1[:off1[2]]) and np.array_equal(t_off, off1[:3])
    finally:
        bf.free_model(h)


@pytest.mark.gpu
def test_settings_changed_after_set_devices_reach_every_shard():
    """SetNoDummyPrefix after BfSetDevices: every range of a sharded batch is tokenised with the new setting (the ranges' handles follow
    the parent's settings), i.e. exactly what one device returns with that setting -- and differently from the old setting"""
    model = "xlm_roberta_base.bin"
    if not bfutil.have_model(model):
        pytest.skip(model)
    text, off = _batch(model, 3000, 5)
    h = bf.load_model(bfutil.model_path(model))
    try:
        with_prefix, _ = bf.text_to_ids_batch(h, (text, off), 64, 3)
        bf.change_settings_dummy_prefix(h, False)
        one_ids, one_off = bf.text_to_ids_batch(h, (text, off), 64, 3)
        assert not np.array_equal(one_ids, with_prefix)
        bf.change_settings_dummy_prefix(h, True)
        bf.set_devices(h, [0, 0, 0])
        bf.change_settings_dummy_prefix(h, False)                  # after the shards exist
        g_ids, g_off = bf.text_to_ids_batch(h, (text, off), 64, 3)
        assert np.array_equal(g_off, one_off) and np.array_equal(g_ids, one_ids)
    finally:
        bf.free_model(h)


@pytest.mark.gpu
def test_set_devices_rejects_devices_the_box_does_not_have():
    import torch
    h = bf.load_model(bfutil.model_path(bfutil.bert_model_name()))
    try:
        with pytest.raises(RuntimeError):
            bf.set_devices(h, [0, torch.cuda.device_count()])
        ids, _ = bf.text_to_ids_batch(h, [b"still works"], 16, 100)
        assert len(ids) == 2
    finally:
        bf.free_model(h)


def _bench(extra, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    cmd = [sys.executable, os.path.join(bfutil.ROOT, "bench.py"), "--docs", "40000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra-timings"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_two_ranks_of_the_bench_share_the_device():
    """torch.distributed with two ranks through the PRODUCT (tests/test_sharding_gloo.py runs the same protocol on the CPU with the oracle)"""
    j = _bench(["--gpus", "2"], {"BF_BENCH_SHARE_GPU": "1"})
    assert j["n_gpus"] == 2 and len(j["ranks"]) == 2 and j["verified_docs"] == 40000 and j["status"] == 0
    assert sum(r["docs"] for r in j["ranks"]) == 40000


@pytest.mark.gpu
def test_in_process_shards_of_the_bench():
    """bench.py --inproc: one process, BfSetDevices, a thread per range on its per-range handle with device-resident shards"""
    j = _bench(["--gpus", "2", "--inproc"], {"BF_BENCH_SHARE_GPU": "1"})
    assert j["n_gpus"] == 2 and len(j["ranks"]) == 2 and j["verified_docs"] == 40000 and j["status"] == 0
    assert j["config"]["launcher"] == "inproc"


@pytest.mark.gpu
def test_concurrent_calls_on_a_sharded_handle():
    """two threads call TextToIdsBatch on ONE sharded handle at the same time (documented as safe: host-buffer calls serialise): each gets its own
    batch's ids -- a range's ids stay in its handle's buffers from its kernels to its copy out, and nobody else may use them meanwhile"""
    import threading
    model = bfutil.bert_model_name()
    ta, oa = _batch(model, 4000, 11)
    tb, ob = _batch(model, 3500, 12)
    h = bf.load_model(bfutil.model_path(model))
    try:
        wa = bf.text_to_ids_batch(h, (ta, oa), 128, 100)
        wb = bf.text_to_ids_batch(h, (tb, ob), 128, 100)
        bf.set_devices(h, [0, 0, 0])
        bad = []

        def work(text, off, want, n):
            for _ in range(n):
                ids, ido = bf.text_to_ids_batch(h, (text, off), 128, 100)
                if not (np.array_equal(ido, want[1]) and np.array_equal(ids, want[0])):
                    bad.append(1)

        ts = [threading.Thread(target=work, args=(ta, oa, wa, 6)), threading.Thread(target=work, args=(tb, ob, wb, 6)),
              threading.Thread(target=work, args=(ta[:oa[50]], oa[:51], (wa[0][:wa[1][50]], wa[1][:51]), 20))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not bad
    finally:
        bf.free_model(h)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--inproc"]])
def test_eight_ranks_of_the_bench_on_one_device(extra):
    """the N = 8 forms of the bench (ranks over torch.distributed; one process with BfSetDevices) before hardware with 8 devices shows up: eight
    disjoint ranges that cover the corpus, every document verified, status 0 (BF_BENCH_SHARE_GPU=1: all on device 0, over gloo)"""
    env = dict(os.environ)
    env["BF_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(bfutil.ROOT, "bench.py"), "--docs", "200000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra-timings", "--gpus", "8"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads(out.stdout.strip().splitlines()[-1])
    assert j["n_gpus"] == 8 and len(j["ranks"]) == 8 and j["verified_docs"] == 200000 and j["status"] == 0
    docs = sorted((r.get("first_doc", None), r["docs"]) for r in j["ranks"])
    assert sum(d for _, d in docs) == 200000
    if all(f is not None for f, _ in docs):                      # the ranges are disjoint and cover the corpus
        at = 0
        for f, d in docs:
            assert f == at
            at += d
    assert j.get("backend") in ("gloo", "nccl", "inproc")


def test_more_gpus_than_devices_is_refused():
    """`bench.py --gpus 8` on a box with fewer devices exits non-zero with the stated message instead of reporting an 8-GPU number from fewer (CPU test:
    this container has none)"""
    env = dict(os.environ)
    env.pop("BF_BENCH_SHARE_GPU", None)
    for extra in ([], ["--inproc"]):
        out = subprocess.run([sys.executable, os.path.join(bfutil.ROOT, "bench.py"), "--gpus", "8"] + extra, capture_output=True, text=True, env=env, timeout=600)
        assert out.returncode != 0
        assert "GPU(s)" in out.stderr
