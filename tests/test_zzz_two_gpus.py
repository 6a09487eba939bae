"""Two PHYSICAL devices: BfSetDevices over devices 0 and 1, and the bench with two `nccl` ranks / two in-process shards -- the code paths
that have never run anywhere, because every box this repository has seen had one GPU.  Skips on such a box.  The file sorts last so that a first-ever failure
here does not hide the rest of the tier behind `-x`."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import bfutil

bf = pytest.importorskip("blingfire_amd")


def _batch(model, ndocs, seed):
    return bfutil.gen_corpus(ndocs, seed=seed, mean=200, sd=80, minlen=1, maxlen=900)


@pytest.mark.gpu
def test_two_physical_devices():
    """The code paths that have never run anywhere: BfSetDevices on two PHYSICAL devices and the bench with two `nccl` ranks.
    Skips on a box with one GPU -- the first multi-GPU box to run this tier exercises them."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (this box has %d)" % torch.cuda.device_count())
    model = bfutil.bert_model_name()
    text, off = _batch(model, 20000, 78)
    h = bf.load_model(bfutil.model_path(model))
    try:
        ids1, off1 = bf.text_to_ids_batch(h, (text, off), 128, 100)
        bf.set_devices(h, [0, 1])
        ids2, off2 = bf.text_to_ids_batch(h, (text, off), 128, 100)
        assert np.array_equal(off2, off1) and np.array_equal(ids2, ids1)
        a = bf.text_to_ids_with_offsets_batch(h, (text, off), 128, 100)
        bf.set_devices(h, [0])
        b = bf.text_to_ids_with_offsets_batch(h, (text, off), 128, 100)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    finally:
        bf.free_model(h)
    env = dict(os.environ)
    env.pop("BF_BENCH_SHARE_GPU", None)
    cmd = [sys.executable, os.path.join(bfutil.ROOT, "bench.py"), "--gpus", "2", "--docs", "200000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra-timings"]
    for extra in ([], ["--inproc"]):
        out = subprocess.run(cmd + extra, capture_output=True, text=True, env=env, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        j = json.loads(out.stdout.strip().splitlines()[-1])
        assert j["n_gpus"] == 2 and j["verified_docs"] == 200000 and j["status"] == 0
        assert j.get("backend") == ("inproc" if extra else "nccl")
        assert len({r["pci"] for r in j["ranks"]}) == 2             # two different devices really ran


