"""CPU, world_size 2 over gloo: the multi-GPU path is static contiguous range sharding with no data-path collective.
Each rank regenerates ITS shard from (seed, first_doc) alone and tokenises it (here with the CPU oracle standing in for
the kernel); the concatenation over ranks must equal the single-process result on the whole corpus, and the timing
protocol of bench.py (barrier + MAX-reduce) must work."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bfutil

NDOCS = 4000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = NDOCS // world
    text, off = bfutil.gen_corpus(per, first_doc=rank * per, nthreads=1, **bfutil.WORKLOADS["config2"]["gen"])
    ora = bfutil.oracle()
    h = ora.load(bfutil.model_path("bert_base_cased_tok.bin"))
    ids, id_off = ora.batch(h, text, off, 512, 100)
    ora.free(h)
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    counts = torch.tensor([len(ids), int(off[-1])], dtype=torch.int64)
    gathered = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered, counts)
    q.put((rank, ids, id_off, float(t.item()), [g.tolist() for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_concatenate_to_the_whole():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    text, off = bfutil.gen_corpus(NDOCS, nthreads=1, **bfutil.WORKLOADS["config2"]["gen"])
    ora = bfutil.oracle()
    h = ora.load(bfutil.model_path("bert_base_cased_tok.bin"))
    ids, id_off = ora.batch(h, text, off, 512, 100)
    ora.free(h)
    cat = np.concatenate([r[1] for r in res])
    assert np.array_equal(cat, ids)
    assert res[0][2][-1] + res[1][2][-1] == id_off[-1]
    assert all(r[3] == float(world) for r in res)               # MAX over ranks
    assert res[0][4] == res[1][4] and sum(g[1] for g in res[0][4]) == off[-1]
