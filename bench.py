#!/usr/bin/env python3
"""bench.py -- throughput of the batch TextToIds hot path on N MI355X (one process per GPU).

A "step" = one pass of the whole pipeline (prep -> tokenise -> scan -> compact) over the rank's shard of
the synthetic corpus, with the input text and document offsets already resident in HBM.  The corpus is
statically range-sharded (rank r owns documents [r*D, (r+1)*D)), no data-path collective exists; the only
torch.distributed traffic is the timing barrier and a MAX-reduce of the elapsed time.

Default workload = the north-star headline (BASELINE.json): bert_base_tok.bin, ~512-byte documents,
1.25 M documents per GPU (= the 10 M-document corpus at 8 GPUs), max_ids 512, unk 100.
`--workload config2` selects BASELINE.json configs[1] (1 M docs ~128 bytes).

Prints ONE JSON line (rank 0) with the driver's contract keys plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="headline512", choices=["headline512", "config2", "config3", "config4", "config5"])
    ap.add_argument("--docs-per-gpu", type=int, default=0, help="override the shard size (documents per GPU)")
    ap.add_argument("--model", default="")
    ap.add_argument("--variant", type=int, default=-1, help="kernel variant (experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-docs", type=int, default=0)
    ap.add_argument("--verify", type=int, default=2000, help="documents checked bit-exact against the CPU checker before timing")
    args = ap.parse_args()

    import numpy as np
    import torch
    import bfutil
    import blingfire_amd as bf

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # BF_BENCH_SHARE_GPU=1 (testing only): several ranks on ONE GPU over gloo, to exercise the N>1 code path on a 1-GPU box
        share = os.environ.get("BF_BENCH_SHARE_GPU") == "1"
        dev_index = 0 if share else local_rank
        torch.cuda.set_device(dev_index)
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
    else:
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)

    wl = bfutil.WORKLOADS[args.workload]
    model_name = args.model or wl["model"] or bfutil.bert_model_name()
    max_ids, unk = wl["max_ids"], wl["unk"]
    default_docs = {"headline512": 1250000, "config2": 1000000, "config3": 1000000, "config4": 1250000, "config5": 1250000}[args.workload]
    docs_per_gpu = args.docs_per_gpu or default_docs

    # ---- the rank's shard, generated on the host and made resident in HBM before any timing
    text, off = bfutil.gen_corpus(docs_per_gpu, first_doc=rank * docs_per_gpu, **wl["gen"])
    d_text = torch.from_numpy(text).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    h = bf.load_model(bfutil.model_path(model_name))
    if args.variant >= 0:
        bf.lib().BfSetVariant(h, args.variant)
    ndocs = docs_per_gpu
    total_bytes = int(off[-1])
    cap = max(1, min(2 * (total_bytes + ndocs), ndocs * max_ids))
    out_ids = torch.empty(cap, dtype=torch.int32, device=dev)
    out_off = torch.empty(ndocs + 1, dtype=torch.int64, device=dev)

    def step():
        bf.text_to_ids_batch_device(h, d_text, d_off, max_ids, unk, out_ids=out_ids, out_off=out_off)

    # ---- parity gate on a prefix of the shard (the checker is never inside the timed region)
    verified = 0
    if args.verify > 0:
        nv = min(args.verify, ndocs)
        lib_path, _kind = bfutil.checker_lib_path()
        _, _, gids, goff = bfutil.cpu_text_to_ids_batch(lib_path, bfutil.model_path(model_name), text[:off[nv]], off[:nv + 1], max_ids, unk)
        step()
        torch.cuda.synchronize(dev)
        g_off = out_off[:nv + 1].cpu().numpy()
        g_ids = out_ids[:int(g_off[-1])].cpu().numpy()
        if not (np.array_equal(g_off, goff) and np.array_equal(g_ids, gids)):
            raise SystemExit("bench: GPU ids differ from the CPU checker on the verification prefix -- refusing to time")
        verified = nv

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    kms = np.zeros(5, dtype=np.float64)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kms += np.array(bf.last_kernel_ms(h), dtype=np.float64)   # HIP events recorded on the launch stream
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kms /= max(args.steps, 1)
    n_ids = int(out_off[-1].item())
    status = bf.lib().BfLastStatus(h)

    if rank == 0:
        docs_total = ndocs * max(world, 1)
        value = docs_total * args.steps / elapsed
        gb_in = total_bytes * max(world, 1) * args.steps / elapsed / 1e9
        # algorithmic bytes of one launch of the dominant kernel: n_in + 4*n_ids + 16 per document
        alg_bytes = total_bytes + 4 * n_ids + 16 * ndocs
        tok_ms = float(kms[1])
        achieved = alg_bytes / (tok_ms * 1e-3) / 1e9 if tok_ms > 0 else 0.0
        traffic = None
        try:   # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            ent = tj.get("%s/%s/%d" % (args.workload, model_name, ndocs))
            if ent:
                traffic = ent["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        res = {
            "metric": "docs/sec", "value": value, "unit": "docs/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s: %s TextToIds, %d docs/GPU, %.0f B/doc avg, max_ids %d, unk %d" % (
                args.workload, model_name, ndocs, total_bytes / ndocs, max_ids, unk),
                "model_file": model_name, "docs_per_gpu": ndocs, "bytes_per_gpu": total_bytes, "ids_per_gpu": n_ids,
                "sharding": "static contiguous document ranges, no collective"},
            "gb_input_per_sec": gb_in,
            "ids_per_sec": n_ids * max(world, 1) * args.steps / elapsed,
            "kernel_ms": {"prep": float(kms[0]), "tokenise": tok_ms, "scan": float(kms[2]), "compact": float(kms[3]), "total": float(kms[4])},
            "roofline": {"bound": "hbm", "kernel": "tokenise (%s)" % {0: "k_lex_wp_flat", 1: "k_seg_unigram_ring"}.get(bf.lib().BfModelKind(h), "k_bpe_fused"), "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes},
            "verified_docs": verified, "status": status,
        }
        if world == 1 and not args.no_cpu_baseline:
            lib_path, kind = bfutil.checker_lib_path()
            cores = os.cpu_count() or 1
            ns = args.cpu_sample_docs or min(ndocs, 25000 * cores)
            secs, _, _, _ = bfutil.cpu_text_to_ids_batch(lib_path, bfutil.model_path(model_name), text[:off[ns]], off[:ns + 1], max_ids, unk,
                                                        nthreads=cores, passes=1, want_ids=False)
            res["cpu_baseline"] = {"value": ns / secs, "unit": "docs/s", "cores": cores, "kind": kind,
                                   "sample": "first %d documents of the same shard, one TextToIds call per document, %d threads sharing one model handle, %.2f s wall" % (ns, cores, secs)}
        print(json.dumps(res), flush=True)
    bf.free_model(h)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
