#!/usr/bin/env python3
"""bench.py -- throughput of the batch TextToIds hot path on N MI355X (one process per GPU).

A "step" = one pass of the whole pipeline (prep -> tokenise -> scan -> compact) over the rank's shard of the synthetic
corpus, with the input text and document offsets already resident in HBM.  The corpus has a FIXED size (strong scaling,
BASELINE.json north_star: "throughput on a synthetic 10M-doc / ~512-byte-per-doc corpus is reported at 1/2/4/8 GPUs"):
rank r of G owns documents [floor(r*N/G), floor((r+1)*N/G)); no data-path collective exists, the only torch.distributed
traffic is the timing barrier, a MAX-reduce of the elapsed time and the gather of the per-rank facts.

    python bench.py                      # N=1: the metric's configuration (bert_base_tok.bin, 10 M documents of ~512 bytes)
    python bench.py --gpus 4             # spawns 4 ranks itself (torch.distributed.run) -- or exits non-zero if the box has < 4 GPUs
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 ... bench.py --gpus 4     # the driver's form

`n_gpus` in the result is the number of ranks that actually ran (WORLD_SIZE), each on its own device (listed under
`ranks`); `--gpus` that disagrees with WORLD_SIZE is an error.  Before timing, EVERY document of the shard is compared
with the CPU checker (oracle/_ref = the compiled reference when present, else the oracle port): id offsets and every id,
exactly -- a mismatch refuses to time.

Prints ONE JSON line (rank 0) with the driver's contract keys plus `roofline`, `cpu_baseline` and `timings`.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

# total documents of each workload; the corpus is sharded over the ranks
TOTAL_DOCS = {"config1": 10000, "headline512": 10000000, "config2": 1000000, "config3": 1000000, "config4": 10000000, "config5": 10000000}


def csrc_sha():
    """identity of the kernels a profile was taken of: sha256 over blingfire_amd/csrc (sorted file names + contents), first 16 hex digits"""
    import hashlib
    d = os.path.join(ROOT, "blingfire_amd", "csrc")
    hsh = hashlib.sha256()
    for f in sorted(f for f in os.listdir(d) if f.endswith((".h", ".hip", ".cpp", ".map"))):
        hsh.update(f.encode())
        hsh.update(open(os.path.join(d, f), "rb").read())
    return hsh.hexdigest()[:16]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)      # 3 warm-up + 10 timed passes
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline512", choices=sorted(TOTAL_DOCS))
    ap.add_argument("--docs", type=int, default=0, help="override the TOTAL number of documents of the corpus")
    ap.add_argument("--docs-per-gpu", type=int, default=0, help="override the corpus size as documents per rank (total = this * ranks)")
    ap.add_argument("--sub-batch-docs", type=int, default=0, help="documents per TextToIdsBatchDevice call (0 = auto: whole shard if the workspace fits)")
    ap.add_argument("--model", default="")
    ap.add_argument("--variant", type=int, default=-1, help="kernel variant (experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-docs", type=int, default=0)
    ap.add_argument("--verify", default="full", help="'full' (default): every document of the shard against the CPU checker; N: the first N; 0: none")
    ap.add_argument("--no-extra-timings", action="store_true", help="skip the PCIe-inclusive / host-API timings and the lexer transition count")
    ap.add_argument("--offsets", action="store_true", help="time TextToIdsWithOffsetsBatchDevice (ids + byte offsets of every id) instead of TextToIdsBatchDevice; single-process form only")
    ap.add_argument("--inproc", action="store_true", help="N GPUs from ONE process through the library (BfSetDevices + the per-range handles, a thread per device) instead of N ranks")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks, one per GPU, or refuse."""
    import torch
    share = os.environ.get("BF_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not share:
        sys.stderr.write("bench: --gpus %d requested but this box exposes %d GPU(s); refusing to report an N-GPU number from fewer devices "
                         "(BF_BENCH_SHARE_GPU=1 runs the ranks on one GPU over gloo, testing only)\n" % (args.gpus, have))
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def device_doc_hashes(torch, ids, id_off, ndocs):
    """bfc_ids_hash (oracle/cpu_baseline.c) of every document, on the device ids: sum_j (id_j + C) * (2j + 1) mod 2^64.
    Checker-side arithmetic only (torch ops on the outputs of the product path)."""
    import bfutil
    out = torch.zeros(ndocs, dtype=torch.int64, device=ids.device)
    chunk = 1 << 20
    for d0 in range(0, ndocs, chunk):
        d1 = min(ndocs, d0 + chunk)
        o = id_off[d0:d1 + 1]
        a, b = int(o[0].item()), int(o[-1].item())
        if b == a:
            continue
        counts = o[1:] - o[:-1]
        doc = torch.repeat_interleave(torch.arange(d1 - d0, device=ids.device), counts)
        j = torch.arange(b - a, device=ids.device, dtype=torch.int64) - (o[:-1] - a)[doc]
        v = (ids[a:b].to(torch.int64) + bfutil.IDS_HASH_C) * (2 * j + 1)
        out[d0:d1].index_add_(0, doc, v)
    return out


def run_config1(args):
    """BASELINE.json configs[0]: the default pattern tokenizer (TextToWords, built-in wbd.bin) on short English lines.  The config names
    the reference's CPU path; the GPU batch entry point (TextToWordsBatchDevice: class stream, lexer -- lanes for the short lines, the
    long-document form of DESIGN.md section 5.3 for the others -- scan, string assembly) is timed beside it.  One GPU only (10,000 lines are
    one launch of each kernel)."""
    import ctypes
    import numpy as np
    import torch
    import bfutil
    import blingfire_amd as bf
    if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        sys.stderr.write("bench: config1 is a single-launch workload; run it with --gpus 1\n")
        sys.exit(2)
    nl = args.docs or TOTAL_DOCS["config1"]
    text, off = bfutil.gen_workload("config1", nl)
    dev = torch.device("cuda", 0)
    L = bf.lib()
    dt, do = torch.from_numpy(text).to(dev), torch.from_numpy(off).to(dev)
    out = torch.empty(2 * len(text) + 64, dtype=torch.uint8, device=dev)
    t_off = torch.empty(nl + 1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        r = L.TextToWordsBatchDevice(None, dt.data_ptr(), do.data_ptr(), nl, len(text), out.data_ptr(), out.numel(), t_off.data_ptr(), ctypes.c_void_p(stream))
        assert r == 0, r

    # every line against the reference's TextToWords (the checker never runs inside the timed region)
    verified = 0
    lib_path, ck_kind = bfutil.checker_lib_path()
    if args.verify != "0" and ck_kind == "reference":
        step()
        torch.cuda.synchronize(dev)
        g_off = t_off.cpu().numpy()
        g_out = out[:int(g_off[-1])].cpu().numpy().tobytes()
        R = ctypes.CDLL(lib_path)
        R.TextToWords.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        raw = text.tobytes()
        nv = nl if args.verify == "full" else min(nl, int(args.verify))
        nv = min(nv, 200000)
        cap = 2 * int((off[1:] - off[:-1]).max()) + 64          # (a line of the reference's file has 8,396 bytes)
        buf = ctypes.create_string_buffer(cap)
        for d in range(nv):
            line = raw[off[d]:off[d + 1]]
            n = R.TextToWords(line, len(line), buf, cap)
            want = buf.raw[:n - 1] if n > 0 else b""
            if g_out[g_off[d]:g_off[d + 1]] != want:
                raise SystemExit("bench: TextToWords output of line %d differs from the reference -- refusing to time" % d)
        verified = nv
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1) / args.steps
    out_bytes = int(t_off[-1].item())
    alg = len(text) + out_bytes + 16 * nl
    res = {"metric": "lines/sec", "value": nl * args.steps / elapsed, "unit": "lines/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8",
           "data": "the reference's own lines (tests/data/config1_lines.txt.gz)" if os.path.exists(bfutil.CONFIG1_LINES) else "synthetic",
           "config": {"workload": "config1: built-in wbd.bin TextToWords (TextToWordsBatchDevice), %d lines, %.1f B/line" % (nl, len(text) / nl), "model_file": "wbd.bin (built in)",
                      "total_docs": nl, "total_bytes": int(len(text)), "output_bytes": out_bytes,
                      "longest_line_bytes": int((off[1:] - off[:-1]).max()),       # (until round 5 one lane walked it: the step's time)
                      "lines_over_1KiB": int(((off[1:] - off[:-1]) > 1024).sum())},
           "gb_input_per_sec": len(text) * args.steps / elapsed / 1e9, "gpu_ms_per_step": gpu_ms,
           "roofline": {"bound": "hbm", "kernel": "whole step (class stream + lexer: lanes and the long-document form + scan + string assembly; latency of 20 small launches, not bandwidth)", "achieved": alg / (gpu_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": alg / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": alg},
           "verified_docs": verified}
    if not args.no_cpu_baseline and ck_kind == "reference":
        T = bfutil.host_threads()
        s1, _ = bfutil.cpu_text_to_words_time(lib_path, text, off, 1, 5)
        sT, _ = bfutil.cpu_text_to_words_time(lib_path, text, off, T, 5)
        res["cpu_baseline"] = {"value": nl / s1, "unit": "lines/s", "cores": 1, "kind": "reference",
                               "sample": "all %d lines, one TextToWords call per line, 1 thread, best of 5 passes (%.4f s)" % (nl, s1),
                               "all_threads": {"value": nl / sT, "threads": T, "seconds": sT}}
    print(json.dumps(res), flush=True)


def run_inproc(args):
    """N GPUs of this node from ONE process, through the library's own multi-device state (BfSetDevices): the corpus is split into the
    library's ranges (BfShardRanges: contiguous, byte-balanced), range g is made resident on device g and tokenised by a host thread of
    its own through the per-range handle (BfShardHandle) -- same static shard, no collective, as the N-rank form, with the tables
    replicated by the library instead of by N processes.  BF_BENCH_SHARE_GPU=1 (testing only) puts every range on device 0."""
    import threading
    import numpy as np
    import torch
    import bfutil
    import blingfire_amd as bf
    share = os.environ.get("BF_BENCH_SHARE_GPU") == "1"
    G = args.gpus
    have = torch.cuda.device_count()
    if have < G and not share:
        sys.stderr.write("bench: --gpus %d --inproc but this box exposes %d GPU(s)\n" % (G, have))
        sys.exit(2)
    dev_ids = [0 if share else g for g in range(G)]
    wl = bfutil.WORKLOADS[args.workload]
    model_name = args.model or wl["model"] or bfutil.bert_model_name()
    max_ids, unk = wl["max_ids"], wl["unk"]
    total_docs = args.docs or (args.docs_per_gpu * G if args.docs_per_gpu else TOTAL_DOCS[args.workload])
    text, off = bfutil.gen_workload(args.workload, total_docs)
    torch.cuda.set_device(dev_ids[0])
    h = bf.load_model(bfutil.model_path(model_name))
    kind = bf.lib().BfModelKind(h)
    if args.variant >= 0 and bf.lib().BfSetVariant(h, args.variant) == -5:
        raise SystemExit("bench.py: --variant %d selects a measurement instance: build with BF_EXPERIMENTS=1" % args.variant)
    bf.set_devices(h, dev_ids)
    bounds = bf.shard_ranges(off, G)
    per_doc_ws = (int(off[-1]) / max(total_docs, 1) + 1) * (6 if kind == 0 else 44)
    parts = []
    for g in range(G):
        lo, hi = int(bounds[g]), int(bounds[g + 1])
        dev = torch.device("cuda", dev_ids[g])
        nd = hi - lo
        sub = max(1, min(max(nd, 1), int(64e9 / per_doc_ws)))
        batches = []
        for d0 in range(lo, hi, sub):
            d1 = min(hi, d0 + sub)
            b0, b1 = int(off[d0]), int(off[d1])
            cap = max(1, min(2 * (b1 - b0 + d1 - d0), (d1 - d0) * max_ids))
            batches.append(dict(d0=d0, d1=d1, text=torch.from_numpy(text[b0:b1]).to(dev), off=torch.from_numpy(off[d0:d1 + 1] - b0).to(dev),
                                ids=torch.empty(cap, dtype=torch.int32, device=dev), id_off=torch.empty(d1 - d0 + 1, dtype=torch.int64, device=dev)))
        parts.append(dict(g=g, dev=dev, h=bf.shard_handle(h, g), lo=lo, hi=hi, batches=batches, secs=0.0, kms=np.zeros(6)))

    def run_part(pt, steps, barrier=None, collect=False):
        torch.cuda.set_device(pt["dev"])
        st = torch.cuda.Stream(pt["dev"])
        with torch.cuda.stream(st):
            if barrier is not None:
                barrier.wait()
            t0 = time.perf_counter()
            for _ in range(steps):
                for b in pt["batches"]:
                    bf.text_to_ids_batch_device(pt["h"], b["text"], b["off"], max_ids, unk, out_ids=b["ids"], out_off=b["id_off"])
                    if collect:
                        pt["kms"] += np.array(bf.last_kernel_ms(pt["h"]), dtype=np.float64)
            st.synchronize()
            pt["secs"] = time.perf_counter() - t0

    def run_all(steps, collect=False):
        bar = threading.Barrier(G + 1)
        th = [threading.Thread(target=run_part, args=(pt, steps, bar, collect)) for pt in parts]
        for t in th:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        return time.perf_counter() - t0

    # ---- parity gate: every document of every range, exactly
    run_all(1)
    lib_path, ck_kind = bfutil.checker_lib_path()
    nv = total_docs if args.verify == "full" else min(int(args.verify), total_docs)
    verified, verify_secs = 0, 0.0
    for pt in parts:
        for b in pt["batches"]:
            if b["d0"] >= nv:
                break
            z = min(b["d1"], nv)
            k = z - b["d0"]
            secs, c_ids, c_off = bfutil.cpu_ids_compact(lib_path, bfutil.model_path(model_name), text[off[b["d0"]]:off[z]], off[b["d0"]:z + 1] - off[b["d0"]], max_ids, unk)
            verify_secs += secs
            g_off = b["id_off"][:k + 1].cpu().numpy()
            if not (np.array_equal(g_off, c_off) and bool(torch.equal(b["ids"][:int(g_off[k])], torch.from_numpy(c_ids).to(pt["dev"])))):
                raise SystemExit("bench: GPU ids of range %d differ from the CPU checker (%s) -- refusing to time" % (pt["g"], ck_kind))
            verified += k
    if args.warmup:
        run_all(args.warmup)
    for pt in parts:
        pt["kms"][:] = 0
    elapsed = run_all(args.steps, collect=True)
    ranks = []
    for pt in parts:
        props = torch.cuda.get_device_properties(pt["dev"])
        try:
            pci = "%04x:%02x:%02x" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        except Exception:
            pci = None
        n_ids = sum(int(b["id_off"][-1].item()) for b in pt["batches"])
        ranks.append({"rank": pt["g"], "device": pt["dev"].index, "name": props.name, "pci": pci, "cus": props.multi_processor_count, "first_doc": int(pt["lo"]), "docs": int(pt["hi"] - pt["lo"]),
                      "bytes": int(off[pt["hi"]] - off[pt["lo"]]), "ids": n_ids, "seconds": pt["secs"], "status": bf.lib().BfLastStatus(pt["h"]),
                      "kernel_ms": [float(x) / max(args.steps, 1) for x in pt["kms"]]})
    bytes_all, ids_all = int(off[-1]), sum(r["ids"] for r in ranks)
    slow = max(parts, key=lambda q: q["secs"])
    tok_ms = float(slow["kms"][5]) / max(args.steps, 1)           # the dominant kernel alone (BfLastKernelMs [5])
    alg = int(off[slow["hi"]] - off[slow["lo"]]) + 4 * ranks[slow["g"]]["ids"] + 16 * (slow["hi"] - slow["lo"])
    bf.lib().BfTokeniseKernel.restype = ctypes.c_char_p
    bf.lib().BfTokeniseKernel.argtypes = [ctypes.c_void_p]
    kernel_name = (bf.lib().BfTokeniseKernel(ctypes.c_void_p(h)) or b"").decode()
    achieved = alg / (tok_ms * 1e-3) / 1e9 if tok_ms > 0 else 0.0
    res = {"metric": "docs/sec", "value": total_docs * args.steps / elapsed, "unit": "docs/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
           "config": {"workload": "%s: %s TextToIds, %d documents in total, %.0f B/doc avg, max_ids %d, unk %d" % (args.workload, model_name, total_docs, bytes_all / max(total_docs, 1), max_ids, unk),
                      "model_file": model_name, "total_docs": total_docs, "total_bytes": bytes_all, "total_ids": ids_all, "launcher": "inproc",
                      "sharding": "BfSetDevices: contiguous byte-balanced document ranges (BfShardRanges), a host thread per device, no collective"},
           "gb_input_per_sec": bytes_all * args.steps / elapsed / 1e9, "ids_per_sec": ids_all * args.steps / elapsed,
           "roofline": {"bound": "hbm", "kernel": "%s, slowest range" % kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": alg},
           "cpu_baseline": None,
           "verified_docs": verified, "verify": {"checker": ck_kind, "seconds": verify_secs, "method": "exact: id offsets and every id of every document against the CPU checker (array equality)"},
           "status": max(r["status"] for r in ranks), "ranks": ranks, "backend": "inproc"}
    bf.free_model(h)
    print(json.dumps(res))
    return 0


def main():
    args = parse_args()
    if args.workload == "config1":
        return run_config1(args)
    if args.inproc:
        return run_inproc(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    import numpy as np
    import torch
    import bfutil
    import blingfire_amd as bf

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench: --gpus %d but WORLD_SIZE=%d: the two must agree (n_gpus is the number of ranks that really run)\n" % (args.gpus, world))
        sys.exit(2)
    share = os.environ.get("BF_BENCH_SHARE_GPU") == "1"      # testing only: several ranks on ONE GPU over gloo
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dev_index = 0 if share else local_rank
        if dev_index >= torch.cuda.device_count():
            sys.stderr.write("bench: rank %d wants cuda:%d but only %d device(s) are visible\n" % (rank, dev_index, torch.cuda.device_count()))
            sys.exit(2)
        torch.cuda.set_device(dev_index)
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
    else:
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)
    props = torch.cuda.get_device_properties(dev)

    wl = bfutil.WORKLOADS[args.workload]
    model_name = args.model or wl["model"] or bfutil.bert_model_name()
    if args.workload in ("headline512", "config2") and not args.model and model_name != "bert_base_tok.bin" and rank == 0:
        sys.stderr.write("bench: models/bert_base_tok.bin is absent -- falling back to %s (NOT the metric's model)\n" % model_name)
    max_ids, unk = wl["max_ids"], wl["unk"]
    total_docs = args.docs or (args.docs_per_gpu * world if args.docs_per_gpu else TOTAL_DOCS[args.workload])
    first = rank * total_docs // world
    ndocs = (rank + 1) * total_docs // world - first

    # ---- the rank's shard, generated on the host and made resident in HBM before any timing
    text, off = bfutil.gen_workload(args.workload, ndocs, first_doc=first)
    total_bytes = int(off[-1])
    h = bf.load_model(bfutil.model_path(model_name))
    kind = bf.lib().BfModelKind(h)
    if args.variant >= 0 and bf.lib().BfSetVariant(h, args.variant) == -5:
        raise SystemExit("bench.py: --variant %d selects a measurement instance: build with BF_EXPERIMENTS=1" % args.variant)
    # sub-batches: one TextToIdsBatchDevice call each.  The _sp segmenters keep up to 16 bytes of workspace per stream element
    # (2 elements per byte with a charmap), so a 5 GB shard is issued in pieces that keep the workspace under ~64 GB.
    sub = args.sub_batch_docs
    if sub <= 0:
        per_doc_ws = (total_bytes / max(ndocs, 1) + 1) * (6 if kind == 0 else 44)
        sub = max(1, min(ndocs, int(64e9 / per_doc_ws)))
    d_text_all = torch.from_numpy(text).to(dev)
    d_off_all = torch.from_numpy(off).to(dev)
    batches = []
    for d0 in range(0, ndocs, sub):
        d1 = min(ndocs, d0 + sub)
        b0, b1 = int(off[d0]), int(off[d1])
        nb, nd = b1 - b0, d1 - d0
        cap = max(1, min(2 * (nb + nd), nd * max_ids))
        batches.append(dict(d0=d0, d1=d1, text=d_text_all[b0:b1], off=(d_off_all[d0:d1 + 1] - b0).contiguous(),
                            ids=torch.empty(cap, dtype=torch.int32, device=dev), id_off=torch.empty(nd + 1, dtype=torch.int64, device=dev)))
        if args.offsets:
            batches[-1]["starts"] = torch.empty(cap, dtype=torch.int32, device=dev); batches[-1]["ends"] = torch.empty(cap, dtype=torch.int32, device=dev)

    def step(collect_ms=None):
        for b in batches:
            if args.offsets:
                r = bf.lib().TextToIdsWithOffsetsBatchDevice(ctypes.c_void_p(h), b["text"].data_ptr(), b["off"].data_ptr(), b["d1"] - b["d0"], b["text"].numel(), b["ids"].data_ptr(),
                                                             b["starts"].data_ptr(), b["ends"].data_ptr(), b["ids"].numel(), b["id_off"].data_ptr(), max_ids, unk,
                                                             ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
                if r != 0:
                    raise SystemExit("bench: TextToIdsWithOffsetsBatchDevice failed (%d)" % r)
            else:
                bf.text_to_ids_batch_device(h, b["text"], b["off"], max_ids, unk, out_ids=b["ids"], out_off=b["id_off"])
            if collect_ms is not None:
                collect_ms += np.array(bf.last_kernel_ms(h), dtype=np.float64)   # HIP events recorded on the launch stream

    # ---- parity gate: EVERY document of the shard (the checker is never inside the timed region)
    verified, verify_secs = 0, 0.0
    nv = ndocs if args.verify == "full" else min(int(args.verify), ndocs)
    lib_path, ck_kind = bfutil.checker_lib_path()
    cpu_threads = max(1, bfutil.host_threads() // (world if not share else 1))
    if nv > 0:
        # exact: every id of every document (SURVEY.md section 8(d) "memcmp"), the reference's ids compact and in document order,
        # compared on the device with the ids of the sub-batch; the CPU side works through the shard in pieces of <= 1 M documents
        step()
        torch.cuda.synchronize(dev)
        piece = 1000000
        for b in batches:
            if b["d0"] >= nv:
                break
            k_all = min(b["d1"], nv) - b["d0"]
            g_off = b["id_off"][:k_all + 1].cpu().numpy()
            for p0 in range(0, k_all, piece):
                p1 = min(k_all, p0 + piece)
                a, z = b["d0"] + p0, b["d0"] + p1
                secs, c_ids, c_off = bfutil.cpu_ids_compact(lib_path, bfutil.model_path(model_name), text[off[a]:off[z]], off[a:z + 1] - off[a], max_ids, unk,
                                                            nthreads=cpu_threads)
                verify_secs += secs
                ok_c = np.array_equal(g_off[p0:p1 + 1] - g_off[p0], c_off)
                ok_i = False
                if ok_c:
                    lo, hi = int(g_off[p0]), int(g_off[p1])
                    ok_i = bool(torch.equal(b["ids"][lo:hi], torch.from_numpy(c_ids).to(dev)))
                if not (ok_c and ok_i):
                    cnt_g, cnt_c = np.diff(g_off[p0:p1 + 1]), np.diff(c_off)
                    bad = np.nonzero(cnt_g != cnt_c)[0]
                    if len(bad):
                        d = int(bad[0])
                    else:
                        g_ids = b["ids"][int(g_off[p0]):int(g_off[p1])].cpu().numpy()
                        ne = np.nonzero(g_ids != c_ids)[0]
                        d = int(np.searchsorted(c_off, ne[0], side="right") - 1) if len(ne) else 0
                    dd = a + d
                    lo, hi = int(g_off[p0 + d]), int(g_off[p0 + d + 1])
                    raise SystemExit("bench: GPU ids differ from the CPU checker (%s), rank %d, first = shard document %d (%r...): GPU %s, CPU %s -- refusing to time"
                                     % (ck_kind, rank, dd, bytes(text[off[dd]:off[dd] + 64]), b["ids"][lo:hi].cpu().numpy().tolist()[:32],
                                        c_ids[int(c_off[d]):int(c_off[d + 1])].tolist()[:32]))
                verified += p1 - p0
                del c_ids

    offsets_verified = 0
    if args.offsets and nv > 0:
        # the byte offsets of every id of a sample of documents against the checker's TextToIdsWithOffsets (one call per document)
        ck = bfutil.reference() if bfutil.have_ref() else bfutil.oracle()
        hck = ck.load(bfutil.model_path(model_name))
        name = "TextToIdsWithOffsets" if bfutil.have_ref() else "bfo_text_to_ids_with_offsets"
        b = batches[0]
        k = min(2000, b["d1"] - b["d0"], nv)
        g_off = b["id_off"][:k + 1].cpu().numpy(); hi = int(g_off[k])
        g_ids, g_st, g_en = b["ids"][:hi].cpu().numpy(), b["starts"][:hi].cpu().numpy(), b["ends"][:hi].cpu().numpy()
        for d in range(k):
            doc = bytes(text[off[d]:off[d + 1]])
            c, gi, gs, ge = ck.with_offsets(hck, doc, max_ids, unk, name)
            a, z = int(g_off[d]), int(g_off[d + 1])
            if (z - a, g_ids[a:z].tolist(), g_st[a:z].tolist(), g_en[a:z].tolist()) != (c, gi[:c], gs[:c], ge[:c]):
                raise SystemExit("bench: offsets of document %d differ from the CPU checker -- refusing to time" % d)
        ck.free(hck)
        offsets_verified = k

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    kms = np.zeros(6, dtype=np.float64)
    per_step = []                       # HIP-event time of every timed step (all its launches), for the median / min of SURVEY.md section 8(d)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one = np.zeros(6, dtype=np.float64)
        step(one)
        kms += one
        per_step.append(float(one[4]))
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    my_elapsed = elapsed
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kms /= max(args.steps, 1)
    n_ids = sum(int(b["id_off"][-1].item()) for b in batches)
    status = bf.lib().BfLastStatus(h)

    # ---- per-rank facts, gathered on rank 0
    try:
        pci = "%04x:%02x:%02x" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
    except Exception:
        pci = None
    me = {"rank": rank, "device": dev_index, "name": props.name, "pci": pci, "cus": props.multi_processor_count, "first_doc": first,
          "docs": ndocs, "bytes": total_bytes, "ids": n_ids, "verified_docs": verified, "seconds": my_elapsed, "status": status}
    ranks = [me]
    if dist:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)

    res = None
    if rank == 0:
        docs_all = sum(r["docs"] for r in ranks)
        bytes_all = sum(r["bytes"] for r in ranks)
        ids_all = sum(r["ids"] for r in ranks)
        value = docs_all * args.steps / elapsed
        # algorithmic bytes of one step's launches of the dominant kernel on THIS rank: n_in + 4*n_ids + 16 per document
        alg_bytes = total_bytes + (12 if args.offsets else 4) * n_ids + 16 * ndocs      # with offsets: id + first byte + last byte per id
        tok_ms = float(kms[1])
        dom_ms = float(kms[5]) if kms[5] > 0 else tok_ms                # the dominant kernel alone (HIP events around it on the launch stream)
        step_ms = float(kms[4])
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic, traffic_stale, l2_hit, traffic_dom, traffic_kernels = None, None, None, None, None
        bf.lib().BfTokeniseKernel.restype = ctypes.c_char_p
        bf.lib().BfTokeniseKernel.argtypes = [ctypes.c_void_p]
        kernel_name = (bf.lib().BfTokeniseKernel(ctypes.c_void_p(h)) or b"").decode()
        bf.lib().BfStepKernels.restype = ctypes.c_char_p
        bf.lib().BfStepKernels.argtypes = [ctypes.c_void_p]
        step_kernels = (bf.lib().BfStepKernels(ctypes.c_void_p(h)) or b"").decode()
        try:   # HBM bytes per step -- of ALL kernels of the step, and of the dominant one -- from the committed rocprofv3 PMC passes of this same command
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            ent = tj.get("%s/%s/%d%s" % (args.workload, model_name, ndocs, "/offsets" if args.offsets else ""))
            if ent:
                traffic = ent.get("hbm_bytes_per_step", ent.get("hbm_bytes_per_launch"))
                traffic_dom = ent.get("dominant_hbm_bytes_per_step")
                traffic_kernels = ent.get("kernels")
                l2_hit = ent.get("l2_hit_rate")
                # the counters belong to the kernels they were taken of: a profile of another build of csrc/ is flagged
                traffic_stale = ent.get("csrc_sha") != csrc_sha() or ent.get("kernel") != kernel_name
        except Exception:
            traffic = None
        res = {
            "metric": "docs/sec", "value": value, "unit": "docs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s: %s TextToIds, %d documents in total (%d on rank 0), %.0f B/doc avg, max_ids %d, unk %d" % (
                args.workload, model_name, docs_all, ndocs, bytes_all / max(docs_all, 1), max_ids, unk),
                "model_file": model_name, "total_docs": docs_all, "total_bytes": bytes_all, "total_ids": ids_all,
                "docs_per_gpu": ndocs, "sub_batches_per_step": len(batches),
                "api": "TextToIdsWithOffsetsBatchDevice (ids + first / last byte of every id)" if args.offsets else "TextToIdsBatchDevice",
                "sharding": "static contiguous document ranges, no collective"},
            "gb_input_per_sec": bytes_all * args.steps / elapsed / 1e9,
            "ids_per_sec": ids_all * args.steps / elapsed,
            "kernel_ms": {"prep": float(kms[0]), "tokenise": tok_ms, "scan": float(kms[2]), "compact": float(kms[3]), "total": float(kms[4]), "dominant": dom_ms, "kernels": step_kernels},
            "kernel_ms_per_step": {"median": float(np.median(per_step)) if per_step else None, "min": float(np.min(per_step)) if per_step else None,
                                   "max": float(np.max(per_step)) if per_step else None, "n": len(per_step),
                                   "what": "HIP-event time of one step's launches (prep .. compact), every timed step"},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_stale": traffic_stale, "l2_hit_rate": l2_hit,
                         "kernel_ms": dom_ms, "algorithmic_bytes_per_launch": alg_bytes, "launches_per_step": len(batches),
                         # the step's ids are WRITTEN by the kernels behind the dominant one (k_wp_merge, k_uni_ids ...), so `achieved` credits the
                         # dominant kernel with bytes it does not move; `step` below is the honest whole-path figure, and this is the dominant kernel on the
                         # bytes of the step's INPUT alone (what every form of the path has to read once in that kernel)
                         "input_only": {"bytes": total_bytes + 8 * ndocs, "achieved": (total_bytes + 8 * ndocs) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0,
                                        "frac": ((total_bytes + 8 * ndocs) / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if dom_ms > 0 else 0.0},
                         "traffic_dominant_kernel": traffic_dom, "traffic_by_kernel": traffic_kernels,
                         "step": {"kernels": step_kernels, "ms": step_ms, "achieved": alg_bytes / (step_ms * 1e-3) / 1e9 if step_ms > 0 else 0.0,
                                  "frac": (alg_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if step_ms > 0 else 0.0,
                                  "what": "the same algorithmic bytes over ALL kernels of the step (prep .. compact)"},
                         "note": "achieved = algorithmic bytes of one step (all its launches) / time of the dominant kernel (`kernel`) in one step, HIP events around it on the launch "
                                 "stream; traffic = FETCH_SIZE + WRITE_SIZE of EVERY kernel of the step of the same command (separate rocprofv3 --pmc passes, "
                                 "profiles/traffic.json: traffic_by_kernel; traffic_dominant_kernel: the dominant kernel's share; traffic_stale: the profile is of another build of csrc/)"},
            "verified_docs": sum(r["verified_docs"] for r in ranks), "verify": {"checker": ck_kind, "threads": cpu_threads, "seconds": verify_secs,
                                                                               "method": "exact: id offsets and every id of every document against the CPU checker (array equality)"},
            "status": max(r["status"] for r in ranks), "ranks": ranks,
            "backend": (dist.get_backend() if dist else None),
        }
        if args.offsets:
            res["verify"]["offsets"] = "start / end byte offsets of every id of the first %d documents against the checker's TextToIdsWithOffsets" % offsets_verified

    # ---- the other two timings of SURVEY.md section 8(d) and the work-rate roofline (rank 0, N=1 only; bounded sample)
    if rank == 0 and world == 1 and not args.no_extra_timings:
        ns = min(ndocs, 1250000)
        nb = int(off[ns])
        s_text, s_off = text[:nb], off[:ns + 1]
        # (2) device end-to-end: pinned host buffers -> H2D -> pipeline -> D2H of ids and offsets
        p_text = torch.from_numpy(s_text).pin_memory()
        p_off = torch.from_numpy(s_off).pin_memory()
        cap = max(1, min(2 * (nb + ns), ns * max_ids))
        e_text = torch.empty(nb, dtype=torch.uint8, device=dev)
        e_off = torch.empty(ns + 1, dtype=torch.int64, device=dev)
        e_ids = torch.empty(cap, dtype=torch.int32, device=dev)
        e_idoff = torch.empty(ns + 1, dtype=torch.int64, device=dev)
        ph_ids = torch.empty(cap, dtype=torch.int32).pin_memory()
        ph_off = torch.empty(ns + 1, dtype=torch.int64).pin_memory()
        e2e = []
        for it in range(4):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            e_text.copy_(p_text, non_blocking=True)
            e_off.copy_(p_off, non_blocking=True)
            bf.text_to_ids_batch_device(h, e_text, e_off, max_ids, unk, out_ids=e_ids, out_off=e_idoff)
            ph_off.copy_(e_idoff, non_blocking=True)
            torch.cuda.synchronize(dev)
            nid = int(ph_off[-1])
            ph_ids[:nid].copy_(e_ids[:nid], non_blocking=True)
            torch.cuda.synchronize(dev)
            if it:
                e2e.append(time.perf_counter() - t1)
        del e_text, e_off, e_ids, e_idoff, ph_ids, p_text, p_off
        # (3) wall clock of the C call on pageable host buffers (TextToIdsBatch)
        # -- the caller's arrays are allocated (and touched) once, outside the timed calls, like a C caller that reuses its buffers: the
        #    Python wrapper bf.text_to_ids_batch allocates a worst-case array per call, whose page faults would be most of the time
        #    On the WHOLE shard (the 10 M-document corpus at N = 1), not the 1.25 M-document sample of the pinned figure.
        api = []
        n_api = ndocs
        cap_api = n_ids + 16
        a_ids = np.zeros(cap_api, dtype=np.int32)
        a_off = np.zeros(n_api + 1, dtype=np.int64)
        for it in range(4):
            t1 = time.perf_counter()
            r = bf.lib().TextToIdsBatch(ctypes.c_void_p(h), text.ctypes.data, off.ctypes.data, n_api, a_ids.ctypes.data, cap_api, a_off.ctypes.data, max_ids, unk)
            if r < 0:
                raise RuntimeError("TextToIdsBatch failed: %d" % r)
            if it:
                api.append(time.perf_counter() - t1)
        if r != n_ids or int(a_off[-1]) != n_ids:
            raise RuntimeError("TextToIdsBatch on host buffers returned %d ids, the device-resident run %d" % (r, n_ids))
        del a_ids, a_off
        # what the link does on this box, each direction alone and both together (pinned memory, 1 GiB each way): the yardstick for the host API's figure
        pcie = None
        try:
            nbytes = 1 << 30
            hp_in = torch.empty(nbytes, dtype=torch.uint8).pin_memory(); hp_out = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
            d_in = torch.empty(nbytes, dtype=torch.uint8, device=dev); d_out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            s_in, s_out = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
            def timed(do_in, do_out):
                best = 1e9
                for _ in range(3):
                    torch.cuda.synchronize(dev)
                    t1 = time.perf_counter()
                    if do_in:
                        with torch.cuda.stream(s_in): d_in.copy_(hp_in, non_blocking=True)
                    if do_out:
                        with torch.cuda.stream(s_out): hp_out.copy_(d_out, non_blocking=True)
                    torch.cuda.synchronize(dev)
                    best = min(best, time.perf_counter() - t1)
                return best
            t_in, t_out, t_both = timed(True, False), timed(False, True), timed(True, True)
            pcie = {"h2d_GBps": nbytes / t_in / 1e9, "d2h_GBps": nbytes / t_out / 1e9, "both_directions_GBps_sum": 2 * nbytes / t_both / 1e9,
                    "what": "1 GiB of pinned memory each way, best of 3, the two directions on two streams"}
            del hp_in, hp_out, d_in, d_out
        except Exception as e:
            pcie = {"error": str(e)}
        b_in, b_out = int(total_bytes) + 8 * (n_api + 1), 4 * int(n_ids) + 8 * (n_api + 1)
        api_t = min(api)
        # the call has two speeds that follow the process, not the build: how much of this process sits on huge pages is recorded beside it
        host_memory = None
        try:
            roll = {l.split(":")[0]: int(l.split()[1]) for l in open("/proc/self/smaps_rollup").read().splitlines() if l.startswith(("Rss:", "AnonHugePages:"))}
            host_memory = {"rss_kB": roll.get("Rss"), "anon_huge_pages_kB": roll.get("AnonHugePages"),
                           "transparent_hugepage": open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip()}
        except Exception:
            pass
        link = None
        if pcie and "h2d_GBps" in pcie:
            serial = b_in / (pcie["h2d_GBps"] * 1e9) + b_out / (pcie["d2h_GBps"] * 1e9)
            ideal = max(b_in / (pcie["h2d_GBps"] * 1e9), b_out / (pcie["d2h_GBps"] * 1e9))
            link = {"bytes_in": b_in, "bytes_out": b_out, "GBps_in": b_in / api_t / 1e9, "GBps_out": b_out / api_t / 1e9,
                    "ms_if_the_copies_ran_one_after_the_other": serial * 1e3, "ms_if_they_overlapped_fully": ideal * 1e3,
                    "overlap_fraction": max(0.0, min(1.0, (serial - api_t) / (serial - ideal))) if serial > ideal else None,
                    "what": "the call's wall clock against this box's pinned copy rates: 1 = as fast as the slower direction alone, 0 = as slow as both in a row (pageable buffers: the staging copies are on top)"}
        res["timings"] = {
            "kernel_only": {"docs_per_s": value, "ms_per_step": res["ms_per_step"], "what": "device-resident input and output, the whole shard (= value)"},
            "device_e2e_pinned": {"docs_per_s": ns / min(e2e), "ms": min(e2e) * 1e3, "median_ms": float(np.median(e2e)) * 1e3, "sample_docs": ns,
                                  "what": "pinned host text -> H2D -> kernels -> D2H ids+offsets, one batch, best of 3"},
            "host_api_wall": {"docs_per_s": n_api / min(api), "ms": min(api) * 1e3, "median_ms": float(np.median(api)) * 1e3, "sample_docs": n_api,
                              "what": "wall clock of TextToIdsBatch on pageable host arrays (chunked through pinned staging, bf_capi.cpp run_host_chunked), output arrays reused, best of 3",
                              "link": link, "host_memory": host_memory},
            "pcie": pcie,
        }
        # work-rate roofline of the lexer: table gathers per second against the measured gather ceiling.
        # Counted by the instrumented instance of the SAME kernel in one extra untimed step (BfSetLexStats).
        if kind == 0:
            try:
                buf = (ctypes.c_ulonglong * 16)()
                bf.lib().BfLexStats.restype = ctypes.c_int
                bf.lib().BfLexStats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
                bf.lib().BfSetLexStats(ctypes.c_void_p(h), 1)
                step()
                torch.cuda.synchronize(dev)
                bf.lib().BfLexStats(ctypes.c_void_p(h), buf, 16)
                bf.lib().BfSetLexStats(ctypes.c_void_p(h), 0)
                if kernel_name == "k_wp_flat":
                    # the flat program is bound by vector instructions, not by gathers: what its instrumented twin counts
                    res["roofline"]["lookups"] = {"chunks_per_step": int(buf[0]), "plain_ascii_chunks": int(buf[1]), "tokens_per_step": int(buf[2]), "table_hits": int(buf[3]),
                                                  "hit_rate": int(buf[3]) / max(int(buf[2]), 1), "words_walked_by_k_wp_units": int(buf[4]), "row_gathers_per_step": 2 * int(buf[2]),
                                                  "documents_handed_to_the_wave_program": int(buf[7]),
                                                  "counted_by": "the STATS instances of k_wp_flat / k_wp_units in one extra untimed step (BfSetLexStats)"}
                    raise StopIteration
                wave = kernel_name == "k_wp_wave"
                issued = int(buf[9]) if wave else int(buf[1])          # gathers issued (the wave program's transition step loads on all 64 lanes)
                transitions = int(buf[10]) if wave else int(buf[1])    # transitions made
                clk_hz = float(getattr(props, "clock_rate", 2400000)) * 1e3
                rate = issued / (tok_ms * 1e-3) / (props.multi_processor_count * clk_hz)
                ceil = None
                try:
                    ceil = json.load(open(os.path.join(ROOT, "profiles", "gather_ceiling.json")))["ceiling_lane_gathers_per_clk_per_cu"]
                except Exception:
                    pass
                res["roofline"]["gather"] = {"achieved": rate, "ceiling": ceil, "frac": (rate / ceil) if ceil else None,
                                             "unit": "table lane-gathers issued / clk / CU", "gathers_per_step": issued, "transitions_per_step": transitions,
                                             "transitions_per_input_byte": transitions / max(total_bytes, 1),
                                             "counted_by": "the STATS instance of %s (every gather the kernel issues; idle lanes of a transition step included)" % kernel_name,
                                             "ceiling_source": "tools/microbench/gather.hip on this GPU, table of the model's size (profiles/gather_ceiling.json)"}
            except StopIteration:
                pass
            except Exception as e:   # instrumentation is optional
                res["roofline"]["gather"] = {"error": str(e)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # reference CPU path on this box's host cores: bounded sample, best of 3 warm passes, 1 thread and all usable threads
        T = bfutil.host_threads()
        ns = args.cpu_sample_docs or min(ndocs, 20000 * T)
        n1 = min(ndocs, 20000)
        mp = bfutil.model_path(model_name)
        s1, _, _, _ = bfutil.cpu_text_to_ids_batch(lib_path, mp, text[:off[n1]], off[:n1 + 1], max_ids, unk, nthreads=1, passes=3, want_ids=False)
        sT, _, _, _ = bfutil.cpu_text_to_ids_batch(lib_path, mp, text[:off[ns]], off[:ns + 1], max_ids, unk, nthreads=T, passes=3, want_ids=False)
        res["cpu_baseline"] = {"value": ns / sT, "unit": "docs/s", "cores": T, "kind": ck_kind,
                               "sample": "first %d documents of the same corpus, one TextToIds call per document, %d threads sharing one model handle, best of 3 passes (%.2f s)" % (ns, T, sT),
                               "one_thread": {"value": n1 / s1, "unit": "docs/s", "sample_docs": n1, "seconds": s1},
                               "host": {"cpu": bfutil.cpu_model_string(), "os_cpu_count": os.cpu_count(),
                                        "affinity": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": bfutil.cgroup_cpu_quota()},
                               "full_shard_pass": {"docs": nv, "threads": cpu_threads, "seconds": verify_secs,
                                                   "docs_per_s": (nv / verify_secs) if verify_secs > 0 else None,
                                                   "what": "the verification pass over the whole shard (one cold pass, includes thread start-up)"}}
    if rank == 0:
        print(json.dumps(res), flush=True)
    bf.free_model(h)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
