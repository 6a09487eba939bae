#!/usr/bin/env python3
"""Where do the instructions of a wave program go, per document -- estimated WITHOUT a GPU ("Where the kernel waits").

The wave programs (bf_wave_body.h, bf_bpe_wave_body.h) compile for the device and for the 64-fibre simulator from the same source.  So:
  1. static side: the kernel's ISA (hipcc -S -gline-tables-only), cut into basic blocks; every block gets its scalar / vector / LDS / memory /
     branch instruction counts and the source lines its instructions come from (.loc);
  2. dynamic side: the simulator built with gcov (g++ -O0 --coverage) runs a sample of a bench workload; a line's execution count / 64 lanes /
     documents = how often per document a wave passes it;
  3. a block's executions per document = that of the source line most of its instructions come from; block counts x executions = the estimate.
It is an estimate: scheduling moves instructions between lines, blocks without a line (spill code, loop-edge copies) are listed apart, a block
that mixes lines of different frequency takes the dominant one, and a block that holds an unrolled source loop (the three transitions of a unit
round, the eight table look-ups of a chunk) is counted once per pass of the loop's line, i.e. too often by the unroll factor.  Good enough to rank: on k_wp_wave it put the scalar copies at the head of the
re-entered fill_step (~140 scalar instructions per document) at the top, which the nested-loop form then removed.

usage: python tools/wave_dyn_profile.py [--trim T] [--cfg C] [--docs N] [--workload W] [--top K]
  --trim T   the TRIM template argument of the k_wp_wave instance to look at (the shipped one: 15)
  --cfg C    the simulator configuration that runs the same TRIM bits (tests/hosttest/bf_wavetest.cpp: 0 = TRIM 0, 3 = TRIM 15, 5 = TRIM 3)
"""
import argparse
import collections
import ctypes
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
CSRC = os.path.join(ROOT, "blingfire_amd", "csrc")
BODY = "bf_wave_body.h"


def isa_blocks(work, trim):
    s_path = os.path.join(work, "kernels.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
                           "-gline-tables-only", os.path.join(CSRC, "bf_kernels.hip"), "-o", s_path], stderr=subprocess.DEVNULL)
    lines = open(s_path).read().split("\n")
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    want = "_ZN3bfa9k_wp_waveINS_5WvLdsILi1024ELi256ELi8ELb0EEELi1ELi3ELi8ELb0ELi0ELi4ELi0ELb0ELi%dEEEv" % trim
    start = next(i for i, l in enumerate(lines) if l.startswith(want) and l.rstrip().split(":")[0].endswith("i"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur, loc = [], None, None
    for l in lines[start:end]:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m or cur is None:
            cur = dict(name=m.group(1) if m else "entry", S=0, V=0, L=0, M=0, B=0, locs=collections.Counter())
            blocks.append(cur)
            if m:
                continue
        m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
        if m:
            loc = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        m = re.match(r'\s+([a-z_0-9]+)\s', l)
        if not m:
            continue
        op = m.group(1)
        k = ("B" if op.startswith("s_cbranch") or op == "s_branch" else None if op in ("s_waitcnt", "s_nop") else "S" if op.startswith("s_") else
             "V" if op.startswith("v_") else "L" if op.startswith("ds_") else "M" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else None)
        if k is None:
            continue
        cur[k] += 1
        if loc and loc[0] == BODY and loc[1] > 0:
            cur["locs"][loc[1]] += 1
    return blocks


def line_frequencies(work, cfg, ndocs, workload):
    import bfutil
    import numpy as np
    obj = os.path.join(work, "bf_oracle.o")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-fPIC", "-c", os.path.join(ROOT, "oracle", "bf_oracle.c"), "-o", obj])
    lib = os.path.join(work, "libcov.so")
    subprocess.check_call(["g++", "-O0", "--coverage", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", lib,
                           os.path.join(ROOT, "tests", "hosttest", "bf_hosttest.cpp"), os.path.join(ROOT, "tests", "hosttest", "bf_wavetest.cpp"),
                           os.path.join(CSRC, "bf_model.cpp"), obj], cwd=work)
    L = ctypes.CDLL(lib)
    L.bft_load.restype = ctypes.c_void_p
    L.bft_load.argtypes = [ctypes.c_char_p]
    L.bft_emu_wave_batch.restype = ctypes.c_long
    L.bft_emu_wave_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
    wl = bfutil.WORKLOADS[workload]
    text, off = bfutil.gen_workload(workload, ndocs)
    h = L.bft_load(bfutil.model_path(wl["model"] or bfutil.bert_model_name()).encode())
    cap = len(text) + 16
    ids = np.zeros(cap, dtype=np.int32)
    ido = np.zeros(ndocs + 1, dtype=np.int64)
    st = np.zeros(16, dtype=np.uint64)
    r = L.bft_emu_wave_batch(h, text.ctypes.data, len(text), off.ctypes.data, ndocs, wl["max_ids"], wl["unk"], 1, 8, cfg, ids.ctypes.data, cap, ido.ctypes.data, st.ctypes.data)
    assert r >= 0, r
    # flush the counters of the library (it stays loaded: ask gcov's runtime to dump)
    try:
        ctypes.CDLL(None).__gcov_dump()
    except Exception:
        pass
    del L
    stats = st.tolist()
    return stats


def read_gcov(work):
    subprocess.call(["gcov", "-o", ".", "libcov.so-bf_wavetest.gcno"], cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ex = {}
    for l in open(os.path.join(work, BODY + ".gcov")):
        m = re.match(r'\s*([0-9]+)\*?:\s*(\d+):', l)
        if m:
            ex[int(m.group(2))] = max(ex.get(int(m.group(2)), 0), int(m.group(1)))
    return ex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trim", type=int, default=15)
    ap.add_argument("--cfg", type=int, default=3)
    ap.add_argument("--docs", type=int, default=1000)
    ap.add_argument("--workload", default="headline512")
    ap.add_argument("--top", type=int, default=30)
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as work:
        blocks = isa_blocks(work, a.trim)
        # the run happens in a child so that the coverage counters are written when it exits
        code = ("import sys; sys.path.insert(0, %r); import wave_dyn_profile as w; print(w.line_frequencies(%r, %d, %d, %r))" %
                (os.path.join(ROOT, "tools"), work, a.cfg, a.docs, a.workload))
        out = subprocess.check_output([sys.executable, "-c", code], cwd=work).decode().strip().split("\n")[-1]
        stats = eval(out)
        ex = read_gcov(work)
    nd = float(a.docs)
    print("simulator, %d documents of %s: %.2f unit rounds, %.2f chunk-wide passes, %.2f decode steps, %.2f retire passes, %.1f tokens per document" %
          (a.docs, a.workload, stats[0] / nd, stats[1] / nd, stats[8] / nd, stats[5] / nd, stats[3] / nd))
    tot = dict(S=sum(b["S"] for b in blocks), V=sum(b["V"] for b in blocks), B=sum(b["B"] for b in blocks))
    print("static, TRIM %d: %d scalar, %d vector, %d branch instructions in %d blocks" % (a.trim, tot["S"], tot["V"], tot["B"], len(blocks)))
    rows, nolines = [], []
    for b in blocks[1:]:                      # the first block is the kernel's prologue
        if not b["locs"]:
            nolines.append(b)
            continue
        ln = b["locs"].most_common(1)[0][0]
        f = ex.get(ln, 0) / 64.0 / nd
        rows.append((b, ln, f))
    dS = sum(b["S"] * f for b, _, f in rows)
    dV = sum(b["V"] * f for b, _, f in rows)
    print("estimate per document (blocks with source lines): %.0f scalar + %.0f vector instructions; blocks without a line: %d scalar, %d vector (static)" %
          (dS, dV, sum(b["S"] for b in nolines), sum(b["V"] for b in nolines)))
    print("\n%-12s %5s %5s %3s  %8s  %8s  %s" % ("block", "S", "V", "B", "per doc", "S x f", "dominant line of " + BODY))
    src = open(os.path.join(CSRC, BODY)).read().split("\n")
    for b, ln, f in sorted(rows, key=lambda r: -r[0]["S"] * r[2])[:a.top]:
        print("%-12s %5d %5d %3d  %8.2f  %8.1f  %4d: %s" % (b["name"], b["S"], b["V"], b["B"], f, b["S"] * f, ln, src[ln - 1].strip()[:80]))


if __name__ == "__main__":
    main()
