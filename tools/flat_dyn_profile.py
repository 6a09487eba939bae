#!/usr/bin/env python3
"""tools/wave_dyn_profile.py for the flat program (bf_flat_body.h, k_wp_flat): where do its instructions go PER 512-BYTE CHUNK -- and how many
of them are SGPR spill traffic (v_readlane / v_writelane / v_readfirstlane) or scratch -- estimated WITHOUT a GPU: the kernel's ISA cut into
basic blocks with their source lines x the simulator's line counts (gcov) on a sample of the metric's corpus.  VERDICT r05 item 3(a).
usage: python tools/flat_dyn_profile.py [--docs N] [--top K] [--kernel SUBSTRING]"""
import argparse, collections, ctypes, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
CSRC = os.path.join(ROOT, "blingfire_amd", "csrc")
BODY = "bf_flat_body.h"


def isa_blocks(work, kernel):
    s_path = os.path.join(work, "kernels.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
                           "-gline-tables-only", os.path.join(CSRC, "bf_kernels.hip"), "-o", s_path], stderr=subprocess.DEVNULL)
    lines = open(s_path).read().split("\n")
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    start = next(i for i, l in enumerate(lines) if re.match(r'^_ZN3bfa\S*%s\S*:' % kernel, l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur, loc = [], None, None
    for l in lines[start:end]:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m or cur is None:
            cur = dict(name=m.group(1) if m else "entry", S=0, V=0, L=0, M=0, B=0, X=0, Q=0, locs=collections.Counter())
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
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            cur["X"] += 1
        if op.startswith("scratch_"):
            cur["Q"] += 1
        if loc and loc[0] == BODY and loc[1] > 0:
            cur["locs"][loc[1]] += 1
    return blocks


def line_frequencies(work, ndocs):
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
    L.bft_emu_flat_batch.restype = ctypes.c_long
    L.bft_emu_flat_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    wl = bfutil.WORKLOADS["headline512"]
    text, off = bfutil.gen_workload("headline512", ndocs)
    h = L.bft_load(bfutil.model_path(bfutil.bert_model_name()).encode())
    cap = len(text) + 16
    ids = np.zeros(cap, dtype=np.int32); ido = np.zeros(ndocs + 1, dtype=np.int64); st = np.zeros(16, dtype=np.uint64)
    r = L.bft_emu_flat_batch(h, text.ctypes.data, len(text), off.ctypes.data, ndocs, wl["max_ids"], wl["unk"], 2, 4, ids.ctypes.data, cap, ido.ctypes.data, st.ctypes.data, 0)
    assert r >= 0, r
    try:
        ctypes.CDLL(None).__gcov_dump()
    except Exception:
        pass
    return [int(len(text))] + st.tolist()


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
    ap.add_argument("--docs", type=int, default=600)
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--kernel", default="9k_wp_flatILi7ELb0E")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as work:
        blocks = isa_blocks(work, a.kernel)
        code = ("import sys; sys.path.insert(0, %r); import flat_dyn_profile as w; print(w.line_frequencies(%r, %d))" % (os.path.join(ROOT, "tools"), work, a.docs))
        out = subprocess.check_output([sys.executable, "-c", code], cwd=work).decode().strip().split("\n")[-1]
        stats = eval(out)
        ex = read_gcov(work)
    nchunks = float(stats[1]) if stats[1] else stats[0] / 512.0
    print("simulator, %d documents of the metric's corpus: %d bytes, %.0f chunks (%.2f per document)" % (a.docs, stats[0], nchunks, nchunks / a.docs))
    tot = {k: sum(b[k] for b in blocks) for k in "SVBXQ"}
    print("static: %d scalar, %d vector (%d of them lane moves = SGPR spill traffic), %d branch, %d scratch instructions in %d blocks" % (tot["S"], tot["V"], tot["X"], tot["B"], tot["Q"], len(blocks)))
    rows, nolines = [], []
    for b in blocks[1:]:
        if not b["locs"]:
            nolines.append(b); continue
        ln = b["locs"].most_common(1)[0][0]
        f = ex.get(ln, 0) / 64.0 / nchunks              # passes of a wave per chunk
        rows.append((b, ln, f))
    d = {k: sum(b[k] * f for b, _, f in rows) for k in "SVBXQL"}
    print("estimate per 512-byte chunk: %.0f scalar + %.0f vector (of them %.0f lane moves) + %.0f LDS + %.0f branch instructions, %.1f scratch; blocks without a line: %d scalar, %d vector, %d lane moves (static)" %
          (d["S"], d["V"], d["X"], d["L"], d["B"], d["Q"], sum(b["S"] for b in nolines), sum(b["V"] for b in nolines), sum(b["X"] for b in nolines)))
    src = open(os.path.join(CSRC, BODY)).read().split("\n")
    print("\nblocks by lane moves x passes per chunk:\n%-12s %5s %5s %5s  %9s  %8s  %s" % ("block", "S", "V", "lane", "per chunk", "lane x f", "dominant line of " + BODY))
    for b, ln, f in sorted(rows, key=lambda r: -r[0]["X"] * r[2])[:a.top]:
        print("%-12s %5d %5d %5d  %9.2f  %8.1f  %4d: %s" % (b["name"], b["S"], b["V"], b["X"], f, b["X"] * f, ln, src[ln - 1].strip()[:90]))


if __name__ == "__main__":
    main()
