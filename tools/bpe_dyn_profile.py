#!/usr/bin/env python3
"""tools/wave_dyn_profile.py for the BPE wave program (bf_bpe_wave_body.h + bf_wave_body.h, k_bpe_wave): where do its instructions go PER DOCUMENT of the
config-3 corpus -- the kernel's ISA cut into basic blocks with their source lines x the simulator's line counts (gcov).  usage: python tools/bpe_dyn_profile.py [--docs N] [--top K] [--notable]"""
import argparse, collections, ctypes, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
CSRC = os.path.join(ROOT, "blingfire_amd", "csrc")
BODIES = ("bf_bpe_wave_body.h", "bf_wave_body.h")


def isa_blocks(work):
    s_path = os.path.join(work, "kernels.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
                           "-gline-tables-only", os.path.join(CSRC, "bf_kernels_sp.hip"), "-o", s_path], stderr=subprocess.DEVNULL)
    lines = open(s_path).read().split("\n")
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    start = next(i for i, l in enumerate(lines) if re.match(r'^_ZN3bfa10k_bpe_wave\S*:', l))
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
        if loc and loc[0] in BODIES and loc[1] > 0:
            cur["locs"][loc] += 1
    return blocks


def line_frequencies(work, ndocs, notable):
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
    L.bft_emu_bpe_wave_batch.restype = ctypes.c_long
    L.bft_emu_bpe_wave_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    text, off = bfutil.gen_workload("config3", ndocs)
    h = L.bft_load(bfutil.model_path("gpt2.bin").encode())
    cap = 2 * len(text) + 2 * ndocs + 16
    ids = np.zeros(cap, dtype=np.int32); ido = np.zeros(ndocs + 1, dtype=np.int64); fl = np.zeros(ndocs + 1, dtype=np.int32); st = np.zeros(16, dtype=np.uint64)
    r = L.bft_emu_bpe_wave_batch(h, text.ctypes.data, len(text), off.ctypes.data, ndocs, 2048, 0, 1, 8, 8 if notable else 0, ids.ctypes.data, cap, ido.ctypes.data, fl.ctypes.data, st.ctypes.data)
    assert r >= 0, r
    try:
        ctypes.CDLL(None).__gcov_dump()
    except Exception:
        pass
    return [int(len(text))] + st.tolist()


def read_gcov(work):
    subprocess.call(["gcov", "-o", ".", "libcov.so-bf_wavetest.gcno"], cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ex = {}
    for body in BODIES:
        try:
            for l in open(os.path.join(work, body + ".gcov")):
                m = re.match(r'\s*([0-9]+)\*?:\s*(\d+):', l)
                if m:
                    ex[(body, int(m.group(2)))] = max(ex.get((body, int(m.group(2))), 0), int(m.group(1)))
        except FileNotFoundError:
            pass
    return ex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=300)
    ap.add_argument("--top", type=int, default=30)
    ap.add_argument("--notable", action="store_true")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as work:
        blocks = isa_blocks(work)
        code = ("import sys; sys.path.insert(0, %r); import bpe_dyn_profile as w; print(w.line_frequencies(%r, %d, %r))" % (os.path.join(ROOT, "tools"), work, a.docs, a.notable))
        out = subprocess.check_output([sys.executable, "-c", code], cwd=work).decode().strip().split("\n")[-1]
        stats = eval(out)
        ex = read_gcov(work)
    nd = float(a.docs)
    print("simulator, %d documents of config 3 (%d bytes), one wave: %.1f words begun by a unit, %.1f taken whole by a unit, %.1f answered by the table, per document" %
          (a.docs, stats[0], stats[1] / nd, stats[2] / nd, stats[13] / nd))
    rows, nolines = [], []
    for b in blocks[1:]:
        if not b["locs"]:
            nolines.append(b); continue
        loc = b["locs"].most_common(1)[0][0]
        f = ex.get(loc, 0) / 64.0 / nd
        rows.append((b, loc, f))
    d = {k: sum(b[k] * f for b, _, f in rows) for k in "SVBLM"}
    print("estimate per document: %.0f scalar + %.0f vector + %.0f LDS + %.0f memory + %.0f branch instructions; blocks without a line: %d scalar, %d vector (static)" %
          (d["S"], d["V"], d["L"], d["M"], d["B"], sum(b["S"] for b in nolines), sum(b["V"] for b in nolines)))
    src = {body: open(os.path.join(CSRC, body)).read().split("\n") for body in BODIES}
    # by source function region: aggregate per dominant line
    print("\n%-12s %5s %5s %4s  %8s  %9s  %s" % ("block", "S", "V", "L", "per doc", "(S+V) x f", "dominant line"))
    for b, loc, f in sorted(rows, key=lambda r: -(r[0]["S"] + r[0]["V"]) * r[2])[:a.top]:
        print("%-12s %5d %5d %4d  %8.2f  %9.1f  %s:%d: %s" % (b["name"], b["S"], b["V"], b["L"], f, (b["S"] + b["V"]) * f, loc[0].replace("bf_", "").replace("_body.h", ""), loc[1], src[loc[0]][loc[1] - 1].strip()[:70]))


if __name__ == "__main__":
    main()
