"""Build recipe for the native pieces (explicit hipcc / gcc invocations, in-tree outputs).

  blingfire_amd/libblingfiretokdll.so   the product: HIP kernels (gfx950) + C-ABI
  oracle/liboracle.so                   TEST INFRA: plain-C restatement of the reference path
  oracle/libcpubaseline.so              TEST INFRA: multi-threaded driver that times a TextToIds .so on host cores
  tools/libcorpusgen.so                 TEST INFRA: deterministic synthetic corpus generator
  tools/single_calls                    TEST INFRA: native-thread harness for single-document calls through the C-ABI
  tools/microbench/{stream,gather,valu_issue}  MEASUREMENT: counter calibration, gather ceiling, VALU / SALU issue rate
  tests/hosttest/libbf_hosttest.so      TEST INFRA: table-equivalence + host emulation of the lane programs
  oracle/_ref/libblingfiretokdll_ref.so TEST INFRA: the unmodified reference, only when /root/reference exists
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "blingfire_amd", "csrc")
PRODUCT = os.path.join(ROOT, "blingfire_amd", "libblingfiretokdll.so")
REF_DIR = os.environ.get("BF_REFERENCE", "/root/reference")


def _run(cmd, cwd=None):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=cwd)


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def build_product(force=False):
    """the product: four translation units compiled side by side (the two kernel files are most of the time), then linked"""
    names = ("bf_kernels.hip", "bf_kernels_sp.hip", "bf_capi.cpp", "bf_model.cpp")
    srcs = [os.path.join(CSRC, f) for f in names]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [
        os.path.join(ROOT, "include", "blingfiretokdll_amd.h"), os.path.join(CSRC, "exports.map"), os.path.join(ROOT, "models", "wbd.bin"), os.path.join(ROOT, "models", "sbd.bin")]
    odir0 = os.path.join(CSRC, "obj")
    os.makedirs(odir0, exist_ok=True)
    # the flag set is part of what the library is built from: toggling BF_EXPERIMENTS must not keep the old objects
    stamp = os.path.join(odir0, "flags.stamp")
    want = "experiments" if os.environ.get("BF_EXPERIMENTS") else "product"
    have = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if have != want:
        force = True
    if force or _newer(PRODUCT, deps):
        # the reference compiles its default word- and sentence-breaking models (wbd.bin, sbd.bin) into the library; same here (data, via .incbin)
        wbd = os.path.join(ROOT, "models", "wbd.bin")
        sbd = os.path.join(ROOT, "models", "sbd.bin")
        odir = os.path.join(CSRC, "obj")
        os.makedirs(odir, exist_ok=True)
        flags = ["--offload-arch=gfx950", "--offload-compress", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-result", "-fvisibility=hidden"] + (
            ["-DBF_EXPERIMENTS"] if os.environ.get("BF_EXPERIMENTS") else []) + ['-DBF_DEFAULT_WBD_PATH="%s"' % wbd, '-DBF_DEFAULT_SBD_PATH="%s"' % sbd]
        objs, procs = [], []
        for src in srcs:
            obj = os.path.join(odir, os.path.basename(src) + ".o")
            cmd = [hipcc()] + flags + ["-c", "-x", "hip", src, "-o", obj]
            print("+", " ".join(cmd), flush=True)
            objs.append(obj); procs.append((cmd, subprocess.Popen(cmd)))
        for cmd, pr in procs:
            if pr.wait() != 0:
                raise subprocess.CalledProcessError(pr.returncode, cmd)
        _run([hipcc(), "--offload-arch=gfx950", "--offload-compress", "-shared", "-fPIC", "-Wl,-s", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map")] + objs + ["-o", PRODUCT])
        with open(stamp, "w") as f:
            f.write(want + "\n")
    return PRODUCT


def build_test_infra(force=False):
    odir = os.path.join(ROOT, "oracle")
    lib = os.path.join(odir, "liboracle.so")
    if force or _newer(lib, [os.path.join(odir, "bf_oracle.c"), os.path.join(odir, "bf_oracle.h"), os.path.join(odir, "bf_tolower_tab.h")]):
        _run(["gcc", "-O2", "-Wall", "-Wextra", "-std=c99", "-fPIC", "-shared", "bf_oracle.c", "-o", "liboracle.so"], cwd=odir)
    drv = os.path.join(odir, "libcpubaseline.so")
    if force or _newer(drv, [os.path.join(odir, "cpu_baseline.c")]):
        _run(["gcc", "-O2", "-Wall", "-std=gnu99", "-fPIC", "-shared", "cpu_baseline.c", "-o", "libcpubaseline.so", "-ldl", "-lpthread"], cwd=odir)
    cg = os.path.join(ROOT, "tools", "libcorpusgen.so")
    if force or _newer(cg, [os.path.join(ROOT, "tools", "corpusgen.c")]):
        _run(["gcc", "-O2", "-Wall", "-std=gnu99", "-fPIC", "-shared", "corpusgen.c", "-o", "libcorpusgen.so", "-lm", "-lpthread"],
             cwd=os.path.join(ROOT, "tools"))
    sc = os.path.join(ROOT, "tools", "single_calls")
    if force or _newer(sc, [os.path.join(ROOT, "tools", "single_calls.c")]):
        _run(["gcc", "-O2", "-Wall", "-o", "single_calls", "single_calls.c", "-ldl", "-lpthread"], cwd=os.path.join(ROOT, "tools"))
    # microbenchmarks the evidence scripts run on the GPU box (counter calibration, gather ceiling, issue rate): binaries, not tracked
    mb = os.path.join(ROOT, "tools", "microbench")
    for name in ("stream", "gather", "valu_issue", "gather_sweep"):
        src, out = os.path.join(mb, name + ".hip"), os.path.join(mb, name)
        if os.path.exists(src) and (force or _newer(out, [src])):
            _run([hipcc(), "--offload-arch=gfx950", "-O2", "-w", "-o", out, src])
    ht = os.path.join(ROOT, "tests", "hosttest", "libbf_hosttest.so")
    ht_src = [os.path.join(ROOT, "tests", "hosttest", "bf_hosttest.cpp"), os.path.join(ROOT, "tests", "hosttest", "bf_wavetest.cpp"), os.path.join(CSRC, "bf_model.cpp")]
    ht_dep = ht_src + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(odir, "bf_oracle.c"),
                                                                                              os.path.join(ROOT, "tests", "hosttest", "wave_emu.h"), os.path.join(ROOT, "tests", "hosttest", "hosttest.h")]
    if force or _newer(ht, ht_dep):
        obj = os.path.join(ROOT, "tests", "hosttest", "bf_oracle.o")
        _run(["gcc", "-O2", "-std=c99", "-fPIC", "-c", os.path.join(odir, "bf_oracle.c"), "-o", obj])
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", ht] + ht_src + [obj])
    ref = os.path.join(odir, "_ref", "libblingfiretokdll_ref.so")
    wrapper = os.path.join(odir, "_ref", "blingfire", "__init__.py")
    dictref = os.path.join(odir, "_ref", "libdictref.so")
    if os.path.isdir(os.path.join(REF_DIR, "blingfireclient.library")) and (
            force or not os.path.exists(ref) or not os.path.exists(wrapper) or _newer(dictref, [os.path.join(odir, "ref_dict_glue.cpp")])):
        _run(["make", "-C", odir, "ref", "REF=" + REF_DIR])


def build_all(force=False):
    build_product(force)
    build_test_infra(force)


if __name__ == "__main__":
    build_all("--force" in sys.argv)
