"""Registers, scratch, LDS and occupancy of every kernel of bf_kernels.hip / bf_kernels_sp.hip as the compiler reports them
(-Rpass-analysis=kernel-resource-usage, device side only, nothing is linked).  usage: python tools/kernel_resources.py [name filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
srcs = [os.path.join(ROOT, "blingfire_amd", "csrc", f) for f in os.environ.get("BF_KSRC", "bf_kernels.hip,bf_kernels_sp.hip").split(",")]
procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-c"] +
                          (["-DBF_EXPERIMENTS"] if os.environ.get("BF_EXPERIMENTS") else []) +
                          ["-Rpass-analysis=kernel-resource-usage", src, "-o", "/dev/null"], stderr=subprocess.PIPE, text=True) for src in srcs]
err = "".join(p.communicate()[1] for p in procs)
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur, rows = None, []
for line in err.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(": ")[1]], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("bfa::", "")}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print("%-78s %5s %5s %8s %6s %4s" % ("kernel", "vgpr", "sgpr", "scratch", "lds", "occ"))
for c in rows:
    if flt in c["name"]:
        print("%-78s %5s %5s %8s %6s %4s" % (c["name"][:78], c.get("VGPRs"), c.get("TotalSGPRs"), c.get("ScratchSize [bytes/lane]"), c.get("LDS Size [bytes/block]"), c.get("Occupancy [waves/SIMD]")))
