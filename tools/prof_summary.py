#!/usr/bin/env python3
"""Turns the rocprofv3 (rocpd sqlite) outputs of tools/gpu_profile.sh into a small text summary for profiles/.
usage: tools/prof_summary.py gpurun_out/prof_<tag> profiles/<name>.txt"""
import glob
import os
import sqlite3
import sys


def rows(db, q):
    cur = db.execute(q)
    return [d[0] for d in cur.description], cur.fetchall()


def main():
    src, dst = sys.argv[1], sys.argv[2]
    out = []
    st = glob.glob(os.path.join(src, "stats", "**", "*.db"), recursive=True)
    if st:
        db = sqlite3.connect(st[0])
        out.append("== rocprofv3 --kernel-trace --stats (durations in ns) ==")
        out.append("%-52s %6s %14s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
        _, rs = rows(db, "select name,total_calls,total_duration,average,percentage from top_kernels")
        for r in rs:
            out.append("%-52s %6d %14.0f %12.0f %7.2f" % (r[0][:52], r[1], r[2] * 1, r[3] * 1, r[4]))
        out.append("(top_kernels view reports microseconds x1000 = ns when multiplied; raw per-dispatch durations below are ns)")
        _, rs = rows(db, "select name, count(*), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), max(scratch_size), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc")
        out.append("")
        out.append("%-44s %5s %12s %12s %12s %5s %5s %8s %6s %10s %4s" % ("kernel", "n", "avg_ns", "min_ns", "max_ns", "vgpr", "sgpr", "scratch", "lds", "grid", "wg"))
        for r in rs:
            out.append("%-44s %5d %12.0f %12.0f %12.0f %5d %5d %8d %6d %10d %4d" % (r[0][:44], r[1], r[2], r[3], r[4], r[5] or 0, r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[10] or 0))
        # bench.py launches the tokenising kernels on two batch sizes (the shard; the 1.25 M-document sample of the PCIe timings):
        # the launches that last at least half as long as the longest one are the shard's
        out.append("")
        out.append("full-size launches (duration >= 0.5 x the kernel's longest): the ones the bench line's kernel_ms is about")
        out.append("%-44s %5s %12s %12s %12s" % ("kernel", "n", "avg_ns", "min_ns", "max_ns"))
        _, ks = rows(db, "select name, max(duration) from kernels where name like '%bfa::%' group by name order by sum(duration) desc")
        for name, mx in ks:
            _, q = rows(db, "select count(*), avg(duration), min(duration), max(duration) from kernels where name = '%s' and duration >= %f" % (name.replace("'", "''"), 0.5 * mx))
            out.append("%-44s %5d %12.0f %12.0f %12.0f" % (name[:44], q[0][0], q[0][1], q[0][2], q[0][3]))
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
        if not dbs:
            continue
        db = sqlite3.connect(dbs[0])
        out.append("")
        out.append("== rocprofv3 --pmc pass %s (per-dispatch averages over the bfa:: kernels) ==" % os.path.basename(d))
        _, rs = rows(db, "select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%bfa::%' group by kernel_name, counter_name order by kernel_name, counter_name")
        out.append("%-44s %-22s %5s %18s %12s" % ("kernel", "counter", "n", "avg_value", "avg_ns"))
        for r in rs:
            out.append("%-44s %-22s %5d %18.1f %12.0f" % (r[0][:44], r[1], r[2], r[3], r[4]))
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
