#!/usr/bin/env python3
"""Extracts HBM traffic of the dominant kernel from tools/gpu_profile.sh PMC passes and records it in profiles/traffic.json.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced streams by 2x (MI355X_MICROARCH.md §HBM) --
the dominant kernel here issues 16-byte-per-lane scattered gathers, not wide coalesced streams, so the raw counter is used and
the caveat is recorded next to the number.
usage: tools/prof_traffic.py gpurun_out/prof_<tag> <key workload/model/ndocs> <kernel substring>"""
import glob, json, os, sqlite3, sys
src, key, kern = sys.argv[1], sys.argv[2], sys.argv[3]
vals = {}
for name, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    db = sqlite3.connect(glob.glob(os.path.join(src, name, "*.db"))[0])
    r = db.execute("select avg(value), avg(duration) from counters_collection where kernel_name like ? and counter_name = ?", ("%" + kern + "%", counter)).fetchone()
    vals[counter] = r[0]
    vals[counter + "_avg_ns"] = r[1]
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
tj = json.load(open(path)) if os.path.exists(path) else {}
tj[key] = {"kernel": kern, "FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
           "hbm_bytes_per_launch": (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, per-launch averages; raw gfx950 counters (FETCH_SIZE may under-count wide coalesced reads 2x; this kernel's reads are 16-byte scattered gathers)",
           "source": src}
json.dump(tj, open(path, "w"), indent=1)
print(key, tj[key])
