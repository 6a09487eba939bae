#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from separate rocprofv3 --pmc passes -> profiles/traffic.json, keyed by workload/model/documents and
stamped with the identity of the kernels it was taken of (sha256 over blingfire_amd/csrc: bench.py prints traffic_stale when it differs).
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies a wide coalesced read at half its bytes (MI355X_MICROARCH.md, HBM); the
factor for the access shape of the kernel's own streaming reads comes from tools/microbench/stream.hip run under the same counters
(profiles/fetch_calibration.json: known bytes / counter).  The kernel's table gathers are scattered 8-byte reads served from L2; only
their misses reach the counter, so applying the streaming factor to the whole counter is an upper bound of the true traffic.
usage: tools/prof_traffic.py <dir with pmc_fetch/ pmc_write/ [pmc_tcc/]> <key workload/model/ndocs> <kernel substring> [launches per step]"""
import glob, hashlib, json, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, key, kern = sys.argv[1], sys.argv[2], sys.argv[3]
per_step = int(sys.argv[4]) if len(sys.argv) > 4 else 1


def csrc_sha():
    d = os.path.join(ROOT, "blingfire_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(f for f in os.listdir(d) if f.endswith((".h", ".hip", ".cpp", ".map"))):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def counter(sub, name):
    dbs = glob.glob(os.path.join(src, sub, "*.db"))
    if not dbs:
        return None, None
    db = sqlite3.connect(dbs[0])
    # the full-size launches only (the shard; smaller launches of the same kernel belong to other timings of the command)
    mx = db.execute("select max(duration) from counters_collection where kernel_name like ? and counter_name = ?", ("%" + kern + "%", name)).fetchone()[0]
    if mx is None:
        return None, None
    r = db.execute("select avg(value), avg(duration) from counters_collection where kernel_name like ? and counter_name = ? and duration >= ?",
                   ("%" + kern + "%", name, 0.5 * mx)).fetchone()
    return r


fetch, fetch_ns = counter("pmc_fetch", "FETCH_SIZE")
write, write_ns = counter("pmc_write", "WRITE_SIZE")
hit, _ = counter("pmc_tcc", "TCC_HIT_sum")
miss, _ = counter("pmc_tcc", "TCC_MISS_sum")
cal = {}
try:
    cal = json.load(open(os.path.join(ROOT, "profiles", "fetch_calibration.json")))
except Exception:
    pass
ffac = float(cal.get("read8_bytes_per_counted_byte", 1.0))
wfac = float(cal.get("write4_bytes_per_counted_byte", 1.0))
path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(path)) if os.path.exists(path) else {}
ent = {"kernel": kern, "csrc_sha": csrc_sha(), "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "launches_per_step": per_step,
       "fetch_factor": ffac, "write_factor": wfac,
       "hbm_bytes_per_launch_raw": (fetch + write) * 1024.0,
       "hbm_bytes_per_launch": (fetch * ffac + write * wfac) * 1024.0,
       "hbm_bytes_per_step": (fetch * ffac + write * wfac) * 1024.0 * per_step,
       "avg_launch_ns_under_counters": fetch_ns,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, averages over the full-size launches; counters x the calibration factors of "
               "profiles/fetch_calibration.json (known bytes / counter for this kernel's streaming access shapes, tools/microbench/stream.hip)",
       "source": os.path.basename(src.rstrip("/"))}
if hit is not None and miss is not None and hit + miss > 0:
    ent["l2_hit_rate"] = hit / (hit + miss)
tj[key] = ent
json.dump(tj, open(path, "w"), indent=1)
print(key, json.dumps(ent))
