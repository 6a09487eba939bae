#!/usr/bin/env python3
"""HBM traffic of ONE STEP of a bench command -- every kernel of the library it launches, and the dominant one on its own -- from separate
rocprofv3 --pmc passes -> profiles/traffic.json, keyed by workload/model/documents[/offsets] and stamped with the identity of the kernels it
was taken of (sha256 over blingfire_amd/csrc: bench.py prints traffic_stale when it differs).

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies a wide coalesced read at half its bytes (MI355X_MICROARCH.md, HBM); the
factor for the access shape of the kernels' own streaming reads comes from tools/microbench/stream.hip run under the same counters
(profiles/fetch_calibration.json: known bytes / counter).  Table gathers are scattered 8-byte reads served from L2; only their misses reach the
counter, so applying the streaming factor to the whole counter is an upper bound of the true traffic: both figures are kept.

A step = one pass of the pipeline over the shard.  The command runs W + K of them; the number is taken from the launches of the dominant kernel
(<launches per step> of it per step), every counter is summed over ALL dispatches of a kernel and divided by that number.

usage: tools/prof_traffic.py <dir with pmc_fetch/ pmc_write/ [pmc_tcc/]> <key workload/model/ndocs[/offsets]> <dominant kernel substring> [launches of it per step]"""
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


def short(name):
    n = name.replace("void ", "").replace("bfa::", "")
    return n.split("(")[0][:60]


def per_kernel(sub, counter):
    """{kernel: (sum of the counter over its dispatches, dispatches, sum of durations ns)} for the library's kernels"""
    dbs = glob.glob(os.path.join(src, sub, "**", "*.db"), recursive=True)
    if not dbs:
        return None
    db = sqlite3.connect(dbs[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = [t for t in tabs if t.startswith("counters_collection")][0]
    out = {}
    for name, v, n, dur in db.execute("select kernel_name, sum(value), count(*), sum(duration) from %s where counter_name = ? group by kernel_name" % view, (counter,)):
        if "bfa::" in name:
            out[short(name)] = (float(v), int(n), float(dur or 0))
    return out


fetch = per_kernel("pmc_fetch", "FETCH_SIZE") or {}
write = per_kernel("pmc_write", "WRITE_SIZE") or {}
hit = per_kernel("pmc_tcc", "TCC_HIT_sum") or {}
miss = per_kernel("pmc_tcc", "TCC_MISS_sum") or {}


def steps_of(tab):
    n = sum(v[1] for k, v in tab.items() if kern in k)
    return max(n / float(per_step), 1.0)


cal = {}
try:
    cal = json.load(open(os.path.join(ROOT, "profiles", "fetch_calibration.json")))
except Exception:
    pass
ffac = float(cal.get("read8_bytes_per_counted_byte", 1.0))
wfac = float(cal.get("write4_bytes_per_counted_byte", 1.0))
sf, sw = steps_of(fetch), steps_of(write)
kernels = {}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, (0, 0, 0))[0] / sf
    w = write.get(k, (0, 0, 0))[0] / sw
    ent = {"FETCH_SIZE_KiB_per_step": f, "WRITE_SIZE_KiB_per_step": w, "launches_per_step": fetch.get(k, write.get(k))[1] / (sf if k in fetch else sw),
           "ms_per_step_under_counters": fetch.get(k, (0, 0, 0))[2] / sf / 1e6}
    if k in hit and k in miss and hit[k][0] + miss[k][0] > 0:
        ent["l2_hit_rate"] = hit[k][0] / (hit[k][0] + miss[k][0])
    kernels[k] = ent
tot_f = sum(v["FETCH_SIZE_KiB_per_step"] for v in kernels.values())
tot_w = sum(v["WRITE_SIZE_KiB_per_step"] for v in kernels.values())
dom_f = sum(v["FETCH_SIZE_KiB_per_step"] for k, v in kernels.items() if kern in k)
dom_w = sum(v["WRITE_SIZE_KiB_per_step"] for k, v in kernels.items() if kern in k)
path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(path)) if os.path.exists(path) else {}
ent = {"kernel": kern, "csrc_sha": csrc_sha(), "steps_profiled": sf, "fetch_factor": ffac, "write_factor": wfac,
       "FETCH_SIZE_KiB_per_step": tot_f, "WRITE_SIZE_KiB_per_step": tot_w,
       "hbm_bytes_per_step_raw": (tot_f + tot_w) * 1024.0,
       "hbm_bytes_per_step": (tot_f * ffac + tot_w * wfac) * 1024.0,
       "dominant_hbm_bytes_per_step_raw": (dom_f + dom_w) * 1024.0,
       "dominant_hbm_bytes_per_step": (dom_f * ffac + dom_w * wfac) * 1024.0,
       "kernels": kernels,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, summed over EVERY kernel of the library the command launches and divided by the number of steps "
               "it ran; `hbm_bytes_per_step` = counters x the calibration factors of profiles/fetch_calibration.json (known bytes / counter for streaming access shapes, "
               "tools/microbench/stream.hip), `..._raw` = the counters as they are; `dominant_...` = the share of the kernel a bench line's roofline is about",
       "source": os.path.basename(src.rstrip("/"))}
th = sum(v[0] for v in hit.values())
tm = sum(v[0] for v in miss.values())
if th + tm > 0:
    ent["l2_hit_rate"] = th / (th + tm)
tj[key] = ent
json.dump(tj, open(path, "w"), indent=1)
print(key, json.dumps({k: v for k, v in ent.items() if k != "kernels"}))
for k, v in kernels.items():
    print("  %-58s fetch %10.0f KiB  write %10.0f KiB  %.3f ms" % (k, v["FETCH_SIZE_KiB_per_step"], v["WRITE_SIZE_KiB_per_step"], v["ms_per_step_under_counters"]))
