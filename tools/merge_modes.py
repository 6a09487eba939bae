#!/usr/bin/env python3
"""One process of `bench.py --offsets` under rocprofv3 --pmc: the average duration of k_wp_merge<true> and of its counters per dispatch (the kernel has
two modes, one per process: DESIGN.md section 6).  usage: tools/merge_modes.py <rocprofv3 output dir> [kernel substring]"""
import glob, os, sqlite3, sys
src = sys.argv[1]; kern = sys.argv[2] if len(sys.argv) > 2 else "k_wp_merge"
dbs = glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
if not dbs: sys.exit("no .db under " + src)
db = sqlite3.connect(dbs[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = [t for t in tabs if t.startswith("counters_collection")][0]
rows = list(db.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration), min(duration), max(duration) from %s where kernel_name like ? group by kernel_name, counter_name" % view, ("%" + kern + "%",)))
if not rows: sys.exit("no dispatch of " + kern)
name = rows[0][0].replace("void ", "").replace("bfa::", "").split("(")[0]
print("%s: %d dispatches, %.3f ms (min %.3f max %.3f) |" % (name, rows[0][3], rows[0][4] / 1e6, rows[0][5] / 1e6, rows[0][6] / 1e6), "  ".join("%s %.4g" % (r[1], r[2]) for r in rows))
