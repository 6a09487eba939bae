#!/usr/bin/env python3
"""Static instruction counts of the kernels of a translation unit (hipcc -S): total, vector, scalar, LDS, vector memory, branches.
usage: python tools/isa_count.py [bf_kernels_sp.hip] [name filter]"""
import os, re, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else "bf_kernels_sp.hip"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = "/tmp/isa_%s.s" % os.path.basename(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
                       os.path.join(ROOT, "blingfire_amd", "csrc", src), "-o", out] + (["-DBF_EXPERIMENTS"] if os.environ.get("BF_EXPERIMENTS") else []), stderr=subprocess.DEVNULL)
s = open(out).read()
for m in re.finditer(r'^(_ZN3bfa\S+):\s*;', s, re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("bfa::", "")
    if flt not in name: continue
    end = s.index('.Lfunc_end', m.end())
    c = collections.Counter()
    for l in s[m.end():end].split('\n'):
        l = l.strip()
        if not l or l.startswith('.') or l.startswith(';') or l.endswith(':'): continue
        op = l.split()[0]
        c['total'] += 1
        if op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'): c['lane'] += 1
        if op.startswith('v_'): c['valu'] += 1
        elif op.startswith('s_cbranch') or op.startswith('s_branch'): c['branch'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        elif op.startswith('ds_'): c['lds'] += 1
        elif op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): c['vmem'] += 1; c['scratch'] += op.startswith('scratch_')
    print("%-60s total %5d valu %5d salu %5d branch %4d lds %4d vmem %4d lane-moves %4d scratch %3d" % (name[:60], c['total'], c['valu'], c['salu'], c['branch'], c['lds'], c['vmem'], c['lane'], c['scratch']))
