#!/bin/bash
# A/B builds of the product library in ONE GPU session (same box, interleaved): tools/gpu_ab.sh libA libB ... -- [bench args]
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "$1" == "--" ] && shift
cp blingfire_amd/libblingfiretokdll.so /tmp/lib_keep.so
for rep in 1 2; do for l in "${libs[@]}"; do
  cp $l blingfire_amd/libblingfiretokdll.so
  timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l', '%.1f M docs/s' % (d['value']/1e6), {k: round(x,3) for k,x in d['kernel_ms'].items()})"
done; done
cp /tmp/lib_keep.so blingfire_amd/libblingfiretokdll.so
