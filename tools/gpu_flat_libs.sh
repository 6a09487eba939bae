#!/bin/bash
# timing experiments: the same bench with other builds of the library (gpurun_tmp/lib_*.so), no verification
set -u
export TMPDIR=/tmp
cp blingfire_amd/libblingfiretokdll.so /tmp/lib_orig.so
for l in "$@"; do
  cp gpurun_tmp/lib_$l.so blingfire_amd/libblingfiretokdll.so
  echo "== $l"; VERIFY=0 bash tools/gpu_flat_occ.sh 3 $((3 + (7<<16))) $((3 + (6<<16))) $((3 + (5<<16)))
done
cp /tmp/lib_orig.so blingfire_amd/libblingfiretokdll.so
