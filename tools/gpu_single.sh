#!/bin/bash
# single-document calls through the drop-in entry point: API tests, calls/s from Python threads and from native threads (tools/single_calls.c)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/single; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_api.py tests/test_golden_api.py tests/test_reference_wrapper.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python tools/bench_single_calls.py > $O/single_calls.txt 2>&1; cat $O/single_calls.txt | grep -v amdgpu
python - <<'PY'
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import bfutil
text, off = bfutil.gen_workload("config2", 4000)
raw = text.tobytes()
with open('/tmp/single_calls_docs.txt', 'wb') as f:
    for i in range(4000):
        d = raw[off[i]:off[i + 1]].replace(b"\n", b" ").replace(b"\r", b" ")
        if d.strip(): f.write(d + b"\n")
PY
model=$(python -c "import sys; sys.path.insert(0,'tests'); import bfutil; print(bfutil.model_path(bfutil.bert_model_name()))")
[ -x tools/single_calls ] || gcc -O2 -Wall -o tools/single_calls tools/single_calls.c -ldl -lpthread
echo "native threads, GPU library:" > $O/single_calls_native.txt
BF_TRACE_ONE=1 timeout 120 ./tools/single_calls blingfire_amd/libblingfiretokdll.so $model /tmp/single_calls_docs.txt 2 >> $O/single_calls_native.txt 2>&1
if [ -f oracle/_ref/libblingfiretokdll_ref.so ]; then
  echo "native threads, reference CPU library (the GPU box's host cores):" >> $O/single_calls_native.txt
  timeout 120 ./tools/single_calls oracle/_ref/libblingfiretokdll_ref.so $model /tmp/single_calls_docs.txt 2 >> $O/single_calls_native.txt 2>&1
fi
grep -v amdgpu $O/single_calls_native.txt
