#!/bin/bash
# round 4, session V: the GPU tests added after the evidence run (instances of k_wp_wave, of the _sp prologue, of the offsets program; the mapped path's capacity error)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4v; mkdir -p $O
timeout 200 python -m pytest "tests/test_gpu_parity_wp.py::test_wave_program_instances_agree" "tests/test_gpu_parity_sp.py::test_prologue_instances_agree" "tests/test_gpu_api.py::test_small_wordpiece_batch_capacity_error_writes_no_ids" "tests/test_offsets.py::test_gpu_offsets_match_checker" -m gpu -x -q > $O/pytest_new.txt 2>&1; tail -4 $O/pytest_new.txt
