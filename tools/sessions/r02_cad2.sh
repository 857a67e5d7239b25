#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
cc -O2 -std=c11 -I include examples/cadence.c -o /tmp/cadence master_amd/libmi_pt.so -Wl,-rpath,$PWD/master_amd || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_cadence_c5.py -m gpu -x -q -k "frames or batched or async" > gpurun_out/r02/cad2_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 gpurun_out/r02/cad2_tests.log
[ $rc = 0 ] || exit 1
: > gpurun_out/r02/cadence_b16.txt
for B in 4 8 16; do for c in 2 4 8; do
  echo "frames per launch $B, frames per wave $c" >> gpurun_out/r02/cadence_b16.txt
  MI_PT_FRAME_CHUNK=$c timeout -k 10 120 /tmp/cadence scenes/CornellBoxDiffuse.miscene 512 512 480 8 $B >> gpurun_out/r02/cadence_b16.txt 2>&1
done; done
for B in 4 8; do
  echo "1080p frames per launch $B" >> gpurun_out/r02/cadence_b16.txt
  timeout -k 10 120 /tmp/cadence scenes/CornellBoxDiffuse.miscene 1920 1080 96 8 $B >> gpurun_out/r02/cadence_b16.txt 2>&1
done
python - <<'PY'
import json
for line in open("gpurun_out/r02/cadence_b16.txt"):
    if line.startswith("{"):
        d = json.loads(line)
        print("  %4dx%-4d sync %.3f async %.3f async+add %.3f ms/frame (x%.2f)  device %.3f / %.3f  batched %.3f %s" % (d["width"], d["height"], d["sync_ms_per_frame"], d["async_ms_per_frame"], d["async_wait_add_ms_per_frame"], d["speedup_wait_add"], d["sync_device_ms_per_frame"], d["async_device_ms_per_frame"], d["batched_call_ms_per_frame"], "" if d["views_bit_identical"] else "VIEWS DIFFER"))
    else:
        print(line.strip())
PY
