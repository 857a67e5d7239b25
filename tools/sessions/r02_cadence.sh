#!/bin/bash
# Frame cadence (one sample per Technique::render call): frames per launch x frames per wave of the frame variant, against the accumulating kernel.
set -u
O=gpurun_out/r02; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_cadence_c5.py -x -q > $O/cadence_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/cadence_tests.log
cc -O2 -std=c11 -I include examples/cadence.c -o /tmp/cadence master_amd/libmi_pt.so -Wl,-rpath,$PWD/master_amd
: > $O/cadence_tuning.txt
run() { # label env...
  echo "$1" >> $O/cadence_tuning.txt; shift
  env "$@" timeout -k 10 120 /tmp/cadence scenes/CornellBoxDiffuse.miscene 512 512 400 8 $B >> $O/cadence_tuning.txt 2>&1
  env "$@" timeout -k 10 120 /tmp/cadence scenes/CornellBoxDiffuse.miscene 1920 1080 96 8 $B >> $O/cadence_tuning.txt 2>&1
  env "$@" timeout -k 10 120 /tmp/cadence scenes/LivingRoomLit.miscene 1920 1080 32 0 $B >> $O/cadence_tuning.txt 2>&1
}
B=4 run "accumulating kernel, 4 launches per batch" MI_PT_FRAME_MODE=0
for B in 4 8; do for c in 1 2 4; do run "frames per launch $B, frames per wave $c" MI_PT_FRAME_CHUNK=$c; done; done
python - <<'PY'
import json
for line in open("gpurun_out/r02/cadence_tuning.txt"):
    if line.startswith("{"):
        d = json.loads(line)
        print("  %-22s %4dx%-4d sync %.3f async %.3f async+add %.3f ms/frame (x%.2f)  device %.3f / %.3f  batched %.3f %s" % (d["scene"].split("/")[-1][:22], d["width"], d["height"], d["sync_ms_per_frame"], d["async_ms_per_frame"], d["async_wait_add_ms_per_frame"], d["speedup_wait_add"], d["sync_device_ms_per_frame"], d["async_device_ms_per_frame"], d["batched_call_ms_per_frame"], "" if d["views_bit_identical"] else "VIEWS DIFFER"))
    else:
        print(line.strip())
PY
