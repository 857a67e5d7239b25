#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
cc -O2 -std=c11 -I include examples/cadence.c -o /tmp/cadence master_amd/libmi_pt.so -Wl,-rpath,$PWD/master_amd || exit 1
: > gpurun_out/r02/cadence_1080.txt
for cfg in "8 4" "8 8" "16 4" "16 8" "16 16"; do set -- $cfg
  echo "1080p frames per launch $1, frames per wave $2" >> gpurun_out/r02/cadence_1080.txt
  MI_PT_FRAME_CHUNK=$2 timeout -k 10 120 /tmp/cadence scenes/CornellBoxDiffuse.miscene 1920 1080 96 8 $1 >> gpurun_out/r02/cadence_1080.txt 2>&1
  echo "LivingRoom 1080p frames per launch $1, frames per wave $2" >> gpurun_out/r02/cadence_1080.txt
  MI_PT_FRAME_CHUNK=$2 timeout -k 10 120 /tmp/cadence scenes/LivingRoomLit.miscene 1920 1080 32 0 $1 >> gpurun_out/r02/cadence_1080.txt 2>&1
done
python - <<'PY'
import json
for line in open("gpurun_out/r02/cadence_1080.txt"):
    if line.startswith("{"):
        d = json.loads(line)
        print("  %4dx%-4d sync %.3f async %.3f async+add %.3f ms/frame (x%.2f)  device %.3f / %.3f  batched %.3f %s" % (d["width"], d["height"], d["sync_ms_per_frame"], d["async_ms_per_frame"], d["async_wait_add_ms_per_frame"], d["speedup_wait_add"], d["sync_device_ms_per_frame"], d["async_device_ms_per_frame"], d["batched_call_ms_per_frame"], "" if d["views_bit_identical"] else "VIEWS DIFFER"))
    else:
        print(line.strip())
PY
