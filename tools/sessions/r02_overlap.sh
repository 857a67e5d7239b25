#!/bin/bash
# Do the frame batches of different streams overlap on the device?  Kernel trace of the cadence loop.
set -u
O=gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cc -O2 -std=c11 -I include examples/cadence.c -o /tmp/cadence master_amd/libmi_pt.so -Wl,-rpath,$PWD/master_amd
for B in 4 8; do timeout -k 10 100 /tmp/cadence scenes/CornellBoxDiffuse.miscene 512 512 400 8 $B; timeout -k 10 100 /tmp/cadence scenes/CornellBoxDiffuse.miscene 1920 1080 96 8 $B; done
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/overlap -- /tmp/cadence scenes/CornellBoxDiffuse.miscene 512 512 64 8 8 > $O/overlap.txt 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r02/overlap/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "megakernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
out = open("gpurun_out/r02/overlap_summary.txt", "w")
prev_end = None
for r in rows[-40:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    line = "q %s  grid %6s  start %10.1f us  end %10.1f us  dur %7.1f us  gap_to_prev_end %8.1f us" % (r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), s / 1e3, e / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end is not None else 0.0)
    print(line); out.write(line + "\n")
    prev_end = e
PY
