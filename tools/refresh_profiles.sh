#!/bin/bash
# Regenerates the measured artefacts of profiles/$R on a GPU box (run through gpurun; outputs under gpurun_out/$R, copied to profiles/$R).
#   R=r04 tools/refresh_profiles.sh core   # rocprof kernel stats of the bench command, the bench line with its live counters -> pmc summaries + traffic.json, cadence, BPT
#   R=r04 tools/refresh_profiles.sh full   # full-size configurations C3'-C5', real mid-size models, the HBM-bound 2 M triangle case
set -u
R=${R:-r04}
O=gpurun_out/$R; mkdir -p $O profiles/$R
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "${1:-core}" = core ]; then
  # r04: the bench line measures its own hardware counters (rocprofv3 --pmc child processes, bench.py collect_live_pmc); tools/bench_to_profiles.py files them
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-time-to-rmse --no-fast-variant > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
  cp $O/prof/*/*kernel_stats.csv $O/bench_kernel_stats.csv
  python bench.py --steps 20 --warmup 5 --pmc-dir $O/pmc_live > $O/bench_n1.json 2> $O/bench_n1.err
  python tools/bench_to_profiles.py $O/bench_n1.json $R > $O/bench_to_profiles.log 2>&1
  cc -O2 -std=c11 -I include examples/cadence.c -o /tmp/cadence master_amd/libmi_pt.so -Wl,-rpath,$PWD/master_amd
  : > $O/cadence.jsonl
  /tmp/cadence scenes/CornellBoxDiffuse.miscene 512 512 480 8 8 >> $O/cadence.jsonl 2>&1
  /tmp/cadence scenes/CornellBoxDiffuse.miscene 1920 1080 96 8 4 >> $O/cadence.jsonl 2>&1
  /tmp/cadence scenes/LivingRoomLit.miscene 1920 1080 32 0 4 >> $O/cadence.jsonl 2>&1
  # BPT: 64 frames of 512^2 per model, per-kernel stats and counters of the largest one
  python tools/bpt_time.py CornellBoxDiffuse CornellBoxSpecular MetalRings LivingRoomLit > $O/bpt_times.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/bpt_prof -- python3 tools/bpt_prof.py LivingRoomLit > $O/bpt_prof.out 2> $O/bpt_prof.err
  cp $O/bpt_prof/*/*kernel_stats.csv $O/bpt_kernel_stats_LivingRoomLit.csv
  tools/pmc_bpt.sh $R/pmc_bpt_livingroom LivingRoomLit > $O/pmc_bpt.log 2>&1
  cp $O/pmc_bpt_livingroom/summary.txt $O/pmc_bpt_livingroom.txt
  for f in bench_under_rocprof.json bench_kernel_stats.csv cadence.jsonl bpt_times.txt bpt_kernel_stats_LivingRoomLit.csv pmc_bpt_livingroom.txt; do cp $O/$f profiles/$R/$f; done
else
  run() { python bench.py --scene "$1" --width $2 --height $3 --spp $4 --max-path 0 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload --no-time-to-rmse --no-fast-variant --no-live-pmc > $O/bench_full_$5.json 2> $O/bench_full_$5.err; }
  run CornellBoxSpecular 1024 1024 512 CornellBoxSpecular
  run atrium 1920 1080 256 atrium
  run clutter 3840 2160 64 clutter
  run MetalRings 1920 1080 128 MetalRings
  run LivingRoomLit 1920 1080 128 LivingRoomLit
  run atrium:2000000 1920 1080 64 atrium2M
  for f in $O/bench_full_*.json; do cp $f profiles/$R/; done
fi
cp -r profiles/$R $O/profiles_copy 2>/dev/null
ls -la $O | head -50
