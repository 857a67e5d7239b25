#!/bin/bash
# Regenerates the measured artefacts of profiles/$R on a GPU box (run through gpurun; outputs under gpurun_out/$R, copied to profiles/$R).
#   R=r03 tools/refresh_profiles.sh core   # rocprof kernel stats, PMC passes + traffic.json (C2 and the HBM-resident workload), bench line, cadence, time-to-RMSE
#   R=r03 tools/refresh_profiles.sh full   # full-size configurations C3'-C5', real mid-size models, the HBM-bound 2 M triangle case
set -u
R=${R:-r03}
O=gpurun_out/$R; mkdir -p $O profiles/$R
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "${1:-core}" = core ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-time-to-rmse > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
  cp $O/prof/*/*kernel_stats.csv $O/bench_kernel_stats.csv
  tools/pmc.sh $R/pmc_c2 > $O/pmc_c2.log 2>&1
  cp $O/pmc_c2/summary.txt $O/pmc_summary_c2.txt; cp $O/pmc_summary_c2.txt profiles/$R/pmc_summary_c2.txt
  python tools/pmc_to_traffic.py profiles/$R/pmc_summary_c2.txt CornellBoxDiffuse_512x512x1024_mp8 > $O/traffic_c2.json
  tools/pmc2.sh $R/pmc_atrium --scene atrium --width 1920 --height 1080 --spp 256 --max-path 0 > $O/pmc_atrium.log 2>&1
  cp $O/pmc_atrium/summary.txt $O/pmc_summary_atrium.txt; cp $O/pmc_summary_atrium.txt profiles/$R/pmc_summary_atrium.txt
  python tools/pmc_to_traffic.py profiles/$R/pmc_summary_atrium.txt atrium_1920x1080x256_mp999 > $O/traffic_atrium.json
  cp profiles/traffic.json $O/traffic.json
  python bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err
  python tests/tools/cpu_baseline_c1.py > $O/cpu_restatement_c1_c2.json 2> $O/cpu_restatement.err
  python tools/time_to_rmse.py > $O/time_to_rmse_c2.json 2> $O/time_to_rmse.err
  cc -O2 -std=c11 -I include examples/cadence.c -o /tmp/cadence master_amd/libmi_pt.so -Wl,-rpath,$PWD/master_amd
  : > $O/cadence.jsonl
  /tmp/cadence scenes/CornellBoxDiffuse.miscene 512 512 480 8 8 >> $O/cadence.jsonl 2>&1
  /tmp/cadence scenes/CornellBoxDiffuse.miscene 1920 1080 96 8 4 >> $O/cadence.jsonl 2>&1
  /tmp/cadence scenes/LivingRoomLit.miscene 1920 1080 32 0 4 >> $O/cadence.jsonl 2>&1
else
  run() { python bench.py --scene "$1" --width $2 --height $3 --spp $4 --max-path 0 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload --no-time-to-rmse > $O/bench_full_$5.json 2> $O/bench_full_$5.err; }
  run CornellBoxSpecular 1024 1024 512 CornellBoxSpecular
  run atrium 1920 1080 256 atrium
  run clutter 3840 2160 64 clutter
  run MetalRings 1920 1080 128 MetalRings
  run LivingRoomLit 1920 1080 128 LivingRoomLit
  run atrium:2000000 1920 1080 64 atrium2M
  tools/pmc2.sh $R/pmc_atrium2M --scene atrium:2000000 --width 1920 --height 1080 --spp 64 --max-path 0 > $O/pmc_atrium2M.log 2>&1
  cp $O/pmc_atrium2M/summary.txt $O/pmc_summary_atrium2M.txt; cp $O/pmc_summary_atrium2M.txt profiles/$R/pmc_summary_atrium2M.txt
  python tools/pmc_to_traffic.py profiles/$R/pmc_summary_atrium2M.txt atrium:2000000_1920x1080x64_mp999 > $O/traffic_atrium2M.json
  cp profiles/traffic.json $O/traffic.json
fi
cp -r profiles/$R $O/profiles_copy 2>/dev/null
ls -la $O | head -50
