#!/bin/bash
# Regenerates the measured artifacts of profiles/r01 on a GPU box (run through gpurun; outputs under gpurun_out/r01).
#   tools/refresh_profiles.sh core   # rocprof kernel stats, PMC passes + traffic.json, bench line, CPU restatement, time-to-RMSE
#   tools/refresh_profiles.sh full   # full-size configurations C3'-C5', real mid-size models, the HBM-bound 2 M triangle case
set -u
O=gpurun_out/r01; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "${1:-core}" = core ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
  cp $O/prof/*/*kernel_stats.csv $O/bench_kernel_stats.csv
  tools/pmc.sh r01/pmc_c2 > $O/pmc_c2.log 2>&1
  cp gpurun_out/r01/pmc_c2/summary.txt $O/pmc_summary_c2.txt
  mkdir -p profiles/r01 && cp $O/pmc_summary_c2.txt profiles/r01/pmc_summary_c2.txt
  python tools/pmc_to_traffic.py profiles/r01/pmc_summary_c2.txt CornellBoxDiffuse_512x512x1024_mp8 > $O/traffic_c2.json
  cp profiles/traffic.json $O/traffic.json
  python bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err
  python tests/tools/cpu_baseline_c1.py > $O/cpu_restatement_c1_c2.json 2> $O/cpu_restatement.err
  python tools/time_to_rmse.py > $O/time_to_rmse_c2.json 2> $O/time_to_rmse.err
else
  run() { python bench.py --scene "$1" --width $2 --height $3 --spp $4 --max-path 0 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_full_$5.json 2> $O/bench_full_$5.err; }
  run CornellBoxSpecular 1024 1024 512 CornellBoxSpecular
  run atrium 1920 1080 256 atrium
  run clutter 3840 2160 64 clutter
  run MetalRings 1920 1080 128 MetalRings
  run LivingRoomLit 1920 1080 128 LivingRoomLit
  run atrium:2000000 1920 1080 64 atrium2M
  tools/pmc2.sh r01/pmc_atrium2M --scene atrium:2000000 --width 1920 --height 1080 --spp 64 --max-path 0 > $O/pmc_atrium2M.log 2>&1
  cp gpurun_out/r01/pmc_atrium2M/summary.txt $O/pmc_summary_atrium2M.txt
fi
ls -la $O | head -40
