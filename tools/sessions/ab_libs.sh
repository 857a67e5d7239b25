#!/bin/bash
# C2 throughput of several builds of the library on one GPU box:  tools/ab_libs.sh <tag> lib1.so lib2.so ...   (libs under master_amd/)
# env EXTRA: more bench.py arguments; env SCENE_ARGS overrides the workload
set -o pipefail
TAG=$1; shift
mkdir -p gpurun_out
for lib in "$@"; do
  MI_PT_LIB=$PWD/master_amd/$lib timeout -k 10 200 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-hbm-workload --no-time-to-rmse $EXTRA > gpurun_out/${TAG}_$lib.json 2> gpurun_out/${TAG}_$lib.err \
    || { echo "$lib FAILED"; tail -3 gpurun_out/${TAG}_$lib.err; continue; }
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_$lib.json')); print('%-28s %7.0f Msamples/s %7.2f ms' % ('$lib', d['value'], d['ms_per_step']))"
done | tee gpurun_out/${TAG}_summary.txt
