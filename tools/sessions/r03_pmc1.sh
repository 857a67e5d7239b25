#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L > gpurun_out/r03_counters_list.txt 2>&1 || true
grep -c "Name" gpurun_out/r03_counters_list.txt
bash tools/pmc.sh r03_pmc_flat2 > gpurun_out/r03_pmc1.txt 2>&1
head -24 gpurun_out/r03_pmc_flat2/summary.txt
