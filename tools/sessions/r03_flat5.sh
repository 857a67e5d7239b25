#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
bash tools/ab_libs.sh r03_flat5 libmi_pt.so libmi_pt_w7.so libmi_pt_w8.so
python -c "
from master_amd import build as mb
mb.build(extra_flags=['-DMI_PHASE_TIMING'], out='libmi_pt_phase.so')
" > gpurun_out/r03_flat5_build.log 2>&1 || { tail -5 gpurun_out/r03_flat5_build.log; exit 1; }
timeout -k 10 300 python tools/gpu_phase.py 2>&1 | tee gpurun_out/r03_flat5_phase.txt
