#!/bin/bash
# r03: where does the flat kernel's time go?  phase stamps (diagnostic build) + PMC passes
set -o pipefail
mkdir -p gpurun_out
python -c "
from master_amd import build as mb
mb.build(extra_flags=['-DMI_PHASE_TIMING'], out='libmi_pt_phase.so')
" > gpurun_out/r03_flat3_build.log 2>&1 || { tail -5 gpurun_out/r03_flat3_build.log; exit 1; }
for f in 0 1; do echo "== MI_PT_FLAT=$f"; MI_PT_FLAT=$f timeout -k 10 300 python tools/gpu_phase.py; done 2>&1 | tee gpurun_out/r03_flat3_phase.txt
MI_PT_FLAT=1 bash tools/pmc.sh r03_pmc_flat > gpurun_out/r03_flat3_pmc.txt 2>&1
head -24 gpurun_out/r03_pmc_flat/summary.txt
