#!/bin/bash
# per-kernel times of the BPT stages with the regenerating tracing kernels (MI_BPT_PERSIST default) on two models
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bpt_persist; mkdir -p $O
for m in LivingRoomLit CornellBoxDiffuse; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$m -- python3 tools/bpt_prof.py $m > $O/$m.out 2> $O/$m.err
  python3 - $O/$m <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:80].ljust(80), r['Calls'], round(float(r['TotalDurationNs'])/1e6,1), round(float(r['AverageNs'])/1e6,3), r['Percentage'])
PY
done
