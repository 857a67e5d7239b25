#!/bin/bash
# why does MI_BPT_PERSIST_BLOCKS != default cost 20x on LivingRoomLit?  per-kernel stats of one run
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bpt_blocks; mkdir -p $O
export MI_BPT_PERSIST_BLOCKS=$1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 tools/bpt_prof.py LivingRoomLit > $O/p.out 2> $O/p.err
cat $O/p.out
python3 - $O/p <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:80].ljust(80), r['Calls'], round(float(r['TotalDurationNs'])/1e6,1), round(float(r['AverageNs'])/1e6,3), r['Percentage'])
PY
