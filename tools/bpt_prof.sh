#!/bin/bash
# rocprofv3 per-kernel breakdown of one BPT render (tools/bpt_prof.py) per scene: tools/bpt_prof.sh CornellBoxDiffuse LivingRoomLit
O=$GRAFT_REPO_ROOT/gpurun_out/bpt_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for sc in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp_$sc -- python3 $GRAFT_REPO_ROOT/tools/bpt_prof.py $sc > $O/$sc.out 2> $O/$sc.err
  f=$(find /tmp/bp_$sc -name "*kernel_stats.csv" | head -1)
  echo "== $sc $(cat $O/$sc.out)"
  if [ -n "$f" ]; then cp $f $O/${sc}_kernel_stats.csv; head -7 $f | cut -d, -f1-7; else echo "no stats file"; tail -3 $O/$sc.err; fi
done
