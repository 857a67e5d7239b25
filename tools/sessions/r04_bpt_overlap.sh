#!/bin/bash
# do the launches in flight overlap?  kernel trace of one BPT render, per queue: busy time, union, per-kernel mean durations
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bpt_overlap; mkdir -p $O
for fl in 1 2 4; do
  export MI_BPT_FLIGHTS=$fl
  rocprofv3 --kernel-trace --output-format csv -d $O/f$fl -- python3 tools/bpt_prof.py LivingRoomLit > $O/f$fl.out 2> $O/f$fl.err
  echo "FLIGHTS=$fl $(cat $O/f$fl.out)"
  python3 - $O/f$fl <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'bpt_' in r['Kernel_Name']]
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('<')[0].split('::')[-1],r.get('Queue_Id','?')) for r in rows]
ev.sort()
# only the timed render: the last 80 % of kernels by time
t0=ev[0][0]; t1=max(e[1] for e in ev)
tot=sum(e[1]-e[0] for e in ev)
# union
u=0; cs,ce=ev[0][0],ev[0][1]
for s,e,_,_ in ev[1:]:
    if s>ce: u+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
u+=ce-cs
print('  span %.1f ms, union busy %.1f ms, sum of kernels %.1f ms, queues %s' % ((t1-t0)/1e6,u/1e6,tot/1e6,sorted(set(e[3] for e in ev))))
d=collections.defaultdict(list)
for s,e,n,q in ev: d[n].append((e-s)/1e6)
for n,v in sorted(d.items(), key=lambda kv:-sum(kv[1])): print('   %-22s calls %4d mean %.3f ms total %.1f' % (n,len(v),sum(v)/len(v),sum(v)))
PY
done
