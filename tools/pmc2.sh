#!/bin/bash
# Memory-side PMC passes for an HBM-resident scene.  usage: tools/pmc2.sh <outdir> [bench args...]
set -u
OUT=gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM" \
  "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pass$i" -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-workload --no-time-to-rmse "$@" > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || { echo "pass $i failed"; tail -2 "$OUT/pass$i.err"; }
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-80:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open(out + "/summary.txt", "w") as o:
    for k, c in agg.items():
        if "megakernel" not in k: continue
        o.write(k + "\n")
        for name, v in sorted(c.items()): o.write("   %-30s %.6g\n" % (name, v))
print(open(out + "/summary.txt").read())
PY
