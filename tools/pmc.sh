#!/bin/bash
# PMC passes over one bench workload (counters in their own runs, no tracing domains mixed in).
# usage: tools/pmc.sh <outdir under gpurun_out> [bench args...]
set -u
OUT=gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
i=0
for set in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
  "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE GRBM_COUNT" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pass$i" -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-workload --no-time-to-rmse "$@" > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || { echo "pass $i failed"; tail -3 "$OUT/pass$i.err"; }
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-80:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen.add((k, r["Dispatch_Id"]))
    for k, _ in seen: calls[(k, f)] += 1
with open(out + "/summary.txt", "w") as o:
    for k, c in agg.items():
        if "megakernel" not in k and "finalize" not in k: continue
        o.write(k + "\n")
        for name, v in sorted(c.items()): o.write("   %-28s %.6g\n" % (name, v))
print(open(out + "/summary.txt").read())
PY
