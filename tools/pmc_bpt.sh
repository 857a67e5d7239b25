#!/bin/bash
# PMC passes over one BPT render (tools/bpt_prof.py <scene>): per-kernel sums.  usage: tools/pmc_bpt.sh <outdir under gpurun_out> <scene>
set -u
OUT=gpurun_out/$1; SC=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
i=0
for set in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
  "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE GRBM_COUNT" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pass$i" -- python3 tools/bpt_prof.py $SC > "$OUT/pass$i.out" 2> "$OUT/pass$i.err" || { echo "pass $i failed"; tail -3 "$OUT/pass$i.err"; }
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open(out + "/summary.txt", "w") as o:
    for k, c in agg.items():
        if "bpt_" not in k or "scan" in k or "commit" in k: continue
        o.write(k + "\n")
        for name, v in sorted(c.items()): o.write("   %-28s %.6g\n" % (name, v))
        if c.get("SQ_ACTIVE_INST_VALU") and c.get("GRBM_GUI_ACTIVE"):
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            o.write("   -> VALU issue %.3f  lanes %.3f  wait_any %.3f  HBM GB %.2f read + %.2f written\n" % (
                c["SQ_ACTIVE_INST_VALU"] * 2.0 / (cyc * 1024.0), c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
                c.get("SQ_WAIT_ANY", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 1)), c.get("FETCH_SIZE", 0) * 2048 / 1e9, c.get("WRITE_SIZE", 0) * 1024 / 1e9))
print(open(out + "/summary.txt").read())
PY
