#!/bin/bash
# counters of the unified-fetch A/B on the 2 M-triangle atrium (one launch each)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_pmc_uni; mkdir -p $O
for lib in libmi_pt_nouni.so libmi_pt.so; do for set in "FETCH_SIZE" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  MI_PT_LIB=$PWD/master_amd/$lib rocprofv3 --pmc $set --output-format csv -d $O/${lib}_$tag -- /usr/bin/python3 tools/one_launch.py $O/${lib}_$tag.json atrium:2000000:1920x1080x32:0 > $O/${lib}_$tag.log 2>&1
  python3 - "$O/${lib}_$tag" "$lib" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pt_megakernel" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
print(sys.argv[2], {k: "%.4g" % v for k, v in agg.items()})
PY
done; done
