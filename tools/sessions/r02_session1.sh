#!/bin/bash
# Round 2, first GPU contact: full -m gpu suite, bench line, FETCH_SIZE calibration on gathers, frame cadence.
set -u
O=gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.log
tail -3 $O/gpu_tests.log
timeout -k 10 300 python bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
tail -c 600 $O/bench_n1.json
cc -O2 -std=c11 -I include examples/cadence.c -o /tmp/cadence master_amd/libmi_pt.so -Wl,-rpath,$PWD/master_amd
timeout -k 10 120 /tmp/cadence scenes/CornellBoxDiffuse.miscene 512 512 400 8 > $O/cadence_512.json 2> $O/cadence_512.err; cat $O/cadence_512.json
timeout -k 10 120 /tmp/cadence scenes/CornellBoxDiffuse.miscene 1920 1080 100 8 > $O/cadence_1080.json 2> $O/cadence_1080.err; cat $O/cadence_1080.json
# FETCH_SIZE on gathers of known size (program directly after --)
timeout -k 10 120 tools/micro/gather_fetch > $O/gather_fetch_plain.txt 2>&1; cat $O/gather_fetch_plain.txt
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/gather_fetch/fetch -- tools/micro/gather_fetch > $O/gather_fetch_pmc1.txt 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum --output-format csv -d $O/gather_fetch/tcc -- tools/micro/gather_fetch > $O/gather_fetch_pmc2.txt 2>&1
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/r02/gather_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]] = agg[r["Kernel_Name"].split("(")[0][-40:]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
with open("gpurun_out/r02/gather_fetch_summary.txt", "w") as o:
    for k, c in sorted(agg.items()):
        o.write(k + "\n")
        for n, v in sorted(c.items()): o.write("   %-26s %.6g\n" % (n, v))
print(open("gpurun_out/r02/gather_fetch_summary.txt").read())
PY
