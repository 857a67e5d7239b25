#!/bin/bash
set -u
O=gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/gpu_tests.log
[ $rc = 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 300 python bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02/bench_n1.json"))
h = d["hbm_workload"]
print("C2 %.1f Msamples/s, %.2f ms/step frac %.5f valu %.3f; hbm_workload %.1f Msamples/s %.1f ms/step dyn=%s frac %.3f eff %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["valu"]["frac"], h["value"], h["ms_per_step"], h["dynamic_fetch_traversal"], h["roofline"]["frac"], h["roofline"]["terms"]["simd_efficiency_unified_traversal"]))
PY
