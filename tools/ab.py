#!/usr/bin/env python3
"""A/B timing of library variants in one GPU session: tools/ab.py libA.so libB.so ... [-- bench args]"""
import json, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
for rep in range(2):
    for lib in args:
        env = dict(os.environ, MI_PT_LIB=os.path.join(ROOT, "master_amd", lib))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print("%-28s %9.1f Msamples/s  %8.2f ms/step  kernel %.2f ms" % (lib, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]), flush=True)
        except Exception as e:
            print(lib, "FAILED", e, r.stderr[-400:])
