#!/usr/bin/env python3
"""the CPU restatement (oracle) on this host's cores.
C1 = CornellBoxDiffuse 256x256, 64 spp, max path 4, median of 3;  C2-cpu = 512x512, 16 spp, max path 8."""
import json, os, statistics, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import master_amd as ma, oracle
sys.path.insert(0, ROOT)
from bench import effective_cpus
s = ma.Scene.load(os.path.join(ROOT, "scenes", "CornellBoxDiffuse.miscene"))
thr = effective_cpus()
out = {"cores": thr, "cpu_model": [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][:1]}
for name, (w, h, spp, mp, reps) in {"C1": (256, 256, 64, 4, 3), "C2-cpu": (512, 512, 16, 8, 3)}.items():
    o = oracle.Oracle(s, max_path=mp); o.render_rgbn(w, h, spp=1, threads=thr)
    runs = []
    for r in range(reps):
        t = time.perf_counter(); o.render_rgbn(w, h, spp=spp, seed=r, threads=thr); dt = time.perf_counter() - t
        st = o.last_stats
        runs.append({"wall_s": dt, "Mpaths_s": st.num_paths / dt / 1e6, "Msamples_s": st.num_basic_rays / dt / 1e6, "Mrays_s": (st.num_basic_rays + st.num_shadow_rays) / dt / 1e6})
    med = sorted(runs, key=lambda x: x["wall_s"])[len(runs) // 2]
    out[name] = {"config": "%dx%d, %d spp, max path %d" % (w, h, spp, mp), "median": med}
print(json.dumps(out, indent=1))
