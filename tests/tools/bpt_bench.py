#!/usr/bin/env python3
"""BPT throughput on the GPU next to the CPU restatement (informational; bench.py stays on the PT north-star metric).
Samples = closest-hit rays (eye + light sub-path segments), like num_basic_rays of the reference's statistics."""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import master_amd as ma
import oracle
from bench import effective_cpus

out = []
for name, W, H, spp in [("CornellBoxDiffuse", 512, 512, 64), ("CornellBoxSpecular", 512, 512, 32), ("TestCase29", 512, 512, 32)]:
    s = ma.Scene.load(os.path.join(ROOT, "scenes", name + ".miscene"))
    pt = ma.PathTracing(s, beta=2.0)
    pt.bpt_render_rgbn(W, H, spp=2, seed=1)
    dt, st = None, None
    for _ in range(2):  # best of two: the frame loop has host read-backs, the first run also pays allocations
        t = time.perf_counter(); img = pt.bpt_render_rgbn(W, H, spp=spp, seed=1); d = time.perf_counter() - t
        if dt is None or d < dt: dt, st = d, pt.last_stats
    thr = effective_cpus()
    orc = oracle.Oracle(s, beta=2.0)
    t = time.perf_counter(); orc.bpt_render_rgbn(W, H, spp=2, seed=1, threads=thr); ct = time.perf_counter() - t
    cs = orc.last_stats
    rec = {"scene": name, "resolution": [W, H], "spp": spp, "gpu_s": dt, "gpu_kernel_ms": st.trace_ms,
           "gpu_Msamples_s": st.num_basic_rays / (st.trace_ms * 1e-3) / 1e6, "gpu_Mrays_s": (st.num_basic_rays + st.num_shadow_rays) / (st.trace_ms * 1e-3) / 1e6,
           "gpu_Mpaths_s": st.num_paths / (st.trace_ms * 1e-3) / 1e6,
           "cpu_threads": thr, "cpu_Msamples_s": cs.num_basic_rays / ct / 1e6, "cpu_Mrays_s": (cs.num_basic_rays + cs.num_shadow_rays) / ct / 1e6,
           "mean": float((img[..., :3] / np.maximum(img[..., 3:], 1)).mean())}
    rec["gpu_over_cpu"] = rec["gpu_Mrays_s"] / rec["cpu_Mrays_s"]
    out.append(rec)
    print(json.dumps(rec), flush=True)
