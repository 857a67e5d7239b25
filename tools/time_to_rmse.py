#!/usr/bin/env python3
"""Secondary metric of BASELINE.json: time-to-target-RMSE.  Renders the scene in frames of `--frame-spp`
samples on the GPU, records (clock_time, rms_error) per frame against a high-spp image of the same
integrator exactly like the reference logs record_t{clock_time, rms_error} (Technique.cpp:61-76,
rms_abs_errors ImageView.cpp:60-85), and reports the wall time at which RMS first drops below the target."""
import argparse, json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import master_amd as ma

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="CornellBoxDiffuse"); ap.add_argument("--width", type=int, default=512); ap.add_argument("--height", type=int, default=512)
ap.add_argument("--max-path", type=int, default=8); ap.add_argument("--target", type=float, default=0.01)
ap.add_argument("--frame-spp", type=int, default=16); ap.add_argument("--ref-spp", type=int, default=65536); ap.add_argument("--max-spp", type=int, default=16384)
a = ap.parse_args()
s = ma.Scene.load(os.path.join(ROOT, "scenes", a.scene + ".miscene"))
pt = ma.PathTracing(s, max_path=a.max_path)
ref = np.zeros((a.height, a.width, 4), np.float64)
for k in range(0, a.ref_spp, 4096):  # reference: disjoint sample range, other seed
    ref += pt.render_rgbn(a.width, a.height, spp=min(4096, a.ref_spp - k), seed=999, sample_offset=k)
ref_rgb = (ref[..., :3] / ref[..., 3:]).astype(np.float32)
pt2 = ma.PathTracing(s, max_path=a.max_path)
view = np.zeros((a.height, a.width, 4), np.float64)
t0 = time.perf_counter(); hit = None; log = []
while pt2.statistics().num_samples < a.max_spp:
    rec = pt2.render(view, seed=1, reference=ref_rgb, spp=a.frame_spp)
    log.append((time.perf_counter() - t0, pt2.statistics().num_samples, rec["rms_error"]))
    if hit is None and rec["rms_error"] <= a.target:
        hit = log[-1]; break
print(json.dumps({"scene": a.scene, "resolution": [a.width, a.height], "max_path": a.max_path, "target_rms": a.target, "reference_spp": a.ref_spp,
                  "time_to_target_s": hit[0] if hit else None, "spp_at_target": hit[1] if hit else None, "rms_at_target": hit[2] if hit else None,
                  "note": "wall time includes the per-frame framebuffer download and the host-side RMS (ImageView.cpp:60-85 semantics)",
                  "trace": [{"t": round(t, 4), "spp": n, "rms": float(r)} for t, n, r in log[:: max(1, len(log) // 12)]]}))
