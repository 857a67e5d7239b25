"""Device time per sample-per-pixel as a function of the samples of one launch (C2 scene): where does a short launch lose?"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import master_amd as ma

scene = ma.Scene.load(os.path.join(ROOT, "scenes", sys.argv[1] if len(sys.argv) > 1 else "CornellBoxDiffuse") + ".miscene")
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 512)
pt = ma.PathTracing(scene, max_path=8)
fb = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
for chunk in (None, 1, 2, 4, 8, 16, 64):
    if chunk is None:
        os.environ.pop("MI_PT_CHUNK_SPP", None)
    else:
        os.environ["MI_PT_CHUNK_SPP"] = str(chunk)
    os.environ["MI_PT_FRAME_MODE"] = "0"
    row = []
    for spp in (1, 2, 4, 8, 16, 32, 64, 256, 1024):
        if chunk is not None and spp < chunk:
            row.append("      -"); continue
        pt.render_device(fb.data_ptr(), W, H, spp=spp, seed=1, sample_offset=0)
        ts = []
        for i in range(3):
            st = pt.render_device(fb.data_ptr(), W, H, spp=spp, seed=1, sample_offset=0)
            ts.append(st.trace_ms)
        li = pt.last_launch()
        row.append("%7.4f" % (min(ts) / spp))
    print("chunk_spp %-5s ms per spp at spp = 1, 2, 4, 8, 16, 32, 64, 256, 1024: %s" % (chunk if chunk else "auto", " ".join(row)), flush=True)
