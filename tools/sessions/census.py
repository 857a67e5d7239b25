"""Loop-body census of the two-loop traversal (instrumented variant): wave-level node / leaf bodies per closest-hit ray."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import master_amd as ma
os.environ["MI_PT_DYN"] = "0"
for name, mp in (("CornellBoxDiffuse", 8), ("TestCaseFurnace", ma.PTRDIFF_MAX), ("CornellBoxSpecular", ma.PTRDIFF_MAX)):
    scene = ma.Scene.load(os.path.join(ROOT, "scenes", name + ".miscene"))
    pt = ma.PathTracing(scene, max_path=mp)
    fb = torch.zeros((512, 512, 4), dtype=torch.float32, device="cuda")
    pt.set_instrumented(True)
    st = pt.render_device(fb.data_ptr(), 512, 512, spp=64, seed=1)
    seg = st.num_basic_rays
    b = list(st.wave_loop_bodies)
    waves_rays = seg / 64.0
    print("%-20s per 64 closest-hit rays: closest loop %.1f node + %.1f leaf bodies (lane visits %.1f + %.1f per ray, slowest-lane steps %.1f); shadow loop %.1f + %.1f (per ray %.1f + %.1f, slowest %.1f); shadow rays / ray %.2f" % (
        name, b[0] / waves_rays, b[1] / waves_rays, st.nodes_closest / seg, st.tris_closest / seg, st.wave_steps_closest / waves_rays,
        b[2] / waves_rays, b[3] / waves_rays, st.nodes_shadow / seg, st.tris_shadow / seg, st.wave_steps_shadow / waves_rays, st.num_shadow_rays / seg))
