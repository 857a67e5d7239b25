"""Host logic of bench.py that runs without a GPU: workload specs of the live-counter passes, the soft failure of those passes, the roofline block's choice of
bound, and the shape of the CPU-baseline leg (oracle as the `port` baseline)."""
import importlib
import os
import sys
import types

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def bench():
    return importlib.import_module("bench")


def test_pmc_workload_specs_round_trip(bench):
    one_launch = importlib.import_module("one_launch")
    cases = [("CornellBoxDiffuse", 512, 512, 1024, 8, None), ("atrium", 1920, 1080, 256, (1 << 63) - 1, None), ("atrium:2000000", 1920, 1080, 64, (1 << 63) - 1, None),
             ("clutter", 3840, 2160, 4096, (1 << 63) - 1, (0, 8))]
    for scene, w, h, spp, mp, shard in cases:
        spec = bench.pmc_spec(scene, w, h, spp, mp, shard)
        got = one_launch.parse_workload(spec)
        assert got == (scene, w, h, spp, 0 if mp >= (1 << 62) else mp, shard), spec


def test_live_pmc_fails_soft_without_a_gpu(bench, tmp_path):
    """No GPU in the build container: the rocprofv3 child process (or rocprofv3 itself) fails, and bench.py must get ({}, reason) — the line then replays
    profiles/traffic.json with pmc_live false instead of dying."""
    out, why = bench.collect_live_pmc([bench.pmc_spec("CornellBoxDiffuse", 64, 64, 2, 4)], timeout_s=120.0, keep_dir=str(tmp_path / "pmc"))
    assert out == {} and isinstance(why, str) and why


def test_roofline_block_names_the_bound_that_binds(bench):
    """LDS-resident scene with counters: bound valu, frac = issue x lanes; scene read from HBM with counters: bound hbm, frac = measured fabric traffic / time /
    8 TB/s with the algorithmic figure kept beside it; without counters the block says so."""
    ma = types.SimpleNamespace(KERNEL_MEGA_LDS=1, KERNEL_MEGA_GLOBAL=2, KERNEL_WAVEFRONT=3)
    ist = types.SimpleNamespace(num_basic_rays=1000, num_hits=900, num_shadow_rays=600, num_paths=250, nodes_closest=16000, tris_closest=3000, nodes_shadow=6000,
                                tris_shadow=1000, wave_steps_closest=400, wave_steps_shadow=300)
    li = types.SimpleNamespace(flat_leaves=16, partial_bytes=2.0e8)
    pt = types.SimpleNamespace(get_kernel=lambda: 1)
    live = {"pmc_live": True, "kernel": "k", "hbm_bytes_per_launch": 2.4e8, "valu_issue_utilisation": 0.7, "valu_thread_utilisation": 0.5, "valu_instructions_per_segment_lane": 2200.0}
    rl = bench.roofline_block(ma, pt, ist, 1.0e9, 46.0, li, "none", live=live)
    assert rl["bound"] == "valu" and rl["unit"] == "TFLOP/s" and abs(rl["frac"] - 0.35) < 1e-12 and rl["pmc_live"] and rl["hbm"]["scene_bytes_served_by"] == "LDS"
    assert abs(rl["achieved"] - 0.35 * bench.FP32_PEAK_TFLOPS) < 1e-9 and rl["traffic"] == 2.4e8
    pt2 = types.SimpleNamespace(get_kernel=lambda: 2)
    li2 = types.SimpleNamespace(flat_leaves=0, partial_bytes=1.0e9)
    live2 = dict(live, hbm_bytes_per_launch=4.0e12)
    rl2 = bench.roofline_block(ma, pt2, ist, 3.0e9, 1000.0, li2, "none", live=live2)
    assert rl2["bound"] == "hbm" and abs(rl2["frac"] - 0.5) < 1e-12 and rl2["algorithmic_frac_cache_served"] > 0 and rl2["valu"]["frac"] == 0.35
    rl3 = bench.roofline_block(ma, pt2, ist, 3.0e9, 1000.0, li2, "no_such_workload_key", live=None)
    assert rl3["bound"] == "hbm" and rl3["pmc_live"] is False and rl3["frac"] <= 1.0 and "no PMC counters" in rl3["note"]
