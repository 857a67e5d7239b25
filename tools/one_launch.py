#!/usr/bin/env python3
"""One launch of the hot path per workload, nothing else: the program bench.py runs under `rocprofv3 --pmc ...` in a fresh child
process to read the hardware counters of exactly the launches it times.

    rocprofv3 --pmc SQ_ACTIVE_INST_VALU ... --output-format csv -d <dir> -- /usr/bin/python3 tools/one_launch.py <out.json> <workload> [<workload> ...]

A workload is  scene:WxHxSPP:max_path[:rank/world]  (max_path 0 = unbounded; rank/world = pixel-tile shard, BASELINE configs[4]).
No torch (start-up stays ~1 s), no warm-up launch, no instrumented variant: the megakernel dispatches of the process are, in order,
the workloads of the command line; <out.json> lists them with the statistics of each launch so the parent can match the counter rows.
"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def parse_workload(spec):
    parts = spec.split(":")
    # the scene name itself may carry a ':' (atrium:2000000): the size field is the first one of the form WxHxS
    k = next(i for i, p in enumerate(parts) if p.count("x") == 2 and p.replace("x", "").isdigit())
    scene = ":".join(parts[:k])
    w, h, spp = [int(v) for v in parts[k].split("x")]
    max_path = int(parts[k + 1]) if len(parts) > k + 1 else 0
    shard = tuple(int(v) for v in parts[k + 2].split("/")) if len(parts) > k + 2 else None
    return scene, w, h, spp, max_path, shard


def main():
    import master_amd as ma
    from bench import load_scene

    out_path, specs = sys.argv[1], sys.argv[2:]
    launches = []
    for spec in specs:
        scene_name, w, h, spp, max_path, shard = parse_workload(spec)
        scene = load_scene(scene_name)
        pt = ma.PathTracing(scene, lights=1.0, roulette=0.9, beta=1.0, max_path=max_path if max_path > 0 else ma.PTRDIFF_MAX, device=0)
        if shard:
            pt.set_tile_shard(*shard)
        # the framebuffer stays on the device side of mi_pt_render's own staging; one call = one megakernel dispatch + pt_finalize
        pt.render_rgbn(w, h, spp=spp, seed=0x5EED, sample_offset=0)
        st, li = pt.last_stats, pt.last_launch()
        launches.append({"workload": spec, "segments": int(st.num_basic_rays), "paths": int(st.num_paths), "shadow_rays": int(st.num_shadow_rays),
                         "trace_ms_under_profiler": float(st.trace_ms), "workgroups": int(li.n_blocks)})
        pt.close()
        del pt, scene
    json.dump({"launches": launches}, open(out_path, "w"))


if __name__ == "__main__":
    main()
