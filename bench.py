#!/usr/bin/env python3
"""bench.py — Msamples/s of the PT hot path on MI355X (BASELINE.json metric).

A "step" is one full pass of the hot path over the workload: CornellBoxDiffuse, PT, 512x512,
1024 spp, max path 8 (BASELINE.json configs[1]) = one mi_pt_render_device call per GPU, followed
(N > 1) by the RCCL sum-reduce of the [H][W][4] framebuffer.  One *sample* = one path segment
(closest-hit ray; the reference's num_basic_rays).  Weak scaling: every GPU renders
`spp` samples per pixel of its own global sample range, the merged image has N * spp.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def algorithmic_bytes_per_sample(st, wavefront=False):
    """megakernel form (no path-state term) or wavefront form (+192 B of path state per segment):
    B = 152*h + 64*(N + N'*s) + 48*(T + T'*s) + 32*s + 16/Lbar   per path segment, where the
    visit counts come from the instrumented kernel variant on the same workload."""
    seg = float(st.num_basic_rays)
    h = st.num_hits / seg
    s = st.num_shadow_rays / seg
    lbar = seg / float(st.num_paths)
    n_c, t_c = st.nodes_closest / seg, st.tris_closest / seg
    n_s, t_s = st.nodes_shadow / seg, st.tris_shadow / seg  # already per segment (= N' * s, T' * s)
    b = 152.0 * h + 64.0 * (n_c + n_s) + 48.0 * (t_c + t_s) + 32.0 * s + 16.0 / lbar + (192.0 if wavefront else 0.0)
    eff_c = (st.nodes_closest + st.tris_closest) / (64.0 * st.wave_steps_closest) if st.wave_steps_closest else None
    eff_s = (st.nodes_shadow + st.tris_shadow) / (64.0 * st.wave_steps_shadow) if st.wave_steps_shadow else None
    return b, dict(h=h, s=s, Lbar=lbar, N=n_c, T=t_c, N_shadow_per_segment=n_s, T_shadow_per_segment=t_s,
                   simd_efficiency_closest_traversal=eff_c, simd_efficiency_shadow_traversal=eff_s)


def effective_cpus():
    """Host cores this process may really use: affinity mask, capped by a cgroup CPU quota if one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 256))


def cpu_baseline(scene, args, budget_s=12.0):
    """CPU restatement (oracle, kind "port") timed on this host's cores on a bounded sample of the
    same workload: same scene / resolution / max path, fewer samples per pixel."""
    import oracle

    threads = effective_cpus()
    orc = oracle.Oracle(scene, lights=1.0, roulette=0.9, beta=1.0, max_path=args.max_path)
    orc.render_rgbn(args.width, args.height, spp=1, seed=1, threads=threads)  # warm-up, page-in
    spp_done, segs, t0 = 0, 0, time.perf_counter()
    while True:
        orc.render_rgbn(args.width, args.height, spp=2, seed=1, sample_offset=spp_done, threads=threads)
        spp_done += 2
        segs += orc.last_stats.num_basic_rays
        dt = time.perf_counter() - t0
        if dt >= budget_s or spp_done >= args.spp:
            break
    return {"value": segs / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": "%s %dx%d max_path %s, %d spp of %d, %.1f s, CPU restatement of reference PT (own BVH, non-Embree), pthreads over 32x32 tiles" % (
                args.scene, args.width, args.height, args.max_path, spp_done, args.spp, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="CornellBoxDiffuse")
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--spp", type=int, default=1024, help="samples per pixel per GPU per step")
    ap.add_argument("--max-path", type=int, default=8, help="0 = unlimited (roulette-terminated), the reference default")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 megakernel with the scene in LDS, 2 megakernel with the scene in HBM, 3 wavefront pipeline")
    ap.add_argument("--shard", choices=("samples", "tiles"), default="samples",
                    help="multi-GPU decomposition: sample ranges (default) or interleaved 32x32 pixel tiles (BASELINE C5); work per GPU is the same")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' + --share-gpu rehearses N ranks on one GPU")
    ap.add_argument("--share-gpu", action="store_true", help="map every rank onto the visible GPUs modulo their count (rehearsal only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import master_amd as ma
    from master_amd import dist as madist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # one rank per GPU: LOCAL_RANK indexes the visible devices.  If the launcher restricts visibility per rank (one visible device
    # each), every rank uses its device 0; --share-gpu does the same mapping on purpose for rehearsals on fewer GPUs than ranks.
    if args.share_gpu or local_rank >= torch.cuda.device_count():
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":  # RCCL over xGMI: one rank per GPU
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    if args.scene.split(":")[0] in ("atrium", "clutter"):  # seeded procedural stand-ins for the missing BASELINE scenes
        from master_amd import scenegen

        scene = scenegen.load(args.scene)
    else:
        scene = ma.Scene.load(os.path.join(ROOT, "scenes", args.scene + ".miscene"))
    if args.max_path <= 0:
        args.max_path = ma.PTRDIFF_MAX
    pt = ma.PathTracing(scene, lights=1.0, roulette=0.9, beta=1.0, max_path=args.max_path, device=local_rank)
    if args.kernel:
        pt.set_kernel(args.kernel)
    W, H = args.width, args.height
    fb = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    seed = 0x5EED
    if world > 1:  # communicator and channel set-up for this message size is not a step: do it before anything is timed, whatever --warmup says
        madist.merge_framebuffers(fb)
        torch.cuda.synchronize()

    tiles = args.shard == "tiles" and world > 1
    if tiles:
        pt.set_tile_shard(rank, world)

    def step(i):
        if tiles:  # this rank's 32x32 tiles, all world x spp samples of the step
            off, n = madist.tile_sample_range(i, world, args.spp)
        else:      # every pixel, this rank's spp samples of the step
            off, n = madist.sample_offset(i, rank, world, args.spp), args.spp
        st = pt.render_device(fb.data_ptr(), W, H, spp=n, seed=seed, sample_offset=off, stream=stream, want_stats=True)
        madist.merge_framebuffers(fb)  # RCCL all-reduce (sum) of (R, G, B, denom): merge_exr semantics
        return st

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    segs = shadow = paths = 0
    kernel_ms = []
    for i in range(args.steps):
        st = step(args.warmup + i)
        segs += st.num_basic_rays
        shadow += st.num_shadow_rays
        paths += st.num_paths
        kernel_ms.append(st.trace_ms)
    fence()
    elapsed = time.perf_counter() - t0

    tot = torch.tensor([elapsed, float(segs), float(shadow), float(paths)], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = tot[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        sums = tot[1:].clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        elapsed = float(tmax.item())
        segs, shadow, paths = [float(x) for x in sums.tolist()]
    denom_ok = bool((fb[..., 3] == float(args.spp * world)).all().item())

    out = None
    if rank == 0:
        # roofline of the dominant kernel (pt_megakernel): algorithmic bytes of ONE launch / its HIP-event duration
        pt.set_instrumented(True)
        ist = pt.render_device(fb.data_ptr(), W, H, spp=min(args.spp, 64), seed=seed, sample_offset=0, stream=stream, want_stats=True)
        pt.set_instrumented(False)
        b_sample, terms = algorithmic_bytes_per_sample(ist, wavefront=pt.get_kernel() == ma.KERNEL_WAVEFRONT)
        seg_per_launch = float(st.num_basic_rays)
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = b_sample * seg_per_launch / (avg_ms * 1e-3) / 1e9
        traffic, pmc = None, None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):
            t = json.load(open(tj))
            key = "%s_%dx%dx%d_mp%d" % (args.scene, W, H, args.spp, min(args.max_path, 999))
            traffic = t.get(key, {}).get("hbm_bytes_per_launch")
            pmc = t.get(key)
        out = {
            "metric": "Msamples/sec (paths x bounces: closest-hit path segments per second)",
            "value": segs / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, PT, %dx%d, %d spp per GPU per step, max path %s, beta 1, roulette 0.9%s" % (
                           args.scene + (".blend" if ":" not in args.scene and args.scene not in ("atrium", "clutter") else " (procedural stand-in)"), W, H, args.spp,
                           "unlimited" if args.max_path >= ma.PTRDIFF_MAX else args.max_path,
                           " (BASELINE configs[1])" if (args.scene, W, H, args.spp, args.max_path) == ("CornellBoxDiffuse", 512, 512, 1024, 8) else ""),
                       "kernel": {1: "pt_megakernel<LDS scene>", 2: "pt_megakernel<HBM scene>", 3: "wavefront pipeline (wf_extend / wf_shade / wf_shadow / wf_regen)"}[pt.get_kernel()],
                       "parallelism": "%s sharded over %d GPU(s), RCCL all-reduce of [H][W][4] f32" % ("32x32 pixel tiles" if tiles else "samples", world),
                       "Mpaths_per_s": paths / elapsed / 1e6, "Mrays_per_s": (segs + shadow) / elapsed / 1e6,
                       "denom_equals_spp": denom_ok},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "wavefront pipeline (all kernels of one step)" if pt.get_kernel() == ma.KERNEL_WAVEFRONT else "pt_megakernel", "avg_launch_ms": avg_ms,
                         "algorithmic_bytes_per_sample": b_sample, "terms": terms, "pmc": pmc,
                         "note": "scene is LDS-resident: the kernel is VALU/latency-bound, not HBM-bound; algorithmic bytes are SURVEY 8(d)'s per-segment figure"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, args, args.cpu_budget)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
