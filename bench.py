#!/usr/bin/env python3
"""bench.py — Msamples/s of the PT hot path on MI355X (BASELINE.json metric).

A "step" is one full pass of the hot path over the workload: CornellBoxDiffuse, PT, 512x512,
1024 spp, max path 8 (BASELINE.json configs[1]) = one mi_pt_render_device call per GPU, followed
(N > 1) by the RCCL sum-reduce of the [H][W][4] framebuffer.  One *sample* = one path segment
(closest-hit ray; the reference's num_basic_rays).  Weak scaling: every GPU renders
`spp` samples per pixel of its own global sample range, the merged image has N * spp.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Its `roofline` block never shows a fraction of a bound the kernel does not touch:

  * The Cornell box is LDS-resident (every workgroup stages the 8.5 KB scene once); the bytes SURVEY 8(d) counts per
    segment are LDS reads there.  `roofline.achieved` is then the HBM traffic that was MEASURED for this workload
    (rocprofv3 PMC passes, profiles/traffic.json) per launch / the launch time measured live here with HIP events, and
    `roofline.valu` carries the bound that matters: VALU issue x lane utilisation against the FP32 vector peak.
  * For a scene that is read from HBM/L2, `achieved` is SURVEY 8(d)'s algorithmic bytes per segment (visit counters of the
    instrumented kernel on the same workload) x the segments of one launch / the launch time.

At N = 1 the line also carries `time_to_rmse` (the second half of BASELINE.json's metric, outside the timed region; --no-time-to-rmse skips it),
`hbm_workload`: one HBM-resident configuration (BASELINE configs[3] through its stand-in, the 269 k-triangle atrium at 1920x1080, 256 spp,
unbounded paths) timed after the primary region with its own ms_per_step, algorithmic bytes and measured traffic, and `hbm_workload_beyond_cache`:
the 2 M-triangle atrium (480 MB of scene: beyond the 256 MB Infinity Cache), the launch that really reaches HBM.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
FP32_PEAK_TFLOPS = 157.3  # FP32 vector peak (same table): 256 CUs x 4 SIMD-32 x 2 flop x 2.4 GHz


def algorithmic_bytes_per_sample(st, wavefront=False, node_bytes=64.0):
    """megakernel form (no path-state term) or wavefront form (+192 B of path state per segment):
    B = 152*h + 64*(N + N'*s) + 48*(T + T'*s) + 32*s + 16/Lbar   per path segment, where the
    visit counts come from the instrumented kernel variant on the same workload."""
    seg = float(st.num_basic_rays)
    h = st.num_hits / seg
    s = st.num_shadow_rays / seg
    lbar = seg / float(st.num_paths)
    n_c, t_c = st.nodes_closest / seg, st.tris_closest / seg
    n_s, t_s = st.nodes_shadow / seg, st.tris_shadow / seg  # already per segment (= N' * s, T' * s)
    b = 152.0 * h + node_bytes * (n_c + n_s) + 48.0 * (t_c + t_s) + 32.0 * s + 16.0 / lbar + (192.0 if wavefront else 0.0)
    eff_c = (st.nodes_closest + st.tris_closest) / (64.0 * st.wave_steps_closest) if st.wave_steps_closest else None
    eff_s = (st.nodes_shadow + st.tris_shadow) / (64.0 * st.wave_steps_shadow) if st.wave_steps_shadow else None
    terms = dict(h=h, s=s, Lbar=lbar, N=n_c, T=t_c, N_shadow_per_segment=n_s, T_shadow_per_segment=t_s)
    if st.wave_steps_closest and not st.wave_steps_shadow and st.nodes_shadow:
        # dynamic-fetch traversal: ONE loop walks the closest-hit rays of a trip and the shadow rays of the trip before, idle lanes refill from the wave's pool
        terms["simd_efficiency_unified_traversal"] = (st.nodes_closest + st.tris_closest + st.nodes_shadow + st.tris_shadow) / (64.0 * st.wave_steps_closest)
        terms["simd_efficiency_closest_traversal"] = terms["simd_efficiency_shadow_traversal"] = terms["simd_efficiency_unified_traversal"]
    else:
        terms["simd_efficiency_closest_traversal"], terms["simd_efficiency_shadow_traversal"] = eff_c, eff_s
    return b, terms


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def native_oracle_build():
    """The CPU-baseline leg runs the restatement compiled for THIS host: -O2 -march=native, default FP contraction — a throughput
    build next to the bit-exact checker build (oracle/Makefile: -mfma -ffp-contract=off), made here because -march=native must match the box."""
    import subprocess
    import tempfile

    out = os.path.join(tempfile.mkdtemp(prefix="mi_oracle_native_"), "libpt_oracle_native.so")
    cmd = ["gcc", "-O2", "-march=native", "-std=gnu11", "-fPIC", "-fvisibility=hidden", "-pthread", "-shared", "-o", out,
           os.path.join(ROOT, "oracle", "pt_oracle.c"), "-lm", "-pthread"]
    subprocess.run(cmd, check=True, capture_output=True)
    return out, " ".join(cmd[1:4])


def embree_on_host():
    import subprocess

    try:
        return "embree" in subprocess.run(["ldconfig", "-p"], capture_output=True, text=True, timeout=10).stdout.lower()
    except Exception:  # noqa: BLE001
        return False


def cpu_baseline(scene, args, budget_s=12.0):
    """CPU restatement (oracle, kind "port") timed on this host's cores on a bounded sample of the
    same workload: same scene / resolution / max path, fewer samples per pixel."""
    import oracle

    threads = effective_cpus()
    flags = "oracle/Makefile build (-O2 -mfma -ffp-contract=off: the bit-exact checker)"
    try:
        lib, flags = native_oracle_build()
        oracle.ORACLE_LIB = lib
        oracle.build = lambda: lib
        oracle._lib = None
    except Exception as e:  # noqa: BLE001  (no compiler on the box: fall back to the checker build and say so)
        flags += "; native build failed: %r" % (e,)
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
    return {"value": segs / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(), "build": flags,
            "embree_on_host": embree_on_host(),
            "sample": "%s %dx%d max_path %s, %d spp of %d, %.1f s, CPU restatement of reference PT (own BVH, non-Embree), pthreads over 32x32 tiles" % (
                args.scene, args.width, args.height, args.max_path, spp_done, args.spp, dt)}


def load_scene(name):
    import master_amd as ma

    if name.split(":")[0] in ("atrium", "clutter"):  # seeded procedural stand-ins for the missing BASELINE scenes
        from master_amd import scenegen

        return scenegen.load(name)
    return ma.Scene.load(os.path.join(ROOT, "scenes", name + ".miscene"))


def workload_key(scene, W, H, spp, max_path):
    return "%s_%dx%dx%d_mp%d" % (scene, W, H, spp, min(max_path, 999))


def traffic_entry(key):
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        return json.load(open(tj)).get(key)
    return None


def roofline_block(ma, pt, ist, seg_per_launch, avg_ms, li, key):
    """The dominant kernel's roofline for one workload.  ist: statistics of the instrumented variant on the same workload."""
    kernel = pt.get_kernel()
    lds_scene = kernel == ma.KERNEL_MEGA_LDS
    node_bytes = 64.0  # SURVEY 8(d) prices a visited node at 64 B whatever the build stores (32-byte quantised nodes read half of that)
    if li.flat_leaves:
        node_bytes = 32.0  # flat leaf list: N counts the leaf-table entries a ray is tested against (32 B each, scalar loads), T two triangles per leaf entered
    b_sample, terms = algorithmic_bytes_per_sample(ist, wavefront=kernel == ma.KERNEL_WAVEFRONT, node_bytes=node_bytes)
    launch_s = avg_ms * 1e-3
    pmc = traffic_entry(key)
    traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
    algorithmic_gbs = b_sample * seg_per_launch / launch_s / 1e9
    # compulsory HBM bytes of a launch of the LDS-resident kernel: the FP64 partial sums (written once, read once by pt_finalize)
    compulsory = float(li.partial_bytes)
    rl = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": "pt_megakernel", "avg_launch_ms": avg_ms,
          "algorithmic_bytes_per_sample": b_sample, "terms": terms, "traffic": traffic, "pmc": pmc}
    if kernel == ma.KERNEL_WAVEFRONT:
        rl["kernel"] = "wavefront pipeline (all kernels of one step)"
    if lds_scene or algorithmic_gbs > HBM_PEAK_GBS:
        # the scene bytes never reach HBM: price the kernel's HBM side by what was measured (or, without a PMC record, by the
        # bytes it must write), and report the algorithmic figure as what it is — bytes served by LDS / caches
        hbm_bytes = traffic if traffic is not None else compulsory
        rl["achieved"] = hbm_bytes / launch_s / 1e9
        rl["achieved_source"] = "measured HBM bytes per launch (rocprofv3 PMC, %s)" % pmc["source"] if traffic is not None else \
            "compulsory HBM bytes of the launch (FP64 partial sums); no PMC record for this workload"
        rl["scene_bytes_served_by"] = "LDS" if lds_scene else "L2 / Infinity Cache"
        rl["algorithmic_GBs_not_hbm"] = algorithmic_gbs
        rl["note"] = ("the scene is LDS-resident: SURVEY 8(d)'s per-segment bytes are LDS reads, so the HBM fraction is tiny by design and the bound "
                      "that matters is `valu`" if lds_scene else "the scene is cache-resident: algorithmic bytes exceed what HBM could deliver")
    else:
        rl["achieved"] = algorithmic_gbs
        rl["achieved_source"] = "SURVEY 8(d) algorithmic bytes per segment x segments per launch / launch time (HIP events)"
        rl["note"] = "bytes served by L2 / Infinity Cache count towards `achieved`; `traffic` is what the PMC counters saw leave the L2"
    rl["frac"] = rl["achieved"] / HBM_PEAK_GBS
    assert rl["frac"] <= 1.0, "roofline.frac must be a fraction"
    if traffic is not None:  # what the PMC counters saw leave the L2, against the HBM peak (VERDICT r02 #4c): the algorithmic figure counts cache-served bytes too
        rl["hbm_measured_frac"] = traffic / launch_s / 1e9 / HBM_PEAK_GBS
    if li.flat_leaves:
        rl["traversal"] = "flat leaf list: %d leaf boxes per ray in one wave-uniform loop (scalar operands), then per-lane tests of the leaves entered" % li.flat_leaves
    # VALU view: issue utilisation and active lanes per issued instruction need PMC counters (separate rocprofv3 passes); the
    # traversal loops' lane efficiency is measured in this run by the instrumented kernel
    valu = {"traversal_lane_efficiency_in_run": {"closest": terms["simd_efficiency_closest_traversal"], "shadow": terms["simd_efficiency_shadow_traversal"]},
            "peak_TFLOPs_fp32": FP32_PEAK_TFLOPS}
    if pmc and "valu_issue_utilisation" in pmc:
        valu.update(issue=pmc["valu_issue_utilisation"], lanes=pmc["valu_thread_utilisation"],
                    frac=pmc["valu_issue_utilisation"] * pmc["valu_thread_utilisation"], source=pmc["source"],
                    note="issue = SQ_ACTIVE_INST_VALU x 2 cycles / (kernel cycles x 1024 SIMDs); lanes = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); "
                         "frac = fraction of the FP32 lane-issue slots of the chip doing path work")
    rl["valu"] = valu
    return rl


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
    ap.add_argument("--no-hbm-workload", action="store_true", help="skip the HBM-resident second workload (N = 1 only)")
    ap.add_argument("--no-time-to-rmse", action="store_true", help="skip the time-to-target-RMSE block (N = 1, default scene; the second half of BASELINE.json's metric).  "
                    "Pass it under rocprofv3: the block launches the same kernel variant at other sample counts, which would mix into the per-kernel average")
    ap.add_argument("--time-to-rmse", action="store_true", help="(default since round 3; kept so older command lines still parse)")
    ap.add_argument("--hbm-scene", default="atrium")
    ap.add_argument("--hbm-size", default="1920x1080x256", help="WxHxSPP of the HBM-resident workload (BASELINE configs[3] shape)")
    ap.add_argument("--hbm-scene2", default="atrium:2000000", help="second HBM workload: a scene larger than the Infinity Cache ('' skips it)")
    ap.add_argument("--hbm-size2", default="1920x1080x64")
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
    gloo_cpu = world > 1 and args.backend != "nccl"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":  # RCCL over xGMI: one rank per GPU
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    scene = load_scene(args.scene)
    if args.max_path <= 0:
        args.max_path = ma.PTRDIFF_MAX
    pt = ma.PathTracing(scene, lights=1.0, roulette=0.9, beta=1.0, max_path=args.max_path, device=local_rank)
    if args.kernel:
        pt.set_kernel(args.kernel)
    W, H = args.width, args.height
    fb = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    seed = 0x5EED

    def merge(t):
        """merge_exr (Options.cpp:1340-1409) as ONE collective: sum of (R, G, B, denom) over the ranks."""
        if world == 1:
            return t
        if gloo_cpu:  # rehearsal backend: the reduce runs on host copies
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            t.copy_(c)
            return t
        return madist.merge_framebuffers(t)

    if world > 1:  # communicator and channel set-up for this message size is not a step: do it before anything is timed, whatever --warmup says
        merge(fb)
        torch.cuda.synchronize()

    tiles = args.shard == "tiles" and world > 1
    if tiles:
        pt.set_tile_shard(rank, world)

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    render_ms, reduce_ms = [], []

    def step(i, timed=False):
        if tiles:  # this rank's 32x32 tiles, all world x spp samples of the step
            off, n = madist.tile_sample_range(i, world, args.spp)
        else:      # every pixel, this rank's spp samples of the step
            off, n = madist.sample_offset(i, rank, world, args.spp), args.spp
        ev[0].record()
        st = pt.render_device(fb.data_ptr(), W, H, spp=n, seed=seed, sample_offset=off, stream=stream, want_stats=True)
        ev[1].record()
        merge(fb)  # RCCL all-reduce (sum) of (R, G, B, denom): merge_exr semantics
        ev[2].record()
        if timed:
            ev[2].synchronize()
            render_ms.append(ev[0].elapsed_time(ev[1]))
            reduce_ms.append(ev[1].elapsed_time(ev[2]))
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
        st = step(args.warmup + i, timed=world > 1)
        segs += st.num_basic_rays
        shadow += st.num_shadow_rays
        paths += st.num_paths
        kernel_ms.append(st.trace_ms)
    fence()
    elapsed = time.perf_counter() - t0
    li = pt.last_launch()

    per_rank = None
    if world > 1:
        dev = "cpu" if gloo_cpu else "cuda"
        tot = torch.tensor([elapsed, float(segs), float(shadow), float(paths)], dtype=torch.float64, device=dev)
        tmax = tot[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        sums = tot[1:].clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        elapsed = float(tmax.item())
        segs, shadow, paths = [float(x) for x in sums.tolist()]
        mine = torch.tensor([sum(render_ms) / len(render_ms), sum(reduce_ms) / len(reduce_ms), sum(kernel_ms) / len(kernel_ms)], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"render_ms": [float(a[0]) for a in allr], "reduce_ms": [float(a[1]) for a in allr], "kernel_ms": [float(a[2]) for a in allr],
                    "reduce_bytes": W * H * 16, "reduce": "all-reduce(sum) of [H][W][4] f32 over %s; reduce_ms of a rank includes waiting for the slowest rank's render" % (
                        "RCCL/xGMI" if args.backend == "nccl" else args.backend + " on host copies (rehearsal)")}
    denom_ok = bool((fb[..., 3] == float(args.spp * world)).all().item())

    out = None
    if rank == 0:
        # roofline of the dominant kernel (pt_megakernel): one launch's bytes / its HIP-event duration
        pt.set_instrumented(True)
        ist = pt.render_device(fb.data_ptr(), W, H, spp=min(args.spp, 64), seed=seed, sample_offset=0, stream=stream, want_stats=True)
        pt.set_instrumented(False)
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        rl = roofline_block(ma, pt, ist, float(st.num_basic_rays), avg_ms, li, workload_key(args.scene, W, H, args.spp, args.max_path))
        is_c2 = (args.scene, W, H, args.spp, args.max_path) == ("CornellBoxDiffuse", 512, 512, 1024, 8)
        procedural = args.scene.split(":")[0] in ("atrium", "clutter")
        out = {
            "metric": "Msamples/sec (paths x bounces: closest-hit path segments per second)",
            "value": segs / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (seeded procedural scene, counter-based sample streams)" if procedural else
                    "the reference's own model (models/%s.blend converted by the build's reader to scenes/%s.miscene); sample streams are seeded counters, there is no data set" % (args.scene, args.scene),
            "config": {"workload": "%s, PT, %dx%d, %d spp per GPU per step, max path %s, beta 1, roulette 0.9%s" % (
                           args.scene + (" (procedural stand-in)" if procedural else ".blend"), W, H, args.spp,
                           "unlimited" if args.max_path >= ma.PTRDIFF_MAX else args.max_path,
                           " (BASELINE configs[1])" if is_c2 else ""),
                       "kernel": {1: "pt_megakernel<LDS scene>", 2: "pt_megakernel<HBM scene>", 3: "wavefront pipeline (wf_extend / wf_shade / wf_shadow / wf_regen)"}[pt.get_kernel()],
                       "launch": {"workgroups": li.n_blocks, "sample_chunks": li.n_chunks, "lds_bytes_per_workgroup": li.lds_bytes, "partial_sum_bytes": li.partial_bytes, "flat_leaves": li.flat_leaves},
                       "parallelism": "%s sharded over %d GPU(s), %s all-reduce of [H][W][4] f32" % ("32x32 pixel tiles" if tiles else "samples", world, "RCCL" if args.backend == "nccl" else args.backend),
                       "Mpaths_per_s": paths / elapsed / 1e6, "Mrays_per_s": (segs + shadow) / elapsed / 1e6,
                       "denom_equals_spp": denom_ok},
            "roofline": rl,
        }
        if per_rank:
            out["per_rank"] = per_rank
    del pt
    if rank == 0 and world == 1 and not args.no_hbm_workload:
        out["hbm_workload"] = hbm_workload(ma, torch, args, seed, args.hbm_scene, args.hbm_size)
        if args.hbm_scene2:  # a scene beyond the 256 MB Infinity Cache: the kernel that really reaches HBM
            out["hbm_workload_beyond_cache"] = hbm_workload(ma, torch, args, seed, args.hbm_scene2, args.hbm_size2)
    if rank == 0 and world == 1 and not args.no_time_to_rmse and args.scene == "CornellBoxDiffuse":
        try:  # the second half of BASELINE.json's metric; never allowed to cost the line
            out["time_to_rmse"] = time_to_rmse(ma, scene, args)
        except Exception as e:  # noqa: BLE001
            out["time_to_rmse"] = {"error": repr(e)}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, args, args.cpu_budget)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


def time_to_rmse(ma, scene, args, target=0.01, frame_spp=16, ref_spp=65536, max_spp=4096):
    """BASELINE.json's second metric: wall time until the RMS error (ImageView.cpp:60-85) of the accumulating view against a high-spp image drops
    below `target`, rendering `frame_spp` samples per Technique::render call and logging (clock_time, rms_error) per call like the reference's
    record_t (Technique.cpp:61-76).  The reference EXRs are missing blobs, so the target image is this integrator's own at `ref_spp` (other seed,
    disjoint samples); the time includes the per-call framebuffer download and the host-side RMS.  Outside the timed region of `value`."""
    import numpy as np
    w, h = args.width, args.height
    pt = ma.PathTracing(scene, max_path=args.max_path)
    ref = np.zeros((h, w, 4), np.float64)
    for k in range(0, ref_spp, 4096):
        ref += pt.render_rgbn(w, h, spp=min(4096, ref_spp - k), seed=999, sample_offset=k)
    ref_rgb = (ref[..., :3] / ref[..., 3:]).astype(np.float32)
    pt2 = ma.PathTracing(scene, max_path=args.max_path)
    view = np.zeros((h, w, 4), np.float64)
    pt2.render(view, seed=1, reference=ref_rgb, spp=frame_spp)  # warm-up call (first launch of a handle), then start over
    pt2 = ma.PathTracing(scene, max_path=args.max_path)
    view[:] = 0
    t0 = time.perf_counter()
    hit = None
    while pt2.statistics().num_samples < max_spp:
        rec = pt2.render(view, seed=1, reference=ref_rgb, spp=frame_spp)
        if rec["rms_error"] <= target:
            hit = (time.perf_counter() - t0, pt2.statistics().num_samples, float(rec["rms_error"]))
            break
    return {"target_rms": target, "seconds": hit[0] if hit else None, "spp_at_target": hit[1] if hit else None, "rms_at_target": hit[2] if hit else None,
            "frame_spp": frame_spp, "reference": "%d spp of the same estimator (the reference's EXRs are not in the tree)" % ref_spp,
            "includes": "per-call framebuffer download and host-side RMS (ImageView.cpp:60-85)"}


def hbm_workload(ma, torch, args, seed, scene_name, size):
    """One HBM-resident configuration after the primary timed region: 1 warm-up + 2 steps, own ms_per_step and roofline."""
    W, H, spp = [int(x) for x in size.split("x")]
    t0 = time.perf_counter()
    scene = load_scene(scene_name)
    t_scene = time.perf_counter() - t0
    pt = ma.PathTracing(scene, lights=1.0, roulette=0.9, beta=1.0, max_path=ma.PTRDIFF_MAX, device=torch.cuda.current_device())
    fb = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    pt.render_device(fb.data_ptr(), W, H, spp=spp, seed=seed, sample_offset=0, stream=stream, want_stats=True)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    segs = shadow = paths = 0
    kernel_ms = []
    steps = 2
    for i in range(steps):
        st = pt.render_device(fb.data_ptr(), W, H, spp=spp, seed=seed, sample_offset=(1 + i) * spp, stream=stream, want_stats=True)
        segs += st.num_basic_rays; shadow += st.num_shadow_rays; paths += st.num_paths
        kernel_ms.append(st.trace_ms)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    li = pt.last_launch()
    denom_ok = bool((fb[..., 3] == float(spp)).all().item())  # of the last timed step: the instrumented pass below renders into fb again
    pt.set_instrumented(True)
    ist = pt.render_device(fb.data_ptr(), W, H, spp=min(spp, 8), seed=seed, sample_offset=0, stream=stream, want_stats=True)
    pt.set_instrumented(False)
    info = pt.bvh_info()
    avg_ms = sum(kernel_ms) / len(kernel_ms)
    rl = roofline_block(ma, pt, ist, float(st.num_basic_rays), avg_ms, li, workload_key(scene_name, W, H, spp, 999))
    return {"workload": "%s (procedural stand-in for BASELINE configs[3] CrytekSponza: %d triangles, BVH depth %d), PT, %dx%d, %d spp, unbounded paths, beta 1, roulette 0.9" % (
                scene_name, info.n_triangles, info.max_depth, W, H, spp),
            "value": segs / elapsed / 1e6, "unit": "Msamples/s", "steps": steps, "warmup": 1, "ms_per_step": elapsed / steps * 1e3,
            "Mpaths_per_s": paths / elapsed / 1e6, "Mrays_per_s": (segs + shadow) / elapsed / 1e6,
            "scene_bytes_in_hbm": li.scene_bytes, "node_records": {0: "32-byte quantised binary", 1: "64-byte quantised wide (4 grandchildren)", 2: "64-byte float binary"}[li.wide_nodes],
            "tables_in_lds": bool(li.lds_tables), "dynamic_fetch_traversal": bool(li.dynamic_fetch), "scene_build_s": t_scene, "bvh_build_ms": info.build_ms,
            "denom_equals_spp": denom_ok,
            "roofline": rl}


if __name__ == "__main__":
    main()
