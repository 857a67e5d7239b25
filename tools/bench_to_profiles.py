#!/usr/bin/env python3
"""Files the counters a bench.py run measured live (rocprofv3 --pmc child processes, bench.py collect_live_pmc) under profiles/:

    python tools/bench_to_profiles.py <bench.json> <round, e.g. r04>

writes profiles/<round>/pmc_summary_<workload>.txt (one line per counter and the derived figures) for the primary workload and every
HBM block of the line, and refreshes profiles/traffic.json — the record bench.py falls back to ("pmc_live": false) where rocprofv3
is missing.  The bench line itself is copied to profiles/<round>/bench_n1.json.
"""
import json
import os
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    line = json.loads(open(src).read().strip().splitlines()[-1])
    out_dir = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(out_dir, exist_ok=True)
    shutil.copyfile(src, os.path.join(out_dir, "bench_n1.json"))
    import bench

    blocks = [("c2", line["config"]["workload"], line["roofline"], line["ms_per_step"], None)]
    for key, tag in (("hbm_workload", "atrium"), ("hbm_workload_beyond_cache", "atrium2M"), ("c5_rank_share", "c5_rank_share")):
        if key in line:
            blocks.append((tag, line[key]["workload"], line[key]["roofline"], line[key]["ms_per_step"], key))
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    traffic = json.load(open(tj)) if os.path.exists(tj) else {}
    keys = {"c2": bench.workload_key("CornellBoxDiffuse", 512, 512, 1024, 8), "atrium": bench.workload_key("atrium", 1920, 1080, 256, 999),
            "atrium2M": bench.workload_key("atrium:2000000", 1920, 1080, 64, 999), "c5_rank_share": bench.workload_key("clutter", 3840, 2160, 4096, 999)}
    for tag, workload, rl, ms, key in blocks:
        pmc = rl.get("pmc") or {}
        if not rl.get("pmc_live"):
            print("%s: no live counters in this line (%s)" % (tag, rl.get("pmc_live_error")))
            continue
        with open(os.path.join(out_dir, "pmc_summary_%s.txt" % tag), "w") as o:
            o.write("%s\n%s\nlaunch (HIP events in the bench run) %.3f ms; counters: one launch per rocprofv3 --pmc pass, child processes of the same bench run\n" % (
                pmc.get("kernel"), workload, rl["avg_launch_ms"]))
            for k in sorted(pmc):
                if k not in ("kernel", "source", "pmc_live"):
                    o.write("   %-38s %s\n" % (k, ("%.6g" % pmc[k]) if isinstance(pmc[k], float) else pmc[k]))
            o.write("   %-38s %s\n" % ("roofline.bound", rl["bound"]))
            o.write("   %-38s %.4f\n" % ("roofline.frac", rl["frac"]))
            if "frac" in rl.get("valu", {}):
                o.write("   %-38s %.4f = issue %.4f x lanes %.4f\n" % ("valu.frac", rl["valu"]["frac"], rl["valu"]["issue"], rl["valu"]["lanes"]))
        e = {k: v for k, v in pmc.items() if k != "pmc_live"}
        e["source"] = "profiles/%s/pmc_summary_%s.txt (measured live by bench.py on the box of profiles/%s/bench_n1.json)" % (rnd, tag, rnd)
        traffic[keys[tag]] = e
    json.dump(traffic, open(tj, "w"), indent=1)
    print("profiles/%s refreshed from %s" % (rnd, src))


if __name__ == "__main__":
    main()
