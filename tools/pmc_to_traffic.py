#!/usr/bin/env python3
"""Turns a tools/pmc.sh (or pmc2.sh) summary into an entry of profiles/traffic.json (read by bench.py for roofline.traffic).

    python tools/pmc_to_traffic.py <summary.txt> <workload key> [kernel substring]

FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled per the gfx950 correction in MI355X_MICROARCH.md.
VALU issue utilisation assumes 2 issue cycles per wave64 VALU instruction (32 FP32 lanes per SIMD per clock).
"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def parse(path, want):
    cur, out = None, {}
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            continue
        args = cur.replace(" ", "").split("<")[-1].rstrip(">").split(",") if "<" in cur else []
        if want in cur and not (len(args) >= 3 and args[2] == "true"):  # skip the instrumented (COUNT) variant
            k, v = line.split()
            out[k] = float(v)
    return out


def main():
    summary, key = sys.argv[1], sys.argv[2]
    want = sys.argv[3] if len(sys.argv) > 3 else "megakernel"
    c = parse(summary, want)
    e = {"source": os.path.relpath(summary, ROOT)}
    if "FETCH_SIZE" in c:
        e["fetch_bytes"] = c["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in c:
        e["write_bytes"] = c["WRITE_SIZE"] * 1024
    if "fetch_bytes" in e:
        e["hbm_bytes_per_launch"] = e["fetch_bytes"] + e.get("write_bytes", 0.0)
    cyc = c.get("GRBM_GUI_ACTIVE")
    if cyc:
        e["kernel_cycles_per_xcd"] = cyc / 8
    if "SQ_ACTIVE_INST_VALU" in c and cyc:
        e["valu_issue_utilisation"] = c["SQ_ACTIVE_INST_VALU"] * 2 / (cyc / 8 * 1024)
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
        e["valu_thread_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        e["wait_any_frac_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVES"):
        if k in c:
            e[k] = c[k]
    e["note"] = ("rocprofv3 --pmc in separate passes (tools/pmc.sh), one launch; FETCH_SIZE doubled per the gfx950 correction; "
                 "writes beyond the FP64 partial buffer are register-spill scratch traffic")
    p = os.path.join(ROOT, "profiles", "traffic.json")
    t = json.load(open(p)) if os.path.exists(p) else {}
    t[key] = e
    json.dump(t, open(p, "w"), indent=1)
    print(json.dumps(e, indent=1))


if __name__ == "__main__":
    main()
