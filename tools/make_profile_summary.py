#!/usr/bin/env python3
"""Condense a tools/gpu_profile.sh output directory into the committed profiles/<tag>_* files."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def counters(root):
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]] += 1
    return acc, calls


N_SE, N_SIMD, N_CU = 32, 1024, 256  # MI355X: 8 XCDs x 4 shader engines (the SQ counters' instances), 256 CUs x 4 SIMDs


def derived_summary(src, tag, out_dir, acc, calls, stats_csv, cmd):
    """<tag>_summary.json: per k_main variant of the run, what the counters say about WHY the launch takes what it takes --
    the figures bench.py's other_workloads legs carry (`pmc`).  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* /
    SQ_ACTIVE_INST_* count quad-cycles; SQ_BUSY_CYCLES counts cycles per shader engine (32 of them): cycles per launch and
    engine / launch duration = the clock the chip actually ran the kernel at (it clocks to its power budget)."""
    dur = {}
    for r in csv.DictReader(open(stats_csv)):
        if "k_main" in r["Name"]:
            dur[r["Name"].split("(")[0]] = (float(r["AverageNs"]) * 1e-9, int(r["Calls"]))
    build_id, line = "unknown", None
    for log in glob.glob(src + "/*.log"):
        for ln in open(log, errors="replace"):
            if ln.startswith("#detail {"):  # (bench.py prints everything behind "#detail ", then the short headline line)
                ln = ln[len("#detail "):]
            elif line is not None and "algorithmic_bytes_per_pair" in line.get("roofline", {}):
                continue  # (the headline line of the same run: the detail line has been read)
            if ln.startswith("{") and "library_build_id" in ln:
                line = json.loads(ln)
                build_id = line.get("library_build_id", build_id)
    out = {"command": cmd, "library_build_id": build_id, "kernels": {}}
    if line:
        out["pairs_per_step"] = line["config"].get("pairs_per_step_per_gpu")
        out["algorithmic_bytes_per_pair"] = line["roofline"].get("algorithmic_bytes_per_pair")
    for k in sorted(acc):
        if "k_main" not in k:
            continue
        per = {c: acc[k][c] / calls[k][c] for c in acc[k]}
        d = {"launches_per_pass": max(calls[k].values()), "counters_per_launch": {c: per[c] for c in sorted(per)}}
        t = dur.get(k)
        if t:
            d["avg_launch_ms"] = t[0] * 1e3
            d["launches_in_stats_pass"] = t[1]
        g = per.get
        if g("SQ_BUSY_CYCLES") and t:
            d["effective_clock_ghz"] = g("SQ_BUSY_CYCLES") / N_SE / t[0] / 1e9
        if g("SQ_BUSY_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
            d["valu_issue_busy_frac"] = g("SQ_ACTIVE_INST_VALU") * 4.0 / (g("SQ_BUSY_CYCLES") / N_SE * N_SIMD)
        if g("SQ_INSTS_VALU") and t:
            d["valu_issue_busy_frac_est_1p8ns"] = g("SQ_INSTS_VALU") * 1.8e-9 / N_SIMD / t[0]
        if g("SQ_WAVE_CYCLES"):
            for c, name in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_any_frac"), ("SQ_WAIT_INST_LDS", "wait_inst_lds_frac")):
                if g(c) is not None:
                    d[name] = g(c) / g("SQ_WAVE_CYCLES") if c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") or True else None
        if g("SQ_LDS_IDX_ACTIVE"):
            if g("SQ_LDS_BANK_CONFLICT") is not None:
                d["lds_bank_conflict_ratio"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
            if g("SQ_BUSY_CYCLES"):
                d["lds_active_frac"] = g("SQ_LDS_IDX_ACTIVE") / (g("SQ_BUSY_CYCLES") / N_SE * N_CU)
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            d["fetch_bytes_per_launch"] = g("FETCH_SIZE") * 1024 * 2.0  # (x 2: the guide's gfx950 correction, calibrated in the default set)
            d["write_bytes_per_launch"] = g("WRITE_SIZE") * 1024
            if out.get("pairs_per_step") and out.get("algorithmic_bytes_per_pair") and isinstance(out["pairs_per_step"], int):
                launches_per_step = max(1, round(d["launches_per_pass"] / max(1, (line or {}).get("steps", 1) + (line or {}).get("warmup", 0))))
                alg = out["pairs_per_step"] * out["algorithmic_bytes_per_pair"] / launches_per_step
                d["k_main_launches_per_step"] = launches_per_step
                d["traffic_over_algorithmic"] = (d["fetch_bytes_per_launch"] + d["write_bytes_per_launch"]) / alg
                if t:
                    d["frac_of_hbm_peak"] = alg / t[0] / 1e9 / 8000.0
        out["kernels"][k] = d
    json.dump(out, open(os.path.join(out_dir, tag + "_summary.json"), "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


def main(src, tag):
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    stats = glob.glob(src + "/stats/*kernel_stats.csv")[0]
    with open(stats) as fh, open(os.path.join(out_dir, tag + "_kernel_stats.csv"), "w") as out:
        cmd = open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else \
            "python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads"
        out.write("# rocprofv3 --kernel-trace --stats -- %s\n" % cmd)
        out.write(fh.read())
    acc, calls = counters(src)
    lines = ["# rocprofv3 --pmc <counters> --kernel-trace (separate passes), same bench command; totals over all dispatches",
             "kernel,counter,total,per_dispatch,dispatches"]
    for k in sorted(acc):
        for c in sorted(acc[k]):
            lines.append('"%s",%s,%.6g,%.6g,%d' % (k, c, acc[k][c], acc[k][c] / calls[k][c], calls[k][c]))  # (template names hold commas)
    open(os.path.join(out_dir, tag + "_pmc.csv"), "w").write("\n".join(lines) + "\n")
    derived_summary(src, tag, out_dir, acc, calls, stats, cmd)
    if not any("FETCH_SIZE" in acc[k] for k in acc):  # tools/prof_indel.sh: SQ counters only, no traffic pass
        return
    if "--model" in cmd or "--indel" in cmd or "--workload" in cmd:  # a side leg: its traffic is in <tag>_summary.json (bench.py takes the
        return                                                          # roofline's traffic from the newest *_traffic.json: the default leg's)
    # HBM-side traffic of k_main per launch, corrected per MI355X_MICROARCH.md (FETCH_SIZE x2 for coalesced
    # reads, calibrated here on k_read_dwordx2; WRITE_SIZE x1, calibrated on k_fill_dword); unit KiB
    cal = {}
    if "k_fill_dword" in acc and "k_read_dwordx2" in acc:
        cal["write_factor"] = (1 << 20) / (acc["k_fill_dword"]["WRITE_SIZE"] / calls["k_fill_dword"]["WRITE_SIZE"])
        cal["fetch_factor"] = (1 << 20) / (acc["k_read_dwordx2"]["FETCH_SIZE"] / calls["k_read_dwordx2"]["FETCH_SIZE"])
    key = [k for k in acc if "k_main" in k][0]  # "iss::k_main" or "void iss::k_main<false>"
    m = acc[key]
    n = calls[key]["FETCH_SIZE"]
    fetch = m["FETCH_SIZE"] / n * 1024 * cal.get("fetch_factor", 2.0)
    write = m["WRITE_SIZE"] / calls[key]["WRITE_SIZE"] * 1024 * cal.get("write_factor", 1.0)
    valu = m.get("SQ_INSTS_VALU", 0.0) / max(calls[key].get("SQ_INSTS_VALU", 1), 1)
    # the build of the library the passes ran on: iss_build_id() as the profiled bench run printed it (bench.py compares it
    # with the id of the library IT has loaded -- "traffic_fresh")
    build_id = "unknown"
    for log in glob.glob(src + "/*.log"):
        for line in open(log, errors="replace"):
            if line.startswith("{") and "library_build_id" in line:
                build_id = json.loads(line).get("library_build_id", build_id)
    summary = {"kernel": "iss::k_main", "launches": n, "calibration": cal, "valu_insts_per_launch": valu,
               "library_build_id": build_id,
               "fetch_bytes_per_launch": fetch,
               "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write,
               "command": "bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads (5,000,000 pairs per step in one k_main launch)",
               "pairs_per_launch_avg": 5_000_000}
    json.dump(summary, open(os.path.join(out_dir, tag + "_traffic.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
