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
    if not any("FETCH_SIZE" in acc[k] for k in acc):  # tools/prof_indel.sh: SQ counters only, no traffic pass
        return
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
