#!/bin/bash
# Round 5, GPU call 5: the worker set with the fast move kernel; the whole default bench line (MT leg in its context).
cd $GRAFT_REPO_ROOT
O=gpurun_out/c5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_mt_compat.py -x -q -p no:cacheprovider -k "worker_set or cpus8" > $O/t_mt.log 2>&1
grep -v WARNING $O/t_mt.log | tail -4
timeout 300 python tools/mt_workers_speed.py novaseq 8 64 256 > $O/mt_speed.log 2>&1
grep "\"value\"\|\"workers\"\|per_worker\|error" $O/mt_speed.log | tail -16
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o mt --output-format csv -- python $GRAFT_REPO_ROOT/tools/mt_workers_speed.py novaseq 64 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
head -9 $O/prof/*kernel_stats.csv | cut -c1-150
(time python bench.py) > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value %.4g frac %.4f main_ms %.4f" % (d["value"], d["roofline"]["frac"], d["kernel_ms_per_step"]["main_ms"]))
for k,v in d["other_workloads"].items():
    if k=="mt_mode": print(k, json.dumps(v)[:1500])
    else: print(k, "value %.4g" % v.get("value",0), "frac", v.get("k_main_frac_of_hbm_peak"), v.get("kernel_ms_per_step",{}).get("main_ms"), v.get("error"))
for k in ("end_to_end","end_to_end_gzip","end_to_end_4_workers"): print(k, d[k].get("value"), d[k].get("error"))
print(d["cpu_baseline"]["value"], d["cpu_baseline_all_cores"]["value"], d["parity_window"][:30])
PY
