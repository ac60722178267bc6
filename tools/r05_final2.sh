#!/bin/bash
# Round 5, last GPU call: the default bench line with the committed counter summaries in it, a longer captured soak.
cd $GRAFT_REPO_ROOT
(time python bench.py) > gpurun_out/bench_r05b.json 2> gpurun_out/bench_r05b.err
tail -2 gpurun_out/bench_r05b.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_r05b.json").read().strip().splitlines()[-1])
print("value %.4g frac %.4f fresh %s pmc %s" % (d["value"], d["roofline"]["frac"], d["roofline"]["traffic_fresh"], json.dumps(d["roofline"]["pmc"])[:300]))
print({k: (v.get("pmc") or {}).get("fresh") for k,v in d["other_workloads"].items() if isinstance(v, dict)})
print("mt", d["other_workloads"]["mt_mode"]["value"], {w: "%.3g" % x.get("value",0) for w,x in d["other_workloads"]["mt_mode"]["worker_sets"].items()})
PY
tools/soak.sh 8 120000 > gpurun_out/soak_r05b.log 2>&1; tail -2 gpurun_out/soak_r05b.log
