#!/bin/bash
# the default step repeated for ~12 s per run, the SMI's clocks / power / temperature sampled meanwhile: is the step time the clock?
cd $GRAFT_REPO_ROOT
for run in 1 2 3; do
  python tools/ramp_probe.py 9 1000 0.5 > /tmp/ramp_$run.log 2>&1 &
  PID=$!
  sleep 6
  for j in 1 2 3; do
    rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|socket graphics\|junction\|hotspot" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'; echo
    sleep 1.5
  done
  wait $PID
  echo "run $run: $(awk '{print $(NF-1)}' /tmp/ramp_$run.log | tr '\n' ' ')"
done
