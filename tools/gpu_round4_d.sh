cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { python bench.py $1 --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-workloads 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$2', 'value %.4g' % d['value'], 'main %.4f scan %.4f setup %.4f fix %.4f' % (k['main_ms'], k['indel_scan_ms'] or 0, k['setup_ms'] or 0, k['indel_fixup_ms'] or 0))"; }
for rep in 1; do


run "--indel 0.001 0.003" indel
done
