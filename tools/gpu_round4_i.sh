cd $GRAFT_REPO_ROOT
run() { python bench.py $1 --steps 12 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-workloads 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$2', 'value %.4g' % d['value'], 'main %.4f' % k['main_ms'], 'frac %.3f' % d['roofline']['frac'], d['parity_window'][:2])"; }
for m in hiseq nextseq miseq novaseq miseq-legacy; do run "--model $m" "$m"; done
run "--workload configs3" configs3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "hip_matches_oracle or both_indel or baseline_sizes or randomized_differential" 2>&1 | tail -3
