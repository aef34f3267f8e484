timeout -k 10 400 python tools/rb_long_rows_bench.py --workload shard 64 32 16 8 4 2>&1 | tail -11
timeout -k 10 400 python tools/rb_long_rows_bench.py --workload netlib 64 16 4 2>&1 | tail -7
timeout -k 10 400 python tools/k1_netlib_bench.py 2>&1 | tail -17
echo "== in-bench, no cpu baseline / no uniform"; SX_BENCH_ONLY_LP_1E6=1 timeout -k 10 300 python bench.py --steps 5 --no-uniform 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['lp_1e6_end_to_end']
print('calls', [round(v) for v in d['gpu_ms_calls']], 'gpp', round(d['gpu_get_perturb_problem_ms'],1), 'resolve', round(d['gpu_resolve_ms'],1))"
