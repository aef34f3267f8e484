for exp in none close_ctx gc close_ctx,gc; do
  echo "== $exp"
  SX_BENCH_EXPERIMENT=$exp SX_BENCH_ONLY_LP_1E6=1 timeout -k 10 300 python bench.py --steps 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['lp_1e6_end_to_end']
print('calls', [round(v) for v in d['gpu_ms_calls']], 'gpp', round(d['gpu_get_perturb_problem_ms'],1), 'resolve', round(d['gpu_resolve_ms'],1))"
done
echo "== no cpu baseline"; SX_BENCH_ONLY_LP_1E6=1 timeout -k 10 300 python bench.py --steps 5 --no-uniform 2>/dev/null | tail -c 300 | cut -c1-300
echo "== K2 tile placement"; timeout -k 10 300 python tools/rb_long_xcd_bench.py --workload shard 50 100 150 200 300 2>&1 | tail -8
timeout -k 10 300 python tools/rb_long_xcd_bench.py --workload netlib 100 200 2>&1 | tail -5
