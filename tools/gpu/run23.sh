timeout -k 10 300 python tools/lp_e2e.py c2 2>/dev/null | cut -c1-300
python bench.py > gpurun_out/r23_bench.json 2> gpurun_out/r23_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r23_bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","crossover_wall_ms","crossover_wall_ms_c5_size","crossover_speedup_vs_cpu_same_host")})
print(d["roofline"])
c=d["crossover"]; print("c2", c["lp_c2_end_to_end"]["gpu_ms"], c["lp_c2_end_to_end"].get("simplex_pivots"))
PY
