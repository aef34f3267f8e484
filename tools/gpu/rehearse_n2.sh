#!/bin/bash
# the N = 2 code path of bench.py on a box with ONE GPU (both ranks on device 0, collectives over gloo): does every leg run?
mkdir -p gpurun_out
SX_BENCH_REHEARSAL=gloo timeout -k 10 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/rehearse_n2.json 2> gpurun_out/rehearse_n2.err
rc=$?
tail -c 600 gpurun_out/rehearse_n2.err; echo
python - <<'PY'
import json
d = json.loads(open("gpurun_out/rehearse_n2.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "n_gpus", "ms_per_step", "scaling")})
print("sharded_pricing", d.get("sharded_pricing"))
print("sharded_resolve", d.get("sharded_resolve"))
PY
exit $rc
