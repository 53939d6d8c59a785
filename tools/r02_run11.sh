#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -m gpu -q -x -k "tps_fit or golden or sharded or tiles or abi" > gpurun_out/r02_gputest12.log 2>&1
tail -8 gpurun_out/r02_gputest12.log | cut -c1-200
timeout 600 python tools/fit_speed.py 2000 5000 10000 20000 2>&1 | grep fixed
MHS_CHOL_K128=1 timeout 600 python tools/fit_speed.py 5000 20000 2>&1 | grep fixed
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --tps-mode tiled > gpurun_out/r02_bench_cfg3_tiled.json 2> gpurun_out/r02_bench_cfg3_tiled.err; tail -3 gpurun_out/r02_bench_cfg3_tiled.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_bench_cfg3_tiled.json').read().strip().split('\n')[-1])
print({k:d.get(k) for k in ('value','ms_per_step','rsq_model','rsq_final')})
PY
