#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -q -x -k "tps_fit or cfg5_gcv" > gpurun_out/r02_gputest8.log 2>&1
tail -5 gpurun_out/r02_gputest8.log
timeout 600 python tools/fit_speed.py 5400 10000 20000 2>&1 | grep -v "^/opt" > gpurun_out/r02_fit_speed_d.txt
cat gpurun_out/r02_fit_speed_d.txt
MHS_BENCH_SKIP_F64=1 timeout 1500 python bench.py --workload cfg5 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_cfg5_a.json 2> gpurun_out/r02_bench_cfg5_a.err
tail -3 gpurun_out/r02_bench_cfg5_a.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_bench_cfg5_a.json').read().strip().split('\n')[-1])
print({k:d[k] for k in ('value','ms_per_step','tps_fit_ms','reference_tiled_tps_ms','rsq_model','rsq_final')})
for r in d['kernels']: print(r['kernel'], round(r['launch_ms'],1), round(r['frac'],3))
PY
