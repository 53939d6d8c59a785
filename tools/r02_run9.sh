#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r02_gputest_full_b.log 2>&1
tail -14 gpurun_out/r02_gputest_full_b.log | cut -c1-200
timeout 900 python bench.py --steps 5 --warmup 1 > gpurun_out/r02_bench_cfg3_b.json 2> gpurun_out/r02_bench_cfg3_b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_bench_cfg3_b.json').read().strip().split('\n')[-1])
print({k:d[k] for k in ('value','ms_per_step','tps_fit_ms','tps_solve_gflops','rsq_model','rsq_final')})
for r in d['kernels']: print(r['kernel'], round(r['launch_ms'],2), round(r['frac'],3))
print(d['f64_boundary'])
PY
