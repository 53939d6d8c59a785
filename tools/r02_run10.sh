#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -m gpu -q -x -k "ensemble or sharded or abi" > gpurun_out/r02_gputest11.log 2>&1
tail -8 gpurun_out/r02_gputest11.log | cut -c1-200
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_cfg3_c.json 2> gpurun_out/r02_bench_cfg3_c.err; tail -3 gpurun_out/r02_bench_cfg3_c.err
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --tps-mode tiled > gpurun_out/r02_bench_cfg3_tiled.json 2> gpurun_out/r02_bench_cfg3_tiled.err; tail -3 gpurun_out/r02_bench_cfg3_tiled.err
python - <<'PY'
import json
for f in ('gpurun_out/r02_bench_cfg3_c.json','gpurun_out/r02_bench_cfg3_tiled.json'):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
    except Exception as e:
        print(f, 'ERR', e); continue
    print(f, {k:d.get(k) for k in ('value','ms_per_step','tps_fit_ms','rsq_model','rsq_final')})
    for r in d['kernels']: print('  ',r['kernel'], round(r['launch_ms'],2), round(r['frac'],3))
PY
