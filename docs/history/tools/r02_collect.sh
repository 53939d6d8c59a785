#!/bin/bash
# round 2: the evidence set for profiles/ (run on the GPU box through gpurun; results land in gpurun_out/r02_final/)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_final
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=5 > $O/gputest_full.log 2>&1; tail -4 $O/gputest_full.log | cut -c1-160
timeout 900 python bench.py > $O/bench_cfg3_n1.json 2> $O/bench_cfg3_n1.err; tail -c 300 $O/bench_cfg3_n1.json; echo
timeout 900 python bench.py --tps-mode tiled --no-cpu-baseline > $O/bench_cfg3_n1_tiled_tps.json 2>/dev/null
timeout 600 python bench.py --workload cfg2 --no-cpu-baseline > $O/bench_cfg2_n1.json 2>/dev/null
timeout 1200 python bench.py --workload cfg4 --steps 2 --warmup 1 > $O/bench_cfg4_n1.json 2>/dev/null
MHS_BENCH_SKIP_F64=1 timeout 1500 python bench.py --workload cfg5 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_cfg5_n1.json 2>/dev/null
( cd /tmp && rm -rf /tmp/kst && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o cfg3 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/tmp/kst.log )
find /tmp/kst -name "*kernel_stats.csv" -exec cp {} $O/cfg3_rocprofv3_kernel_stats.csv \;
find /tmp/kst -name "*domain_stats.csv" -exec cp {} $O/cfg3_rocprofv3_domain_stats.csv \;
head -8 $O/cfg3_rocprofv3_kernel_stats.csv | cut -c1-150
timeout 600 python tools/fit_speed.py 500 2000 5000 10000 20000 2>&1 | grep -v "^/opt" > $O/fit_speed.txt; cat $O/fit_speed.txt
bash tools/fit_chain_gaps.sh > $O/fit_chain_gaps.txt 2>&1
timeout 600 python tools/fit_beside_member.py 2>&1 | grep "reserve\|alone" > $O/fit_beside_member.txt
timeout 600 python tools/learn_fit_speed.py 5000 20000 2>&1 | grep -v "^/opt" > $O/learn_fit_speed.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/cu_mask_probe.hip -o /tmp/cu_mask_probe 2>/dev/null && /tmp/cu_mask_probe > $O/cu_mask_probe.txt 2>&1
bash tools/r02_chol_timeline.sh 20000 > $O/chol_timeline_n20000.txt 2>&1
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(sys.argv[1].split('/')[-1], {k:d.get(k) for k in ('value','ms_per_step','tps_fit_ms','rsq_model','rsq_final')})
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
