#!/bin/bash
# round 2, GPU session 1: new tests + bench lines (cfg3 default, cfg4-mini, 2-rank plumbing, cfg4)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --durations=10 -k "sharded or golden or cfg1 or cfg5 or ensemble or r_capture" > gpurun_out/r02_gputest3.log 2>&1
tail -15 gpurun_out/r02_gputest3.log
timeout 900 python bench.py --workload cfg4-mini --steps 2 --warmup 1 > gpurun_out/r02_bench_cfg4mini.json 2> gpurun_out/r02_bench_cfg4mini.err; tail -c 1500 gpurun_out/r02_bench_cfg4mini.json; tail -5 gpurun_out/r02_bench_cfg4mini.err
MHS_BENCH_BACKEND=gloo MHS_BENCH_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload cfg3-mini --steps 2 --warmup 1 > gpurun_out/r02_bench_2rank_mini.json 2> gpurun_out/r02_bench_2rank_mini.err; tail -c 2500 gpurun_out/r02_bench_2rank_mini.json; tail -5 gpurun_out/r02_bench_2rank_mini.err
timeout 1500 python bench.py --steps 3 --warmup 1 > gpurun_out/r02_bench_cfg3_a.json 2> gpurun_out/r02_bench_cfg3_a.err; tail -c 6000 gpurun_out/r02_bench_cfg3_a.json; tail -5 gpurun_out/r02_bench_cfg3_a.err
timeout 1500 python bench.py --workload cfg4 --steps 1 --warmup 1 > gpurun_out/r02_bench_cfg4.json 2> gpurun_out/r02_bench_cfg4.err; tail -c 2000 gpurun_out/r02_bench_cfg4.json; tail -5 gpurun_out/r02_bench_cfg4.err
