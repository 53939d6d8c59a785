cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_tps_fit_gpu.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -o fit -- python $GRAFT_REPO_ROOT/tools/fit_prof.py > /tmp/pf.log 2>&1
F=$(find /tmp/pf -name "*kernel_stats.csv" | head -1)
head -6 "$F" | cut -c1-60,100-200
cd $GRAFT_REPO_ROOT
MHS_TIMING=1 timeout 300 python tools/fit_speed.py 2>&1 | tail -6
