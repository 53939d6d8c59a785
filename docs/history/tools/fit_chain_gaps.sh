cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pf -o fit -- python $GRAFT_REPO_ROOT/tools/fit_prof.py > /tmp/pf.log 2>&1
python $GRAFT_REPO_ROOT/tools/fit_chain_gaps.py
cd $GRAFT_REPO_ROOT
MHS_TIMING=1 timeout 300 python tools/fit_prof.py 2>&1 | tail -9
