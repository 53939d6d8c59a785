#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ct
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ct -o c -- python $GRAFT_REPO_ROOT/tools/fit_pmc.py fixed ${1:-20000} 2 > /tmp/ct.log 2>&1
python $GRAFT_REPO_ROOT/tools/chol_timeline.py
