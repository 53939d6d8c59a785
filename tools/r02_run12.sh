#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests -m gpu -q -x -k "tps_fit or golden" 2>&1 | tail -3
timeout 600 python tools/fit_speed.py 5000 10000 20000 2>&1 | grep "fixed\|GCV"
