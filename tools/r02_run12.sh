#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -q -x -k "tps_fit or golden or tiles or cfg1 or determinism" 2>&1 | tail -4 | cut -c1-200
timeout 600 python tools/fit_speed.py 500 2000 5000 20000 2>&1 | grep "fixed\|GCV\|rel err"
python -c "import __graft_entry__ as g; g.smoke()"
