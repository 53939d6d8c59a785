#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_learn_fit_gpu.py -m gpu -q -x 2>&1 | tail -30
