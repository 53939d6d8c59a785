#!/bin/bash
# The ONE command that can move `parity` from "unpinned" to pinned, on any box with R (>= 4.3) and fields + terra installed
# (gbm / randomForest / kernlab / nnet are captured where present):
#     bash tests/golden/capture_and_check.sh
# 1. tests/golden/capture_from_R.R runs the reference's own packages on the committed inputs (tests/golden/r_inputs/*.csv: the
#    fixture station sets, the G6 grids' tile boxes incl. the fragile 1 501^2 ones, the member structures) and writes
#    tests/golden/r_capture/ (fields::Tps objects $c $d $lambda $transform, predict() surfaces, terra::crop windows, package versions);
# 2. tests/test_r_capture.py (skipped while that directory is absent) checks the oracle -- and, on a GPU box, the HIP path --
#    against the capture: lambda (both search modes), c, d, surfaces, integer windows bit for bit, member predictions.
set -e
cd "$(dirname "$0")/../.."
command -v Rscript >/dev/null || { echo "Rscript not found: this script needs R with fields + terra" >&2; exit 2; }
Rscript tests/golden/capture_from_R.R
python -m pytest tests/test_r_capture.py -q "$@"
