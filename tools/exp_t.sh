#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2 3 4 5 6 7 8; do timeout 600 python -m pytest tests/test_gpu_zz_timing.py -x -q -m gpu 2>&1 | grep -E "AssertionError|passed|failed"; done
