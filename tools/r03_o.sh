#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03o
mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_peer_mailbox.py tests/test_gpu_bench_flow.py tests/test_gpu_linesearch.py -q -m gpu -x > $OUT/t.log 2>&1; echo "tests rc=$?"
tail -30 $OUT/t.log
