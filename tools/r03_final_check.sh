#!/bin/bash
# what the driver runs at round end, in one call: the GPU suite, smoke(), the default bench line
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03check
mkdir -p $OUT
timeout 2400 python3 -m pytest tests -q -m gpu > $OUT/t.log 2>&1; echo "tests rc=$?"
tail -5 $OUT/t.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench.json
python3 -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['gpu_vs_cpu_same_step_rel_dev'])"
