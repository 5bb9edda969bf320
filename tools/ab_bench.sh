#!/bin/bash
# A/B of two builds of the library on the same box: tools/ab_bench.sh <workload> <steps> [extra bench args]
# A = firstordersolvers.jl_amd/csrc/libfoship_ab.so (the comparison build), B = libfoship.so (the tree's); two repetitions each, interleaved.
W=${1:-C5}; K=${2:-10}; shift; shift
for rep in 1 2; do
  for lib in libfoship_ab.so libfoship.so; do
    FOSHIP_LIB=$PWD/firstordersolvers.jl_amd/csrc/$lib python bench.py --workload $W --steps $K --no-cpu-baseline --no-raw-instance --no-direct-extra "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline'] if d['roofline_kkt']=='= roofline' else d['roofline_kkt']
print('$lib', 'it/s', d['value'], 'ms/step', d['ms_per_step'], 'sweep_ms', r['avg_kernel_ms'], 'cg/step', d['config']['cg_iters_per_step'])"
  done
done
