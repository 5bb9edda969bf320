cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests -q -m gpu -x 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^$\|^RCCL" | tail -6
run() { tag=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python3 bench.py "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json
d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], d['config'].get('cg_iters_per_step'), d['config'].get('cg_variant'))"; }
run shard A=1 -- --small --no-cpu-baseline
run shard_dist FOS_FORCE_DIST=1 -- --small --no-cpu-baseline
run shard_dist2 FOS_FORCE_DIST=1 -- --small --no-cpu-baseline
run c4 A=1 -- --no-cpu-baseline
