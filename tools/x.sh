cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_certificates.py tests/test_gpu_edge_cases.py -q -m gpu -x -k "psd or PSD or cert or cone" 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^$\|^RCCL" | tail -4
python3 bench.py --workload C4 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'): d=json.loads(l); print('C4', d['value'], d['ms_per_step'], json.dumps(d['roofline_psd']), d['time_shares'], d['config']['residuals_after_run'])"
