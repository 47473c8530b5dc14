cd ${GRAFT_REPO_ROOT:-/root/repo}; timeout 600 python3 -m pytest tests/test_ops_gpu.py -x -q -k "qknorm" 2>&1 | tail -30
