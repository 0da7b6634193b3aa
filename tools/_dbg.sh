cd /root/repo
for env in "X=1" "NMF_STEP_CORE=0" "NMF_HOST_EXT=0" "NMF_OVERLAP=0"; do echo "== $env"; env $env timeout 300 python -m pytest tests/test_hip_e2e.py -m gpu -x -q -s -k "retrace_order_steady_state" 2>&1 | grep "retrace order\|passed\|failed"; done
