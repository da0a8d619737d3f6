cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python3 -m pytest tests/test_gpu_concurrent_moves.py -x -q -m gpu 2>&1 | grep -v "^\[Multi-H\]" | tail -15
SECONDS=150 SEED=6 timeout 400 python3 tools/stress_parity.py 2>&1 | tail -3
timeout 3000 python3 -m pytest tests -x -q -m gpu 2>&1 | grep "passed\|failed\|rror" | tail -5
