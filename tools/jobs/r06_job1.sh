set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_barrsmith.py -x -q -m gpu -s 2>&1 | grep -v "^\[Multi-H\]" | tail -30
