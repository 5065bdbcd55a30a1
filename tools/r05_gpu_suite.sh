# full GPU suite on the box: bash tools/r05_gpu_suite.sh TAG
set -u
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/${1:-suite}; mkdir -p $O
export TMPDIR=/tmp
timeout 1400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc $?"; tail -22 $O/pytest_gpu.log | cut -c1-200
