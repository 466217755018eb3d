cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
WORKLOADS="config3 config5" REPS=3 bash tools/gpu_quick_ab.sh
timeout 1500 python -m pytest tests -m gpu -q -x -k "not soak" > gpurun_out/pytest_patch.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_patch.log | cut -c1-300
