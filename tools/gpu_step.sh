#!/bin/bash
# round 6: the -m gpu tests ($K filters, TESTS=0 skips), then the rocprofv3 summary of the default text step of $W (config3 | config5 | config2)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
if [ "${TESTS:-1}" = 1 ]; then
  timeout 2800 python -m pytest ${FILES:-tests} -m gpu -q -x -k "${K:-not soak}" > gpurun_out/pytest_step.log 2>&1; echo "pytest rc=$?"; grep -v "^{\|options:$" gpurun_out/pytest_step.log | tail -${TAILN:-6} | cut -c1-300
fi
for W in ${WL:-config3}; do
  EXTRA="--text-step-only" bash tools/gpu_profile.sh $W gpurun_out/step_${TAG:-new}_$W.txt 16
done
