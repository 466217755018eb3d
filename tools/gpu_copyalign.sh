#!/bin/bash
# round 6: A/B of the copy kernels' switches (AQC_COPY_ALIGN / AQC_GEN_ALIGN / AQC_GEN_SMALL / AQC_GEN_DECODE_LDS / AQC_GEN_PREFETCH, aqc_text.hpp), of
# global against flat memory instructions (AQC_FLAT_AS, aqc_kernels.hpp) and of the verdict kernel's loads on the 16-byte grid (AQC_ABL=129) —
# the GPU tests with the tree's library, then the device step of each workload with it and with the build/ablate/*.so named for it, interleaved
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
if [ "${TESTS:-1}" = 1 ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x -k "${K:-not soak}" > gpurun_out/pytest_copyalign.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_copyalign.log | cut -c1-300
fi
: > gpurun_out/copyalign_ab.log
A=build/ablate
run() { W=$1; shift; P=5000000; [ $W = config5 ] && P=3000000
  echo "## $W" | tee -a gpurun_out/copyalign_ab.log
  LIBS="$*" REPS=${REPS:-2} bash tools/gpu_libs_ab.sh --workload $W --pairs $P | tee -a gpurun_out/copyalign_ab.log; }
run config3 $A/lib_flat.so $A/lib_abl1.so $A/lib_abl129.so
run config5 $A/lib_flat.so $A/lib_declds0.so $A/lib_prefetch.so
run config2 $A/lib_flat.so $A/lib_declds0.so
for W in ${TRACE:-config3 config5 config2}; do
  P=5000000; [ $W = config5 ] && P=3000000
  for f in afterqc_amd/csrc/libafterqc_hip.so $A/lib_flat.so; do
    rm -rf /tmp/kt; (cd /tmp && AQC_LIB=$GRAFT_REPO_ROOT/$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --workload $W --pairs $P --cpu-sample 0 --device-only --device-steps 10 --no-pmc --no-fused-step --text-step-only > /dev/null 2>&1)
    echo "# $W $f"; grep -h "fmt_\|text_index\|fast_filter\|kmer\|qc_stat" /tmp/kt/*kernel_stats.csv | sed 's/(aqc::[^"]*"/"/' | cut -c1-120 | head -10
  done
done | tee -a gpurun_out/copyalign_ab.log
