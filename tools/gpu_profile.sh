#!/bin/bash
# rocprofv3 summary of the HBM-resident device step of one workload: kernel trace + the HBM / instruction counters, each in its own pass
# usage: tools/gpu_profile.sh config3|config5|config2 [out-file]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; W=${1:-config3}; P=5000000; [ $W = config5 ] && P=3000000
DST=${2:-gpurun_out/profile_summary_$W.txt}
# EXTRA: more bench.py flags, e.g. --spans-step-only / --text-step-only (which variant of the device step the timed region runs)
B="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --device-only --device-steps 10 --workload $W --pairs $P --no-pmc --no-fused-step $EXTRA"
rm -rf $OUT/prof_kt $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kt -o kt -- $B > /dev/null 2>&1)
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o f -- $B > /dev/null 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/prof_write -o w -- $B > /dev/null 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES VALUBusy GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_sq -o s -- $B > /dev/null 2>&1)
python tools/pmc_summary.py $OUT/prof_kt $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq | grep -v -E "rocclr|kmer_compact" > $DST
cut -c1-170 $DST | head -${3:-40}
