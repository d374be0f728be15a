#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace + separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ counters: the TCC
# counters do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") of one workload.
# Usage: tools/profile_gpu.sh <tag> [c5|late|c2]     -> gpurun_out/prof_<tag>/summary_<tag>.{md,json}
#   c5   : the bench workload (C5 exploded + 2,000 warm-up steps), 100 timed steps
#   late : C5 on its 7 original contigs, the late stage's reference-arithmetic scoring steps of bench.py (--late-only): 3 + 3 x 12 steps
#   c4   : the C4 stand-in (40,000 bins, 8 M contacts), explode + 2 cycles of a headless run (PROF_TRACE_ONLY=1: kernel trace only)
#   c3   : the C3 stand-in (3,500 bins x 3 sub-fragments, 600 k contacts), explode + 20 cycles (PROF_TRACE_ONLY=1)
#   c2   : the C2 stand-in (1,086 bins x 3 sub-fragments) on its 7 contigs, 400 full MCMC steps (tools/step_breakdown.py)
set -u
TAG=${1:-r02}
WHAT=${2:-c5}
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
case $WHAT in
  c5)   ARGS="$REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-late-stage --no-hbm-control" ;;
  late) ARGS="$REPO/bench.py --late-only --late-repeats 3" ;;   # exactly the launches bench.py's late_stage.roofline is quoted on (3 warm-up + 3 x 12 timed steps)
  c2)   ARGS="$REPO/tools/step_breakdown.py --n-bins 1086 --nnz 120000 --n-sub 3 --original --steps 400" ;;
  c4)   ARGS="$REPO/tools/run_configs.py C4 --cycles ${C4_CYCLES:-2}" ;;
  c3)   ARGS="$REPO/tools/run_configs.py C3 --cycles ${C3_CYCLES:-20}" ;;
esac
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
echo "trace rc=$?"
if [ -z "${PROF_TRACE_ONLY:-}" ] && [ -z "${PROF_SQ_ONLY:-}" ]; then
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
echo "fetch rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
echo "write rc=$?"
fi
if [ -z "${PROF_TRACE_ONLY:-}" ]; then
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
echo "sq rc=$?"
fi
if [ -n "${PROF_VALU_MIX:-}" ]; then
# the instruction MIX of the vector instructions (round-4 review item 5: a wave64 float64 FMA and a float32 add do not cost a SIMD the same):
# per-type counters in passes of their own (SQ counter slots), priced by tools/summarize_prof.py with the rates of tools/valu_issue_micro.hip
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace --output-format csv -d $OUT/mix64 -- python3 $ARGS > $OUT/mix64.log 2>&1
echo "mix64 rc=$?"
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --kernel-trace --output-format csv -d $OUT/mix32 -- python3 $ARGS > $OUT/mix32.log 2>&1
echo "mix32 rc=$?"
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/mixint -- python3 $ARGS > $OUT/mixint.log 2>&1
echo "mixint rc=$?"
fi
cd $REPO
python3 tools/summarize_prof.py $OUT $TAG
# raw CSVs are too big to travel back (64 MiB cap): keep only the summaries and logs
find $OUT -name "*.csv" -delete
find $OUT -name "*.db" -delete
du -sh $OUT
