#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace + two separate PMC passes of the default bench workload.
# Usage: tools/profile_gpu.sh <tag>      -> gpurun_out/prof_<tag>/{trace,fetch,write}
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-late-stage"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
echo "trace rc=$?"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
echo "fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
echo "write rc=$?"
python3 tools/summarize_prof.py $OUT $TAG
# raw CSVs are too big to travel back (64 MiB cap): keep only the summaries and logs
find $OUT -name "*.csv" -delete
du -sh $OUT
