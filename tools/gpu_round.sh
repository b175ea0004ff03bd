#!/bin/bash
# One GPU-box session: bench JSON, rocprofv3 kernel-trace stats, PMC traffic passes.  Usage: tools/gpu_round.sh <tag>
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --no-cpu-baseline > $OUT/trace.log 2>&1; echo "trace rc=$?"
for SET in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc_$SET -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $OUT/pmc_$SET.log 2>&1; echo "pmc $SET rc=$?"
done
python $R/tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_summary.txt 2>&1
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
ls -la $OUT
