#!/bin/bash
# GPU box: the three rocprofv3 passes behind profiles/rNN_* (ROUND=r02 by default) (kernel trace + stats; PMC FETCH_SIZE; PMC WRITE_SIZE - counters in
# their own runs, never combined with trace domains).  Run from the repo root: bash tools/collect_profiles.sh
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_${ROUND:-r02}
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/trace.json 2> $O/trace.err
echo "trace pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/fetch.json 2> $O/fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/write.json 2> $O/write.err
echo "write pass done"
cd $R
S=$(find $O/trace -name "*kernel_stats.csv" | head -1)
F=$(find $O/fetch -name "*counter_collection.csv" | head -1)
W=$(find $O/write -name "*counter_collection.csv" | head -1)
cp $S $O/kernel_stats.csv
python tools/pmc_summary.py $F $W $O/pmc_traffic.json 4     # the product's graph-replayed steps only (2 eager + capture + 1)
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T > $O/timeline.txt || true
rm -rf $O/trace $O/fetch $O/write
ls -la $O
