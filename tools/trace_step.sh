#!/bin/bash
# GPU box: kernel trace of the default bench, timeline of one graph-replayed step -> gpurun_out/<dir>/timeline.txt
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-trace}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cd $R
T=$(find $O/raw -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T > $O/timeline.txt
rm -rf $O/raw
tail -1 $O/timeline.txt
