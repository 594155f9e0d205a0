#!/bin/bash
# usage: tools/bisect_ddp.sh <commit> [reps]   - check out <commit> into .bisect/old, remove the optimiser drain, build, run the two-rank graph step on the GPU box
set -e
C=$1; REPS=${2:-12}
cd /root/repo/.bisect/old
git checkout -q -f $C
sed -i 's/^        if runner is not None:$/        if runner is not None and False:/' audio-ssl_amd/src/upstream/common.py
sed -i 's/if runner is not None and _DDP_DRAIN:/if False:/' audio-ssl_amd/src/upstream/common.py
grep -c "synchronize()" audio-ssl_amd/src/upstream/common.py
cp /root/repo/tools/ddp_flaky.py tools/ddp_flaky.py
python audio-ssl_amd/build.py > /tmp/bisect_build.log 2>&1 || { tail -20 /tmp/bisect_build.log; exit 1; }
cd /root/repo
/usr/local/graft/bin/gpurun --timeout 900 -- "mkdir -p gpurun_out/r2a; cd .bisect/old && timeout -k 10 600 python tools/ddp_flaky.py delores_s $REPS graph > ../../gpurun_out/r2a/bisect_$C.log 2>&1; grep '^rep\|Error' ../../gpurun_out/r2a/bisect_$C.log" 2>&1 | grep "^rep\|Error\|status" | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10}'
