#!/bin/bash
# GPU box: SQ / LDS counters of the forward 3x3 conv kernel, one counter group per rocprofv3 run (counters only, no trace domains).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmc_conv; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- python3 $R/tools/conv_one.py > $O/g$i.log 2>&1 || echo "group $i failed: $grp"
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_conv/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3x3" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} {sum(v)/len(v):16.0f}   (n={len(v)})")
PY
rm -rf $O/g*/
