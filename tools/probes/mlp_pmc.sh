#!/bin/bash
# PMC counters of the fused feed-forward kernel (separate passes; no tracing domains besides kernel-trace)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for PASS in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"; do
  tag=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $R/gpurun_out/mlp_pmc_$tag -- python3 $R/tools/mlp_probe.py 32768 > /dev/null 2>&1 || echo "pass failed: $PASS"
done
python3 - "$R/gpurun_out" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(f"{d}/mlp_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mlp_fused" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(agg): print(f"{k:32s} {agg[k] / n[k]:16.0f}  per launch ({n[k]} launches)")
PY
