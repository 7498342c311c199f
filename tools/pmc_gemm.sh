set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for T in 1 10 5; do
  for PASS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM"; do
    rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_t$T -- python3 $R/tools/gemm_one.py "640->640 @32 up" $T 1 3 > /dev/null 2>&1 || echo fail $T
  done
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
for T in (1,10,5):
    agg=collections.defaultdict(float); n=collections.defaultdict(int)
    for f in glob.glob(f"{R}/gpurun_out/pmc_t{T}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if 'gemm' in r['Kernel_Name'] and 'splitk' not in r['Kernel_Name']:
                agg[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
    print("tile",T, {k: round(v/max(1,n[k])) for k,v in sorted(agg.items())})
PY
