# PMC passes (SQ counters) over the flash-attention micro-benchmark (tools/attn_bench.py):  bash tools/pmc_attn.sh
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for PASS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU"; do
  rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_attn -- python3 $R/tools/attn_bench.py > /dev/null 2>&1 || echo "pass failed: $PASS"
done
python3 - "$R/gpurun_out/pmc_attn" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'attn_fwd' in r['Kernel_Name']:
            key = (r['Kernel_Name'][:60], r.get('Grid_Size', ''))
            agg[key][r['Counter_Name']] += float(r['Counter_Value']); n[key][r['Counter_Name']] += 1
for key in agg:
    v = {k: agg[key][k] / max(1, n[key][k]) for k in agg[key]}
    print("==", key)
    for k in sorted(v): print(f"  {k:28s} {v[k]:16.0f}")
    if 'SQ_WAVE_CYCLES' in v:
        w = v['SQ_WAVE_CYCLES']
        print(f"  per wave-cycle: wait_any {v.get('SQ_WAIT_ANY',0)/w:.2f} wait_inst {v.get('SQ_WAIT_INST_ANY',0)/w:.2f} active {v.get('SQ_ACTIVE_INST_ANY',0)/w:.2f} valu_active {v.get('SQ_ACTIVE_INST_VALU',0)/w:.2f} lds_active {v.get('SQ_ACTIVE_INST_LDS',0)/w:.2f} wait_lds {v.get('SQ_WAIT_INST_LDS',0)/w:.2f}")
PY
