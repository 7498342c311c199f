# PMC passes (SQ counters, 8 per pass) over one GEMM shape of tools/gemm_sweep.py with a forced tile.
#   bash tools/pmc_gemm.sh "<shape substring>" <tile> <splitk>
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SHAPE="$1"; T="$2"; S="$3"
cd /tmp
TAG=$(echo "$SHAPE" | tr -c 'a-zA-Z0-9' '_')_t${T}
for PASS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 $R/tools/gemm_one.py "$SHAPE" $T $S 3 > /dev/null 2>&1 || echo "pass failed: $PASS"
done
python3 - "$R/gpurun_out/pmc_$TAG" "$SHAPE" "$T" <<'PY'
import csv, glob, sys, collections
d, shape, t = sys.argv[1:4]
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm' in r['Kernel_Name'] and 'splitk' not in r['Kernel_Name'] and 'fill' not in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
v = {k: agg[k] / max(1, n[k]) for k in agg}
print(f"== {shape} tile {t}")
for k in sorted(v): print(f"  {k:28s} {v[k]:16.0f}")
if 'SQ_WAVE_CYCLES' in v:
    w = v['SQ_WAVE_CYCLES']
    print(f"  wait_any {v.get('SQ_WAIT_ANY',0)/w:.2f}  wait_inst {v.get('SQ_WAIT_INST_ANY',0)/w:.2f}  active {v.get('SQ_ACTIVE_INST_ANY',0)/w:.2f}  (fractions of wave-cycles)")
PY
