# HBM / L2 traffic of one GEMM shape (separate PMC passes; FETCH_SIZE under-reports wide streaming reads by 2x on gfx950:
# guides/MI355X_MICROARCH.md "HBM").   bash tools/pmc_tcc.sh "<shape substring>" <tile> <splitk>
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SHAPE="$1"; T="$2"; S="$3"
cd /tmp
TAG=tcc_$(echo "$SHAPE" | tr -c 'a-zA-Z0-9' '_')_t${T}
for PASS in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/tools/gemm_one.py "$SHAPE" $T $S 3 > /dev/null 2>&1 || echo "pass failed: $PASS"
done
python3 - "$R/gpurun_out/$TAG" "$SHAPE" "$T" <<'PY'
import csv, glob, sys, collections
d, shape, t = sys.argv[1:4]
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm' in r['Kernel_Name'] and 'splitk' not in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print(f"== {shape} tile {t}")
for k in sorted(agg): print(f"  {k:24s} {agg[k]/max(1,n[k]):16.1f}")
PY
