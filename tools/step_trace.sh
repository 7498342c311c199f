#!/bin/bash
# In-graph per-dispatch budget of one denoising step: rocprofv3 kernel trace of a short bench run -> tools/step_sequence.py
#   gpurun -- bash tools/step_trace.sh <tag> [extra env assignments...]      (outputs gpurun_out/<tag>_step_sequence.log, _kernel_stats.csv)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-trace}; shift
for kv in "$@"; do export "$kv"; done
mkdir -p $R/gpurun_out && cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err || { tail -5 $R/gpurun_out/${TAG}_bench.err; exit 1; }
KT=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
KS=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
python3 $R/tools/step_sequence.py $KT --all > $R/gpurun_out/${TAG}_step_sequence.log
cp $KS $R/gpurun_out/${TAG}_kernel_stats.csv
head -45 $R/gpurun_out/${TAG}_step_sequence.log
python3 -c "import json;d=json.load(open('$R/gpurun_out/${TAG}_bench.json'));print('slices/s',d['value'],'step ms',d['denoise_step_ms'])"
