# A/B of the fused feed-forward kernel's DMA schedule on the bench workload (MRISR_MLP_ROT x MRISR_MLP_SPREAD)
for cfg in "0 0" "1 0" "0 1" "1 1" "0 0" "1 1"; do
  set -- $cfg
  MRISR_MLP_ROT=$1 MRISR_MLP_SPREAD=$2 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/mlp_ab.json 2> gpurun_out/mlp_ab.err || exit 1
  python - "$1" "$2" <<P
import json, sys
d = json.loads(open("gpurun_out/mlp_ab.json").read().strip().splitlines()[-1])
print("rot", sys.argv[1], "spread", sys.argv[2], "slices/s", round(d["value"], 2), "step ms", round(d["denoise_step_ms"], 3), "mlp ms/step", d["roofline"]["classes_ms_per_step"].get("mlp_fused_c320"))
P
done
