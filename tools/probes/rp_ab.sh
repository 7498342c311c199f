# A/B of the row-panel kernels' weight-DMA schedule on the bench workload: MRISR_GEMM_FLAGS 4096 = no per-workgroup piece rotation,
# 8192 = next chunk's DMA spread over the K loop
for fl in 4096 0 8192 12288 4096 0 8192; do
  MRISR_GEMM_FLAGS=$fl timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/rp_ab.json 2> gpurun_out/rp_ab.err || exit 1
  python - "$fl" <<P
import json, sys
d = json.loads(open("gpurun_out/rp_ab.json").read().strip().splitlines()[-1])
c = d["roofline"]["classes_ms_per_step"]
print("flags", sys.argv[1], "slices/s", round(d["value"], 2), "step ms", round(d["denoise_step_ms"], 3), {k: v for k, v in c.items() if "rp" in k})
P
done
