# A/B of the 2-stage tiled kernel's DMA placement on the bench workload: MRISR_GEMM_FLAGS 16384 = both halves of a stage at the top of the tile
for fl in 16384 0 16384 0; do
  MRISR_GEMM_FLAGS=$fl timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bl_ab.json 2> gpurun_out/bl_ab.err || exit 1
  python - "$fl" <<P
import json, sys
d = json.loads(open("gpurun_out/bl_ab.json").read().strip().splitlines()[-1])
c = d["roofline"]["classes_ms_per_step"]
print("flags", sys.argv[1], "slices/s", round(d["value"], 2), "step ms", round(d["denoise_step_ms"], 3), {k: v for k, v in c.items() if "_bl" in k})
P
done
