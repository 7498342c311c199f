# A/B of the level-0 GroupNorm geometry on the bench workload: MRISR_GN_WIDE=<channels per slab> (512-thread workgroups, 16-byte vectors); 0 = shipped
for w in 0 40 80 160 0 80; do
  MRISR_GN_WIDE=$w timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/gn_ab.json 2> gpurun_out/gn_ab.err || { tail -3 gpurun_out/gn_ab.err; exit 1; }
  python - "$w" <<P
import json, sys
d = json.loads(open("gpurun_out/gn_ab.json").read().strip().splitlines()[-1])
c = d["roofline"]["classes_ms_per_step"]
print("wide", sys.argv[1], "slices/s", round(d["value"], 2), "step ms", round(d["denoise_step_ms"], 3), {k: v for k, v in c.items() if "groupnorm" in k}, "finite", d["finite"])
P
done
