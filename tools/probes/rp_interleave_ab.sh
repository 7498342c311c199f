C=mri-diffusion-superresolution_amd/csrc
cp $C/libmrisr.so /tmp/base.so
for v in base il base il; do
  if [ $v = il ]; then cp $C/libmrisr_il.so $C/libmrisr.so; else cp /tmp/base.so $C/libmrisr.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/il_ab.json 2> gpurun_out/il_ab.err || exit 1
  python - "$v" <<P
import json, sys
d = json.loads(open("gpurun_out/il_ab.json").read().strip().splitlines()[-1])
c = d["roofline"]["classes_ms_per_step"]
print(sys.argv[1], "slices/s", round(d["value"], 2), "step ms", round(d["denoise_step_ms"], 3), {k: v for k, v in c.items() if "rp" in k})
P
done
cp /tmp/base.so $C/libmrisr.so
