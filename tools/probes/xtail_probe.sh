for d in 0 1 2 4 3 7; do
  MRISR_XTAIL_DBG=$d timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03s_xtail_dbg$d.json 2> gpurun_out/r03s_xtail_dbg$d.err || exit 1
done
python - <<P
import json
for n in (0,1,2,4,3,7):
    d=json.loads(open("gpurun_out/r03s_xtail_dbg%d.json"%n).read().strip().splitlines()[-1])
    print("dbg", n, "step ms", round(d["denoise_step_ms"],3), "xtail ms/step", d["roofline"]["classes_ms_per_step"].get("xattn_tail_c320"))
P
