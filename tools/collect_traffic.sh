# HBM traffic per kernel class for bench.py's workload (PMC, separate passes as guides/MI355X_MICROARCH.md prescribes):
#   bash tools/collect_traffic.sh  ->  gpurun_out/traffic.json   (copy to profiles/rNN_traffic.json)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
# the shipped tile table (what bench.py uses), copied so that a missing signature would be appended to the copy only
cp $R/profiles/r03_tune_cache.tsv $R/gpurun_out/tune_cache.tsv
export MRISR_TUNE_CACHE=$R/gpurun_out/tune_cache.tsv
# fill the autotune table first so that the profiled runs contain no tuning launches
python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph --ddim-steps 2 > $R/gpurun_out/traffic_warm.log 2>&1
for PASS in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $R/gpurun_out/traffic_$PASS -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph --ddim-steps 4 > $R/gpurun_out/traffic_$PASS.log 2>&1 || echo "pass failed: $PASS"
done
python3 - "$R/gpurun_out" <<'PY'
import csv, glob, json, re, sys, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
def cls(name):
    m = re.search(r"gemm_bl_kernelILi(\d+)ELi(\d+)E", name) or re.search(r"gemm_bl_kernel<(\d+), (\d+),", name)
    if m: return f"gemm_bf16_bl{m.group(1)}x{m.group(2)}"
    m = re.search(r"gemm_halo_kernelILi(\d+)ELi(\d+)E", name) or re.search(r"gemm_halo_kernel<(\d+), (\d+),", name)
    if m: return f"gemm_bf16_halo{m.group(1)}x{m.group(2)}"
    m = re.search(r"gemm_rp_kernelILi(\d+)ELi(\d+)ELb[01]ELi[01]ELb([01])E", name) or re.search(r"gemm_rp_kernel<(\d+), (\d+), (?:true|false), [01], (true|false), \d+>", name)
    if m: return f"gemm_{'fp8' if m.group(3) in ('1', 'true') else 'bf16'}_rp{int(m.group(1)) * 32}x{int(m.group(2)) * 16}"
    m = re.search(r"gemm_kernelI(DF16b|f)Li(\d+)ELi(\d+)E", name)
    if m: return f"gemm_{'bf16' if m.group(1) != 'f' else 'f32'}_{m.group(2)}x{m.group(3)}"
    if "xattn_tail" in name: return "xattn_tail_c320"
    if "gn_fused" in name: return "groupnorm_from_slabs" if re.search(r"gn_fused_kernelILi\d+ELi\d+ELb1E", name) or re.search(r"gn_fused_kernel<\d+, \d+, true>", name) else "groupnorm_fused"
    if "mlp_fused" in name: return "mlp_fused_c320"
    for k, v in (("attn_fwd", "flash_attention"), ("gn_stats", "groupnorm_stats"), ("gn_apply", "groupnorm_apply"), ("layernorm", "layernorm"),
                 ("splitk_reduce", "splitk_reduce"), ("lora_down", "lora_down"), ("small_conv", "small_conv"), ("gemv_rows", "time_embed_gemv")):
        if k in name: return v
    return None
for p in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{d}/traffic_{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            c = cls(r["Kernel_Name"])
            if c and r["Counter_Name"] == p:
                agg[c][p] += float(r["Counter_Value"]); cnt[c][p] += 1
out = {}
for c in agg:
    n = max(1, cnt[c].get("FETCH_SIZE", 0))
    fetch_kb = agg[c].get("FETCH_SIZE", 0.0) / n
    write_kb = agg[c].get("WRITE_SIZE", 0.0) / max(1, cnt[c].get("WRITE_SIZE", 0))
    # gfx950: FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) streaming reads -> x2 (guide, "HBM")
    out[c] = {"launches": n, "fetch_kb_raw": fetch_kb, "write_kb": write_kb, "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0}
json.dump(out, open(f"{d}/traffic.json", "w"), indent=1, sort_keys=True)
for c, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
    print(f"{c:28s} n={v['launches']:5d} fetch_raw={v['fetch_kb_raw']/1024:9.2f} MB write={v['write_kb']/1024:9.2f} MB  hbm/launch={v['hbm_bytes_per_launch']/1e6:9.2f} MB")
PY
