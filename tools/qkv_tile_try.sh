# in-model timing of the three QKV projections under alternative tiles (edits a copy of the shipped tile table)
R=$GRAFT_REPO_ROOT
for T in 16 25 14 26 18 28; do
  sed -e "s/^\(32768,960,320,.*o1,.*\)\t[0-9]*\t1$/\1\t$T\t1/" -e "s/^\(8192,1920,640,.*o1,.*\)\t[0-9]*\t1$/\1\t$T\t1/" -e "s/^\(2048,3840,1280,.*o1,.*\)\t[0-9]*\t1$/\1\t$T\t1/" $R/profiles/r01_tune_cache.tsv > $R/gpurun_out/tc_$T.tsv
  MRISR_TUNE_CACHE=$R/gpurun_out/tc_$T.tsv timeout -k 10 200 python $R/tools/shape_profile.py > $R/gpurun_out/sp_qkv_$T.log 2>&1 || exit 1
  echo "== tile $T"; grep -h "total\|N=960 K=320\|N=1920 K=640\|N=3840 K=1280" $R/gpurun_out/sp_qkv_$T.log
done
