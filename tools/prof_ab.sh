cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in default emptyfin; do
  lib=$R/invpref_kdd_2022_amd/variants/$v.so; [ $v = default ] && lib=$R/invpref_kdd_2022_amd/libinvpref_hip.so
  export INVPREF_LIB=$lib KB3_SHORT=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab_$v -- python3 $R/tools/kb3.py > $R/gpurun_out/prof_ab_$v.log 2>&1
  echo "## $v"; grep -E "pp=0" $R/gpurun_out/prof_ab_$v.log
  f=$(ls -t $R/gpurun_out/prof_ab_$v/*/*kernel_stats.csv | head -1); head -4 $f | cut -c1-200
done
