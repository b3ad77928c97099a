for v in "-DREPLICAS=16" "-DREPLICAS=32"; do echo "=== extra: $v"; INVPREF_HIPCC_EXTRA="$v" python -c "from invpref_kdd_2022_amd import build; build.build(force=True)"; python tools/kbench.py 2>&1 | grep "rows per" | head -3; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof4 -- python3 $GRAFT_REPO_ROOT/tools/kbench.py > /dev/null 2>&1
head -4 $GRAFT_REPO_ROOT/gpurun_out/prof4/*/*kernel_stats.csv | cut -c1-150
