# round-3 soak on the GPU box (gpurun): randomised plan sweep, repeated determinism / trajectory tests, long graph-vs-eager run
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak_r03
O=gpurun_out/soak_r03
INVPREF_FUZZ=${SOAK_FUZZ:-300} timeout 1500 python -m pytest tests/test_edge_cases_gpu.py -m gpu -q -k test_random_plan_parameters_and_shapes 2>&1 | tail -3 > $O/fuzz300.txt
for i in $(seq 1 ${SOAK_REPEATS:-50}); do timeout 300 python -m pytest tests/test_manager_gpu.py -m gpu -q -k "test_sharded_epochs_graph_vs_eager or test_g4_yahoo_like or test_reference_loop_through_train_a_batch or test_alpha_schedule" > $O/.one.txt 2>&1; echo "rc=$? $(grep -E "passed|failed|error" $O/.one.txt | tail -1)"; done > $O/repeat50_manager.txt
for i in $(seq 1 ${SOAK_PARITY:-20}); do timeout 600 python -m pytest tests/test_large_traj_gpu.py tests/test_hip_parity.py -m gpu -q 2>&1 | tail -1; done > $O/repeat20_large_and_parity.txt
timeout 1500 python tools/soak.py > $O/soak_1700_epochs.txt 2>&1
for g in 0 1; do INVPREF_NO_GRAPH=$g timeout 300 python tools/nd_check.py 2>&1 | grep -v amdgpu; done > $O/nd_check.txt
tail -2 $O/fuzz300.txt; sort $O/repeat50_manager.txt | uniq -c; sort $O/repeat20_large_and_parity.txt | uniq -c | sed 's/in [0-9.]*s//' | head; tail -3 $O/soak_1700_epochs.txt; cat $O/nd_check.txt | cut -c1-250
