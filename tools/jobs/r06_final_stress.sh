cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06/final_stress.txt
: > $O
echo "== tools/stress_parity.py SECONDS=500 SEED=6001 (batch size and first-cycle rule at random)" >> $O
SECONDS=500 SEED=6001 timeout 900 python3 tools/stress_parity.py 2>&1 | grep -v "^\[Multi" | tail -3 >> $O
echo "== tools/stress_alternation.py SECONDS=400 SEED=6002 (whole Process() against the oracle's loop)" >> $O
SECONDS=400 SEED=6002 timeout 900 python3 tools/stress_alternation.py 2>&1 | grep -v "^\[Multi\|^Median\|^Iteration\|^$" | tail -3 >> $O
echo "== tools/stress_select_refit.py CASES=120 SEED=6003" >> $O
CASES=120 SEED=6003 timeout 600 python3 tools/stress_select_refit.py 2>&1 | grep -v "^\[Multi" | tail -2 >> $O
echo "== tools/stress_mean_shift.py CASES=100 SEED=6004" >> $O
CASES=100 SEED=6004 timeout 600 python3 tools/stress_mean_shift.py 2>&1 | grep -v "^\[Multi" | tail -2 >> $O
echo "== tools/stress_residual_edges.py SECONDS=120 SEED=6005" >> $O
SECONDS=120 SEED=6005 timeout 400 python3 tools/stress_residual_edges.py 2>&1 | grep -v "^\[Multi" | tail -2 >> $O
cat $O
