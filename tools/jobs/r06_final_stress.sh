# the round's last stress session: every randomised parity tool on the final code, results under gpurun_out/r06/final_stress.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06/final_stress.txt
: > $O
rm -f gpurun_out/stress_alternation.txt
echo "== tools/stress_parity.py SECONDS=${PARITY_S:-500} SEED=${SEED0:-6001} (batch size and first-cycle rule at random)" >> $O
SECONDS=${PARITY_S:-500} SEED=${SEED0:-6001} timeout 1500 python3 tools/stress_parity.py 2>&1 | grep -a "stress ok\|MISMATCH\|differ\|rror" | tail -3 >> $O
echo "== tools/stress_alternation.py SECONDS=${ALT_S:-400} SEED=$((${SEED0:-6001} + 1)) (whole Process() against the oracle's loop)" >> $O
SECONDS=${ALT_S:-400} SEED=$((${SEED0:-6001} + 1)) timeout 1500 python3 tools/stress_alternation.py > /tmp/alt.txt 2>&1; echo "   exit code $?" >> $O
tail -1 gpurun_out/stress_alternation.txt >> $O
echo "== tools/stress_select_refit.py CASES=120 SEED=$((${SEED0:-6001} + 2))" >> $O
CASES=120 SEED=$((${SEED0:-6001} + 2)) timeout 600 python3 tools/stress_select_refit.py 2>&1 | grep -a "stress_select_refit" | tail -2 >> $O
echo "== tools/stress_mean_shift.py CASES=100 SEED=$((${SEED0:-6001} + 3))" >> $O
CASES=100 SEED=$((${SEED0:-6001} + 3)) timeout 600 python3 tools/stress_mean_shift.py 2>&1 | grep -a "stress_mean_shift" | tail -2 >> $O
echo "== tools/stress_residual_edges.py SECONDS=120 SEED=$((${SEED0:-6001} + 4))" >> $O
SECONDS=120 SEED=$((${SEED0:-6001} + 4)) timeout 400 python3 tools/stress_residual_edges.py 2>&1 | grep -a "residual edge" | tail -2 >> $O
cat $O
