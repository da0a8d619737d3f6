cd $GRAFT_REPO_ROOT
for m in 0 16; do echo "== MINL=$m"; MINL=$m CTX=16 CASES=separated,r04 REPS=5 timeout 600 python3 tools/batch_probe.py 2>&1 | grep -v "^\[Multi-H\]" | tail -5; done
