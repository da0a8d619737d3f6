set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^\[Multi-H\]" | tail -15 > gpurun_out/r06/gputests_a.txt
tail -5 gpurun_out/r06/gputests_a.txt
timeout 1200 python bench.py > gpurun_out/r06/bench_a.json 2> gpurun_out/r06/bench_a.err
tail -c 600 gpurun_out/r06/bench_a.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06/bench_a.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["transport"])
for k in ("labeling","labeling_on_the_intermediate_scene","labeling_on_the_r04_scene"):
    print(k, {q:d[k].get(q) for q in ("gpu_labeling_step_ms","cycles","moves_run","moves_solved","core_max","barriers","cpu_reference_expansion_ms","labels_identical")})
fl=d["full_loop"]
print({k:fl.get(k) for k in ("iterations_run","loop_s","ms_per_iteration","process_s_second_call","cpu_baseline_loop","reference_route")})
PY
