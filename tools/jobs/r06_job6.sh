cd $GRAFT_REPO_ROOT
MH_LIB=tools/jobs/libmultih_hip_ticks.so python3 tools/jobs/ticks_probe.py 2>&1 | grep -v "^\[Multi" | cut -c1-900
