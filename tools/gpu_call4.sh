#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O; cd $R
export TMPDIR=/tmp
RSR_PERF_VARIANTS="xcd_order=1;xcd_order=0;xcd_order=1;xcd_order=0;xcd_order=1;alternate_order=0;alternate_order=1" timeout 600 python tools/flow_diag.py perf > $O/perf.log 2>&1; echo "perf rc=$?"
grep -E "ms/frame" $O/perf.log
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_all.log 2>&1; echo "all rc=$?"; tail -15 $O/pytest_all.log
