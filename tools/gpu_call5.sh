#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "parity rc=$?"; tail -5 $O/pytest_parity.log
timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "options or fused or corrupt or bgr or concurrent_process or c5 or rows" > $O/pytest_r2.log 2>&1; echo "r2 rc=$?"; tail -5 $O/pytest_r2.log
RSR_PERF_VARIANTS="flow_flags=0;flow_flags=8;flow_flags=0;flow_flags=8;flow_flags=0" timeout 600 python tools/flow_diag.py perf > $O/perf.log 2>&1; echo "perf rc=$?"
grep -E "ms/frame|64->3 |64->64 @4x" $O/perf.log
