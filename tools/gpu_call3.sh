#!/bin/bash
# round-3 GPU session 3: resident-weight kernels -- parity, then A/B against streaming (flow_flags 4)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "parity rc=$?"; tail -5 $O/pytest_parity.log
RSR_PERF_VARIANTS="flow_flags=0;flow_flags=4;flow_flags=0;flow_flags=4;flow_flags=0" timeout 600 python tools/flow_diag.py perf > $O/perf.log 2>&1; echo "perf rc=$?"
grep -E "ms/frame|->32 @1x|192->64|64->64|64->3 " $O/perf.log
timeout 600 python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "options or fused or guards or concurrent or c5" > $O/pytest_r2.log 2>&1; echo "r2 rc=$?"; tail -5 $O/pytest_r2.log
