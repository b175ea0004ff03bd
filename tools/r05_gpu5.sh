set -x
O=gpurun_out/r05d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -s --durations=15 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -32 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python tests/full_frame_sweep.py c3 > $O/sweep_c3.txt 2>&1; echo "sweep rc=$?"; grep -E "^C[0-9]|OUTSIDE" $O/sweep_c3.txt
