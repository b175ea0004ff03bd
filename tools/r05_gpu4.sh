set -x
O=gpurun_out/r05c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -s -k "c5_tta or folded_last or engine_options" > $O/t.log 2>&1; echo "t rc=$?"; tail -8 $O/t.log
timeout 900 python tests/full_frame_sweep.py c5 c2 > $O/sweep.txt 2>&1; echo "sweep rc=$?"; grep -E "^C[0-9]|OUTSIDE" $O/sweep.txt
