set -x
O=gpurun_out/r05e; mkdir -p $O
timeout 1200 python tests/stress_geometries.py 120 5 > $O/stress.txt 2>&1; echo "stress rc=$?"; tail -5 $O/stress.txt
bash tools/gpu_round.sh r05
