set -x
O=gpurun_out/r05b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -s --durations=15 --deselect tests/test_gpu_round2.py::test_c3_tiles_against_oracle_and_batching --deselect tests/test_gpu_round2.py::test_c5_tta_against_oracle > $O/tests.log 2>&1; echo "tests rc=$?"; tail -30 $O/tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json; tail -5 $O/bench.err
