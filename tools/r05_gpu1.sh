set -x
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv3x3 or folded or rows_below or residual or kernel_paths or native" > $O/t1.log 2>&1; echo "t1 rc=$?"; tail -5 $O/t1.log
timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -s -k "folded or engine_options or c2_tiles" > $O/t2.log 2>&1; echo "t2 rc=$?"; tail -8 $O/t2.log
timeout 300 python tools/ab_options.py --config C2 --variants "fold=1;fold=0" --rounds 4 > $O/ab_c2_fold.txt 2>&1; cat $O/ab_c2_fold.txt
timeout 400 python tools/ab_options.py --config C3 --variants "fold=1;fold=0" --rounds 3 --frames 3 > $O/ab_c3_fold.txt 2>&1; cat $O/ab_c3_fold.txt
timeout 300 python tools/ab_options.py --config C2 --variants "flow_flags=0;flow_flags=16" --rounds 4 > $O/ab_c2_ring.txt 2>&1; cat $O/ab_c2_ring.txt
for i in 1 2; do for L in realsr-ncnn-vulkan_amd/lib/librealsr_hip.so realsr-ncnn-vulkan_amd/lib/exp/wr4.so realsr-ncnn-vulkan_amd/lib/exp/r04_base.so; do RSR_LIB=$L timeout 200 python tools/ab_options.py --config C2 --variants "num_cu=256" --rounds 3 2>&1 | tail -2; done; done > $O/ab_c2_builds.txt 2>&1; cat $O/ab_c2_builds.txt
