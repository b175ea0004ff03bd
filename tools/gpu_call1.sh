#!/bin/bash
# round-3 GPU session 1: parity of the new kernel paths, A/B of the skipped work, bench line, power sources
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O; cd $R
export TMPDIR=/tmp
( ls /sys/class/drm/card*/device/hwmon/*/ 2>&1 | head -60; for f in /sys/class/drm/card*/device/hwmon/*/{power1_average,power1_input,power1_cap,freq1_input,freq1_label}; do echo "$f: $(cat $f 2>&1)"; done; python -c "import amdsmi; print('amdsmi ok', amdsmi.__file__)" 2>&1 | tail -1; nproc ) > $O/sys.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "parity rc=$?"; tail -5 $O/pytest_parity.log
RSR_PERF_VARIANTS="flow_flags=0;dbg=32,trim=0;dbg=0,trim=1;dbg=32,trim=0;dbg=0,trim=1;dbg=32,trim=1;dbg=0,trim=0" timeout 600 python tools/flow_diag.py perf > $O/perf.log 2>&1; echo "perf rc=$?"; grep -E "ms/frame" $O/perf.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-1500 $O/bench.json
timeout 300 python tools/power_sampler.py --out $O/power_clock.txt --hz 20 -- python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-host --no-other-configs --no-profile > $O/power.log 2>&1; echo "power rc=$?"; head -8 $O/power_clock.txt
timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py --durations=15 > $O/pytest_rest.log 2>&1; echo "rest rc=$?"; tail -30 $O/pytest_rest.log
