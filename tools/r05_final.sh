set -x
O=gpurun_out/r05g; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; j=json.load(open('$O/bench.json')); r=j['roofline']
print(j['value'], j['ms_per_step'], j['ms_per_step_profiled'], j['config']['whole_path_frac_of_peak'], r['frac'], r['bound'], r['hbm_frac'], r['traffic_source'], r['vendor_gemm_same_board']['all_convs_over_randn_gemm'], {k:v.get('frac_of_peak') for k,v in j['other_configs'].items()}, j['cpu_baseline']['c1_single_thread']['gpu_c1_max_abs_diff_uint8'], j['group_mode']['value'])"
