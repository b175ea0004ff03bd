#!/bin/bash
# One GPU-box session: bench JSON, rocprofv3 kernel-trace stats, PMC passes (HBM traffic + MFMA / LDS / clock counters).
# Usage: tools/gpu_round.sh <tag>  (default r06)     -> gpurun_out/<tag>/{bench.json,power_clock.txt,kernel_stats.csv,pmc_traffic.txt,pmc_counters.txt}
# Copy the summaries you want judged into profiles/ (tracked).
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
# board power + sclk at 20 Hz over a longer timed region of the same command (bench.py writes its marks): profiles/<tag>_power_clock.txt
timeout 300 python $R/tools/power_sampler.py --out $OUT/power_clock.txt --hz 20 -- python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-host --no-other-configs --no-profile --no-group > $OUT/power.log 2>&1; echo "power rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --no-cpu-baseline --no-host --no-other-configs --no-group > $OUT/trace.log 2>&1; echo "trace rc=$?"
BA="--steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-host --no-other-configs --no-group"
i=0
# separate --pmc passes, kernel-trace only (no other trace domain): FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2; SQ has 8 slots
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -- python $R/bench.py $BA > $OUT/pmc$i.log 2>&1; echo "pmc pass $i ($SET) rc=$?"
done
# executed matrix work of a C3 frame (every tile ends in a folded 4-pixel block column) and of a C2 frame, against the algorithmic count
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d $OUT/pmc5 -- python $R/tools/ab_options.py --config C3 --variants "fold=1" --rounds 1 --frames 1 > $OUT/pmc5.log 2>&1; echo "pmc pass 5 (C3 MOPS) rc=$?"
python $R/tools/mops_check.py $OUT/pmc5 C3 4 > $OUT/mops_c3.txt 2>&1
python $R/tools/mops_check.py $OUT/pmc3 C2 2 > $OUT/mops_c2.txt 2>&1
cat $OUT/mops_c3.txt $OUT/mops_c2.txt
python $R/tools/pmc_summary.py --traffic $OUT/pmc1 $OUT/pmc2 > $OUT/pmc_traffic.txt 2>&1
python $R/tools/pmc_summary.py $OUT/pmc3 $OUT/pmc4 > $OUT/pmc_counters.txt 2>&1
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5
ls -la $OUT; cat $OUT/pmc_traffic.txt | head -30
# A/B of the wave layout for the 64-output-channel convs (flow_flags 1) and of the streamed-weight build (4), alternating
timeout 300 python $R/tools/ab_options.py --config C2 --variants "flow_flags=0;flow_flags=1;flow_flags=4" --rounds 3 > $OUT/ab_flags.log 2>&1; grep -v amdgpu.ids $OUT/ab_flags.log
# precise mode (rsr_set_option precise 1): alternating A/B on C2 and C3, the per-class cost, and the kernel trace of a precise C2 frame
( timeout 300 python $R/tools/ab_options.py --config C2 --variants "precise=0;precise=1" --rounds 4; timeout 300 python $R/tools/ab_options.py --config C3 --variants "precise=0;precise=1" --rounds 2 --frames 3; timeout 300 python $R/tools/precise_cost.py 3 ) 2>&1 | grep -v amdgpu.ids > $OUT/precise_cost.txt; cat $OUT/precise_cost.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_p -- python $R/tools/ab_options.py --config C2 --variants "precise=1" --rounds 1 --frames 5 > $OUT/trace_p.log 2>&1; echo "precise trace rc=$?"
find $OUT/trace_p -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/precise_kernel_stats.csv; rm -rf $OUT/trace_p
# small images of concurrent calls merged into one tile batch
timeout 600 python $R/tools/small_image_probe.py 64 2>&1 | grep -v amdgpu.ids > $OUT/small_images.txt; cat $OUT/small_images.txt
