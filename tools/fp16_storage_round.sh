for m in "seed=42" "seed=43" "seed=44 hot=32.0 last_gain=0.15" "seed=45 chan_sigma=1.0 last_gain=0.12" "seed=45 chan_sigma=1.0 last_gain=0.2"; do timeout 280 python tools/fp16_storage_probe.py $m 2>&1 | grep -v amdgpu; done
timeout 600 python tools/check_real_model.py /tmp/rsr_models_probe/m_45_chan_sigma1_last_gain0.2 --no-bench 2>&1 | grep -v amdgpu | tail -12
timeout 600 python -m pytest tests/test_gpu_round2.py -x -q -k "real_model_harness" 2>&1 | tail -3
