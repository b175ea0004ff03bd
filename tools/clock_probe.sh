#!/bin/bash
# Sample GPU clocks / power while the bench runs (GPU box):  tools/clock_probe.sh
cd $GRAFT_REPO_ROOT
python bench.py --steps 100 --warmup 2 --no-cpu-baseline --no-profile > /tmp/bench_bg.json 2>/tmp/bench_bg.err &
BP=$!
for i in $(seq 1 60); do
  L=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -E 's/.*(sclk clock level: [^ ]+ \(([0-9]+)Mhz\)).*/sclk=\2/; s/.*Power \(W\): ([0-9.]+)/P=\1/' | tr '\n' ' ')
  echo "t=$i $L"
  kill -0 $BP 2>/dev/null || break
  sleep 0.4
done
wait $BP
cut -c1-160 /tmp/bench_bg.json
