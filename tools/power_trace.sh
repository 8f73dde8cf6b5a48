#!/bin/bash
# Samples GPU clock and socket power (rocm-smi, read-only) while bench.py runs: is the training step clock- / power-limited?
#   tools/power_trace.sh [bench.py arguments] > log
cd "$(dirname "$0")/.."
python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-kernel-timing "$@" > /tmp/power_bench.json 2>/dev/null &
pid=$!
sleep 25                      # import + warm-up
for i in $(seq 1 12); do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | tr '\n' ' '
    echo
    sleep 0.3
done
wait $pid
cat /tmp/power_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], 'img/s', d['ms_per_step'], 'ms/step')"
echo "--- idle ---"
sleep 3
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
