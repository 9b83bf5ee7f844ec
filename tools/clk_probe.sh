rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -5
rocm-smi --showperflevel 2>/dev/null | grep -i perf | head -3
(python tools/quick_lap_bench.py 20000 > /tmp/b1.log 2>&1 &)
sleep 9
for i in 1 2 3; do rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -2; rocm-smi --showpower 2>/dev/null | grep -i "power" | head -2; sleep 1; done
wait; sleep 3; tail -1 /tmp/b1.log | cut -c1-200
echo "--- set perf level high"
rocm-smi --setperflevel high 2>&1 | tail -2
rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -2
python tools/quick_lap_bench.py 1000 20000 2>&1 | tail -2 | cut -c1-330
rocm-smi --setperflevel auto 2>&1 | tail -1
