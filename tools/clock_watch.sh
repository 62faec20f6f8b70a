# sample GPU clocks / power while the headline bench runs (read-only rocm-smi queries)
python bench.py --steps 5000 --no-cpu > gpurun_out/clock_bench.json 2>/dev/null &
BP=$!
sleep 16
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -6; echo ---; sleep 0.7; done
wait $BP
cut -c1-110 gpurun_out/clock_bench.json
echo "idle:"; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -6
