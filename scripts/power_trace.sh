# Samples rocm-smi (power, clocks, temperature) every ~0.25 s while bench.py runs: is the iteration power-limited?
cd $GRAFT_REPO_ROOT
O=gpurun_out/power_trace.log
: > $O
rocm-smi --showpower --showclocks --showtemp --showuse > gpurun_out/power_idle.log 2>&1
( python bench.py --steps 2500 --warmup 20 --no-cpu-baseline --no-opt-in > gpurun_out/power_bench.log 2>&1 ) &
BP=$!
sleep 25      # import + warm-up
for i in $(seq 1 40); do
  echo "--- sample $i $(date +%s.%N)" >> $O
  rocm-smi --showpower --showclocks --showuse 2>&1 | grep -E "Power|sclk|mclk|fclk|GPU use" >> $O
  sleep 0.25
done
wait $BP
tail -c 400 gpurun_out/power_bench.log
