# Sustained clock / power while a dense MFMA layer loops: random operands vs zero operands (tools/bench_conv.py --zeros).
# usage (GPU box): bash tools/clock_probe.sh > gpurun_out/clock_probe.txt
for mode in "" "--zeros"; do
  echo "== conv4_2 fwd+dgrad loop, operands: ${mode:-random}"
  python tools/bench_conv.py --layers conv4_2 --what fwd,dgrad --iters 40000 $mode > /tmp/bc.log 2>&1 &
  pid=$!
  sleep 6
  for i in 1 2 3 4; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | tr -s ' ' | head -4
    sleep 1.5
  done
  wait $pid
  tail -2 /tmp/bc.log
done
