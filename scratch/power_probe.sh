# samples rocm-smi power / clocks while a workload loops: bash scratch/power_probe.sh "<label>" <command...>
label=$1; shift
( "$@" > /dev/null 2>&1 ) &
pid=$!
sleep 1.5
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|GPU use" | sed "s/^/$label  /" | tr '\n' ';'; echo
  sleep 0.4
done
kill $pid 2>/dev/null; wait $pid 2>/dev/null
