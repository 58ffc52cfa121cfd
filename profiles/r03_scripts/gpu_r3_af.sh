#!/bin/bash
# Round 3, session AF: clock / power over a 5 s run of each kernel, and the rate as a function of how long the run is.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
D=$(ls -d /sys/class/drm/card*/device | head -1)
H=$(ls -d $D/hwmon/hwmon* | head -1)
{
echo "power cap $(cat $H/power1_cap) default $(cat $H/power1_cap_default) max $(cat $H/power1_cap_max)"
for P in 'foobardoesnotexist' '[0-9]{16}' '[A-Za-z_][A-Za-z0-9_]{15,}'; do
  for IT in 8 100 1500; do
    echo "## $P, $IT launches"
    ( while true; do echo "$(date +%s.%N | cut -c1-14) sclk $(( $(cat $H/freq1_input) / 1000000 )) MHz power $(( $(cat $H/power1_input) / 1000000 )) W temp $(( $(cat $H/temp2_input) / 1000 )) $(( $(cat $H/temp3_input) / 1000 ))"; sleep 0.1; done ) > gpurun_out/af_samples.tmp &
    SP=$!
    timeout 300 $SW --gib 16 --iters $IT --variants 38 --bpc 0 --pattern "$P" 2>&1 | grep -E "^variant"
    kill $SP; wait $SP 2>/dev/null
    [ $IT = 1500 ] && awk 'NR%4==1' gpurun_out/af_samples.tmp | tail -16
  done
done
} 2>&1 | tee gpurun_out/af_clocks_long.txt
rm -f gpurun_out/af_samples.tmp
