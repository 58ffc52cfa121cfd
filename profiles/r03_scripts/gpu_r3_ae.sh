#!/bin/bash
# Round 3, session AE: does the GPU clock or power state differ between the kernels?  sysfs sampled every 50 ms while the sweep
# runs K1, the lane form without records, and the identifier scan, 200 launches each over 16 GiB.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
D=$(ls -d /sys/class/drm/card*/device | head -1)
H=$(ls -d $D/hwmon/hwmon* | head -1)
{
echo "device $D hwmon $H"; ls $H | tr '\n' ' '; echo
cat $D/pp_dpm_sclk 2>/dev/null | tr '\n' ' '; echo
for P in 'foobardoesnotexist' '[0-9]{16}' '[A-Za-z_][A-Za-z0-9_]{15,}'; do
  echo "## $P"
  ( while true; do echo "$(date +%s.%N | cut -c1-14) sclk $(cat $H/freq1_input 2>/dev/null) mclk $(cat $H/freq2_input 2>/dev/null) power $(cat $H/power1_average 2>/dev/null || cat $H/power1_input 2>/dev/null) temp $(cat $H/temp1_input 2>/dev/null) $(cat $H/temp2_input 2>/dev/null) $(cat $H/temp3_input 2>/dev/null)"; sleep 0.05; done ) > gpurun_out/ae_samples.tmp &
  SP=$!
  timeout 300 $SW --gib 16 --iters 200 --variants 38 --bpc 0 --pattern "$P" 2>&1 | grep -E "^variant"
  kill $SP; wait $SP 2>/dev/null
  awk 'NR%8==1' gpurun_out/ae_samples.tmp | tail -25
done
} 2>&1 | tee gpurun_out/ae_clocks.txt
rm -f gpurun_out/ae_samples.tmp
