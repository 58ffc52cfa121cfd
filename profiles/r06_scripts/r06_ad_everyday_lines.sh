# Round 6, session AD: the LINE-PRINTING mode (grab's default output) for patterns the device resolves -- the host formats the lines
# from the mapped window -- against the reference on all cores, 4 GiB, output compared by digest
timeout 400 python scripts/everyday_patterns.py 4 "" 7,12,16,18,21,23,31 > gpurun_out/r06_ad_everyday_lines.jsonl 2> gpurun_out/r06_ad_err.txt
cut -c1-400 gpurun_out/r06_ad_everyday_lines.jsonl; tail -3 gpurun_out/r06_ad_err.txt
