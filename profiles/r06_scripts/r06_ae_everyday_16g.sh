# Round 6, session AE: the 32 everyday patterns at 16 GiB (`-n 8 -r -O -l`): at 4 GiB (sessions U, W) a run is 0.2 s of which 0.17 are
# the process's fixed cost, on both sides
timeout 640 python scripts/everyday_patterns.py 16 > gpurun_out/r06_ae_everyday_patterns_16g.jsonl 2> gpurun_out/r06_ae_err.txt
wc -l gpurun_out/r06_ae_everyday_patterns_16g.jsonl; tail -3 gpurun_out/r06_ae_err.txt
