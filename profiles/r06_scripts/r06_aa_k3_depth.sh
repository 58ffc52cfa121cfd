# Round 6, session AA: K3 with TWO filter positions (VERDICT r5 task 4), same box, same process, launches interleaved round-robin
python -m pytest tests/test_gpu_engine.py -q -m gpu -k "k3_filter_depths or kernels_against_libpcre" > gpurun_out/r06_aa_pytest.txt 2>&1; tail -3 gpurun_out/r06_aa_pytest.txt
S=grab_amd/bin/gscan_sweep
$S --gib 16 --seg-mib 64 --iters 12 --variants -1 --bpc 0 --k3-depth 3,2,4 \
   --pattern 'foobardoesnotexist|Linus|555-1234' \
   --pattern 'error|warning|fatal|critical' \
   --pattern 'foo|bar' \
   --pattern 'qzxj|wvkq|jjxz|zqqv|xkcd|vvvv|qqqq|zzzz' \
   --pattern '[0-9]+\.[0-9]+' \
   --pattern '[a-z][0-9][A-Z][.,][;:]' > gpurun_out/r06_aa_k3_depth_sweep.txt 2>&1
cat gpurun_out/r06_aa_k3_depth_sweep.txt
