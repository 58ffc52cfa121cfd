# Round 6, session AB: session Z's `pytest -m gpu` stopped after 282 tests and ran into gpurun's limit.  The whole suite again with
# a per-test timeout and a watchdog that writes down what a child process older than 75 s is doing; then the K3 filter-depth sweep.
timeout 900 python -m pytest tests -q -m gpu --timeout 150 -p no:cacheprovider > gpurun_out/r06_ab_pytest_gpu.txt 2>&1 &
PYT=$!
python scripts/hang_watch.py --root $PYT --age 75 --out gpurun_out/r06_ab_hang.txt &
wait $PYT; echo "pytest rc $?"
tail -5 gpurun_out/r06_ab_pytest_gpu.txt
bash profiles/r06_scripts/r06_aa_k3_depth.sh 2>&1 | tail -40
