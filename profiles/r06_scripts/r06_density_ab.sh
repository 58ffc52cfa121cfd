# Round 6, session E: patterns whose START windows list a large part of the text (\w+(?=\() : three bytes in four) --
# the round-5 path (K3 with the VM in its cold path, hit windows, host matcher on every kept hit: GSCAN_NO_RESOLVE=1) against
# the resolve pass over every start-window hit (GSCAN_RESOLVE_MAX_DENSITY=2), 8 GiB, -n 8, -O -l, output to a file.
O=gpurun_out/r06_e_density_ab.jsonl; : > $O
for p in '\w+(?=\()' '(\w)\1{3,}x|foobardoes(?=not)' '[a-z]+\([a-z0-9, ]*\);' 'a+b+c' '[a-z]+@[a-z]+\.com' '.*foobar'; do
  for mode in old new; do
    if [ $mode = old ]; then export GSCAN_NO_RESOLVE=1; unset GSCAN_RESOLVE_MAX_DENSITY; else unset GSCAN_NO_RESOLVE; export GSCAN_RESOLVE_MAX_DENSITY=2; fi
    python scripts/e2e_cli.py --files 128 --file-kib 65536 --pattern "$p" --flags "-O -l" --workers 8 --reps 2 --tag $mode >> $O 2>gpurun_out/r06_e_err.txt || tail -3 gpurun_out/r06_e_err.txt
  done
done
python - <<'PY'
import json
for ln in open('gpurun_out/r06_e_density_ab.jsonl'):
    r = json.loads(ln); g = r['grab']['8']
    print("%-36s %-4s grab %.3fs %6.2f GB/s lines %s same %s | ref %.3fs" % (r['pattern'], r['tag'], g['s'] or -1, g['GBps'] or -1, g['lines'], g['same_as_reference'], r['reference']['s']))
PY
