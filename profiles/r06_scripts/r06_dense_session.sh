python -m pytest tests/test_gpu_resolve.py -q -m gpu > gpurun_out/r06_d_resolve.log 2>&1; tail -3 gpurun_out/r06_d_resolve.log
: > gpurun_out/r06_d_dense_e2e.jsonl
for p in '\b[A-Za-z_]\w*\s*\(' '(?<=\$)\d+' '\s\w{8,}\s' '\b[A-Z][a-z]+\b' '\([^()]*\)' '\b[a-z]{3,}\b' 'foo|bar|[0-9]{5}' '\d+\.\d+'; do
  python scripts/e2e_cli.py --files 128 --file-kib 65536 --pattern "$p" --flags "-O -l" --workers 8,16,32 --reps 2 --tag dense >> gpurun_out/r06_d_dense_e2e.jsonl 2>gpurun_out/r06_d_err.txt || tail -3 gpurun_out/r06_d_err.txt
done
python - <<'PY'
import json
for ln in open('gpurun_out/r06_d_dense_e2e.jsonl'):
    r = json.loads(ln)
    print("%-26s" % r['pattern'], " | ".join("-n %s: %.3fs %6.2f GB/s same %s" % (w, g['s'], g['GBps'], g['same_as_reference']) for w, g in r['grab'].items()), "| lines %s | ref %s cores %.3fs %.2f GB/s" % (r['grab']['8']['lines'], r['reference']['cores'], r['reference']['s'], r['reference']['GBps']))
PY
