#!/bin/bash
# Round 5, session Q: bench.py (two processes in a row) once more, on another box: the range of the line that
# no GPU context; corpora from generator children) -- the line, and cfg5 in it.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
( time timeout 1500 python bench.py ) > gpurun_out/r5q_bench.json 2> gpurun_out/r5q_bench.err
tail -5 gpurun_out/r5q_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/r5q_bench.json').read().strip().splitlines()[-1])
print("lines of stdout:", len(open('gpurun_out/r5q_bench.json').read().strip().splitlines()))
print("value", r['value'], "check", r['check'], "roofline", {k: r['roofline'][k] for k in ('frac', 'kernel_ms', 'traffic')}, r.get('e2e_context'))
print({k: (v['frac'], v['frac_min'], v['frac_max'], v['check']) for k, v in r['kernels'].items()})
for k in ("e2e", "cpu_baseline", "e2e_cfg3", "e2e_cfg1", "e2e_cfg5", "e2e_cfg4"):
    v = r.get(k) or {}
    print(k, {x: v.get(x) for x in ("value", "wall_s", "startup_s", "exit_s", "scan_phase_GBps", "lines", "lines_ok", "back_to_back_wall_s", "vs_cpu_baseline", "same_as_reference", "at_16GiB", "at_8GiB", "corpus_write_s", "error")})
print("n8", json.dumps((r.get("n8_model") or {}).get("forecast_N8")))
PY
