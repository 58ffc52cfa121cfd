#!/bin/bash
# Round 4, session I: the whole GPU suite, then bench.py with default flags (the new blocks: checks in every kernel block,
# interleaved passes, e2e_cfg1, cfg4 / cfg5 at full size).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
( time timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > gpurun_out/i_pytest.txt 2>&1
tail -6 gpurun_out/i_pytest.txt
( time python bench.py ) > gpurun_out/i_bench.json 2> gpurun_out/i_bench.err
tail -3 gpurun_out/i_bench.err
python -c "
import json
d=json.loads(open('gpurun_out/i_bench.json').read().strip().splitlines()[0])
print(json.dumps({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if not isinstance(vv,(dict,list))}) for k,v in d.items()}, indent=1)[:6000])
"
