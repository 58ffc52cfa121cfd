#!/bin/bash
# Round 5, session O: the CLI-level GPU tests at the refactored engine (streams per device, second stream on demand, read-ahead, 3 readers per index) with EIGHT device indices presented by the engine
# (GSCAN_VIRTUAL_DEVICES=8, a faked two-socket sysfs tree): every `grab` of the suite then deals its workers / the windows of
# its multi-window files over eight indices, each with its own reader pool, blocks and streams.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
PCI=/tmp/fake_pci
python - <<'PY'
import os
cpus = sorted(os.sched_getaffinity(0)); half = len(cpus) // 2
for v in range(8):
    d = "/tmp/fake_pci/0000:%02x:00.0" % (0x0c + 0x10 * v)
    os.makedirs(d, exist_ok=True)
    lst = cpus[:half] if v < 4 else cpus[half:]
    open(d + "/local_cpulist", "w").write("%d-%d\n" % (lst[0], lst[-1]))
PY
GSCAN_VIRTUAL_DEVICES=8 GSCAN_SYSFS_PCI=$PCI timeout 1500 python -m pytest tests/test_gpu_filegrep.py tests/test_gpu_geometry.py tests/test_integration.py -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r5o_cli_suite_under_eight_virtual_devices.txt
# ... and once with every device's second copy stream made at once (GSCAN_SECOND_STREAM_MIB=0: the two-stream path in every test, also the small ones)
GSCAN_SECOND_STREAM_MIB=0 timeout 1500 python -m pytest tests/test_gpu_filegrep.py tests/test_gpu_engine.py -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r5o_suite_with_the_second_stream_from_the_start.txt
