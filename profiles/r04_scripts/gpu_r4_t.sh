#!/bin/bash
# Round 4, session T: the CLI-level GPU tests once more with EIGHT device indices presented by the engine
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
GSCAN_VIRTUAL_DEVICES=8 GSCAN_SYSFS_PCI=$PCI timeout 1500 python -m pytest tests/test_gpu_filegrep.py tests/test_gpu_geometry.py tests/test_integration.py -m gpu -q 2>&1 | tail -8 | tee gpurun_out/t_cli_suite_under_eight_virtual_devices.txt
