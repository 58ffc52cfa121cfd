#!/bin/bash
# End-to-end CLI rates (PCIe-inclusive) vs the reference binary, cfg2-like and cfg4-like corpora.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== cfg2-like: 256 x 64 MiB =="
timeout 900 python scripts/e2e_cli.py --files 256 --file-kib 65536 --workers 1,4,16,32 --tag cfg2_16GiB 2>&1 | tail -1 | tee gpurun_out/g_e2e_cfg2.json
echo "== cfg4-like: 16384 x 512 KiB in a 16x16 tree =="
timeout 900 python scripts/e2e_cli.py --files 16384 --file-kib 512 --fanout 16 --workers 1,4,16,32 --tag cfg4_8GiB 2>&1 | tail -1 | tee gpurun_out/g_e2e_cfg4.json
echo "== one 8 GiB file, default chunk (cfg5-like), serial =="
timeout 900 python scripts/e2e_cli.py --files 1 --file-kib 8388608 --workers 1 --ref-cores 2 --tag cfg5_8GiB 2>&1 | tail -1 | tee gpurun_out/g_e2e_cfg5.json
