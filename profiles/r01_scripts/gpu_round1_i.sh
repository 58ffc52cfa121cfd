#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== startup cost: empty dir"; mkdir -p /dev/shm/empty_d; echo x > /dev/shm/empty_d/a; for i in 1 2 3; do ( time grab_amd/bin/grab -r foo /dev/shm/empty_d ) 2>&1 | grep real; done; ( time grab_amd/bin/grab -n 16 -r foo /dev/shm/empty_d ) 2>&1 | grep real
timeout 900 python scripts/e2e_cli.py --files 256 --file-kib 65536 --workers 1,2,4 --tag cfg2_16GiB 2>&1 | tail -1 | tee gpurun_out/i_e2e_cfg2.json
GSCAN_READERS=16 timeout 900 python scripts/e2e_cli.py --files 256 --file-kib 65536 --workers 1,4 --tag cfg2_16GiB_r16 2>&1 | tail -1 | tee gpurun_out/i_e2e_cfg2_r16.json
