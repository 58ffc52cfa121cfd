#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p /dev/shm/empty_d; echo x > /dev/shm/empty_d/a
echo "== startup marks, serial"; GSCAN_TIMING=1 GRAB_TIMING=1 grab_amd/bin/grab -r foo /dev/shm/empty_d 2>&1 | grep "+"; ( time grab_amd/bin/grab -r foo /dev/shm/empty_d ) 2>&1 | grep real
echo "== startup marks, -n 4"; GRAB_TIMING=1 grab_amd/bin/grab -n 4 -r foo /dev/shm/empty_d 2>&1 | grep "+"; ( time grab_amd/bin/grab -n 4 -r foo /dev/shm/empty_d ) 2>&1 | grep real
echo "== reference startup"; ( time oracle/_ref/grab_jit -r foo /dev/shm/empty_d ) 2>&1 | grep real
echo "== ldd"; ldd grab_amd/bin/grab | wc -l; ( time /bin/true ) 2>&1 | grep real
echo "== AMD_LOG? hip init only"; ( time grab_amd/bin/host_probe /dev/shm/p.bin ) 2>&1 | head -2
