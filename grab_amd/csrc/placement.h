// placement.h -- where the threads of `grab -n N` run on a multi-GPU node: which device each worker drives and which CPUs
// it may use.  Pure functions of (workers, devices, each device's local CPU list, the process's CPU mask), so that the
// 8-GPU layout can be checked on a box without a GPU (tests/test_host_cpu.py).
//
// The reference pins thread i to CPU i (/root/reference/src/main.cc:200-215) and has no devices; here worker i drives
// device i mod #devices (the unit of work is the file, the queue is shared) and runs on the CPUs of that
// device's NUMA node -- its batch reads go into pinned blocks it touches first and its share of the report walks the page
// cache, so it belongs next to the PCIe root of its GPU.
#pragma once
#include <sched.h>

#include <vector>

struct WorkerPlace {
    int device;      // HIP device of the worker's FileGrep
    cpu_set_t cpus;  // the CPUs the worker thread is bound to
    bool local;      // cpus is the device's NUMA-local list (cut to the process's mask), not a fall-back
};

// pin: nullptr / "" = NUMA-local (the default); "cpu" = the reference's rule, worker i on CPU i; "none" = the process's mask.
// dev_cpus[d] = the CPUs local to device d (empty: unknown).
std::vector<WorkerPlace> grab_place_workers(int workers, int ndev, const std::vector<std::vector<int>> &dev_cpus, const cpu_set_t &allowed,
                                            const char *pin);
