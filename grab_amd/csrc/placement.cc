// placement.cc -- see placement.h
#include "placement.h"

#include <cstring>

std::vector<WorkerPlace> grab_place_workers(int workers, int ndev, const std::vector<std::vector<int>> &dev_cpus, const cpu_set_t &allowed,
                                            const char *pin)
{
    std::vector<WorkerPlace> out;
    if (ndev < 1) ndev = 1;
    const bool by_cpu = pin && !strcmp(pin, "cpu"), none = pin && !strcmp(pin, "none");
    for (int i = 0; i < workers; i++) {
        WorkerPlace w;
        w.device = i % ndev; // round robin: with N >= #devices every device gets floor or ceil of N / #devices workers
        w.local = false;
        CPU_ZERO(&w.cpus);
        if (by_cpu) {
            if (i < CPU_SETSIZE) CPU_SET(i, &w.cpus);
            out.push_back(w);
            continue;
        }
        int got = 0;
        if (!none && (size_t)w.device < dev_cpus.size())
            for (int c : dev_cpus[(size_t)w.device])
                if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) {
                    CPU_SET(c, &w.cpus);
                    got++;
                }
        if (got) w.local = true;
        else w.cpus = allowed; // nothing known about the device's node, or none of it is ours: the process's own mask
        out.push_back(w);
    }
    return out;
}
