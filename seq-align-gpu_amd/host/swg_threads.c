/*
 * swg_threads.c -- how many threads the host helpers' parallel loops use.
 *
 * omp_get_max_threads() reports the machine's hardware threads; inside a container the process
 * may own far fewer (a cpuset, or a CPU-time quota), and a parallel loop over 256 spinning threads
 * on 16 CPUs runs several times slower than on 16 threads.  The reference sizes its OpenMP loop by
 * omp_get_max_threads() alone (src/alignment_cmdline.c:341-347).
 */
#define _GNU_SOURCE
#include "../../include/swg_host.h"

#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static long quota_cpus(void)
{
    long cpus = 0;
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r"); /* cgroup v2: "<quota|max> <period>" */
    if (f) {
        char q[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long quota = atol(q);
            if (quota > 0) cpus = (quota + period - 1) / period;
        }
        fclose(f);
        return cpus;
    }
    long quota = 0, period = 0; /* cgroup v1 */
    f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
    if (f) {
        if (fscanf(f, "%ld", &quota) != 1) quota = 0;
        fclose(f);
    }
    f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
    if (f) {
        if (fscanf(f, "%ld", &period) != 1) period = 0;
        fclose(f);
    }
    if (quota > 0 && period > 0) cpus = (quota + period - 1) / period;
    return cpus;
}

int swg_host_threads(void)
{
    static int cached = 0;
    if (cached > 0) return cached;
    long n = 1;
#ifdef _OPENMP
    n = omp_get_max_threads(); /* honours OMP_NUM_THREADS */
#endif
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const long c = CPU_COUNT(&set);
        if (c > 0 && c < n) n = c;
    }
    const long q = quota_cpus();
    if (q > 0 && q < n) n = q;
    cached = n < 1 ? 1 : (int)n;
    return cached;
}
