#include "rand.h"
#include <cstdlib>
#include <cstring>
#include <stdint.h>

// The reference seeds with srand(t) / rand() (rand.cpp:6-15): libc's ONE generator per process.  Several models are built
// at the same time here (gcn-hip runs one host thread per GPU; the tests run up to eight logical ranks as threads), and with
// the process-wide generator two threads inside this function at once drew each other's numbers: a rank started from other
// weights than its peers (found in round 5: rank traces differing in the L2 term from epoch 0 on, a few runs in a hundred).
// glibc's reentrant interface on a private 128-byte state is the same generator (TYPE_3, the default of srand / rand):
// the same two numbers for the same t, whoever else is drawing.
#ifndef __GLIBC__
#include <mutex>
static std::mutex libc_rand_mutex;       // other libcs: srand / rand under a lock (their numbers are that libc's, as the reference's would be)
#endif
void HostRng::seed_time(unsigned t) {
#ifndef __GLIBC__
    std::lock_guard<std::mutex> hold(libc_rand_mutex);
    srand(t);
    int x = 0, y = 0;
    while (x == 0 || y == 0) { x = rand(); y = rand(); }
    s[0] = (uint64_t)x;
    s[1] = (uint64_t)y;
#else
    struct random_data rd;
    char state[128];
    memset(&rd, 0, sizeof rd);
    memset(state, 0, sizeof state);
    initstate_r(t, state, sizeof state, &rd);
    int32_t x = 0, y = 0;
    while (x == 0 || y == 0) {      // both words must be non-zero (rand.cpp:9-12)
        random_r(&rd, &x);
        random_r(&rd, &y);
    }
    s[0] = (uint64_t)x;
    s[1] = (uint64_t)y;
#endif
}

uint32_t HostRng::next() {
    uint64_t t = s[0];
    const uint64_t u = s[1];
    s[0] = u;
    t ^= t << 23;
    t ^= t >> 17;
    t ^= u ^ (u >> 26);
    s[1] = t;
    return (uint32_t)((t + u) & 0x7fffffff);
}
