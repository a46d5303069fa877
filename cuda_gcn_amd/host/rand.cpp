#include "rand.h"
#include <cstdlib>

void HostRng::seed_time(unsigned t) {
    srand(t);
    int x = 0, y = 0;
    while (x == 0 || y == 0) {      // both words must be non-zero (rand.cpp:9-12)
        x = rand();
        y = rand();
    }
    s[0] = (uint64_t)x;
    s[1] = (uint64_t)y;
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
