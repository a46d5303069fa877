// rand.h — the reference's host RNG (src/seq/rand.h:6-11, rand.cpp:6-28):
// xorshift128+ seeded from srand(seed)/rand().  gcn-hip keeps it so that the
// same seed gives the same Glorot weights as gcn-seq, and so that parity runs
// can replay the CPU path's dropout decisions (HOST_MASKS mode).
#pragma once
#include <cstdint>

#define MY_RAND_MAX 0x7fffffff

struct HostRng {
    uint64_t s[2];
    void seed_time(unsigned t);     // what init_rand_state() does with time(NULL) == t
    uint32_t next();
};
