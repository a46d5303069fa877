// timer.h — the reference's 13 named accumulators (src/common/timer.h:5-20),
// re-done with device events: on the GPU the reference's chrono timers measure
// launch overhead only because nothing synchronises.  Here
// timer_start/stop record hipEvents on the context's stream; totals are
// resolved lazily (one synchronisation when a total is read).
#pragma once
#include <vector>
#include "gcnhip_driver.h"

typedef enum {
    TMR_TRAIN = 0, TMR_TEST, TMR_MATMUL_FW, TMR_MATMUL_BW, TMR_SPMATMUL_FW, TMR_SPMATMUL_BW,
    TMR_GRAPHSUM_FW, TMR_GRAPHSUM_BW, TMR_LOSS_FW, TMR_RELU_FW, TMR_RELU_BW, TMR_DROPOUT_FW, TMR_DROPOUT_BW,
    TMR_ADAM, TMR_COMM, TMR_GRAPHSUM_WIDE,     // additions: optimiser, collectives, GraphSum at the hidden width only
    __NUM_TMR
} timer_instance;

class DeviceTimers {
public:
    explicit DeviceTimers(gcnhip_ctx *ctx) : ctx_(ctx) {}
    ~DeviceTimers();
    bool enabled = false;
    void start(timer_instance t);
    void stop(timer_instance t);
    // seconds and number of start/stop pairs since the last reset (synchronises)
    double total(timer_instance t, long *count = nullptr);
    void reset();
private:
    struct Pair { void *a, *b; };
    gcnhip_ctx *ctx_;
    std::vector<Pair> pending_[__NUM_TMR];
    std::vector<void *> pool_;
    void *open_[__NUM_TMR] = {};
    double sum_[__NUM_TMR] = {};
    long cnt_[__NUM_TMR] = {};
    void *get_event();
    void resolve();
};
