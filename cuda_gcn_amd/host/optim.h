// optim.h — Adam (src/seq/optim.h:6-27, optim.cpp:6-37; CUDA twin
// cuda_module.cu:229-263).  One fused launch updates every variable.
#pragma once
#include <utility>
#include <vector>
#include "module.h"

struct AdamParams {
    float lr, beta1, beta2, eps, weight_decay;
    static AdamParams get_default();        // {0.001, 0.9, 0.999, 1e-8, 0}
};

class HipAdam {
    HipEnv *env = nullptr;
    AdamParams params;
    int step_count = 0;
    std::vector<gcnhip_adam_var> vars;
    std::vector<float *> state;             // m, v buffers (owned)
    float *d_step_sizes = nullptr;          // step size of step t at [t-1] (device table for graph replay)
    int table_len = 0;
public:
    float *d_sumsq = nullptr;               // sum(w0^2) after the latest update (gcn.cpp:98-105's L2 term)
    HipAdam() {}
    HipAdam(const HipAdam &) = delete;
    ~HipAdam();
    void init(HipEnv *env, std::vector<std::pair<HipVariable *, bool>> vars, AdamParams params, int max_steps);
    static float step_size(const AdamParams &p, int step_count);    // optim.cpp:26
    void step();                            // uses the device epoch word as step index when a table exists
    // graph replay: n steps happened on the device; true if all of them were inside the table
    bool can_replay(int n) const { return step_count + n <= table_len; }
    void note_replayed(int n) { step_count += n; }
    int steps() const { return step_count; }
};
