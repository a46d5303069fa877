#include "optim.h"
#include <cmath>
#include "hip_check.h"

AdamParams AdamParams::get_default() { return {0.001, 0.9, 0.999, 1e-8, 0.0}; }

float HipAdam::step_size(const AdamParams &p, int t) {
    return p.lr * sqrtf(1 - powf(p.beta2, t)) / (1 - powf(p.beta1, t));
}

void HipAdam::init(HipEnv *e, std::vector<std::pair<HipVariable *, bool>> vs, AdamParams p, int max_steps) {
    env = e; params = p; step_count = 0;
    for (auto &pr : vs) {
        HipVariable *v = pr.first;
        gcnhip_adam_var av;
        av.w = v->data; av.g = v->grad; av.n = (int64_t)v->elems(); av.decay = pr.second ? 1 : 0;
        void *m, *vv;
        GCNHIP_CHECK(gcnhip_malloc(env->ctx, &m, v->elems() * sizeof(float)));
        GCNHIP_CHECK(gcnhip_malloc(env->ctx, &vv, v->elems() * sizeof(float)));
        GCNHIP_CHECK(gcnhip_memset_async(env->ctx, m, 0, v->elems() * sizeof(float)));
        GCNHIP_CHECK(gcnhip_memset_async(env->ctx, vv, 0, v->elems() * sizeof(float)));
        av.m = (float *)m; av.v = (float *)vv;
        state.push_back(av.m); state.push_back(av.v);
        vars.push_back(av);
    }
    void *q;
    GCNHIP_CHECK(gcnhip_malloc(env->ctx, &q, sizeof(float)));
    d_sumsq = (float *)q;
    // sum(w0^2) of the initial weights, for the loss reported before the first update
    GCNHIP_CHECK(gcnhip_sumsq(env->ctx, vars[0].w, vars[0].n, d_sumsq));
    table_len = max_steps > 0 ? max_steps : 1;
    std::vector<float> tab(table_len);
    for (int t = 1; t <= table_len; t++) tab[t - 1] = step_size(params, t);   // host libm, as the CPU path
    GCNHIP_CHECK(gcnhip_malloc(env->ctx, &q, table_len * sizeof(float)));
    d_step_sizes = (float *)q;
    GCNHIP_CHECK(gcnhip_h2d(env->ctx, d_step_sizes, tab.data(), table_len * sizeof(float)));
}

HipAdam::~HipAdam() {
    if (!env) return;
    for (float *p : state) gcnhip_free(env->ctx, p);
    gcnhip_free(env->ctx, d_sumsq);
    gcnhip_free(env->ctx, d_step_sizes);
}

void HipAdam::step() {
    step_count++;
    env->timers->start(TMR_ADAM);
    // inside the table the step size is read on the device at index *d_epoch (== step_count - 1),
    // which keeps a captured epoch replayable; past it the host value is passed
    const bool use_table = step_count <= table_len;
    if (env->d_epoch_done)      // the launch also advances the epoch word: the next training pass needs no counter launch
        GCNHIP_CHECK(gcnhip_adam_step_advance(env->ctx, vars.data(), (int)vars.size(), step_size(params, step_count),
                                              use_table ? d_step_sizes : nullptr, use_table ? env->d_epoch : nullptr,
                                              params.beta1, params.beta2, params.eps, params.weight_decay, d_sumsq,
                                              env->d_epoch, env->d_epoch_done));
    else
        GCNHIP_CHECK(gcnhip_adam_step(env->ctx, vars.data(), (int)vars.size(), step_size(params, step_count),
                                      use_table ? d_step_sizes : nullptr, use_table ? env->d_epoch : nullptr,
                                      params.beta1, params.beta2, params.eps, params.weight_decay, d_sumsq));
    env->timers->stop(TMR_ADAM);
}
