#include "timer.h"
#include "hip_check.h"

DeviceTimers::~DeviceTimers() {
    for (int t = 0; t < __NUM_TMR; t++) {
        for (auto &p : pending_[t]) { gcnhip_event_destroy(p.a); gcnhip_event_destroy(p.b); }
        if (open_[t]) gcnhip_event_destroy(open_[t]);
    }
    for (void *e : pool_) gcnhip_event_destroy(e);
}

void *DeviceTimers::get_event() {
    if (!pool_.empty()) { void *e = pool_.back(); pool_.pop_back(); return e; }
    void *e = nullptr;
    GCNHIP_CHECK(gcnhip_event_create(&e));
    return e;
}

void DeviceTimers::start(timer_instance t) {
    if (!enabled) return;
    void *e = get_event();
    GCNHIP_CHECK(gcnhip_event_record(ctx_, e));
    if (open_[t]) pool_.push_back(open_[t]);
    open_[t] = e;
}

void DeviceTimers::stop(timer_instance t) {
    if (!enabled || !open_[t]) return;
    void *e = get_event();
    GCNHIP_CHECK(gcnhip_event_record(ctx_, e));
    pending_[t].push_back({open_[t], e});
    open_[t] = nullptr;
    if (pending_[t].size() >= 4096) resolve();
}

void DeviceTimers::resolve() {
    for (int t = 0; t < __NUM_TMR; t++) {
        for (auto &p : pending_[t]) {
            float ms = 0.f;
            GCNHIP_CHECK(gcnhip_event_elapsed_ms(p.a, p.b, &ms));
            sum_[t] += ms * 1e-3;
            cnt_[t]++;
            pool_.push_back(p.a);
            pool_.push_back(p.b);
        }
        pending_[t].clear();
    }
}

double DeviceTimers::total(timer_instance t, long *count) {
    resolve();
    if (count) *count = cnt_[t];
    return sum_[t];
}

void DeviceTimers::reset() {
    resolve();
    for (int t = 0; t < __NUM_TMR; t++) { sum_[t] = 0; cnt_[t] = 0; }
}
