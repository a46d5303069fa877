// variable.h — Variable (host, src/seq/variable.h:4-12) and HipVariable, the
// device-resident twin (reference: CUDAVariable, src/cuda/cuda_variable.cuh:7-20).
#pragma once
#include <cstddef>
#include <vector>
#include "gcnhip_driver.h"
#include "partition.h"
#include "rand.h"

struct Variable {
    std::vector<float> data, grad;
    Variable(int size, bool requires_grad = true) : data(size), grad(requires_grad ? size : 0) {}
    void glorot(int in_size, int out_size, HostRng &rng);     // variable.cpp:11-18
};

// A row-major [rows x cols] f32 matrix on the GPU with leading dimension ld
// (multiple of 4 floats: every row 16-byte aligned for the vector kernels).
// For variables that GraphSum gathers from other ranks, `full` is the base of
// the gathered table ([plan.table_rows x ld], partition.h) and data/grad point
// at this rank's block inside it (so the exchange completes it in place).
struct HipVariable {
    gcnhip_ctx *ctx = nullptr;
    float *data = nullptr, *grad = nullptr;          // this rank's rows
    float *full = nullptr, *full_grad = nullptr;     // gather buffers (== data/grad when world == 1)
    int rows = 0, cols = 0, ld = 0;
    bool requires_grad = false;
    size_t full_elems = 0;

    HipVariable() {}
    HipVariable(const HipVariable &) = delete;
    HipVariable &operator=(const HipVariable &) = delete;
    ~HipVariable();
    // gather_*: data / grad is a gathered table of plan->table_rows rows with this rank's block at plan->own_offset
    // (only when the plan spans more than one rank)
    void alloc(gcnhip_ctx *ctx, int rows, int cols, bool requires_grad, bool gather_data = false,
               bool gather_grad = false, const ExchangePlan *plan = nullptr);
    // replicated input of a GraphSum: every rank computes all `total_rows` rows itself (no all-gather);
    // data points at this rank's rows inside the full matrix
    void alloc_replicated(gcnhip_ctx *ctx, int total_rows, int local_rows, int row_start, int cols, bool requires_grad);
    bool replicated = false;
    size_t elems() const { return (size_t)rows * ld; }
    void zero();
    void zero_grad();
    void upload(const float *host_rowmajor);                 // [rows x cols] contiguous
    void download(float *host_rowmajor, bool want_grad = false) const;
};
