// options.cpp — the one place where the host library reads HIPGCN_* environment variables.
// main.cpp and capi.cpp call HipGCNOptions::from_environment() once per model; HipGCN works from the struct alone.
#include <cstdlib>
#include <cstring>
#include "gcn.h"

HipGCNOptions HipGCNOptions::from_environment(HipGCNOptions o) {
    static const struct { const char *name; int bit; } flag_vars[] = {
        {"HIPGCN_EDGE_COEF", HIPGCN_EDGE_COEF},
        {"HIPGCN_PACKED_DH1", HIPGCN_PACKED_DH1},
        {"HIPGCN_BWD_PIPELINE", HIPGCN_BWD_PIPELINE},
        {"HIPGCN_STRUCTURE_PARTITION", HIPGCN_STRUCTURE_PARTITION},
        {"HIPGCN_ID_PARTITION", HIPGCN_ID_PARTITION},
        {"HIPGCN_NO_LABEL_HINT", HIPGCN_NO_LABEL_HINT},
        {"HIPGCN_ALL_ROWS", HIPGCN_ALL_ROWS},
        {"HIPGCN_MASKED_BWD", HIPGCN_MASKED_BWD},
        {"HIPGCN_OVERLAP_EXCHANGE", HIPGCN_OVERLAP_EXCHANGE},
        {"HIPGCN_NO_AGG_FIRST_EVAL", HIPGCN_NO_AGG_FIRST_EVAL},
        {"HIPGCN_EVAL_LANE", HIPGCN_EVAL_LANE},
        {"HIPGCN_SYNC_EPOCHS", HIPGCN_SYNC_EPOCHS},
    };
    for (const auto &f : flag_vars)
        if (getenv(f.name)) o.flags |= f.bit;
    if (getenv("HIPGCN_VERBOSE")) o.verbose = true;
    if (const char *e = getenv("HIPGCN_EXCHANGE"))
        o.exchange = !strcmp(e, "halo") ? 2 : (!strcmp(e, "allgather") ? 1 : (!strcmp(e, "auto") ? 0 : o.exchange));
    if (getenv("HIPGCN_NO_STRUCTURE_GROUPS")) o.structure_groups = false;
    if (getenv("HIPGCN_NO_MASK_BITS")) o.mask_bits = false;
    if (getenv("HIPGCN_NO_LOSS_EPILOGUE")) o.loss_epilogue = false;
    if (getenv("HIPGCN_NO_EVAL_FUSION")) o.eval_fusion = false;
    if (getenv("HIPGCN_NO_SLICE_TUNING")) o.slice_tuning = false;
    if (getenv("HIPGCN_FOLD_TRAINING")) o.fold_training = true;
    if (getenv("HIPGCN_RECORD_LAUNCH")) o.loss_records_metrics = false;
    if (const char *e = getenv("HIPGCN_BWD_CHUNKS")) o.bwd_chunks = atoi(e);
    if (const char *e = getenv("HIPGCN_READBACK_STREAM")) o.readback_stream = atoi(e) != 0;
    if (const char *e = getenv("HIPGCN_READBACK_GROUP")) o.readback_group = atoi(e);
    if (const char *e = getenv("HIPGCN_SCHEDULE")) {
        if (!strcmp(e, "degree")) o.schedule = 0;
        else if (!strcmp(e, "label")) o.schedule = 1;
        else if (!strncmp(e, "dealt", 5)) { o.schedule = 2; if (e[5] == '-' && atoi(e + 6) > 0) o.schedule_groups = atoi(e + 6); }
        else if (!strcmp(e, "structure")) o.schedule = 3;
    }
    if (const char *e = getenv("HIPGCN_GEMM")) o.gemm = !strcmp(e, "bf16x3") ? 1 : (!strcmp(e, "f32") ? 0 : o.gemm);
    return o;
}
