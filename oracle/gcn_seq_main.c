/*
 * gcn_seq_main.c — command line around the oracle (TEST INFRASTRUCTURE / CPU
 * baseline only).  Mirrors the reference's `gcn-seq <graph_name>` entry
 * (src/main.cpp:15-48) and implements the positional hyper-parameters its
 * usage string advertises but never reads (src/main.cpp:24-25).
 *
 *   gcn-seq graph_name [num_nodes input_dim hidden_dim output_dim dropout
 *                       learning_rate weight_decay epochs early_stopping]
 *
 * A value of "-" (or a non-positive num_nodes/input_dim/output_dim) keeps
 * what the parser found.  GCN_SEED plays the role of time(NULL) in
 * src/seq/rand.cpp:7; GCN_DATA_ROOT replaces the hard-coded "data/".
 */
#include "gcn_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

int main(int argc, char **argv) {
    setbuf(stdout, NULL);
    if (argc < 2) {
        printf("gcn-seq graph_name [num_nodes input_dim hidden_dim output_dim dropout "
               "learning_rate, weight_decay epochs early_stopping]\n");
        return EXIT_FAILURE;
    }
    or_params p = or_params_default();
    or_data d;
    const char *root = getenv("GCN_DATA_ROOT");
    if (!root) root = "data/";
    if (or_parse(root, argv[1], &p, &d) != 0) {
        fprintf(stderr, "Cannot read input: %s\n", argv[1]);
        return EXIT_FAILURE;
    }
#define ARG(i) (argc > (i) && strcmp(argv[i], "-") != 0)
    if (ARG(4)) p.hidden_dim = atoi(argv[4]);
    if (ARG(6)) p.dropout = (float)atof(argv[6]);
    if (ARG(7)) p.learning_rate = (float)atof(argv[7]);
    if (ARG(8)) p.weight_decay = (float)atof(argv[8]);
    if (ARG(9)) p.epochs = atoi(argv[9]);
    if (ARG(10)) p.early_stopping = atoi(argv[10]);
    const char *seed = getenv("GCN_SEED");
    long t = seed ? atol(seed) : (long)time(NULL);
    printf("RUNNING ON CPU\n");
    or_gcn *g = or_gcn_create(p, &d, t);
    or_gcn_run(g, NULL, 0);
    or_gcn_destroy(g);
    or_data_free(&d);
    return EXIT_SUCCESS;
}
