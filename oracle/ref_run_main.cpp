/*
 * ref_run_main.cpp — our harness `main` around the REFERENCE's GCN class.
 * TEST INFRASTRUCTURE ONLY; built into oracle/_ref/ref_run (container only).
 *
 * Same flow as the reference's entry point (src/main.cpp:29-45) but it
 * applies the hyper-parameters that entry point ignores, through the public
 * GCN(GCNParams, GCNData*) constructor (src/seq/gcn.h:39).  time(NULL) is
 * wrapped at link time so GCN_SEED makes runs reproducible.
 *
 *   ref_run graph_name [- - hidden_dim - dropout lr weight_decay epochs early_stopping]
 * Run from a directory that has data/<name>.{graph,split,svmlight}.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <string>
#include "gcn.h"
#include "parser.h"
#include "timer.h"

extern "C" time_t __wrap_time(time_t *t) {
    const char *s = getenv("GCN_SEED");
    time_t v = (time_t)(s ? atol(s) : 0);
    if (t) *t = v;
    return v;
}

int main(int argc, char **argv) {
    setbuf(stdout, NULL);
    if (argc < 2) { fprintf(stderr, "usage: ref_run graph_name [...]\n"); return EXIT_FAILURE; }
    GCNParams params = GCNParams::get_default();
    GCNData data;
    Parser parser(&params, &data, std::string(argv[1]));
    if (!parser.parse()) { std::cerr << "Cannot read input: " << argv[1] << std::endl; return EXIT_FAILURE; }
#define ARG(i) (argc > (i) && strcmp(argv[i], "-") != 0)
    if (ARG(4)) params.hidden_dim = atoi(argv[4]);
    if (ARG(6)) params.dropout = (float)atof(argv[6]);
    if (ARG(7)) params.learning_rate = (float)atof(argv[7]);
    if (ARG(8)) params.weight_decay = (float)atof(argv[8]);
    if (ARG(9)) params.epochs = atoi(argv[9]);
    if (ARG(10)) params.early_stopping = atoi(argv[10]);
    std::cout << "RUNNING ON CPU" << std::endl;
    GCN gcn(params, &data);
    gcn.run();
    if (getenv("GCN_TIMERS"))
        for (int i = 0; i < __NUM_TMR; i++) printf("timer[%d]=%.6f\n", i, timer_total((timer_instance)i));
    return EXIT_SUCCESS;
}
