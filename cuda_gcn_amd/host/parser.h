// parser.h — loader of the reference's three text files
// (src/common/parser.cpp:11-118): data/<name>.graph / .split / .svmlight ->
// GCNData with the self loop stored first in every adjacency row.  Same
// observable behaviour, including the dropped final line without '\n'
// (parser.cpp:27-28) and the derived dims (parser.cpp:45,90-91); a single-pass
// tokenizer instead of one istringstream per token.
// Also: a binary cache (<name>.gcnbin) so large graphs do not pay the text
// parse on every run.
#pragma once
#include <string>
#include "gcn.h"

class Parser {
public:
    // root defaults to $GCN_DATA_ROOT or "data/" (the reference hard-codes "data/", parser.cpp:12)
    Parser(GCNParams *gcnParams, GCNData *gcnData, std::string graph_name, std::string root = "");
    bool parse();
    static bool save_binary(const std::string &path, const GCNParams &p, const GCNData &d);
    bool from_cache() const { return from_cache_; }          // the last parse() was served by <name>.gcnbin
    std::string cache_path() const { return root + name + ".gcnbin"; }
    static bool load_binary(const std::string &path, GCNParams *p, GCNData *d);
private:
    bool from_cache_ = false;
    std::string root, name;
    GCNParams *gcnParams;
    GCNData *gcnData;
    bool parseGraph(const std::string &path);
    bool parseNode(const std::string &path);
    bool parseSplit(const std::string &path);
};
