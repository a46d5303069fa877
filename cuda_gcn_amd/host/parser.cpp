#include "parser.h"
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <algorithm>

namespace {
// whole file in memory; lines are [begin, end) without the '\n'.  A trailing
// piece with no '\n' is not a line (getline + eof() test of parser.cpp:27-28).
struct FileLines {
    std::string buf;
    bool ok = false;
    explicit FileLines(const std::string &path) {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) return;
        fseek(f, 0, SEEK_END);
        long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        buf.resize((size_t)n);
        if (n && fread(&buf[0], 1, (size_t)n, f) != (size_t)n) { fclose(f); return; }
        fclose(f);
        ok = true;
    }
    template <class Fn>
    void for_each(Fn fn) {
        size_t pos = 0;
        while (pos < buf.size()) {
            const void *nl = memchr(buf.data() + pos, '\n', buf.size() - pos);
            if (!nl) break;
            const size_t end = (const char *)nl - buf.data();
            buf[end] = 0;
            fn(buf.data() + pos);
            pos = end + 1;
        }
    }
};

// istream >> int
inline bool scan_int(const char *&s, int &out) {
    while (*s && isspace((unsigned char)*s)) s++;
    char *end;
    long v = strtol(s, &end, 10);
    if (end == s) return false;
    out = (int)v;
    s = end;
    return true;
}
}  // namespace

Parser::Parser(GCNParams *p, GCNData *d, std::string graph_name, std::string r) : root(r), name(graph_name), gcnParams(p), gcnData(d) {
    if (root.empty()) {
        const char *e = getenv("GCN_DATA_ROOT");
        root = e ? e : "data/";
    }
    if (!root.empty() && root.back() != '/') root += '/';
}

bool Parser::parseGraph(const std::string &path) {          // parser.cpp:20-46
    FileLines f(path);
    if (!f.ok) return false;
    auto &g = gcnData->graph;
    g.indptr.push_back(0);
    int node = 0;
    f.for_each([&](const char *line) {
        g.indices.push_back(node);                           // implicit self connection, first in the row
        g.indptr.push_back(g.indptr.back() + 1);
        node++;
        int nb;
        while (scan_int(line, nb)) { g.indices.push_back(nb); g.indptr.back() += 1; }
    });
    gcnParams->num_nodes = node;
    return true;
}

bool Parser::parseNode(const std::string &path) {           // parser.cpp:52-92
    FileLines f(path);
    if (!f.ok) return false;
    auto &idx = gcnData->feature_index;
    auto &val = gcnData->feature_value;
    auto &labels = gcnData->label;
    idx.indptr.push_back(0);
    int max_idx = 0, max_label = 0;
    f.for_each([&](const char *s) {
        idx.indptr.push_back(idx.indptr.back());
        int label = -1;
        const char *q = s;
        while (*q && isspace((unsigned char)*q)) q++;
        bool ok = false;
        if (*q) { ok = scan_int(s, label); if (!ok) label = 0; }   // C++11 extraction failure stores 0
        labels.push_back(label);
        if (!ok) return;
        max_label = std::max(max_label, label);
        for (;;) {
            while (*s && isspace((unsigned char)*s)) s++;
            if (!*s) break;
            char *end;
            const long k = strtol(s, &end, 10);              // "k:v"
            const char *t = end;
            float v = 0;
            if (*t && !isspace((unsigned char)*t)) { t++; v = strtof(t, &end); t = end; }
            while (*t && !isspace((unsigned char)*t)) t++;
            s = t;
            val.push_back(v);
            idx.indices.push_back((int)k);
            idx.indptr.back() += 1;
            max_idx = std::max(max_idx, (int)k);
        }
    });
    gcnParams->input_dim = max_idx + 1;
    gcnParams->output_dim = max_label + 1;
    return true;
}

bool Parser::parseSplit(const std::string &path) {          // parser.cpp:94-103
    FileLines f(path);
    if (!f.ok) return false;
    f.for_each([&](const char *line) { gcnData->split.push_back((int)strtol(line, nullptr, 10)); });
    return true;
}

bool Parser::parse() {
    // a binary cache, when present, wins
    from_cache_ = false;
    if (load_binary(root + name + ".gcnbin", gcnParams, gcnData)) {
        std::cout << "Loaded binary cache." << std::endl;
        from_cache_ = true;
        return true;
    }
    *gcnData = GCNData();      // a rejected cache leaves nothing behind: the text parsers append
    // all three must open before anything is parsed (parser.cpp:48-50,111)
    for (const char *ext : {".graph", ".split", ".svmlight"}) {
        FILE *f = fopen((root + name + ext).c_str(), "rb");
        if (!f) return false;
        fclose(f);
    }
    if (!parseGraph(root + name + ".graph")) return false;
    std::cout << "Parse Graph Succeeded." << std::endl;
    if (!parseNode(root + name + ".svmlight")) return false;
    std::cout << "Parse Node Succeeded." << std::endl;
    if (!parseSplit(root + name + ".split")) return false;
    std::cout << "Parse Split Succeeded." << std::endl;
    return true;
}

// ---- binary cache: header + raw arrays --------------------------------------
namespace {
const char MAGIC[8] = {'G', 'C', 'N', 'B', 'I', 'N', '0', '1'};
template <class T>
bool wr(FILE *f, const std::vector<T> &v) {
    uint64_t n = v.size();
    return fwrite(&n, sizeof n, 1, f) == 1 && (n == 0 || fwrite(v.data(), sizeof(T), n, f) == n);
}
// `left` = bytes of the file not yet consumed: a length prefix larger than that is a truncated or
// corrupt cache, rejected before anything is allocated
template <class T>
bool rd(FILE *f, std::vector<T> &v, uint64_t &left) {
    uint64_t n;
    if (left < sizeof n || fread(&n, sizeof n, 1, f) != 1) return false;
    left -= sizeof n;
    if (n > left / sizeof(T)) return false;
    v.resize(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) return false;
    left -= n * sizeof(T);
    return true;
}
}  // namespace

bool Parser::save_binary(const std::string &path, const GCNParams &p, const GCNData &d) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    int dims[3] = {p.num_nodes, p.input_dim, p.output_dim};
    bool ok = fwrite(MAGIC, 8, 1, f) == 1 && fwrite(dims, sizeof dims, 1, f) == 1 &&
              wr(f, d.graph.indptr) && wr(f, d.graph.indices) && wr(f, d.feature_index.indptr) &&
              wr(f, d.feature_index.indices) && wr(f, d.feature_value) && wr(f, d.split) && wr(f, d.label);
    fclose(f);
    return ok;
}

bool Parser::load_binary(const std::string &path, GCNParams *p, GCNData *d) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[8];
    int dims[3];
    uint64_t left = 0;
    if (fseek(f, 0, SEEK_END) == 0) { const long sz = ftell(f); if (sz > 0) left = (uint64_t)sz; }
    rewind(f);
    // read into a scratch object and validate it against the header: the caller's GCNData is touched only by a
    // complete, self-consistent cache (a stale or cut-off file must leave it empty for the text parsers)
    GCNData t;
    bool ok = left >= 8 + sizeof dims && fread(magic, 8, 1, f) == 1 && memcmp(magic, MAGIC, 8) == 0 &&
              fread(dims, sizeof dims, 1, f) == 1;
    if (ok) left -= 8 + sizeof dims;
    ok = ok && rd(f, t.graph.indptr, left) && rd(f, t.graph.indices, left) && rd(f, t.feature_index.indptr, left) &&
         rd(f, t.feature_index.indices, left) && rd(f, t.feature_value, left) && rd(f, t.split, left) && rd(f, t.label, left);
    fclose(f);
    if (!ok) return false;
    const int64_t N = dims[0];
    if (N < 0 || dims[1] < 0 || dims[2] < 0) return false;
    if ((int64_t)t.graph.indptr.size() != N + 1 || (int64_t)t.feature_index.indptr.size() != N + 1 ||
        (int64_t)t.split.size() != N || (int64_t)t.label.size() != N)
        return false;
    if (t.graph.indptr.front() != 0 || (size_t)t.graph.indptr.back() != t.graph.indices.size()) return false;
    if (t.feature_index.indptr.front() != 0 || (size_t)t.feature_index.indptr.back() != t.feature_value.size()) return false;
    // indices may be omitted for a dense X (every row = columns 0..input_dim-1)
    if (t.feature_index.indices.size() != t.feature_value.size() &&
        !(t.feature_index.indices.empty() && (int64_t)t.feature_value.size() == N * dims[1]))
        return false;
    for (int64_t i = 0; i < N; i++)
        if (t.graph.indptr[i + 1] < t.graph.indptr[i] || t.feature_index.indptr[i + 1] < t.feature_index.indptr[i]) return false;
    for (int j : t.graph.indices)
        if (j < 0 || j >= N) return false;
    for (int k : t.feature_index.indices)
        if (k < 0 || k >= dims[1]) return false;
    *d = std::move(t);
    p->num_nodes = dims[0]; p->input_dim = dims[1]; p->output_dim = dims[2];
    return true;
}
