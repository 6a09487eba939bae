// fast_iwec — build helper, NOT part of the product or of the oracle.
//
// Drop-in replacement for ONE stage of the reference's lexer compile pipeline,
// `fa_fsm2fsm_iwec --fsm-type=rs-nfa --iw-base=B --iw-max=M --new-iw-base=N --out-map=F`
// (reference: blingfiretools/fa_fsm2fsm_iwec/fa_fsm2fsm_iwec.cpp,
//  blingfirecompile.library/src/FACalcIwEqClasses.cpp:220-313, FASplitSets.cpp:52-110),
// used only to rebuild `bert_base_tok.bin` from /root/reference/ldbsrc/bert_base_tok in
// feasible time: the reference stage enumerates states x alphabet (~2.5e5 x 1.3e5 cells
// for the BERT vocabularies, many CPU-hours); this one groups the transitions by input
// weight and compares the per-weight (state, destination-set) lists, O(T log T).
//
// Same result by construction, including the class numbering: two input weights in
// [B, M] fall in one class iff every state sends them to the same destination set, and
// classes are numbered in order of first appearance over ascending input weight
// (FASplitSets::Classify assigns ids by first insertion while walking i upwards).
// Weights outside [B, M] map to themselves.  Output text formats follow
// FAAutIOTools::PrintNfaCommon / FAMapIOTools::Print.  Verified byte-for-byte against the
// reference tool on the wbd NFA and by rebuilding the checked-in bert_base_cased_tok.bin
// (tools/modelbuild/README.md).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

struct Tr { int32_t iw, src, dst; };

int main(int argc, char** argv) {
    long long iw_base = 0, iw_max = 0, new_base = 0;
    const char* out_map = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!strncmp(argv[i], "--iw-base=", 10)) iw_base = atoll(argv[i] + 10);
        else if (!strncmp(argv[i], "--iw-max=", 9)) iw_max = atoll(argv[i] + 9);
        else if (!strncmp(argv[i], "--new-iw-base=", 14)) new_base = atoll(argv[i] + 14);
        else if (!strncmp(argv[i], "--out-map=", 10)) out_map = argv[i] + 10;
        else if (!strcmp(argv[i], "--fsm-type=rs-nfa")) {}
        else { fprintf(stderr, "fast_iwec: unsupported argument %s\n", argv[i]); return 2; }
    }
    if (!out_map) { fprintf(stderr, "fast_iwec: --out-map required\n"); return 2; }

    std::string header;
    std::vector<Tr> tr;
    char line[256];
    bool in_header = true;
    while (fgets(line, sizeof line, stdin)) {
        if (in_header && !(line[0] >= '0' && line[0] <= '9') && line[0] != '-' && line[0] != '\n') {
            header += line;
            continue;
        }
        in_header = false;
        if (line[0] == '\n') break;
        Tr t;
        if (sscanf(line, "%d %d %d", &t.src, &t.dst, &t.iw) != 3) {
            fprintf(stderr, "fast_iwec: bad transition line: %s", line); return 3;
        }
        if (t.dst < 0) { fprintf(stderr, "fast_iwec: dead-state transitions not handled\n"); return 3; }
        tr.push_back(t);
    }
    std::sort(tr.begin(), tr.end(), [](const Tr& a, const Tr& b) {
        if (a.iw != b.iw) return a.iw < b.iw;
        if (a.src != b.src) return a.src < b.src;
        return a.dst < b.dst;
    });

    // classes over ascending iw
    struct Run { size_t b, e; };
    std::vector<Run> reps;                               // representative run of each class
    std::unordered_map<uint64_t, std::vector<int>> by_hash;
    std::vector<std::pair<int32_t, int32_t>> iw2new;     // every alphabet weight, ascending
    for (size_t b = 0; b < tr.size();) {
        size_t e = b;
        while (e < tr.size() && tr[e].iw == tr[b].iw) ++e;
        const int32_t iw = tr[b].iw;
        if (iw < iw_base || iw > iw_max) {
            iw2new.push_back({iw, iw});
        } else {
            uint64_t h = 1469598103934665603ull;
            for (size_t i = b; i < e; ++i) {
                h = (h ^ (uint32_t)tr[i].src) * 1099511628211ull;
                h = (h ^ (uint32_t)tr[i].dst) * 1099511628211ull;
            }
            int cls = -1;
            auto& cand = by_hash[h];
            for (int c : cand) {
                const Run& r = reps[c];
                if (r.e - r.b != e - b) continue;
                bool same = true;
                for (size_t i = 0; i < e - b && same; ++i)
                    same = tr[r.b + i].src == tr[b + i].src && tr[r.b + i].dst == tr[b + i].dst;
                if (same) { cls = c; break; }
            }
            if (cls < 0) { cls = (int)reps.size(); reps.push_back({b, e}); cand.push_back(cls); }
            iw2new.push_back({iw, (int32_t)(cls + new_base)});
        }
        b = e;
    }

    // remap in place (tr is sorted by iw, iw2new too)
    {
        size_t k = 0;
        for (auto& t : tr) {
            while (iw2new[k].first != t.iw) ++k;
            t.iw = iw2new[k].second;
        }
    }
    std::sort(tr.begin(), tr.end(), [](const Tr& a, const Tr& b) {
        if (a.src != b.src) return a.src < b.src;
        if (a.iw != b.iw) return a.iw < b.iw;
        return a.dst < b.dst;
    });
    fputs(header.c_str(), stdout);
    const Tr* prev = nullptr;
    for (const Tr& t : tr) {
        if (prev && prev->src == t.src && prev->iw == t.iw && prev->dst == t.dst) continue;
        printf("%d %d %d\n", t.src, t.dst, t.iw);
        prev = &t;
    }
    putchar('\n');

    FILE* f = fopen(out_map, "w");
    if (!f) { perror(out_map); return 4; }
    for (auto& p : iw2new) fprintf(f, "%d -> %d\n", p.first, p.second);
    fputc('\n', f);
    fclose(f);
    fprintf(stderr, "fast_iwec: %zu transitions, %zu weights, %zu classes\n", tr.size(), iw2new.size(), reps.size());
    return 0;
}
