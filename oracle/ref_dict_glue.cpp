// ref_dict_glue.cpp -- TEST INFRASTRUCTURE: a C entry to the reference's own FADictInterpreter_t<int>.
//
// blingfiretokdll exports nothing that reaches FADictInterpreter_t, so this file (our code, no reference code in it) sets the
// interpreter up over a model's [pos-dict] exactly as blingfiretokdll.cpp:945-956 sets up the segmentation engines
// (FALDB::SetImage, header Get(FUNC_POS_DICT), FADictConfKeeper::SetLDB + Init) plus FADictInterpreter_t::SetConf (conf, NULL),
// and is compiled by oracle/Makefile TOGETHER WITH the reference sources where they lie under /root/reference into
// oracle/_ref/libdictref.so (git-ignored; travels to the GPU box).  tests/test_dict_lookup.py pins the oracle restatement
// (bfo_dict_get_info*) against it.
#include "FAConfig.h"
#include "FAImageDump.h"
#include "FALDB.h"
#include "FAFsmConst.h"
#include "FADictConfKeeper.h"
#include "FADictInterpreter_t.h"

#include <new>

using namespace BlingFire;

namespace {
struct RefDict {
    FAImageDump img;
    FALDB ldb;
    FADictConfKeeper conf;
    FADictInterpreter_t<int> dict;
};
}

extern "C" {

void *refdict_load(const char *path)
{
    RefDict *r = new (std::nothrow) RefDict();
    if (!r) return nullptr;
    try {
        r->img.Load(path);
        r->ldb.SetImage(r->img.GetImageDump());
        const int *vals = nullptr;
        const int n = r->ldb.GetHeader()->Get(FAFsmConst::FUNC_POS_DICT, &vals);
        if (n == -1) { delete r; return nullptr; }
        r->conf.SetLDB(&r->ldb);
        r->conf.Init(vals, n);
        r->dict.SetConf(&r->conf, nullptr);
    } catch (...) {
        delete r;
        return nullptr;
    }
    return r;
}

void refdict_free(void *h) { delete (RefDict *)h; }

int refdict_get_info_id(void *h, const int *word, int n) { return ((RefDict *)h)->dict.GetInfoId(word, n); }

int refdict_get_info(void *h, const int *word, int n, int *out, int max_out) { return ((RefDict *)h)->dict.GetInfo(word, n, out, max_out); }

}
