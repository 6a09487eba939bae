// hosttest.h -- TEST-ONLY: the handle shared by the translation units of libbf_hosttest.so
#pragma once
#include <string>
#include "../../blingfire_amd/csrc/bf_model.h"

struct Handle { bfa::Model m; std::string path; };

#include <vector>
extern "C" {
// bf_hosttest.cpp: the class stream the _sp prologue makes of one document (false: TextToIds returns 0 for it)
bool bft_sp_stream(const bfa::Model &m, const char *s, int n, std::vector<uint16_t> &st, std::vector<int> *src_off);
// bf_hosttest.cpp: the sequential restatement of the whole _sp path on one document (what the lane-per-document kernels compute)
int bft_emu_sp_doc(const bfa::Model &m, const char *s, int n, int32_t *ids, int max_ids, int unk);
}
