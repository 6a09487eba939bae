// hosttest.h -- TEST-ONLY: the handle shared by the translation units of libbf_hosttest.so
#pragma once
#include <string>
#include "../../blingfire_amd/csrc/bf_model.h"

struct Handle { bfa::Model m; std::string path; };
