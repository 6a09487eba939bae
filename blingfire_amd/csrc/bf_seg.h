// bf_seg.h -- per-document segmenter programs (Unigram-LM / BPE); filled in below.
#pragma once
#include "bf_lex.h"
