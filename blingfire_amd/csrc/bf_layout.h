// bf_layout.h -- bit layout of the device lexer table, shared by the loader (bf_model.cpp) and the lane program (bf_lex.h)
#pragma once
#include <stdint.h>

namespace bfa {

// Device entry of the lexer table (low word): [final:1 | next:18 | cls:13]; an empty slot stores cls = 0x1FFF.
// Class-stream value of a code point outside the alphabet: 0x1FFE (classes are < 0x1FFE, checked at load), so the
// probe T[state + cls] needs no clamp and can never match: a real entry at that slot belongs to a class < 0x1FFE.
constexpr uint32_t LX_CLS_NONE = 0x1FFEu;
constexpr uint32_t LX_T_CLS_MASK = 0x1FFFu, LX_T_FINAL = 1u << 31, LX_T_NEXT_MASK = 0x3FFFFu;
constexpr int LX_T_NEXT_SHIFT = 13;
// Class-stream flag (bit 14 of the 16-bit stream element, above the 13 class bits): the class is one of the self-loop
// symbols of the model's "loop state" (bf_model.h Model::loop_base) -- the lane program fast-forwards over runs of such
// elements without touching the table.  Set in the fused code-point maps at load; masked off before every table probe.
constexpr uint32_t LX_C_LOOP = 0x4000u;

} // namespace bfa
