// bf_batch.h -- the batch a kernel works on (concatenated documents) and the device form of a code-point map; shared by every
// parameter block (bf_kernels.h, bf_wave.h).  Plain C++ (no HIP types) so that tests/hosttest can build parameter blocks on the host.
#pragma once
#include <stdint.h>

namespace bfa {

// Inputs shared by the prep kernels: the batch (concatenated documents) and where results go.
struct Batch {
    const uint8_t *text;        // concatenated UTF-8 documents
    const int64_t *doc_off;     // [ndocs+1] byte offsets, doc_off[0] == 0 and doc_off[ndocs] == total_bytes (include/*.h: precondition)
    int64_t ndocs;
    int64_t total_bytes;        // bytes of `text`; a document whose range leaves [0, total_bytes] is treated as empty (status bit 3)
    int *status;
};
constexpr int BF_STATUS_BAD_OFFSETS = 8;
constexpr int BF_STATUS_DOC_FAILED = 32;   // a BPE document has no answer (the reference does not terminate on it, ..._bpe_t.h:299-313); its count is 0
constexpr int BF_STATUS_POOL = 64;         // a BPE document's arcs did not fit the pool (bf_bpe_seg_body.h); its count is 0, the need is added up for the host
constexpr int BF_STATUS_INTERNAL = 16;   // a kernel met a state its load-time checks exclude (results of the batch are not to be trusted)

// code point map (TwoLevelMap on the device)
struct DevCpMap { const uint16_t *l1; const uint32_t *pages; };

} // namespace bfa
