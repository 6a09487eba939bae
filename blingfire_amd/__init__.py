"""blingfire_amd -- Python mirror of the reference wrapper for the TextToIds path.

Same function names, argument meaning and return conventions as the reference
``dist-pypi/blingfire/__init__.py`` (load_model :229-234, free_model :237-240, text_to_ids :243-253,
change_settings_dummy_prefix :287-288, text_to_words :85-102, text_to_words_with_model :105-122), bound to the MI355X-native ``libblingfiretokdll.so`` built
in-tree by ``blingfire_amd/build.py``.  Additive: ``text_to_ids_batch`` (host buffers) and
``text_to_ids_batch_device`` (torch tensors already resident in HBM).

There is no CPU fallback: if the shared library is missing or no HIP device is visible the calls
raise / return the reference's error value, loudly.
"""
import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libblingfiretokdll.so")

_lib = None


def lib():
    """The loaded C-ABI library (built by blingfire_amd/build.py)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: run `python -m blingfire_amd.build` (hipcc, gfx950) first; "
                               "there is no CPU fallback" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.LoadModel.restype = c_void_p
        L.LoadModel.argtypes = [c_char_p]
        L.SetModel.restype = c_void_p
        L.SetModel.argtypes = [c_char_p, c_int]
        L.FreeModel.restype = c_int
        L.FreeModel.argtypes = [c_void_p]
        for name in ("TextToIds", "TextToIds_wp", "TextToIds_sp"):
            f = getattr(L, name)
            f.restype = c_int
            f.argtypes = [c_void_p, c_char_p, c_int, c_void_p, c_int, c_int]
        for name in ("TextToIdsWithOffsets", "TextToIdsWithOffsets_wp", "TextToIdsWithOffsets_sp"):
            f = getattr(L, name)
            f.restype = c_int
            f.argtypes = [c_void_p, c_char_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int]
        L.TextToIdsWithOffsetsBatch.restype = c_int64
        L.TextToIdsWithOffsetsBatch.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int]
        L.TextToIdsWithOffsetsBatchDevice.restype = c_int
        L.TextToIdsWithOffsetsBatchDevice.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                                                      c_void_p, c_int, c_int, c_void_p]
        L.SetNoDummyPrefix.restype = c_int
        L.SetNoDummyPrefix.argtypes = [c_void_p, ctypes.c_bool]
        L.GetBlingFireTokVersion.restype = c_int
        L.TextToIdsBatch.restype = c_int64
        L.TextToIdsBatch.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int]
        L.TextToIdsBatchDevice.restype = c_int
        L.TextToIdsBatchDevice.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p,
                                           c_int, c_int, c_void_p]
        L.TextToWords.restype = c_int
        L.TextToWords.argtypes = [c_char_p, c_int, c_void_p, c_int]
        L.TextToWordsWithModel.restype = c_int
        L.TextToWordsWithModel.argtypes = [c_char_p, c_int, c_void_p, c_int, c_void_p]
        L.TextToWordsWithOffsets.restype = c_int
        L.TextToWordsWithOffsets.argtypes = [c_char_p, c_int, c_void_p, c_void_p, c_void_p, c_int]
        L.TextToWordsWithOffsetsWithModel.restype = c_int
        L.TextToWordsWithOffsetsWithModel.argtypes = [c_char_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]
        L.TextToSentences.restype = c_int
        L.TextToSentences.argtypes = [c_char_p, c_int, c_void_p, c_int]
        L.TextToSentencesWithModel.restype = c_int
        L.TextToSentencesWithModel.argtypes = [c_char_p, c_int, c_void_p, c_int, c_void_p]
        L.TextToSentencesWithOffsets.restype = c_int
        L.TextToSentencesWithOffsets.argtypes = [c_char_p, c_int, c_void_p, c_void_p, c_void_p, c_int]
        L.TextToSentencesWithOffsetsWithModel.restype = c_int
        L.TextToSentencesWithOffsetsWithModel.argtypes = [c_char_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]
        L.TextToWordsBatch.restype = c_int64
        L.TextToWordsBatch.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p]
        L.TextToWordsBatchDevice.restype = c_int
        L.TextToWordsBatchDevice.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p]
        L.TextToSentencesBatch.restype = c_int64
        L.TextToSentencesBatch.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p]
        L.TextToSentencesBatchDevice.restype = c_int
        L.TextToSentencesBatchDevice.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p]
        L.IdsToText.restype = c_int
        L.IdsToText.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_int, ctypes.c_bool]
        L.IdsToTextBatch.restype = c_int64
        L.IdsToTextBatch.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int]
        L.IdsToTextBatchDevice.restype = c_int
        L.IdsToTextBatchDevice.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p]
        L.BfLastKernelMs.restype = c_int
        L.BfLastKernelMs.argtypes = [c_void_p, POINTER(c_float), c_int]
        L.BfLastStatus.restype = c_int
        L.BfLastStatus.argtypes = [c_void_p]
        L.BfLastError.restype = c_char_p
        L.BfModelKind.restype = c_int
        L.BfModelKind.argtypes = [c_void_p]
        L.BfSetVariant.restype = c_int
        L.BfSetVariant.argtypes = [c_void_p, c_int]
        L.DictGetInfoBatch.restype = c_int64
        L.DictGetInfoBatch.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]
        L.DictGetInfoBatchDevice.restype = c_int
        L.DictGetInfoBatchDevice.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]
        L.BfSetDevices.restype = c_int
        L.BfSetDevices.argtypes = [c_void_p, c_void_p, c_int]
        L.BfShardHandle.restype = c_void_p
        L.BfShardHandle.argtypes = [c_void_p, c_int]
        L.BfShardRanges.restype = c_int
        L.BfShardRanges.argtypes = [c_void_p, c_int64, c_int, c_void_p]
        L.BfSetLexStats.restype = c_int
        L.BfSetLexStats.argtypes = [c_void_p, c_int]
        L.BfSetBpePoolBytes.restype = c_int64
        L.BfSetBpePoolBytes.argtypes = [c_void_p, c_int64]
        L.BfReserve.restype = c_int
        L.BfReserve.argtypes = [c_void_p, c_int64, c_int64, c_int]
        L.NormalizeSpaces.restype = c_int
        L.NormalizeSpaces.argtypes = [c_char_p, c_int, c_void_p, c_int, c_int]
        L.TextToHashes.restype = c_int
        L.TextToHashes.argtypes = [c_char_p, c_int, c_void_p, c_int, c_int, c_int]
        L.WordHyphenationWithModel.restype = c_int
        L.WordHyphenationWithModel.argtypes = [c_char_p, c_int, c_void_p, c_int, c_void_p, c_int]
        _lib = L
    return _lib


def get_blingfiretok_version():
    return lib().GetBlingFireTokVersion()


def load_model(file_name):
    """reference __init__.py:229-234; raises instead of returning a NULL handle."""
    h = lib().LoadModel(file_name.encode("utf-8"))
    if not h:
        raise RuntimeError("LoadModel(%r) failed: %s" % (file_name, lib().BfLastError().decode("utf-8", "replace")))
    return h


def free_model(h):
    lib().FreeModel(c_void_p(h))


def text_to_ids(h, s, max_len, unk=0, no_padding=False):
    """reference __init__.py:243-253: zero-padded uint32 array of max_len (or the exact count with no_padding)."""
    s_bytes = s.encode("utf-8") if isinstance(s, str) else bytes(s)
    o = (c_int32 * max_len)()
    t = lib().TextToIds(c_void_p(h), s_bytes, len(s_bytes), byref(o), max_len, unk)
    n = min(max_len, t) if no_padding else max_len
    return np.frombuffer(o, dtype=np.uint32, count=n)


def utf8text_to_ids_with_offsets(h, s_bytes, max_len, unk=0, no_padding=False):
    """reference __init__.py:272-284: (ids uint32, start offsets int32, end offsets int32), zero padded to max_len."""
    o = (c_int32 * max_len)()
    o_s = (c_int32 * max_len)()
    o_e = (c_int32 * max_len)()
    t = lib().TextToIdsWithOffsets(c_void_p(h), s_bytes, len(s_bytes), byref(o), byref(o_s), byref(o_e), max_len, unk)
    n = min(max_len, t) if no_padding else max_len
    return (np.frombuffer(o, dtype=np.uint32, count=n), np.frombuffer(o_s, dtype=np.int32, count=n),
            np.frombuffer(o_e, dtype=np.int32, count=n))


def normalize_spaces(s, uSpace=0x20):
    """reference __init__.py:65-82: runs of white space -> one uSpace, leading / trailing dropped; '' on error."""
    s_bytes = s.encode("utf-8")
    o = ctypes.create_string_buffer(len(s_bytes) * 2 + 4)
    n = lib().NormalizeSpaces(s_bytes, len(s_bytes), o, len(o), uSpace)
    return "" if n == -1 or n > len(o) else o.value.decode("utf-8")


def text_to_hashes(s, word_n_grams, bucketSize):
    """reference __init__.py:151-167: int32 array of the token hashes followed by the word n-gram hashes; '' on error."""
    s_bytes = s.encode("utf-8")
    cap = max(word_n_grams * len(s_bytes), 1)
    o = (c_int32 * cap)()
    n = lib().TextToHashes(s_bytes, len(s_bytes), byref(o), cap, word_n_grams, bucketSize)
    if n == -1 or n > cap:
        return ""
    return np.frombuffer(o, dtype=np.int32, count=n)


def word_hyphenation_with_model(h, s, uHy=0x2D):
    """reference __init__.py:125-142.  The hyphenation engine is not part of this library (not on the TextToIds path): the C entry
    point exists and returns the reference's error value, so this returns '' for every non-empty word."""
    s_bytes = s.encode("utf-8")
    o = ctypes.create_string_buffer(len(s_bytes) * 4 + 1)
    n = lib().WordHyphenationWithModel(s_bytes, len(s_bytes), o, len(o), c_void_p(h) if h else None, uHy)
    return "" if n == -1 or n > len(o) else o.value.decode("utf-8")


def text_to_words(s):
    """reference __init__.py:85-102: space-joined words by the built-in wbd.bin; '' on error."""
    s_bytes = s.encode("utf-8")
    o = ctypes.create_string_buffer(len(s_bytes) * 3)
    n = lib().TextToWords(s_bytes, len(s_bytes), o, len(o))
    return "" if n == -1 or n > len(o) else o.value.decode("utf-8")


def text_to_words_with_model(h, s):
    """reference __init__.py:105-122"""
    s_bytes = s.encode("utf-8")
    o = ctypes.create_string_buffer(len(s_bytes) * 3)
    n = lib().TextToWordsWithModel(s_bytes, len(s_bytes), o, len(o), c_void_p(h))
    return "" if n == -1 or n > len(o) else o.value.decode("utf-8")


def text_to_words_with_offsets(s, h=None):
    """reference __init__.py:161-223: (string, [(begin, end) per word]) as character offsets into `s`, end exclusive."""
    s_bytes = s.encode("utf-8")
    cap = len(s_bytes) * 3
    o = ctypes.create_string_buffer(max(cap, 1))
    st = (c_int32 * max(cap, 1))()
    en = (c_int32 * max(cap, 1))()
    n = lib().TextToWordsWithOffsetsWithModel(s_bytes, len(s_bytes), o, byref(st), byref(en), cap, c_void_p(h) if h else None)
    if n == -1 or n > cap:
        return "", []
    words = o.value.decode("utf-8")
    k = len(words.split(" ")) if words else 0
    lead = np.frombuffer(s_bytes, dtype=np.uint8) & 0xC0 != 0x80
    char_at = np.concatenate([[0], np.cumsum(lead)])          # byte offset -> number of characters that start before it
    return words, [(int(char_at[st[i]]), int(char_at[en[i] + 1])) for i in range(k)]


def text_to_sentences(s):
    """reference __init__.py:45-62: one sentence per line by the built-in sbd.bin; '' on error."""
    s_bytes = s.encode("utf-8")
    o = ctypes.create_string_buffer(len(s_bytes) * 2)
    n = lib().TextToSentences(s_bytes, len(s_bytes), o, len(o))
    return "" if n == -1 or n > len(o) else o.value.decode("utf-8")


def text_to_sentences_with_model(h, s):
    """reference __init__.py:65-82"""
    s_bytes = s.encode("utf-8")
    o = ctypes.create_string_buffer(len(s_bytes) * 2)
    n = lib().TextToSentencesWithModel(s_bytes, len(s_bytes), o, len(o), c_void_p(h))
    return "" if n == -1 or n > len(o) else o.value.decode("utf-8")


def text_to_sentences_and_offsets(s, h=None):
    """reference __init__.py:225-226: (string, [(begin, end) per sentence]) as character offsets into `s`, end exclusive."""
    s_bytes = s.encode("utf-8")
    cap = len(s_bytes) * 2
    o = ctypes.create_string_buffer(max(cap, 1))
    st = (c_int32 * max(cap, 1))()
    en = (c_int32 * max(cap, 1))()
    n = lib().TextToSentencesWithOffsetsWithModel(s_bytes, len(s_bytes), o, byref(st), byref(en), cap, c_void_p(h) if h else None)
    if n == -1 or n > cap:
        return "", []
    sents = o.value.decode("utf-8")
    k = sents.count("\n") + 1 if sents else 0
    lead = np.frombuffer(s_bytes, dtype=np.uint8) & 0xC0 != 0x80
    char_at = np.concatenate([[0], np.cumsum(lead)])
    return sents, [(int(char_at[st[i]]), int(char_at[en[i] + 1])) for i in range(k)]


def text_to_sentences_batch(docs, h=None):
    """additive: TextToSentences over many documents; h None = the built-in sbd.bin.  Same conventions as text_to_words_batch."""
    return text_to_words_batch(docs, h, _fn="TextToSentencesBatch")


def text_to_words_batch(docs, h=None, _fn="TextToWordsBatch"):
    """additive: TextToWords over many documents (list of bytes, or a (text uint8, offsets int64) pair) -> (text uint8, offsets int64[ndocs+1]);
    h None = the built-in wbd.bin."""
    text, off = docs if isinstance(docs, tuple) else pack_docs(docs)
    text = np.ascontiguousarray(text, dtype=np.uint8); off = np.ascontiguousarray(off, dtype=np.int64)
    nd = len(off) - 1
    t_off = np.zeros(nd + 1, dtype=np.int64)
    hp = c_void_p(h) if h else None
    fn = getattr(lib(), _fn)
    n = fn(hp, c_void_p(text.ctypes.data), c_void_p(off.ctypes.data), nd, None, 0, c_void_p(t_off.ctypes.data))
    out = np.empty(0, dtype=np.uint8)
    if n == -3:                                           # BF_E_CAPACITY: the offsets tell the size
        out = np.empty(int(t_off[-1]), dtype=np.uint8)
        n = fn(hp, c_void_p(text.ctypes.data), c_void_p(off.ctypes.data), nd, c_void_p(out.ctypes.data), len(out), c_void_p(t_off.ctypes.data))
    if n < 0:
        raise RuntimeError("%s failed: %d (%s)" % (_fn, n, lib().BfLastError().decode("utf-8", "replace")))
    return out, t_off


def ids_to_text(h, ids, skip_special_tokens=True, output_buffer_size=None):
    """reference __init__.py:256-269: text of an id array (numpy int32 / uint32); '' on error or if the buffer is too small."""
    ids = np.ascontiguousarray(ids).view(np.int32) if isinstance(ids, np.ndarray) and ids.dtype.itemsize == 4 else np.asarray(ids, dtype=np.int32)
    if output_buffer_size is None:
        output_buffer_size = len(ids) * 32
    o = ctypes.create_string_buffer(max(output_buffer_size, 1))
    n = lib().IdsToText(c_void_p(h), c_void_p(ids.ctypes.data), len(ids), o, output_buffer_size, bool(skip_special_tokens))
    if n == -1 or n > output_buffer_size:
        return ""
    return o.value.decode("utf-8")


def ids_to_text_batch(h, ids, id_off, skip_special_tokens=True):
    """additive: (text bytes as uint8 array, text offsets int64[nseq+1]) for many id sequences at once (IdsToTextBatch)."""
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    id_off = np.ascontiguousarray(id_off, dtype=np.int64)
    nseq = len(id_off) - 1
    t_off = np.zeros(nseq + 1, dtype=np.int64)
    n = lib().IdsToTextBatch(c_void_p(h), c_void_p(ids.ctypes.data), c_void_p(id_off.ctypes.data), nseq, None, 0, c_void_p(t_off.ctypes.data),
                             int(bool(skip_special_tokens)))
    if n == -3:                                           # BF_E_CAPACITY: the offsets tell the size
        text = np.empty(int(t_off[-1]), dtype=np.uint8)
        n = lib().IdsToTextBatch(c_void_p(h), c_void_p(ids.ctypes.data), c_void_p(id_off.ctypes.data), nseq, c_void_p(text.ctypes.data), len(text),
                                 c_void_p(t_off.ctypes.data), int(bool(skip_special_tokens)))
    else:
        text = np.empty(0, dtype=np.uint8)
    if n < 0:
        raise RuntimeError("IdsToTextBatch failed: %d (%s)" % (n, lib().BfLastError().decode("utf-8", "replace")))
    return text, t_off


def change_settings_dummy_prefix(h, add_prefix):
    lib().SetNoDummyPrefix(c_void_p(h), bool(not add_prefix))


# ---------------------------------------------------------------------------------------------
# additive batch API
# ---------------------------------------------------------------------------------------------

def pack_docs(docs):
    """list of str/bytes -> (uint8 array, int64 offsets[ndocs+1])"""
    bs = [d.encode("utf-8") if isinstance(d, str) else bytes(d) for d in docs]
    off = np.zeros(len(bs) + 1, dtype=np.int64)
    if bs:
        np.cumsum([len(b) for b in bs], out=off[1:])
    text = np.frombuffer(b"".join(bs), dtype=np.uint8) if bs else np.zeros(0, dtype=np.uint8)
    return text, off


def set_devices(h, device_ids):
    """Range-shards the host-buffer batch calls of h over the listed devices of this node (BfSetDevices: tables replicated, contiguous
    byte-balanced document ranges, one host thread per device, no collective).  A device may be listed more than once."""
    arr = (c_int * len(device_ids))(*[int(d) for d in device_ids])
    r = lib().BfSetDevices(c_void_p(h), arr, len(device_ids))
    if r != 0:
        raise RuntimeError("BfSetDevices failed (%d): %s" % (r, lib().BfLastError().decode("utf-8", "replace")))


def shard_handle(h, g):
    """the handle that tokenises range g after set_devices (None past the last range); owned by h"""
    return lib().BfShardHandle(c_void_p(h), int(g))


def shard_ranges(doc_offsets, G):
    """document ranges a batch is split into over G devices: int64[G + 1] (BfShardRanges; no device needed)"""
    off = np.ascontiguousarray(doc_offsets, dtype=np.int64)
    bounds = np.zeros(G + 1, dtype=np.int64)
    r = lib().BfShardRanges(off.ctypes.data, len(off) - 1, G, bounds.ctypes.data)
    if r != 0:
        raise RuntimeError("BfShardRanges failed (%d)" % r)
    return bounds


def text_to_ids_batch(h, docs, max_len, unk=0):
    """For every document exactly what TextToIds(h, doc, len, buf, max_len, unk) writes, concatenated.

    docs: list of str/bytes, or a (uint8 ndarray, int64 offsets) pair.  Returns (ids int32[total], offsets int64[ndocs+1]).
    """
    text, off = docs if isinstance(docs, tuple) else pack_docs(docs)
    text = np.ascontiguousarray(text, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    ndocs = len(off) - 1
    total = int(off[-1] - off[0]) if ndocs > 0 else 0
    cap = min(2 * (total + ndocs), ndocs * max(max_len, 0)) + 1      # every id covers >= 1 element of <= 2*(n+1) per document
    ids = np.empty(cap, dtype=np.int32)
    id_off = np.zeros(ndocs + 1, dtype=np.int64)
    r = lib().TextToIdsBatch(c_void_p(h), text.ctypes.data, off.ctypes.data, ndocs, ids.ctypes.data, cap, id_off.ctypes.data,
                             max_len, unk)
    if r < 0:
        raise RuntimeError("TextToIdsBatch failed (%d): %s" % (r, lib().BfLastError().decode("utf-8", "replace")))
    return ids[:r], id_off


def text_to_ids_with_offsets_batch(h, docs, max_len, unk=0):
    """Batch form of TextToIdsWithOffsets: (ids, starts, ends, id_offsets); starts/ends are byte offsets inside each document."""
    text, off = docs if isinstance(docs, tuple) else pack_docs(docs)
    text = np.ascontiguousarray(text, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    ndocs = len(off) - 1
    total = int(off[-1] - off[0]) if ndocs > 0 else 0
    cap = min(2 * (total + ndocs), ndocs * max(max_len, 0)) + 1
    ids = np.empty(cap, dtype=np.int32)
    starts = np.empty(cap, dtype=np.int32)
    ends = np.empty(cap, dtype=np.int32)
    id_off = np.zeros(ndocs + 1, dtype=np.int64)
    r = lib().TextToIdsWithOffsetsBatch(c_void_p(h), text.ctypes.data, off.ctypes.data, ndocs, ids.ctypes.data, starts.ctypes.data,
                                        ends.ctypes.data, cap, id_off.ctypes.data, max_len, unk)
    if r < 0:
        raise RuntimeError("TextToIdsWithOffsetsBatch failed (%d): %s" % (r, lib().BfLastError().decode("utf-8", "replace")))
    return ids[:r], starts[:r], ends[:r], id_off


def text_to_ids_batch_device(h, d_text, d_doc_off, max_len, unk=0, out_ids=None, out_off=None, stream=None):
    """Device-resident batch: torch uint8 text tensor + int64 offsets tensor (both on the model's GPU).

    Enqueues on torch's current stream (or `stream`) and returns (ids int32[cap], id_offsets int64[ndocs+1]) tensors
    without synchronising; the valid id count is id_offsets[-1].
    """
    import torch
    ndocs = d_doc_off.numel() - 1
    total = d_text.numel()
    cap = max(1, min(2 * (total + ndocs), ndocs * max(max_len, 0)))
    if out_ids is None:
        out_ids = torch.empty(cap, dtype=torch.int32, device=d_text.device)
    if out_off is None:
        out_off = torch.empty(ndocs + 1, dtype=torch.int64, device=d_text.device)
    s = stream if stream is not None else torch.cuda.current_stream(d_text.device).cuda_stream
    r = lib().TextToIdsBatchDevice(c_void_p(h), d_text.data_ptr(), d_doc_off.data_ptr(), ndocs, total, out_ids.data_ptr(),
                                   out_ids.numel(), out_off.data_ptr(), max_len, unk, c_void_p(s))
    if r != 0:
        raise RuntimeError("TextToIdsBatchDevice failed (%d): %s" % (r, lib().BfLastError().decode("utf-8", "replace")))
    return out_ids, out_off


def ids_to_text_batch_device(h, d_ids, d_id_off, out_text=None, out_off=None, skip_special_tokens=True, stream=None):
    """Device-resident IdsToText: torch int32 ids + int64 offsets on the model's GPU -> (text uint8[cap], text offsets int64[nseq+1]).
    Without `out_text` a buffer of 16 bytes per id is used (IdsToTextBatchDevice never writes past it)."""
    import torch
    nseq = d_id_off.numel() - 1
    if out_text is None:
        out_text = torch.empty(max(16 * d_ids.numel(), 1), dtype=torch.uint8, device=d_ids.device)
    if out_off is None:
        out_off = torch.empty(nseq + 1, dtype=torch.int64, device=d_ids.device)
    s = stream if stream is not None else torch.cuda.current_stream(d_ids.device).cuda_stream
    r = lib().IdsToTextBatchDevice(c_void_p(h), d_ids.data_ptr(), d_id_off.data_ptr(), nseq, out_text.data_ptr(), out_text.numel(),
                                   out_off.data_ptr(), int(bool(skip_special_tokens)), c_void_p(s))
    if r != 0:
        raise RuntimeError("IdsToTextBatchDevice failed (%d): %s" % (r, lib().BfLastError().decode("utf-8", "replace")))
    return out_text, out_off


def dict_get_info_batch(h, keys):
    """additive: the reference's FADictInterpreter_t<int>::GetInfo over the model's [pos-dict] for many keys at once (DictGetInfoBatch).
    keys: list of str (code points) or of int sequences, or an (int32 array, int64 offsets) pair.
    Returns (ret int32[nkeys] = value count or -1, info_ids int32[nkeys], values int32[total], value_offsets int64[nkeys+1])."""
    if isinstance(keys, tuple):
        flat, off = keys
    else:
        seqs = [[ord(c) for c in k] if isinstance(k, str) else list(k) for k in keys]
        off = np.zeros(len(seqs) + 1, dtype=np.int64)
        if seqs:
            np.cumsum([len(q) for q in seqs], out=off[1:])
        flat = np.array([c for q in seqs for c in q], dtype=np.int32)
    flat = np.ascontiguousarray(flat, dtype=np.int32)
    off = np.ascontiguousarray(off, dtype=np.int64)
    nk = len(off) - 1
    ret = np.zeros(nk, dtype=np.int32)
    ids = np.zeros(nk, dtype=np.int32)
    v_off = np.zeros(nk + 1, dtype=np.int64)
    L = lib()
    n = L.DictGetInfoBatch(c_void_p(h), flat.ctypes.data, off.ctypes.data, nk, ret.ctypes.data, ids.ctypes.data, None, 0, v_off.ctypes.data)
    vals = np.empty(0, dtype=np.int32)
    if n == -3:                                           # BF_E_CAPACITY: the offsets tell the size
        vals = np.empty(int(v_off[-1]), dtype=np.int32)
        n = L.DictGetInfoBatch(c_void_p(h), flat.ctypes.data, off.ctypes.data, nk, ret.ctypes.data, ids.ctypes.data, vals.ctypes.data, len(vals), v_off.ctypes.data)
    if n < 0:
        raise RuntimeError("DictGetInfoBatch failed: %d (%s)" % (n, L.BfLastError().decode("utf-8", "replace")))
    return ret, ids, vals, v_off


def reserve(h, max_docs, max_bytes, want_offsets=False):
    """additive: size the handle's workspaces once (BfReserve) so that later device-resident batches of that size allocate nothing."""
    r = lib().BfReserve(c_void_p(h), int(max_docs), int(max_bytes), int(bool(want_offsets)))
    if r != 0:
        raise RuntimeError("BfReserve failed (%d): %s" % (r, lib().BfLastError().decode("utf-8", "replace")))


def last_kernel_ms(h):
    """[prep, tokenise, scan, compact, total, dominant kernel] GPU milliseconds of the last batch call (HIP events on its stream)."""
    buf = (c_float * 6)()
    n = lib().BfLastKernelMs(c_void_p(h), buf, 6)
    return [buf[i] for i in range(max(n, 0))]
