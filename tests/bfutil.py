"""Shared TEST / BENCH helpers (not part of the product): synthetic corpora, the oracle and the
compiled reference as checkers, the CPU-baseline timer.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg use this module; the product (blingfire_amd/) never imports it."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = os.path.join(ROOT, "models")
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libblingfiretokdll_ref.so")
CPUBASE_LIB = os.path.join(ROOT, "oracle", "libcpubaseline.so")
CORPUSGEN_LIB = os.path.join(ROOT, "tools", "libcorpusgen.so")
HOSTTEST_LIB = os.path.join(ROOT, "tests", "hosttest", "libbf_hosttest.so")
WORDS_EN = os.path.join(ROOT, "tests", "data", "words_en.txt")


def model_path(name):
    return os.path.join(MODELS, name)


def have_model(name):
    return os.path.exists(model_path(name))


def bert_model_name():
    """headline model: bert_base_tok.bin when the rebuilt file is present, else the checked-in cased sibling
    (same engine and format, SURVEY.md §7 'Minimum slice')."""
    return "bert_base_tok.bin" if have_model("bert_base_tok.bin") else "bert_base_cased_tok.bin"


# ------------------------------------------------------------------------------------------------
# oracle (plain-C restatement) and compiled reference
# ------------------------------------------------------------------------------------------------
class _T2I:
    """TextToIds-style checker around a C library."""

    def __init__(self, lib_path, load_name, t2i_name, free_name):
        self.lib = ctypes.CDLL(lib_path)
        self._load = getattr(self.lib, load_name)
        self._load.restype = ctypes.c_void_p
        self._load.argtypes = [ctypes.c_char_p]
        self._t2i = getattr(self.lib, t2i_name)
        self._t2i.restype = ctypes.c_int
        self._t2i.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        self._free = getattr(self.lib, free_name)
        self._free.argtypes = [ctypes.c_void_p]

    def load(self, path):
        h = self._load(path.encode())
        if not h:
            raise RuntimeError("checker could not load %s" % path)
        return h

    def free(self, h):
        self._free(ctypes.c_void_p(h))

    def text_to_ids(self, h, b, max_ids, unk, sentinel=-7):
        """returns (count, full buffer as list) with the buffer pre-filled with `sentinel`"""
        n = max(max_ids, 1)
        arr = (ctypes.c_int32 * n)(*([sentinel] * n))
        c = self._t2i(ctypes.c_void_p(h), b, len(b), arr, max_ids, unk)
        return c, list(arr)

    def with_offsets(self, h, b, max_ids, unk, name):
        """TextToIdsWithOffsets through `name` (reference: TextToIdsWithOffsets, oracle: bfo_text_to_ids_with_offsets).
        The text is passed at buffer+1 behind a fixed ASCII byte: the reference reads the byte BEFORE the string for a
        token made of the dummy prefix alone (tokdll:1527 with ToOffset == -1)."""
        f = getattr(self.lib, name)
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        n = max(max_ids, 1)
        i = (ctypes.c_int32 * n)()
        s = (ctypes.c_int32 * n)()
        e = (ctypes.c_int32 * n)()
        buf = ctypes.create_string_buffer(b"A" + b, len(b) + 1)
        c = f(ctypes.c_void_p(h), ctypes.addressof(buf) + 1, len(b), i, s, e, max_ids, unk)
        c = max(c, 0)
        return c, list(i)[:c], list(s)[:c], list(e)[:c]

    def batch(self, h, text, off, max_ids, unk):
        """per-document loop -> (ids int32[total], id_offsets int64[ndocs+1]) -- the golden form of TextToIdsBatch"""
        ndocs = len(off) - 1
        buf = (ctypes.c_int32 * max(max_ids, 1))()
        out = []
        id_off = np.zeros(ndocs + 1, dtype=np.int64)
        raw = text.tobytes() if isinstance(text, np.ndarray) else bytes(text)
        for d in range(ndocs):
            b = raw[off[d]:off[d + 1]]
            c = self._t2i(ctypes.c_void_p(h), b, len(b), buf, max_ids, unk)
            out.append(np.frombuffer(buf, dtype=np.int32, count=c).copy())
            id_off[d + 1] = id_off[d] + c
        ids = np.concatenate(out) if out else np.zeros(0, dtype=np.int32)
        return ids, id_off


def oracle():
    return _T2I(ORACLE_LIB, "bfo_load_model", "bfo_text_to_ids", "bfo_free_model")


def have_ref():
    return os.path.exists(REF_LIB)


def reference():
    return _T2I(REF_LIB, "LoadModel", "TextToIds", "FreeModel")


def checker_lib_path():
    """the CPU TextToIds used as golden/CPU baseline: the compiled reference when present, else the oracle port"""
    return (REF_LIB, "reference") if have_ref() else (ORACLE_LIB, "port")


def cpu_text_to_ids_batch(lib_path, model, text, off, max_ids, unk, nthreads=None, passes=1, want_ids=True):
    """Runs TextToIds per document on host threads through oracle/libcpubaseline.so.
    Returns (seconds, total_ids, ids int32[total] or None, id_offsets or None)."""
    L = ctypes.CDLL(CPUBASE_LIB)
    f = L.bfc_time_text_to_ids
    f.restype = ctypes.c_double
    f.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                  ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    text = np.ascontiguousarray(text, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    ndocs = len(off) - 1
    if nthreads is None:
        nthreads = host_threads()
    total = ctypes.c_int64(0)
    out_ids = out_counts = None
    if want_ids:
        out_ids = np.empty((ndocs, max(max_ids, 1)), dtype=np.int32)
        out_counts = np.zeros(ndocs, dtype=np.int64)
    # oracle library exports bfo_* names; the driver wants the reference names -> use a tiny shim for the port
    path = lib_path
    secs = f(path.encode(), model.encode(), text.ctypes.data, off.ctypes.data, ndocs, max_ids, unk, nthreads, passes,
             ctypes.byref(total), out_ids.ctypes.data if want_ids else None, out_counts.ctypes.data if want_ids else None)
    if secs < 0:
        raise RuntimeError("cpu baseline driver failed (%s)" % secs)
    if not want_ids:
        return secs, total.value, None, None
    id_off = np.zeros(ndocs + 1, dtype=np.int64)
    np.cumsum(out_counts, out=id_off[1:])
    mask = np.arange(out_ids.shape[1])[None, :] < out_counts[:, None]
    return secs, total.value, out_ids[mask], id_off


def cpu_doc_hashes(lib_path, model, text, off, max_ids, unk, nthreads=None):
    """One CPU pass that keeps, per document, the id count and the 64-bit hash of its ids (oracle/cpu_baseline.c
    bfc_ids_hash) -- the checker side of bench.py's full-shard comparison.  Returns (seconds, counts int64[ndocs], hashes uint64[ndocs])."""
    L = ctypes.CDLL(CPUBASE_LIB)
    f = L.bfc_text_to_ids_hashes
    f.restype = ctypes.c_double
    f.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                  ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    text = np.ascontiguousarray(text, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    ndocs = len(off) - 1
    counts = np.zeros(ndocs, dtype=np.int64)
    hashes = np.zeros(ndocs, dtype=np.uint64)
    total = ctypes.c_int64(0)
    secs = f(lib_path.encode(), model.encode(), text.ctypes.data, off.ctypes.data, ndocs, max_ids, unk, nthreads or host_threads(),
             ctypes.byref(total), counts.ctypes.data, hashes.ctypes.data)
    if secs < 0:
        raise RuntimeError("cpu baseline driver failed (%s)" % secs)
    return secs, counts, hashes


def cpu_ids_compact(lib_path, model, text, off, max_ids, unk, nthreads=None):
    """TextToIds per document on host threads, every id kept: (seconds, ids int32[total], id_offsets int64[ndocs + 1]) -- the CPU side
    of bench.py's exact full-shard check (oracle/cpu_baseline.c bfc_text_to_ids_compact)."""
    L = ctypes.CDLL(CPUBASE_LIB)
    f = L.bfc_text_to_ids_compact
    f.restype = ctypes.c_double
    f.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                  ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)), ctypes.c_void_p]
    L.bfc_free.argtypes = [ctypes.c_void_p]
    text = np.ascontiguousarray(text, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    ndocs = len(off) - 1
    id_off = np.zeros(ndocs + 1, dtype=np.int64)
    ptr = ctypes.POINTER(ctypes.c_int32)()
    secs = f(lib_path.encode(), model.encode(), text.ctypes.data, off.ctypes.data, ndocs, max_ids, unk, nthreads or host_threads(), ctypes.byref(ptr), id_off.ctypes.data)
    if secs < 0:
        raise RuntimeError("cpu baseline driver failed (%s)" % secs)
    n = int(id_off[-1])
    ids = np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].copy()
    L.bfc_free(ptr)
    return secs, ids, id_off


def cpu_text_to_words_time(lib_path, text, off, nthreads=1, passes=3):
    """best-of-`passes` seconds for one TextToWords call per line (built-in model) through oracle/libcpubaseline.so; returns (seconds, output bytes)"""
    L = ctypes.CDLL(CPUBASE_LIB)
    f = L.bfc_time_text_to_words
    f.restype = ctypes.c_double
    f.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    text = np.ascontiguousarray(text, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    ob = ctypes.c_int64(0)
    secs = f(lib_path.encode(), text.ctypes.data, off.ctypes.data, len(off) - 1, nthreads, passes, ctypes.byref(ob))
    if secs < 0:
        raise RuntimeError("cpu baseline driver failed (%s)" % secs)
    return secs, ob.value


IDS_HASH_C = -7046029254386353131          # 0x9E3779B97F4A7C15 as int64 (oracle/cpu_baseline.c bfc_ids_hash)


def ids_hash_np(ids, id_off):
    """bfc_ids_hash of every document, numpy (tests): sum_j (id_j + C) * (2j + 1) modulo 2^64 through a wrapping prefix sum."""
    ids = np.asarray(ids, dtype=np.int64)
    id_off = np.asarray(id_off, dtype=np.int64)
    nd = len(id_off) - 1
    doc = np.repeat(np.arange(nd), np.diff(id_off))
    j = np.arange(len(ids), dtype=np.int64) - id_off[:-1][doc]
    with np.errstate(over="ignore"):
        v = (ids.astype(np.uint64) + np.uint64(IDS_HASH_C & 0xFFFFFFFFFFFFFFFF)) * (2 * j + 1).astype(np.uint64)
        cs = np.concatenate([np.zeros(1, dtype=np.uint64), np.cumsum(v, dtype=np.uint64)])
        return cs[id_off[1:]] - cs[id_off[:-1]]


def host_threads():
    """CPU threads this process may actually use: the affinity mask, capped by a cgroup CPU quota if there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    q = cgroup_cpu_quota()
    if q:
        n = max(1, min(n, int(q + 0.999)))
    return n


def cgroup_cpu_quota():
    """CPU quota of the container in cores (cgroup v2 cpu.max or v1 cfs quota), None if unlimited / unknown."""
    try:
        a = open("/sys/fs/cgroup/cpu.max").read().split()
        if a[0] != "max":
            return float(a[0]) / float(a[1])
        return None
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / p if q > 0 else None
    except Exception:
        return None


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


# ------------------------------------------------------------------------------------------------
# synthetic corpora (tools/corpusgen.c)
# ------------------------------------------------------------------------------------------------
_words_cache = None


def _words():
    global _words_cache
    if _words_cache is None:
        ws = [w.encode() for w in open(WORDS_EN).read().split()]
        blob = np.frombuffer(b"".join(ws), dtype=np.uint8).copy()
        woff = np.zeros(len(ws) + 1, dtype=np.int32)
        np.cumsum([len(w) for w in ws], out=woff[1:])
        ranks = np.arange(1, len(ws) + 1, dtype=np.float64)
        p = ranks ** -1.07
        cdf = np.cumsum(p / p.sum())
        cdf[-1] = 1.0
        _words_cache = (blob, woff, cdf)
    return _words_cache


def gen_corpus(ndocs, seed=20240202, mean=128, sd=16, minlen=32, maxlen=256, loguniform=False, multibyte=False,
               first_doc=0, nthreads=None):
    """Deterministic synthetic corpus.  Returns (text uint8[total], doc_off int64[ndocs+1])."""
    L = ctypes.CDLL(CORPUSGEN_LIB)
    f = L.bfc_gen
    f.restype = ctypes.c_int64
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64,
                  ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                  ctypes.c_void_p, ctypes.c_int]
    blob, woff, cdf = _words()
    off = np.zeros(ndocs + 1, dtype=np.int64)
    args = [blob.ctypes.data, woff.ctypes.data, len(woff) - 1, cdf.ctypes.data, seed, first_doc, ndocs,
            1 if loguniform else 0, float(mean), float(sd), minlen, maxlen, 1 if multibyte else 0]
    total = f(*args, None, off.ctypes.data, 1)
    text = np.empty(total, dtype=np.uint8)
    f(*args, text.ctypes.data, off.ctypes.data, nthreads or min(host_threads(), 64))
    return text, off


_pieces_cache = {}
MULTI_BUCKETS = ["latin", "cyrillic", "cjk", "arabic", "devanagari", "greek", "thai"]
MULTI_BUCKET_P = [0.50, 0.15, 0.10, 0.08, 0.05, 0.04, 0.03]          # + 5 % "mixed" 


def _pieces(name):
    """tests/data/pieces_<name>.tsv.gz (tools/make_piece_lists.py) as flat arrays for tools/corpusgen.c bfc_gen_multi"""
    if name not in _pieces_cache:
        import gzip
        per = {b: [] for b in MULTI_BUCKETS}
        cm = []
        with gzip.open(os.path.join(ROOT, "tests", "data", "pieces_%s.tsv.gz" % name), "rb") as f:
            for line in f:
                b, piece = line.rstrip(b"\n").split(b"\t", 1)
                if b == b"charmap":
                    cm.append(piece)
                else:
                    per[b.decode()].append(piece)
        pieces, boff, cdf = [], [0], []
        for b in MULTI_BUCKETS:
            ps = per[b]
            pieces += ps
            boff.append(len(pieces))
            if ps:
                w = np.arange(1, len(ps) + 1, dtype=np.float64) ** -1.07
                c = np.cumsum(w / w.sum())
                c[-1] = 1.0
                cdf.append(c)
        blob = np.frombuffer(b"".join(pieces), dtype=np.uint8).copy()
        poff = np.zeros(len(pieces) + 1, dtype=np.int32)
        np.cumsum([len(x) for x in pieces], out=poff[1:])
        cm_blob = np.frombuffer(b"".join(cm) or b"\0", dtype=np.uint8).copy()
        cm_off = np.zeros(len(cm) + 1, dtype=np.int32)
        if cm:
            np.cumsum([len(x) for x in cm], out=cm_off[1:])
        bucket_cdf = np.cumsum(np.array(MULTI_BUCKET_P + [1.0 - sum(MULTI_BUCKET_P)], dtype=np.float64))
        bucket_cdf[-1] = 1.0
        _pieces_cache[name] = (blob, poff, np.array(boff, dtype=np.int32), np.concatenate(cdf), bucket_cdf, cm_blob, cm_off, len(cm))
    return _pieces_cache[name]


def gen_corpus_multi(ndocs, pieces="xlmr", seed=4, mean=512, sd=64, minlen=128, maxlen=1024, first_doc=0, nthreads=None):
    """Deterministic multilingual corpus.  Returns (text uint8[total], doc_off int64[ndocs+1])."""
    L = ctypes.CDLL(CORPUSGEN_LIB)
    f = L.bfc_gen_multi
    f.restype = ctypes.c_int64
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                  ctypes.c_int, ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    blob, poff, boff, cdf, bucket_cdf, cm_blob, cm_off, ncm = _pieces(pieces)
    off = np.zeros(ndocs + 1, dtype=np.int64)
    args = [blob.ctypes.data, poff.ctypes.data, boff.ctypes.data, len(MULTI_BUCKETS), cdf.ctypes.data, bucket_cdf.ctypes.data, cm_blob.ctypes.data,
            cm_off.ctypes.data, ncm, seed, first_doc, ndocs, float(mean), float(sd), minlen, maxlen]
    total = f(*args, None, off.ctypes.data, 1)
    text = np.empty(total, dtype=np.uint8)
    f(*args, text.ctypes.data, off.ctypes.data, nthreads or min(host_threads(), 64))
    return text, off


CONFIG1_LINES = os.path.join(ROOT, "tests", "data", "config1_lines.txt.gz")


def config1_lines(ndocs, first_doc=0):
    """BASELINE.json configs[0] / SURVEY.md section 8(d): the first 10,000 non-empty lines of the reference's own
    ldbsrc/bert_multi_cased/test.legacy.txt.zip:test.txt (a reference-held fixture, staged under tests/data/ by the snippet in
    tests/data/README.md: 427,735 bytes, 42.8 per line, real English, no RNG); more than 10,000 documents: the lines again from the top"""
    import gzip
    with gzip.open(CONFIG1_LINES, "rb") as f:
        lines = f.read().split(b"\n")[:-1]
    docs = [lines[(first_doc + i) % len(lines)] for i in range(ndocs)]
    off = np.zeros(ndocs + 1, dtype=np.int64)
    np.cumsum([len(d) for d in docs], out=off[1:])
    return np.frombuffer(b"".join(docs), dtype=np.uint8).copy(), off


def gen_workload(name, ndocs, first_doc=0):
    """the corpus of a named workload (WORKLOADS below): documents [first_doc, first_doc + ndocs)"""
    wl = WORKLOADS[name]
    if name == "config1" and os.path.exists(CONFIG1_LINES):
        return config1_lines(ndocs, first_doc)
    if wl.get("multi"):
        return gen_corpus_multi(ndocs, first_doc=first_doc, **wl["multi"])
    return gen_corpus(ndocs, first_doc=first_doc, **wl["gen"])


WORKLOADS = {
    # BASELINE.json configs[0]: the default pattern tokenizer (built-in wbd.bin), TextToWords on short English lines: the 10,000 lines SURVEY.md
    # section 8(d) names (tests/data/config1_lines.txt.gz, config1_lines() above); the generator only when that fixture is absent
    "config1": dict(model="wbd.bin", gen=dict(seed=1, mean=43, sd=12, minlen=8, maxlen=120), max_ids=0, unk=0),
    # name: generator kwargs + tokenizer call parameters
    "headline512": dict(model=None, gen=dict(seed=20240201, mean=512, sd=64, minlen=128, maxlen=1024), max_ids=512, unk=100),
    "config2": dict(model=None, gen=dict(seed=20240202, mean=128, sd=16, minlen=32, maxlen=256), max_ids=512, unk=100),
    "config3": dict(model="gpt2.bin", gen=dict(seed=3, minlen=32, maxlen=2048, loguniform=True, multibyte=True), max_ids=2048, unk=0),
    # configs 4 / 5: the multilingual generator of SURVEY.md section 8d over per-script piece lists derived from the models' vocabularies
    "config4": dict(model="xlm_roberta_base.bin", multi=dict(pieces="xlmr", seed=4, mean=512, sd=64, minlen=128, maxlen=1024), max_ids=1024, unk=3),
    "config5": dict(model="laser500k.bin", multi=dict(pieces="laser500k", seed=5, mean=512, sd=64, minlen=128, maxlen=1024), max_ids=1024, unk=0),
}


ADVERSARIAL = [
    b"a", b" ", b"   ", b"\xef\xbb\xbf", b"\xef\xbb\xbfhello", b"hello", b"   hello  ", b"Hello, world! This is a test.",
    b"Hello, world! This is a test of unaffable.", b"Hello unaffable qzxjkvw [UNK] world", b"Hello unaffable world again",
    b"ab\xff cd", b"ab\xff\xfecd", b"ab\x00cd", b"ab\x01\x02cd", "café naïve".encode(), b"a" * 400, "́".encode(),
    "Эpple pie.".encode(), "Sergei Alonichau I saw a girl with a \ttelescope.".encode(), "好好好 ok".encode(),
    "à la".encode(), b"a la", b"\xc0\xaf", b"\xe0\x80\xaf", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", b"\xf8\x88\x80\x80\x80",
    b"\xe2\x82", b"abc\xe2\x82", b"\x80abc", b"abc\x80", b"abc\xc3", "\U0001F600 smile \U00010000".encode(),
    "a b c‏d e f⁠g␠h␤i　j﻿k".encode(), "ª ﬁ ㍿".encode(),
    "▁▁ a ▁ b  ▁".encode(), b"[UNK] [CLS] [SEP] [MASK] [PAD]", b"don't U.S.A. e-mail 3,000.50", b"x" * 299 + b" " + b"y" * 301,
    ("word " * 200).encode(), b"\t\n\r\x0b\x0c", b"\x7f\x1f", "İstanbul ǅ ß".encode(), b"A" * 1025,
]


def fuzz_docs(n, seed=1, maxwords=60):
    """mixed adversarial documents: real words, control chars, astral code points, random bytes, broken UTF-8"""
    import random
    rnd = random.Random(seed)
    words = ("the of and to in a is that for it as was with be by on not he I this are or his from at which but have an had "
             "they you were their one all we can her has there been if more when will would who so no Hello unaffable qzxjkvw "
             "[UNK] [CLS] world café naïve 好好 好 привет мир Sergei Alonichau telescope . , ! ? ; : ' \" ( ) - 123 4.5 \t ▁ ﬁ ª ㍿ "
             "́ é Ünïcödé ß Straße İstanbul ǅ 𝒳 😀 don't U.S.A. e-mail 3,000.50 http://a.b/c?d=e&f x@y.com ##ing á "
             "​   ١٢٣ עברית").split(" ")
    alpha = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"
    docs = []
    while len(docs) < n:
        r = rnd.random()
        if r < 0.05:
            b = bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 40)))
        elif r < 0.1:
            b = "".join(rnd.choice(alpha) for _ in range(rnd.randint(250, 700))).encode()
        else:
            parts = []
            for _ in range(rnd.randint(0, maxwords)):
                q = rnd.random()
                if q < 0.08:
                    parts.append("".join(rnd.choice(alpha) for _ in range(rnd.randint(1, 20))))
                elif q < 0.11:
                    parts.append(chr(rnd.choice([0, 1, 2, 3, 0x7f, 0x85, 0x2028, 0xfeff, 0x10000, 0x10ffff, 0xe000, 0xd7ff])))
                else:
                    parts.append(rnd.choice(words))
            b = rnd.choice([" ", " ", "  ", "\n", ""]).join(parts).encode("utf-8")
            if rnd.random() < 0.05 and len(b) > 2:
                i = rnd.randrange(len(b))
                b = b[:i] + bytes([rnd.randrange(256)]) + b[i + 1:]
            if rnd.random() < 0.03:
                b = b"\xef\xbb\xbf" + b
        if b:
            docs.append(b)
    return docs
