#!/usr/bin/env python3
"""Derives the compact per-script piece lists behind the multilingual corpora of BASELINE.json configs[3] / [4]
("per doc pick a script bucket ... emit pieces sampled from pos.dict.utf8 entries of that script
(Zipf over the piece's score rank), U+2581 -> space; 1 % chars drawn from charmap keys").

Runs ONCE in the dev container (it reads the vocabulary sources of the reference checkout, which do not travel to the GPU
box); the output is committed under tests/data/ and is all the generator (tools/corpusgen.c bfc_gen_multi) needs:

    python tools/make_piece_lists.py            # -> tests/data/pieces_xlmr.tsv.gz, tests/data/pieces_laser500k.tsv.gz

File format (UTF-8, gzip): one line per entry, `bucket<TAB>piece`, pieces of a bucket in vocabulary order (= descending
unigram score, the rank the Zipf draw uses).  Buckets: latin cyrillic cjk arabic devanagari greek thai, plus `charmap`
(single characters that are keys of the model's charmap: the 1 % normalisation exercise; empty for models without one).
Pieces without a letter (punctuation, digits, the bare U+2581) are listed in EVERY script bucket at their own rank.
"""
import gzip
import io
import os
import sys
import zipfile

REF = os.environ.get("BF_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PER_BUCKET = 6000
CHARMAP_KEYS = 1500
COMMON_MAX = 400        # script-neutral pieces (digits, ASCII punctuation) shared by every bucket: the 400 most frequent

RANGES = {
    "latin": [(0x41, 0x5A), (0x61, 0x7A), (0xC0, 0x24F), (0x1E00, 0x1EFF)],
    "cyrillic": [(0x400, 0x52F)],
    "cjk": [(0x3040, 0x30FF), (0x3400, 0x4DBF), (0x4E00, 0x9FFF), (0xAC00, 0xD7AF)],
    "arabic": [(0x600, 0x6FF), (0x750, 0x77F)],
    "devanagari": [(0x900, 0x97F)],
    "greek": [(0x370, 0x3FF), (0x1F00, 0x1FFF)],
    "thai": [(0xE00, 0xE7F)],
}
ORDER = ["latin", "cyrillic", "cjk", "arabic", "devanagari", "greek", "thai"]


def script_of(ch):
    c = ord(ch)
    for name in ORDER:
        for lo, hi in RANGES[name]:
            if lo <= c <= hi:
                return name
    return None


def classify(piece):
    """bucket of a piece: the script all its non-ASCII characters and ASCII letters share; None = mixed / other script;
    'common' = ASCII digits / punctuation (and the bare U+2581) only"""
    seen = set()
    for ch in piece:
        if ch == "\u2581":
            continue
        if ord(ch) < 0x80:
            if ch.isalpha():
                seen.add("latin")
            continue
        s = script_of(ch)
        if s is None:
            return None
        seen.add(s)
    if not seen:
        return "common"
    return seen.pop() if len(seen) == 1 else None


def read_vocab(zip_path, member=None):
    z = zipfile.ZipFile(zip_path)
    name = member or z.namelist()[0]
    pieces = []
    for line in io.TextIOWrapper(z.open(name), encoding="utf-8", newline="\n"):
        f = line.rstrip("\n").split("\t")
        if f and f[0] and not (f[0].startswith("<") and f[0].endswith(">")):
            pieces.append(f[0])
    return pieces


def charmap_keys(path, limit):
    """single code points on the left-hand side of the reference's charmap.utf8 (`\\xHEX` or literal), in file order"""
    keys = []
    if not path or not os.path.exists(path):
        return keys
    for line in open(path, encoding="utf-8"):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        lhs = line.split(" ")[0].split("\t")[0]
        try:
            ch = chr(int(lhs[2:], 16)) if lhs.startswith("\\x") else lhs
        except ValueError:
            continue
        if len(ch) == 1 and 0x80 <= ord(ch) <= 0x10FFFF and not (0xD800 <= ord(ch) <= 0xDFFF):
            keys.append(ch)
    step = max(1, len(keys) // limit)
    return keys[::step][:limit]


def build(name, pieces, cm_path):
    buckets = {b: [] for b in ORDER}
    ncommon = 0
    for p in pieces:
        if any(ord(ch) < 0x20 for ch in p) or "\t" in p or "\n" in p:
            continue
        b = classify(p)
        if b == "common":
            ncommon += 1
            if ncommon > COMMON_MAX:
                continue
            for k in ORDER:
                if len(buckets[k]) < PER_BUCKET:
                    buckets[k].append(p)
        elif b in buckets and len(buckets[b]) < PER_BUCKET:
            buckets[b].append(p)
    out = os.path.join(ROOT, "tests", "data", "pieces_%s.tsv.gz" % name)
    with gzip.GzipFile(out, "wb", mtime=0) as g:
        for b in ORDER:
            for p in buckets[b]:
                g.write(("%s\t%s\n" % (b, p)).encode("utf-8"))
        for ch in charmap_keys(cm_path, CHARMAP_KEYS):
            g.write(("charmap\t%s\n" % ch).encode("utf-8"))
    print(out, {b: len(buckets[b]) for b in ORDER}, os.path.getsize(out), "bytes")


def main():
    ld = os.path.join(REF, "ldbsrc")
    build("xlmr", read_vocab(os.path.join(ld, "xlm_roberta_base", "pos.dict.utf8.zip")), os.path.join(ld, "xlm_roberta_base", "charmap.utf8"))
    build("laser500k", read_vocab(os.path.join(ld, "laser500k", "laser_500k.vocab.zip")), None)


if __name__ == "__main__":
    main()
