"""Generates tests/golden/*.json from the UNMODIFIED reference (oracle/_ref/libblingfiretokdll_ref.so, compiled
from /root/reference by oracle/Makefile).  Run in the dev container; the JSON fixtures are committed so the GPU
box (where /root/reference does not exist) can still pin both the oracle and the HIP path to the reference.

  python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bfutil  # noqa: E402

CASES = [  # (model, max_ids, unk)
    ("bert_base_tok.bin", 128, 100), ("bert_base_cased_tok.bin", 64, 100), ("bert_chinese.bin", 64, 100), ("wbd.bin", 64, 0),
    ("gpt2.bin", 64, 0), ("roberta.bin", 64, 0), ("xlnet.bin", 64, 0), ("xlnet_nonorm.bin", 64, 0), ("bpe_example.bin", 64, 1),
    ("laser100k.bin", 64, 0), ("xlm_roberta_base.bin", 128, 0), ("laser500k.bin", 64, 0),
    ("uri100k.bin", 64, 0), ("uri100kint.bin", 64, 0), ("laser50k.bin", 64, 0), ("bpe_example2.bin", 64, 1),
]


def main():
    ref = bfutil.reference()
    docs = list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(150, seed=4242, maxwords=25)
    for model, max_ids, unk in CASES:
        if not bfutil.have_model(model):
            print("skip", model)
            continue
        h = ref.load(bfutil.model_path(model))
        rows = []
        for b in docs:
            for mx, uk in ((max_ids, unk), (3, unk)):
                c, buf = ref.text_to_ids(h, b, mx, uk)
                rows.append({"hex": b.hex(), "max": mx, "unk": uk, "count": c, "ids": buf[:max(c, 0)]})
        ref.free(h)
        out = os.path.join(HERE, model.replace(".bin", "") + ".json")
        json.dump({"model": model, "source": "oracle/_ref (unmodified reference v0.1.8)", "rows": rows}, open(out, "w"))
        print(model, len(rows), "rows ->", out)


def _text_fn(lib, name):
    import ctypes
    g = getattr(lib, name)
    g.restype = ctypes.c_int
    g.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    return g


def main_api():
    """Fixtures of the other entry points on the path: TextToIdsWithOffsets, TextToWords / TextToSentences (with offsets), IdsToText."""
    import ctypes
    import random
    ref = bfutil.reference()
    docs = list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(60, seed=777, maxwords=25)
    out = {"source": "oracle/_ref (unmodified reference v0.1.8)", "offsets": [], "words": [], "sentences": [], "ids_to_text": []}
    for model, max_ids, unk in (("bert_base_cased_tok.bin", 64, 100), ("gpt2.bin", 64, 0), ("xlm_roberta_base.bin", 64, 3), ("xlnet.bin", 64, 0)):
        h = ref.load(bfutil.model_path(model))
        for b in docs:
            c, ids, st, en = ref.with_offsets(h, b, max_ids, unk, "TextToIdsWithOffsets")
            out["offsets"].append({"model": model, "hex": b.hex(), "max": max_ids, "unk": unk, "count": c, "ids": ids[:max(c, 0)],
                                   "starts": st[:max(c, 0)], "ends": en[:max(c, 0)]})
        ref.free(h)
    for key, fn_name, models in (("words", "TextToWordsWithOffsetsWithModel", ("wbd.bin", None, "bert_base_cased_tok.bin")),
                                 ("sentences", "TextToSentencesWithOffsetsWithModel", ("sbd.bin", None))):
        g = _text_fn(ref.lib, fn_name)
        for model in models:
            h = ref.load(bfutil.model_path(model)) if model else None
            for b in docs:
                mx = 4 * len(b) + 8
                o = ctypes.create_string_buffer(max(mx, 1) + 4); s = (ctypes.c_int32 * max(mx, 1))(); e = (ctypes.c_int32 * max(mx, 1))()
                r = g(b, len(b), o, s, e, mx, ctypes.c_void_p(h) if h else None)
                txt = o.raw[:r] if 0 < r <= mx else b""
                k = (txt[:-1].count(b" " if key == "words" else b"\n") + 1) if r > 1 else 0
                out[key].append({"model": model, "hex": b.hex(), "ret": r, "out_hex": txt.hex(), "starts": list(s[:k]), "ends": list(e[:k])})
            if h:
                ref.free(h)
    g = ref.lib.IdsToText
    g.restype = ctypes.c_int
    g.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_bool]
    rng = random.Random(99)
    for model, cnt in (("gpt2.i2w", 50257), ("bert_base_cased_tok.i2w", 28996), ("xlnet.i2w", 32000)):
        h = ref.load(bfutil.model_path(model))
        for t in range(60):
            n = rng.choice([1, 2, 5, 20, 60])
            ids = [rng.randrange(0, cnt) for _ in range(n)]
            if t % 9 == 0:
                ids[rng.randrange(n)] = rng.choice([-1, cnt, cnt + 7])
            for skip in (False, True):
                arr = (ctypes.c_int32 * n)(*ids)
                o = ctypes.create_string_buffer(4096)
                r = g(ctypes.c_void_p(h), arr, n, o, 4096, skip)
                out["ids_to_text"].append({"model": model, "ids": ids, "skip": int(skip), "ret": r, "out_hex": o.raw[:max(r, 0)].hex()})
        ref.free(h)
    path = os.path.join(HERE, "api", "fixtures.json")
    json.dump(out, open(path, "w"))
    print({k: len(v) for k, v in out.items() if isinstance(v, list)}, "->", path)


if __name__ == "__main__":
    main()
    main_api()
