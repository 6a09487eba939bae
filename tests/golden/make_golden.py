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


if __name__ == "__main__":
    main()
