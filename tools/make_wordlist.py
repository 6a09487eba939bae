"""Regenerates tests/data/words_en.txt (the sampling vocabulary of the synthetic corpora):
whole-word entries of the reference's ldbsrc/bert_base_tok/vocab.txt with id >= 1996, alphabetic ASCII, no '##'.
Run in the dev container only (needs /root/reference); the output is committed so the GPU box never needs it."""
import sys
src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/ldbsrc/bert_base_tok/vocab.txt"
out = sys.argv[2] if len(sys.argv) > 2 else "tests/data/words_en.txt"
words = []
for i, line in enumerate(open(src, encoding="utf-8")):
    w = line.rstrip("\n")
    if i >= 1996 and w.isascii() and w.isalpha() and not w.startswith("##"):
        words.append(w)
open(out, "w").write("\n".join(words) + "\n")
print(len(words), "words ->", out)
