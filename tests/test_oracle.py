"""CPU: pins the oracle (plain-C restatement) to the reference -- known-answer vectors held by the reference's own
docs/comments, committed golden fixtures generated from the compiled reference, and (when oracle/_ref is present)
a live differential run."""
import json
import os

import pytest

import bfutil

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# (model, unk, max, text, expected ids)   sources: reference README.md / tokdll comments / SURVEY.md Appendix C
KATS = [
    ("bert_base_tok.bin", 100, 128,
     "Эpple pie. How do I renew my virtual smart card?: /Microsoft IT/ 'virtual' smart card certificates for DirectAccess are valid "
     "for one year. In order to get to microsoft.com we need to type pi@1.2.1.2.",
     [1208, 9397, 2571, 11345, 1012, 2129, 2079, 1045, 20687, 2026, 7484, 6047, 4003, 1029, 1024, 1013, 7513, 2009, 1013, 1005, 7484,
      1005, 6047, 4003, 17987, 2005, 3622, 6305, 9623, 2015, 2024, 9398, 2005, 2028, 2095, 1012, 1999, 2344, 2000, 2131, 2000, 7513,
      1012, 4012, 2057, 2342, 2000, 2828, 14255, 1030, 1015, 1012, 1016, 1012, 1015, 1012, 1016, 1012]),           # README.md:111-141
    ("bert_base_tok.bin", 100, 64, "Эpple pie.", [1208, 9397, 2571, 11345, 1012]),                                 # tokdll:1103-1106
    ("xlm_roberta_base.bin", 0, 128,
     "Autophobia, also called monophobia, isolophobia, or eremophobia, is the specific phobia of isolation. I saw a girl with a "
     "telescope. Я увидел девушку с телескопом.",
     [4396, 22014, 9166, 4, 2843, 35839, 22460, 22014, 9166, 4, 83, 7537, 22014, 9166, 4, 707, 6, 102835, 22014, 9166, 4, 83, 70, 29458,
      53073, 9166, 111, 6, 219488, 5, 87, 24124, 10, 23040, 678, 10, 5501, 70820, 5, 1509, 79132, 29513, 105, 46009, 135, 18293, 41333,
      419, 5]),                                                                                                   # README.md:232-268
    ("xlnet.bin", 0, 64, "Sergei Alonichau I saw a girl with a \ttelescope.",
     [14363, 651, 7201, 25263, 35, 685, 24, 1615, 33, 24, 16163, 9]),                                              # tokdll:1341-1347
    ("xlnet.bin", 0, 64, "好好好 ok", [17, 0, 17, 3518]),
    ("gpt2.bin", 0, 64, "Hello, world! This is a test.", [18435, 11, 995, 0, 770, 318, 257, 1332, 13]),
    ("roberta.bin", 0, 64, "Hello, world! This is a test.", [20920, 6, 232, 328, 152, 16, 10, 1296, 4]),
    ("gpt2.bin", 0, 64, "à la", [6184, 8591]),
    ("gpt2.bin", 0, 64, "a la", [257, 8591]),
    ("gpt2.bin", 0, 64, b"ab\xff\xfecd", [450, 187, 186, 10210]),
    ("gpt2.bin", 0, 64, b"\xef\xbb\xbfhello", [23748]),
    ("gpt2.bin", 0, 64, "   hello  ", [23748]),
    ("gpt2.bin", 0, 64, " ", [220]),
    ("gpt2.bin", 0, 64, b"\xef\xbb\xbf", []),
    ("gpt2.bin", 0, 64, " pedia", [50236 - 1 + 1 - 1]),   # ldbsrc/gpt2/README.TXT:40-47: pos-dict entry 'pedia' has id 50236; TextToIds adds id-offset -1
    ("bert_base_cased_tok.bin", 100, 16, "Hello, world! This is a test of unaffable.",
     [8667, 117, 1362, 106, 1188, 1110, 170, 2774, 1104, 8362, 9823, 8057, 2165, 119]),
    ("bert_base_cased_tok.bin", 100, 64, "Hello unaffable qzxjkvw [UNK] world",
     [8667, 8362, 9823, 8057, 2165, 186, 1584, 1775, 17187, 1964, 2246, 100, 1362]),
    ("bert_base_cased_tok.bin", 100, 3, "Hello unaffable world again", [8667, 8362, 9823]),
    ("bert_base_cased_tok.bin", 100, 64, b"ab\xff cd", []),
    ("bert_base_cased_tok.bin", 100, 64, b"ab\x00cd", [170, 1830, 172, 1181]),
    ("bert_base_cased_tok.bin", 100, 64, "café naïve", [20583, 9468, 28203, 2707]),
    ("bert_base_cased_tok.bin", 100, 64, "a" * 400, [170] + [22118] * 63),
    ("bert_base_cased_tok.bin", 100, 64, "   ", []),
    ("bert_base_cased_tok.bin", 100, 64, "́", []),
    ("laser500k.bin", 0, 64, "Hello world, this is a test. Привет мир", [83744, 9393, 4, 3306, 107, 11, 4469, 3, 2458, 83657, 11552]),
]


@pytest.fixture(scope="module")
def ora():
    return bfutil.oracle()


@pytest.mark.parametrize("i", range(len(KATS)))
def test_known_answer_vectors(ora, i):
    model, unk, mx, text, want = KATS[i]
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)
    if text == " pedia":
        pytest.skip("documentation-only vector (pos-dict entry, not a TextToIds output)")
    b = text.encode("utf-8") if isinstance(text, str) else text
    h = ora.load(bfutil.model_path(model))
    c, buf = ora.text_to_ids(h, b, mx, unk)
    ora.free(h)
    assert c == len(want) and buf[:c] == want
    assert all(v == -7 for v in buf[c:]), "ids beyond the count must stay untouched"


GOLDEN_FILES = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".json") and f != "tolower_pairs.json")      # (that one is the fold table: tests/test_ignore_case.py)


@pytest.mark.parametrize("fname", GOLDEN_FILES)
def test_oracle_matches_committed_reference_fixtures(ora, fname):
    g = json.load(open(os.path.join(GOLDEN, fname)))
    if not bfutil.have_model(g["model"]):
        pytest.skip("%s not present" % g["model"])
    h = ora.load(bfutil.model_path(g["model"]))
    for r in g["rows"]:
        b = bytes.fromhex(r["hex"])
        c, buf = ora.text_to_ids(h, b, r["max"], r["unk"])
        assert c == r["count"] and buf[:max(c, 0)] == r["ids"], (g["model"], b[:60])
    ora.free(h)


@pytest.mark.skipif(not bfutil.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("model", ["bert_base_tok.bin", "bert_base_cased_tok.bin", "wbd.bin", "gpt2.bin", "roberta.bin", "xlnet.bin",
                                   "bpe_example.bin", "xlm_roberta_base.bin", "laser500k.bin", "bert_chinese.bin"])
def test_oracle_vs_live_reference(ora, model):
    if not bfutil.have_model(model):
        pytest.skip("%s not present" % model)
    ref = bfutil.reference()
    ho, hr = ora.load(bfutil.model_path(model)), ref.load(bfutil.model_path(model))
    docs = list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(1500, seed=77)
    for k, b in enumerate(docs):
        mx = (0, 1, 3, 16, 64, 512)[k % 6]
        unk = (0, 100, 3, 257)[k % 4]
        assert ora.text_to_ids(ho, b, mx, unk) == ref.text_to_ids(hr, b, mx, unk), (model, b[:60], mx, unk)
    ora.free(ho)
    ref.free(hr)
