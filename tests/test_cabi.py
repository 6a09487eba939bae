"""CPU: the product library loads without a GPU and exports every symbol include/blingfiretokdll_amd.h declares
(no compute calls here -- those need a device and live in the -m gpu tests)."""
import ctypes
import os
import re

import bfutil


def declared_symbols():
    src = open(os.path.join(bfutil.ROOT, "include", "blingfiretokdll_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))
    return sorted(set(re.findall(r"\b([A-Z][A-Za-z0-9_]+)\s*\(", src)))


def test_library_exports_declared_symbols():
    import blingfire_amd as bf
    assert os.path.exists(bf.LIB_PATH)
    L = ctypes.CDLL(bf.LIB_PATH)
    names = declared_symbols()
    assert {"LoadModel", "SetModel", "FreeModel", "TextToIds", "TextToIds_wp", "TextToIds_sp", "TextToIdsBatch",
            "TextToIdsBatchDevice", "GetBlingFireTokVersion", "SetNoDummyPrefix"} <= set(names)
    for n in names:
        assert hasattr(L, n), "missing export %s" % n


def test_reference_error_conventions_without_device():
    import blingfire_amd as bf
    L = bf.lib()
    assert L.GetBlingFireTokVersion() == 18000          # reference tokdll:36-37,108-111
    assert L.FreeModel(None) == 0                       # tokdll:1654-1656
    assert L.TextToIds(None, b"abc", 3, None, 8, 0) == 0  # NULL model -> 0 (tokdll:1629-1631)


def test_product_never_links_the_oracle():
    """no symbol of the oracle / reference checker is reachable from the product library"""
    import blingfire_amd as bf
    import subprocess
    out = subprocess.run(["nm", "-D", bf.LIB_PATH], capture_output=True, text=True).stdout
    assert "bfo_" not in out
    ldd = subprocess.run(["ldd", bf.LIB_PATH], capture_output=True, text=True).stdout
    assert "liboracle" not in ldd and "_ref" not in ldd


# blingfiretools/blingfiretokdll/blingfiretokdll.def:3-26 -- every symbol the reference library exports
REFERENCE_DEF = ["TextToSentences", "TextToWords", "TextToSentencesWithOffsets", "TextToWordsWithOffsets", "GetBlingFireTokVersion", "TextToHashes",
                 "LoadModel", "TextToIds", "FreeModel", "TextToSentencesWithOffsetsWithModel", "TextToSentencesWithModel",
                 "TextToWordsWithOffsetsWithModel", "TextToWordsWithModel", "TextToIds_sp", "TextToIds_wp", "TextToIdsWithOffsets_sp",
                 "TextToIdsWithOffsets_wp", "TextToIdsWithOffsets", "NormalizeSpaces", "SetModel", "WordHyphenationWithModel", "SetNoDummyPrefix",
                 "IdsToText"]


def exported_symbols():
    import blingfire_amd as bf
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", bf.LIB_PATH], capture_output=True, text=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if l.strip())


def test_every_reference_export_is_present():
    exp = set(exported_symbols())
    missing = [n for n in REFERENCE_DEF if n not in exp]
    assert not missing, missing
    ref_def = "/root/reference/blingfiretools/blingfiretokdll/blingfiretokdll.def"
    if os.path.exists(ref_def):          # dev container: the list above is the reference's
        names = [l.strip() for l in open(ref_def) if l.startswith("    ")]
        assert sorted(names) == sorted(REFERENCE_DEF)


def test_nothing_but_the_c_abi_is_exported():
    """-fvisibility=hidden + csrc/exports.map: no C++ symbols (kernel stubs, template instantiations), no data"""
    internal = re.findall(r"BF_API [^;(]*?\b([A-Za-z_][A-Za-z0-9_]*)\(", open(os.path.join(bfutil.ROOT, "blingfire_amd", "csrc", "bf_internal.h")).read())
    allowed = set(declared_symbols()) | set(internal)
    exp = exported_symbols()
    assert not [n for n in exp if n.startswith("_Z")], "C++ symbols leak from the drop-in library"
    assert set(exp) <= allowed, sorted(set(exp) - allowed)


def test_word_hyphenation_stub_fails_loudly():
    import blingfire_amd as bf
    L = bf.lib()
    L.WordHyphenationWithModel.restype = ctypes.c_int
    L.WordHyphenationWithModel.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    buf = ctypes.create_string_buffer(64)
    assert L.WordHyphenationWithModel(b"", 0, buf, 64, None, 0x2D) == 0       # tokdll:832-834
    assert L.WordHyphenationWithModel(b"hyphenation", 11, buf, 64, None, 0x2D) == -1
