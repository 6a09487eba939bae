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
