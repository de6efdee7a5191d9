"""The C-ABI library loads and exports every symbol include/diffhe_hip.h declares
(no compute call: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

from diffhe import _hip

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "diffhe_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(diffhe_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_hip.SIGNATURES)


@pytest.mark.skipif(not os.path.exists(_hip.LIB_PATH), reason="libdiffhe_hip.so not built (run __graft_entry__.build())")
def test_library_exports_every_declared_symbol():
    handle = ctypes.CDLL(_hip.LIB_PATH)
    for name in _declared():
        assert hasattr(handle, name), name
    L = _hip.lib()
    assert L.diffhe_abi_version() == _hip.ABI_VERSION
    assert L.diffhe_status_string(0) == b"ok"
    assert b"batch" in L.diffhe_status_string(-4)
    assert L.diffhe_cg_workspace_doubles(1000, 64) > 4 * 1000 * 64


def test_binding_and_header_agree_on_the_abi_version():
    import re
    m = re.search(r"#define\s+DIFFHE_ABI_VERSION\s+(\d+)", open(HEADER).read())
    assert m and int(m.group(1)) == _hip.ABI_VERSION
