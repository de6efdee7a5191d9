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


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof of the two structs that cross the boundary, as a C compiler sees the header, against the ctypes
    mirrors in diffhe/_hip.py (the header is plain C: gcc compiles it without HIP)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    fields = {"diffhe_mg_level": [f for f, _ in _hip.MgLevel._fields_], "diffhe_amg_level": [f for f, _ in _hip.AmgLevel._fields_]}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for struct, names in fields.items():
        lines.append(f'  printf("{struct} %zu", sizeof({struct}));')
        for f in names:
            lines.append(f'  printf(" %zu", offsetof({struct}, {f}));')
        lines.append('  printf("\\n");')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    for line, (struct, cls) in zip(out, (("diffhe_mg_level", _hip.MgLevel), ("diffhe_amg_level", _hip.AmgLevel))):
        nums = [int(v) for v in line.split()[1:]]
        assert line.split()[0] == struct
        assert nums[0] == ctypes.sizeof(cls)
        assert nums[1:] == [getattr(cls, f).offset for f, _ in cls._fields_]
