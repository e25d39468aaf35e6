"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/nrays_abi.h declares, and the ctypes mirror matches the C struct layout."""
import ctypes as C
import os
import re
import subprocess
import tempfile

from nrays_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "nrays_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nrays_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(built):
    lib = abi.load_hip_lib()
    names = declared_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), "libnrays_hip.so does not export %s" % n
        assert n in abi.HIP_SYMBOLS, "ctypes signature missing for %s" % n
    assert sorted(abi.HIP_SYMBOLS) == names


def test_abi_version_and_error_string(built):
    lib = abi.load_hip_lib()
    assert lib.nrays_abi_version() == abi.ABI_VERSION
    assert lib.nrays_last_error() is not None


def test_struct_layout_matches_c(built):
    """Compiles a tiny C program printing sizeof/offsetof of every ABI struct."""
    structs = {"NraysLight": abi.NraysLight, "NraysTexture": abi.NraysTexture, "NraysMaterial": abi.NraysMaterial,
               "NraysMesh": abi.NraysMesh, "NraysNode": abi.NraysNode, "NraysSceneDesc": abi.NraysSceneDesc,
               "NraysRenderParams": abi.NraysRenderParams, "NraysStats": abi.NraysStats}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "nrays_abi.h"', 'int main(void){']
    for name, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (name, name))
        for f, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (name, f, name, f))
    lines.append('return 0;}')
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "l.c")
        open(src, "w").write("\n".join(lines))
        exe = os.path.join(d, "l")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", exe, src])
        out = subprocess.check_output([exe]).decode().split("\n")
    got = dict(l.split() for l in out if l)
    for name, cls in structs.items():
        assert int(got[name]) == C.sizeof(cls), name
        for f, _ in cls._fields_:
            assert int(got["%s.%s" % (name, f)]) == getattr(cls, f).offset, (name, f)


def test_tile_rows_helper(built):
    lib = abi.load_hip_lib()
    p = abi.NraysRenderParams()
    p.width, p.height = 64, 100
    assert lib.nrays_tile_rows(C.byref(p)) == 100
    p.band_rows, p.band_owners, p.band_owner = 16, 4, 1
    # 100 rows -> 7 bands of 16 -> 2 bands per owner (padded) -> 32 rows
    assert lib.nrays_tile_rows(C.byref(p)) == 32


def test_null_arguments_are_errors_not_crashes(built):
    lib = abi.load_hip_lib()
    assert lib.nrays_scene_create(None, None) == abi.ERR_BAD_ARG
    assert lib.nrays_render(None, None, None) == abi.ERR_BAD_ARG
    assert b"null" in lib.nrays_last_error()
    lib.nrays_scene_destroy(None)


def test_product_package_never_references_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "nrays_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "nrays_oracle" not in text and "import oracle" not in text and "from oracle" not in text, \
                    "%s references the oracle" % os.path.join(dirpath, f)
