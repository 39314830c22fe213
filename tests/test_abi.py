"""The C-ABI library builds, loads and exports every symbol include/grape_hip.h declares (CPU only:
no compute calls).  Also checks that the product path has no route into oracle/."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "grape_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(grape_[a-z_]+)\s*\(", src)))


def test_library_builds_and_exports_header_symbols():
    import __graft_entry__ as entry
    entry.build()
    from grape_jl_amd import api
    lib = ctypes.CDLL(api.library_path())
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in grape_hip.h but not exported"
    assert sorted(api.EXPORTS) == names
    lib.grape_abi_version.restype = ctypes.c_int
    assert lib.grape_abi_version() == api.ABI_VERSION


def test_create_rejects_bad_problems_without_gpu():
    # argument validation happens before any HIP call, so it is testable on a CPU box
    import numpy as np
    from grape_jl_amd import api
    lib = api.load_library()
    p = api._Problem()
    p.abi_version = 99
    h = ctypes.c_void_p()
    assert lib.grape_create(ctypes.byref(h), ctypes.byref(p)) == -1
    assert b"abi_version" in lib.grape_last_error(None)
    p.abi_version = api.ABI_VERSION
    p.N, p.K, p.N_T, p.L = 4, 1, 3, 0
    assert lib.grape_create(ctypes.byref(h), ctypes.byref(p)) == -6  # "no controls" (workspace.jl:155-157)
    assert b"no controls" in lib.grape_last_error(None)
    p.L = 1
    assert lib.grape_create(ctypes.byref(h), ctypes.byref(p)) == -1  # null arrays
    _ = np


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "grape.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "grape_oracle" not in text and "grape_ref" not in text and "oracle/" not in text.replace(
                    "oracle/ ", ""), f"{f} references the oracle"


def test_problem_struct_layout_matches_the_header(tmp_path):
    """sizeof / offsetof of every grape_problem field as the C compiler lays it out (include/grape_hip.h) against the
    ctypes mirror in api.py, and the field list of the Julia mirror (julia/GrapeHIP.jl) against both."""
    import subprocess
    from grape_jl_amd import api
    fields = [f[0] for f in api._Problem._fields_]
    src = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "grape_hip.h")}"',
             'int main(void) {', '  printf("%zu\\n", sizeof(grape_problem));']
    lines += [f'  printf("{f} %zu\\n", offsetof(grape_problem, {f}));' for f in fields]
    lines += ['  return 0;', '}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    assert int(out[0]) == ctypes.sizeof(api._Problem)
    for line in out[1:]:
        if line.strip():
            name, off = line.split()
            assert getattr(api._Problem, name).offset == int(off), name
    # header field order == ctypes field order == Julia struct field order
    hdr = open(os.path.join(ROOT, "include", "grape_hip.h")).read()
    body = re.sub(r"/\*.*?\*/", "", hdr[hdr.index("typedef struct {"):hdr.index("} grape_problem;")], flags=re.S)
    hdr_fields = re.findall(r"\b(\w+)\s*;", body)
    assert hdr_fields == fields
    jl = open(os.path.join(ROOT, "julia", "GrapeHIP.jl")).read()
    jbody = jl[jl.index("struct GrapeProblem"):]
    jbody = jbody[:jbody.index("\nend")]
    jl_fields = re.findall(r"^\s+(\w+)::", jbody, flags=re.M)
    assert jl_fields == fields
