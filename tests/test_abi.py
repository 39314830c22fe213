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


def _tiny_problem(api):
    import numpy as np
    N, K, L, N_T = 4, 1, 1, 3
    keep = dict(tlist=np.linspace(0.0, 1.0, N_T + 1), H0=np.zeros((K, N, N), complex), Hc=np.zeros((L, N, N), complex),
                psi0=np.ones((K, N), complex), target=np.ones((K, N), complex))
    p = api._Problem()
    p.abi_version, p.N, p.K, p.K_total, p.N_T, p.L = api.ABI_VERSION, N, K, K, N_T, L
    for name, arr in keep.items():
        setattr(p, name, arr.ctypes.data)
    return p, keep


def test_exception_barrier_of_grape_create(monkeypatch):
    """A host-side C++ exception inside grape_create must come back as a status (GRAPE_ERR_HOST) with a message, never
    cross the extern "C" boundary (std::terminate would abort this very test process).  The fault is injected by the
    library's test hook BEFORE the first HIP call, so the barrier is exercised on a CPU box; the GPU suite repeats it at
    the point where the handle already owns device memory (tests/test_gpu_boundary.py)."""
    from grape_jl_amd import api
    lib = api.load_library()
    p, keep = _tiny_problem(api)
    h = ctypes.c_void_p()
    monkeypatch.setenv("GRAPE_TEST_HOOKS", "1")
    monkeypatch.setenv("GRAPE_TEST_THROW_AT", "early")
    monkeypatch.setenv("GRAPE_TEST_THROW", "bad_alloc")
    assert lib.grape_create(ctypes.byref(h), ctypes.byref(p)) == -8
    assert not h.value
    assert b"bad_alloc" in lib.grape_last_error(None) and b"C boundary" in lib.grape_last_error(None)
    monkeypatch.setenv("GRAPE_TEST_THROW", "something else went wrong")
    assert lib.grape_create(ctypes.byref(h), ctypes.byref(p)) == -8
    assert b"something else went wrong" in lib.grape_last_error(None)
    # without the hook switch the variables are inert: the call gets as far as the device (none here: a HIP error status,
    # or a handle on a GPU box)
    monkeypatch.delenv("GRAPE_TEST_HOOKS")
    rc = lib.grape_create(ctypes.byref(h), ctypes.byref(p))
    assert rc in (0, -2)
    if rc == 0:
        lib.grape_destroy(h)
    assert api.STATUS[-8] == "GRAPE_ERR_HOST"
    _ = keep


def test_every_entry_point_is_an_exception_barrier():
    """source-level: every `extern "C"` function of the host file that returns a status is a function-try-block ending in
    the barrier macro (grape_destroy / grape_last_error / grape_abi_version cannot throw: no allocation)"""
    src = open(os.path.join(ROOT, "grape.jl_amd", "csrc", "grape_hip.hip")).read()
    names = [n for n in _declared() if n not in ("grape_destroy", "grape_last_error", "grape_abi_version")]
    for n in names:
        m = re.search(r"\nint %s\([^{;]*\) try \{" % n, src)
        assert m, f"{n} is not a function-try-block"
        tail = src[m.end():]
        end = tail.index("\n}\n")
        assert tail[end + 3:].startswith("GRAPE_BARRIER("), n


def test_concurrent_builds_do_not_race(tmp_path):
    """several ranks importing at once: build_library is serialised by a lock and hands out a complete library"""
    import subprocess
    import sys
    from grape_jl_amd import api
    api.build_library()
    before = os.path.getmtime(api.library_path())
    code = "import sys; sys.path.insert(0, %r); from grape_jl_amd import api; print(api.build_library())" % ROOT
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True) for _ in range(3)]
    outs = [p.communicate()[0].strip() for p in procs]
    assert all(p.returncode == 0 for p in procs) and all(o == api.library_path() for o in outs)
    assert os.path.getmtime(api.library_path()) == before          # up to date: nobody rebuilt
    assert not [d for d in os.listdir(os.path.join(ROOT, "grape.jl_amd", "csrc")) if d.startswith("_build_")]


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
