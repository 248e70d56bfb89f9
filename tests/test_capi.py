"""The C-ABI library: loads, exports every symbol include/smo.h declares, and refuses to run without a GPU
(no CPU fallback).  No compute calls here — this file runs in the GPU-less build container."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from spheremanopt_amd import _capi


def _declared():
    hdr = open(os.path.join(ROOT, "include", "smo.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(smo_[a-z_]+)\s*\(", hdr)))


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _capi.lib()


def test_exports_match_header(L):
    names = _declared()
    assert len(names) >= 19
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_capi.EXPORTS) == names           # the Python binding covers exactly the header


def test_version_and_error_strings(L):
    assert b"libsmo" in L.smo_version()
    assert isinstance(L.smo_last_error(), bytes)


def test_bad_config_rejected(L):
    h = C.c_void_p()
    cfg = _capi.smo_config(_capi.SMO_SH23, 256, 0., 1., -1.0, 10, -0.3, 0, 1, 0, 0, 1, 1)      # dt < 0
    assert L.smo_create(C.byref(cfg), C.byref(h)) == 1 and not h.value
    assert b"bad config" in L.smo_last_error()
    cfg = _capi.smo_config(99, 256, 0., 1., 0.1, 10, -0.3, 0, 1, 0, 0, 1, 1)                     # unknown kind
    assert L.smo_create(C.byref(cfg), C.byref(h)) == 1
    assert L.smo_forward(None, None, None) == 1                                               # null context


def test_no_cpu_fallback(L):
    """Without a HIP device the product path must fail loudly, not compute on the CPU."""
    if _capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_capi.SmoError) as e:
        _capi.Context(_capi.SMO_SH23, 64, (0., 1.), 0.1, 4, -0.3)
    assert e.value.code == 2 and "no CPU fallback" in str(e.value)
    from spheremanopt_amd import sh23
    dom, X = sh23.Generate_IC(0.0725, Npts=64)
    with pytest.raises(_capi.SmoError):
        sh23.FWD_Solve_IVP_Lin([X], dom, 0.1, 4, 4, sh23.GEN_BUFFER(dom, 4))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "spheremanopt_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)


def test_header_is_plain_c_and_matches_the_ctypes_struct(tmp_path):
    """include/smo.h compiles as C99 and as C++; sizeof / field offsets of smo_config equal the ctypes mirror in _capi.py
    (and therefore the stub printed in INTEGRATION.md, which lists the same fields)."""
    import ctypes
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    fields = [f[0] for f in _capi.smo_config._fields_]
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "smo.h"\nint main(void){ printf("%zu", sizeof(smo_config));\n' +
                   "".join('printf(" %%zu", offsetof(smo_config, %s));\n' % f for f in fields) + "return 0; }\n")
    inc = os.path.join(ROOT, "include")
    exe = str(tmp_path / "abi")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", inc, str(src), "-o", exe], check=True)
    subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)], check=True)
    vals = [int(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    assert vals[0] == ctypes.sizeof(_capi.smo_config)
    assert vals[1:] == [getattr(_capi.smo_config, f).offset for f in fields]
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for f in fields:
        assert '("%s"' % f in doc, f


def _build_c_smoke(tmpdir):
    """gcc -std=c99 -Wall -Wextra -Werror: include/smo.h must be valid C, and a plain C program must link against libsmo.so."""
    import subprocess
    exe = os.path.join(str(tmpdir), "abi_smoke")
    libdir = os.path.join(ROOT, "spheremanopt_amd", "lib")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
           "-o", exe, "-L" + libdir, "-lsmo", "-lm", "-Wl,-rpath," + libdir]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return exe


def test_header_is_valid_c_and_a_c_program_links(tmp_path):
    assert os.path.exists(_build_c_smoke(tmp_path))


def _run_py(code, env_extra):
    import subprocess
    import sys
    env = dict(os.environ, **env_extra)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)


def test_missing_librccl_is_an_error_code_not_a_crash(L):
    """with no loadable librccl, smo_comm_unique_id must return SMO_ERR_UNSUPPORTED with a message (the old loader called
    dlerror() twice and built a std::string from NULL).  SMO_RCCL_LIB replaces the loader's candidate list; own process because the
    library is bound once per process."""
    code = ("import ctypes as C\nfrom spheremanopt_amd import _capi\nL = _capi.lib()\nb = C.create_string_buffer(128)\n"
            "rc = L.smo_comm_unique_id(b)\nprint('RC', rc, '|', L.smo_last_error().decode(), '|', repr(L.smo_comm_library().decode()))\n")
    p = _run_py(code, {"SMO_RCCL_LIB": "/nonexistent/librccl-not-here.so"})
    assert p.returncode == 0, p.stderr
    line = [l for l in p.stdout.splitlines() if l.startswith("RC")][0]
    assert line.startswith("RC 6 |"), line                        # SMO_ERR_UNSUPPORTED
    assert "cannot load librccl" in line and "librccl-not-here.so" in line
    assert line.rstrip().endswith("''")                            # and no library path is claimed


def test_bound_librccl_is_reported(L):
    """smo_comm_library(): the file the bound ncclGetUniqueId lives in (which of a process' RCCL copies carries the transposes)."""
    if not os.path.exists("/opt/rocm/lib/librccl.so.1"):
        pytest.skip("no system librccl")
    code = ("from spheremanopt_amd import _capi\nprint('LIB', _capi.lib().smo_comm_library().decode())\n")
    p = _run_py(code, {"SMO_RCCL_LIB": "/opt/rocm/lib/librccl.so.1"})
    assert p.returncode == 0, p.stderr
    line = [l for l in p.stdout.splitlines() if l.startswith("LIB")][0]
    assert "librccl" in line and os.path.exists(line.split(None, 1)[1])


def test_product_library_directory_holds_only_the_product(L):
    """experimental (deliberately miscompiled) builds live under tools/bin/ or xp_tmp/, never beside the product
    library, where a glob or a stray SMO_LIB could pick them up."""
    libdir = os.path.join(ROOT, "spheremanopt_amd", "lib")
    assert sorted(f for f in os.listdir(libdir) if not f.startswith(".")) == ["libsmo.so"]
