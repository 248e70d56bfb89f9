"""Build guards for the code-generation defect behind round 2's "out-of-line device call" failure.

ROCm 7.2's hipcc miscompiles gfx950 kernels that CALL a device function under register pressure: the values that must survive the call are
saved by VGPR copies placed ahead of the `s_or_b64 exec` that ends an `if (tid < N)` region, so masked lanes are never saved and get
garbage back (tools/scan_exec_masked_saves.py explains and detects the pattern; tools/run_outline_abi.sh holds the GPU experiments).  The
sources therefore force every helper inline and the library is built with IPRA off; these tests make both properties CHECKED ones instead
of consequences of one compiler version's inliner.  They disassemble the gfx950 code objects of the shipped libsmo.so; no GPU needed."""
import os
import re
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT
from spheremanopt_amd import _capi

sys.path.insert(0, os.path.join(ROOT, "tools"))
import scan_exec_masked_saves as scan      # noqa: E402

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
OBJDUMP = os.path.join(scan.LLVM, "llvm-objdump")
needs_llvm = pytest.mark.skipif(not (os.path.exists(OBJDUMP) and os.path.exists(HIPCC)), reason="no ROCm LLVM tools in this image")


@needs_llvm
def test_shipped_library_has_no_device_calls_and_no_masked_saves():
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    rows = scan.scan_library(_capi.LIB_PATH)
    assert len(rows) >= 5, [r[0] for r in rows]                  # vecops, sh23, shb23, kdyn, pois
    assert sum(r[1] for r in rows) > 500                         # every instantiation was looked at (kernels only: see below)
    for co, nfun, hits, calls in rows:
        assert not calls, "%s: device-side call(s) at %s" % (co, calls[:3])
        assert not hits, "%s: VGPR writes ahead of an exec restore: %s" % (co, hits[:2])
    # ... and nothing but kernels in the device code: an out-of-line helper would show up as a function symbol without a kernel descriptor
    import tempfile
    with tempfile.TemporaryDirectory() as w:
        for co in scan.code_objects(_capi.LIB_PATH, w):
            syms = subprocess.run([OBJDUMP, "-t", co], capture_output=True, text=True, check=True).stdout
            funcs = {l.split()[-1] for l in syms.splitlines() if re.search(r"\sF\s+\.text", l)}
            kd = {l.split()[-1][:-3] for l in syms.splitlines() if l.rstrip().endswith(".kd")}
            assert funcs and not (funcs - kd), "%s: device functions that are not kernels: %s" % (os.path.basename(co), sorted(funcs - kd)[:3])


@needs_llvm
def test_library_is_built_without_ipra():
    mk = open(os.path.join(ROOT, "spheremanopt_amd", "csrc", "Makefile")).read()
    assert "-enable-ipra=0" in mk


@needs_llvm
def test_the_scanner_flags_the_round2_failure(tmp_path):
    """The guard itself, on the real thing: csrc/shb23.hip with dct3<0> marked noinline (variant B of tools/run_outline_abi.sh, which returns
    NaN on the GPU) must be flagged — 7 device calls and VGPR saves under a narrowed exec mask in the forward and Continuous-adjoint kernels;
    the same source with IPRA off (variant C, correct on the GPU) has the calls but no masked saves."""
    csrc = os.path.join(ROOT, "spheremanopt_amd", "csrc")
    src = open(os.path.join(csrc, "shb23.hip")).read()
    marked = src.replace("template <> __device__ __forceinline__ void dct3<0>", "template <> __device__ __attribute__((noinline)) void dct3<0>")
    assert marked != src
    f = tmp_path / "shb23_noinline.hip"
    f.write_text(marked.replace('"fft_lds.hpp"', '"%s/fft_lds.hpp"' % csrc))
    res = {}
    for name, extra in (("ipra", []), ("no_ipra", ["-mllvm", "-enable-ipra=0"])):
        obj = str(tmp_path / (name + ".o"))
        subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-I", csrc] + extra +
                       ["-c", str(f), "-o", obj], check=True, capture_output=True)
        (row,) = scan.scan_library(obj)
        res[name] = row
    assert len(res["ipra"][3]) == 7 and len(res["no_ipra"][3]) == 7            # the calls are there in both
    flagged = {h[0] for h in res["ipra"][2]}
    assert any("shb_forward_kernelILi0E" in k for k in flagged) and any("shb_adjoint_cnts_kernelILi0E" in k for k in flagged), flagged
    assert not res["no_ipra"][2]


@needs_llvm
def test_axpby_kernels_are_not_contracted_into_fma():
    """smo_vec_axpby promises NumPy's rounding, fl(fl(a*x) + fl(b*y)) (include/smo.h).  hipcc's default -ffp-contract=fast had turned the
    sum into v_fmac_f64 (one rounding less whenever neither factor is +-1); the kernels now pin the two-rounding form and this test keeps
    the ISA honest: no fused multiply-add in any vec_axpby kernel of the shipped library."""
    import tempfile
    found = 0
    with tempfile.TemporaryDirectory() as w:
        for co in scan.code_objects(_capi.LIB_PATH, w):
            dis = subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True, check=True).stdout
            cur = None
            for ln in dis.splitlines():
                m = re.match(r"^[0-9a-f]{16} <(\S+)>:", ln)
                if m:
                    cur = m.group(1)
                    found += "vec_axpby" in cur
                elif cur and "vec_axpby" in cur:
                    assert not re.search(r"\bv_(pk_)?fma?c?_f64\b|\bv_fma_f64\b|\bv_fmac_f64", ln), (cur, ln.strip())
    assert found >= 4                                            # <true>/<false> of the 16-byte and the 8-byte kernel
