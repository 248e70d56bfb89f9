"""The ctypes stub printed in INTEGRATION.md (section 2) is executed as written (only the problem size is reduced) and must agree with
the shipped binding."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_documented_ctypes_stub_runs():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "class smo_config" in b]
    assert len(stub) == 1
    code = stub[0].replace('"spheremanopt_amd/lib/libsmo.so"', repr(os.path.join(ROOT, "spheremanopt_amd", "lib", "libsmo.so")))
    code = code.replace("smo_config(3, 128, 0.0, 2*np.pi, 1e-3, 1000,", "smo_config(3, 16, 0.0, 2*np.pi, 1e-3, 5,")
    assert "smo_config(3, 16," in code
    from spheremanopt_amd import _capi, kdyn
    _capi.lib()                                               # same HIP runtime preload the shipped binding performs
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    dom, B, U = kdyn.Generate_IC(16, U_Noise=True)
    J = ns["FWD_Solve_IVP_Lin"]([B, U])
    g = ns["ADJ_Solve_IVP_Lin"]([B, U])
    ip = ns["Inner_Prod_3"](B, g[0])
    buf = kdyn.GEN_BUFFER(16, dom, 5)
    args = [dom, 1., 1e-3, 5, 5, buf, "Final", "Discrete"]
    assert J == kdyn.FWD_Solve_IVP_Lin([B, U], *args)
    g2 = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
    assert np.array_equal(g[0], g2[0]) and np.array_equal(g[1], g2[1])
    assert ip == kdyn.Inner_Prod_3(B, g2[0], dom)
    # section 2c: the same stub on a multi-device context (ONE process, the box's GPU listed twice), as documented
    multi = [b for b in blocks if "smo_create_multi" in b]
    assert len(multi) == 1
    mcode = multi[0].replace("smo_config(3, 128, 0.0, 2*np.pi, 1e-3, 1000,", "smo_config(3, 16, 0.0, 2*np.pi, 1e-3, 5,")
    assert "smo_config(3, 16," in mcode
    exec(compile(mcode, "INTEGRATION.md#2c", "exec"), ns)
    Jm = ns["FWD_Solve_IVP_Lin_multi"]([B, U])
    gm = ns["ADJ_Solve_IVP_Lin_multi"]([B, U])
    assert abs(Jm - J) <= 1e-13 * abs(J)
    assert np.linalg.norm(gm[0] - g[0]) <= 1e-12 * np.linalg.norm(g[0]) and np.linalg.norm(gm[1] - g[1]) <= 1e-12 * np.linalg.norm(g[1])
    ns["L"].smo_destroy(ns["mctx"])
    ns["L"].smo_destroy(ns["ctx"])
