"""CPU-side checks of the drop-in boundary: libdr3hip.so loads without a GPU, exports every entry
point include/dr3hip.h declares, fails loudly (no CPU fallback) and keeps the reference's literals."""
import ctypes as C
import os
import re
import subprocess

import pytest


def _header_functions(dr3):
    txt = open(dr3.HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dr3hip_[a-z0-9_]+)\s*\(", txt)))


def test_library_loads_and_exports_every_declared_symbol(dr3):
    L = dr3.lib()                                   # dlopen + bind: AttributeError if one is missing
    declared = _header_functions(dr3)
    assert len(declared) >= 30
    assert declared == dr3.exported_symbols()       # the Python mirror binds exactly the header's set
    out = subprocess.check_output(["nm", "-D", "--defined-only", dr3.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (dr3hip_[a-z0-9_]+)", out))
    assert set(declared) <= exported
    assert L.dr3hip_version() == 104


def test_no_cpu_fallback(dr3):
    L = dr3.lib()
    if L.dr3hip_device_count() > 0:
        pytest.skip("a GPU is visible; the failure path is covered on the CPU container")
    h = C.c_void_p()
    assert L.dr3hip_init(0, C.byref(h)) == -2       # DR3HIP_ENODEV
    assert b"no CPU fallback" in L.dr3hip_last_error(None)
    with pytest.raises(dr3.Dr3HipError):
        dr3.Context(0)


def test_reference_literals_and_level_rule(dr3):
    p = dr3.lk_params_reference()                   # src/initialization.cpp:593-613
    assert (p.win_w, p.win_h, p.max_level, p.crit_type, p.max_iters, p.flags) == (30, 30, 4, 3, 1000, 4)
    assert p.eps == 1e-3 and p.min_eig_threshold == 1e-4
    # host-only geometry helper needs no GPU
    assert dr3.lk_effective_max_level(1241, 376, 30, 30, 4) == 3
    assert dr3.lk_effective_max_level(1241, 376, 15, 15, 3) == 3
    assert dr3.lk_effective_max_level(1920, 1080, 21, 21, 4) == 4


def test_constants_match_oracle_and_header(dr3, O):
    txt = open(dr3.HEADER).read()
    for name, val in [("DR3HIP_8UC1", dr3.T_8UC1), ("DR3HIP_8UC3", dr3.T_8UC3), ("DR3HIP_32FC4", dr3.T_32FC4),
                      ("DR3HIP_LK_USE_INITIAL_FLOW", dr3.LK_USE_INITIAL_FLOW), ("DR3HIP_WARP_INVERSE_MAP", dr3.WARP_INVERSE_MAP),
                      ("DR3HIP_HALF_REF_X86", dr3.HALF_REF_X86), ("DR3HIP_PROF_PYR_L0", dr3.PROF_PYR_L0)]:
        m = re.search(r"#define\s+%s\s+(-?\d+)" % name, txt)
        assert m and int(m.group(1)) == val, name
    assert (dr3.T_8UC1, dr3.T_8UC3, dr3.T_32FC4) == (O.T_8UC1, O.T_8UC3, O.T_32FC4)
    assert C.sizeof(dr3.LKParams) == C.sizeof(O.LKParams) == 48


def test_product_never_touches_the_oracle():
    """the oracle and the bench's launch-path stand-in are test infrastructure: nothing under 3dr_amd/ or include/ may
    reference either"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for base in ("3dr_amd", "include"):
        for dp, _, files in os.walk(os.path.join(root, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"dr3_oracle|libdr3oracle|from oracle|import oracle|dr3o_|bench_standin|from tests|import tests", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
    out = subprocess.check_output(["ldd", os.path.join(root, "3dr_amd", "libdr3hip.so")], text=True)
    assert "oracle" not in out
