"""CPU: static checks of the LuaJIT-FFI shim (mc-cnn_amd/lua/adcensus.lua).  LuaJIT / Torch7 are not in the
image, so the file cannot be executed here; what CAN go wrong silently -- a prototype in its ffi.cdef block
drifting from include/mc_adcensus.h, the ABI version constant, a library function used but never declared,
the mc_params field order -- is checked against the header token by token."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LUA = os.path.join(ROOT, "mc-cnn_amd", "lua", "adcensus.lua")
HDR = os.path.join(ROOT, "include", "mc_adcensus.h")


def _strip_c(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def _decls(src):
    """{name: normalised declaration} for every `... mc_xxx(...);` prototype"""
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?\bmc_[a-z0-9_]+\s*\([^;{]*\))\s*;", src, flags=re.S):
        d = re.sub(r"\s+", " ", m.group(1)).strip()
        d = re.sub(r"\s*([\*\(\),])\s*", r"\1", d)
        name = re.search(r"\b(mc_[a-z0-9_]+)\(", d).group(1)
        out[name] = d
    return out


def _struct_fields(src):
    m = re.search(r"typedef struct mc_params \{(.*?)\} mc_params;", src, flags=re.S)
    assert m, "mc_params not found"
    return [re.sub(r"\s+", " ", f).strip() for f in m.group(1).split(";") if f.strip()]


def _cdef(lua):
    m = re.search(r"ffi\.cdef\[\[(.*?)\]\]", lua, flags=re.S)
    assert m, "no ffi.cdef block"
    return m.group(1)


def test_cdef_prototypes_match_the_header():
    lua = open(LUA).read()
    hdr = _decls(_strip_c(open(HDR).read()))
    shim = _decls(_strip_c(_cdef(lua)))
    assert len(shim) >= 25
    for name, d in shim.items():
        assert name in hdr, "%s is declared in the shim but not in include/mc_adcensus.h" % name
        assert d == hdr[name], "prototype drift for %s:\n shim  : %s\n header: %s" % (name, d, hdr[name])


def test_every_lib_call_is_declared():
    lua = open(LUA).read()
    declared = set(_decls(_strip_c(_cdef(lua))))
    used = set(re.findall(r"\blib\.(mc_[a-z0-9_]+)", lua))
    assert used, "the shim calls nothing?"
    assert used <= declared, "used but not declared in ffi.cdef: %s" % sorted(used - declared)
    # the fused entry point that bench.py times is reachable from Lua
    assert {"mc_predict", "mc_predict_workspace_bytes"} <= used


def test_abi_version_constant_matches_the_header():
    lua = open(LUA).read()
    hdr = open(HDR).read()
    want = int(re.search(r"#define MC_ABI_VERSION (\d+)", hdr).group(1))
    got = int(re.search(r"local MC_ABI_VERSION = (\d+)", lua).group(1))
    assert got == want
    assert "lib.mc_version() == MC_ABI_VERSION" in lua
    assert not re.search(r"mc_version\(\)\s*==\s*\d", lua), "hard-wired version literal in the shim"


def test_mc_params_layout_matches_the_header(mc):
    lua = open(LUA).read()
    hf = _struct_fields(_strip_c(open(HDR).read()))
    lf = _struct_fields(_strip_c(_cdef(lua)))
    assert lf == hf, "mc_params differs:\n shim  : %s\n header: %s" % (lf, hf)
    # ... and the ctypes mirror has the same field names in the same order
    names = [f.split()[-1] for f in hf]
    assert [n for n, _ in mc.params.McParams._fields_] == names


def test_the_reference_table_is_covered():
    """every adcensus.* function stereo_predict calls (main.lua:929-1082) exists in the shim's table"""
    lua = open(LUA).read()
    for fn in ("ad", "census", "StereoJoin", "cross", "cbca", "sgm2", "outlier_detection", "interpolate_occlusion",
               "interpolate_mismatch", "subpixel_enchancement", "median2d", "mean2d", "spatial_argmin",
               "Normalize_forward", "predict", "fc_stack"):
        assert re.search(r"function adcensus\.%s\(" % fn, lua), fn
    # sm stage names are the reference's (main.lua:25-26) and agree with the Python host's tables
    import mc_cnn_amd as mc
    for tbl, ref in (("SM_TERMINATE", mc.params.SM_TERMINATE), ("SM_SKIP", mc.params.SM_SKIP)):
        body = re.search(r"local %s = \{(.*?)\}" % tbl, lua, flags=re.S).group(1)
        got = {k: int(v) for k, v in re.findall(r"(\w+) = (\d+)", body)}
        got[""] = int(re.search(r"\[''\] = (\d+)", body).group(1))
        assert got == ref
