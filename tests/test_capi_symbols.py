"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/scenerf_hip.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

from scenerf_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "scenerf_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(scenerf_hip_[a-z0-9_]+)\s*\(", src)))


def test_library_is_built_in_tree():
    assert os.path.exists(_capi.LIB_PATH), "run `python -m scenerf_amd.build` (hipcc cross-compiles gfx950 without a GPU)"
    assert os.path.dirname(_capi.LIB_PATH).startswith(ROOT)


def test_every_declared_symbol_is_exported_and_bound():
    declared = _declared_functions()
    assert len(declared) >= 15
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "header declares %s but the library does not export it" % name
    assert set(declared) == set(_capi.EXPORTED_SYMBOLS), "ctypes binding and header disagree: %s" % (
        set(declared) ^ set(_capi.EXPORTED_SYMBOLS))


def test_abi_version_and_error_string():
    lib = _capi.load()
    assert lib.scenerf_hip_abi_version() == _capi.ABI_VERSION
    m = re.search(r"#define SCENERF_HIP_ABI_VERSION (\d+)", open(HEADER).read())
    assert int(m.group(1)) == _capi.ABI_VERSION
    assert isinstance(lib.scenerf_hip_last_error(), bytes)


def test_struct_layouts_match_header_constants():
    src = open(HEADER).read()
    consts = dict((k, int(v)) for k, v in re.findall(r"#define (SCENERF_[A-Z_]+) (\d+)", src))
    assert consts["SCENERF_N_SCALES"] == _capi.N_SCALES
    assert consts["SCENERF_D_LATENT"] == _capi.D_LATENT
    assert consts["SCENERF_D_HIDDEN"] == _capi.D_HIDDEN
    assert consts["SCENERF_D_XENC"] == _capi.D_XENC
    assert consts["SCENERF_TILE_ROWS"] == _capi.TILE_ROWS
    # scenerf_cfg: 6 int32 + 10 float + 5*5 int32 + precision + map_chw[5] + fused_min_rows + fwd_kernel + flags
    assert ctypes.sizeof(_capi.Cfg) == 4 * (6 + 10 + 25 + 1 + 5 + 3)
    assert ctypes.sizeof(_capi.MlpActs) == 8 * 11   # H[4], Nn[3], h0pre, logits, sign_bits, x3_ready (+ padding)
    assert ctypes.sizeof(_capi.ProfRec) == 48 + 4 + 4 + 8 + 8


def test_struct_layouts_match_the_c_compiler(tmp_path):
    """sizeof / offsetof of every ABI struct as gcc lays them out from include/scenerf_hip.h == the ctypes mirror."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    structs = {"scenerf_cfg": _capi.Cfg, "scenerf_mlp_weights": _capi.MlpWeights, "scenerf_mlp_params": _capi.MlpParams,
               "scenerf_mlp_grads": _capi.MlpGrads, "scenerf_mlp_acts": _capi.MlpActs, "scenerf_prof_rec": _capi.ProfRec}
    lines = []
    for cname, cls in structs.items():
        lines.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in cls._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, f[0], cname, f[0]))
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(void) {\n%s\nreturn 0; }\n' % (HEADER, "\n".join(lines)))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-o", str(exe), str(src)])
    got = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        c, f, v = ln.split()
        got[(c, f)] = int(v)
    for cname, cls in structs.items():
        assert got[(cname, "sizeof")] == ctypes.sizeof(cls), cname
        for f in cls._fields_:
            assert got[(cname, f[0])] == getattr(cls, f[0]).offset, "%s.%s" % (cname, f[0])


def test_bad_arguments_are_reported_not_fatal():
    """NULL pointers / inconsistent configs must come back as error codes with a message (never abort)."""
    lib = _capi.load()
    rc = lib.scenerf_hip_composite_forward(None, None, None, 4, 64, None, None, None, None, None, None, None, None, None)
    assert rc != 0 and b"NULL" in lib.scenerf_hip_last_error()
    from scenerf_amd.config import RenderConfig
    cc = RenderConfig.kitti().to_c()
    cc.n_samples = 999
    rc = lib.scenerf_hip_raysom_forward(ctypes.byref(cc), None, None, None, None, 1, None, None, None, None, None, None)
    assert rc != 0 and b"n_samples" in lib.scenerf_hip_last_error()
    # the capturable optimizer step: no device-side [lr, t], no launch
    arr = (_capi.AdamWTensor * 1)()
    rc = lib.scenerf_hip_adamw_step_dev(1, arr, None, 0.9, 0.999, 1e-8, 0.0, None)
    assert rc != 0 and b"hyper" in lib.scenerf_hip_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", "/nonexistent/libscenerf_hip.so")
    with pytest.raises(RuntimeError, match="no CPU / eager fallback"):
        _capi.load()
