"""The C-ABI library loads and exports every symbol include/crt.h declares; without a GPU the
device entry points refuse to run (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _header_functions():
    src = open(os.path.join(ROOT, "include", "crt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(crt_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(cr):
    from caitlynrenderer_amd import _lib
    names = _header_functions()
    assert len(names) >= 45
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"libcrt.so does not export {n}"
        assert n in _lib.SYMBOLS, f"python binding does not declare {n}"
    assert sorted(_lib.SYMBOLS) == names


def test_header_cites_reference_for_each_device_entry():
    src = open(os.path.join(ROOT, "include", "crt.h")).read()
    for anchor in ("Scene.h:1000-1156", "Scene.h:1233-1246", "Scene.h:1208-1213", "Scene.h:1160-1172",
                   "output.fs:9-20", "sbvh.h:99-153", "cwbvh.h:58-73", "Scene.h:742-926", "Camera.h:7-19", "Rnd.h:21-40"):
        assert anchor in src


def test_struct_sizes_match_reference_layouts(cr, survey):
    from caitlynrenderer_amd import _lib
    assert C.sizeof(_lib.crt_camera) == 15 * 4
    assert cr.RAY_DT.itemsize == 32 and cr.HIT_DT.itemsize == 16 and cr.STATS_DT.itemsize == 4
    assert survey["struct_sizes"] == {"FlatNode": 32, "Triangle": 48, "node8": 80}


def test_abi_version_and_error_string(cr):
    from caitlynrenderer_amd import _lib
    L = _lib.lib()
    assert L.crt_abi_version() == 6
    h = C.c_void_p()
    rc = L.crt_load_obj(b"/nonexistent/file.obj", None, C.byref(h))
    assert rc == _lib.CRT_ERR_IO and b"not found" in L.crt_last_error()
    assert L.crt_scene_create(None, C.byref(h)) == _lib.CRT_ERR_INVALID


def test_device_path_fails_loudly_without_gpu(cr, cornell_data):
    from caitlynrenderer_amd import _lib
    if _lib.lib().crt_device_count() > 0:
        pytest.skip("a GPU is visible; covered by the gpu tests")
    with pytest.raises(cr.CrtError) as e:
        cr.Scene(cornell_data, 64, 64, 1)
    assert e.value.code == _lib.CRT_ERR_NO_DEVICE


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under caitlynrenderer_amd/ may mention it."""
    pkg = os.path.join(ROOT, "caitlynrenderer_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                assert "liboracle" not in txt and "oracle.binding" not in txt and "from oracle" not in txt, fn


def test_scene_create_rejects_bad_indices(cr, cornell_data):
    import copy
    from caitlynrenderer_amd import _lib
    d = copy.copy(cornell_data)
    d.triangles = d.triangles.copy()
    d.triangles[0, 0] = 10 ** 6
    with pytest.raises(cr.CrtError) as e:
        cr.Scene(d, 32, 32, 1)
    assert e.value.code == _lib.CRT_ERR_INVALID


def test_scene_create_rejects_non_finite_vertices(cr, cornell_data):
    """an inf / NaN / absurdly large coordinate turns boxes and surface areas into inf or NaN (one such cluster never
    found a neighbour and the PLOC tail spun); refused up front, before any device work."""
    import copy
    from caitlynrenderer_amd import _lib
    for bad in (np.inf, -np.inf, np.nan, 3.0e19):
        d = copy.copy(cornell_data)
        d.vertices = d.vertices.copy()
        d.vertices[5, 1] = bad
        with pytest.raises(cr.CrtError) as e:
            cr.Scene(d, 32, 32, 1)
        assert e.value.code == _lib.CRT_ERR_INVALID and "vertex coordinate" in str(e.value)
