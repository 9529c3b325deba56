"""The drop-in claim, checked against the REAL reference (build container only: needs oracle/_ref/ref_binding, i.e.
/root/reference at build time).

oracle/ref_binding.cpp is the binding of INTEGRATION.md compiled against the reference's own headers and linked with its own
translation units: it loads a .scene with the reference's Scene(path) and fills an rtx_scene_desc from the reference's public
members (scene.h:68-100, objects.h:24-200, lights.h:21-73).  This repo's host (rendering_amd/host/src/scene.cpp, flattenScene)
fills one from ITS Scene for the same file.  Both descriptions are serialised by rtx_desc_serialize (include/rtx_debug.h:
every scalar and every array rtx_scene_create reads) and must be byte-identical -- options, camera matrix, objects, the
flattened acceleration structure in the reference's visiting order, triangles, texture maps, lights, area-light sample
points, skybox faces."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIND = os.path.join(ROOT, "oracle", "_ref", "ref_binding")
pytestmark = pytest.mark.skipif(not os.path.exists(BIND), reason="oracle/_ref/ref_binding not built (no /root/reference here)")


def host_bytes(ra, scene, w, h):
    rtx, host = ra.load()
    s = ra.Scene(scene, w, h)
    try:
        flat = C.c_void_p(host.rah_flatten(s.h))
        desc = C.c_void_p(host.rah_flat_desc(flat))
        rtx.rtx_desc_serialize.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        need = C.c_size_t(0)
        assert rtx.rtx_desc_serialize(desc, None, 0, C.byref(need)) == 0
        buf = np.zeros(need.value, np.uint8)
        assert rtx.rtx_desc_serialize(desc, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(need)) == 0
        host.rah_flat_free(flat)
        return buf.tobytes()
    finally:
        s.close()


def first_difference(a, b):
    n = min(len(a), len(b))
    x, y = np.frombuffer(a[:n], np.uint8), np.frombuffer(b[:n], np.uint8)
    d = np.nonzero(x != y)[0]
    return ("lengths %d / %d" % (len(a), len(b))) if d.size == 0 else "first differing byte %d of %d / %d" % (d[0], len(a), len(b))


# every object / material / light type, textures (diffuse + normal + specular maps), the skybox, the 25k-triangle mesh
@pytest.mark.parametrize("name,w,h", [("cfg1_simple_shapes", 512, 512), ("cfg2_smooth_4k", 160, 120), ("cfg2_smooth_25k", 320, 200),
                                      ("cfg3_reflective_refractive", 1920, 1080), ("cfg4_textured_256", 256, 256), ("mixed_materials", 200, 152),
                                      ("area_light", 200, 152), ("coincident", 160, 120), ("uv_out_of_range", 128, 128)])
def test_binding_description_equals_the_hosts(ra, tmp_path, name, w, h):
    out = tmp_path / "ref.bin"
    r = subprocess.run([BIND, "dump", ROOT, "scenes/%s.scene" % name, str(w), str(h), str(out)], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    ref = out.read_bytes()
    mine = host_bytes(ra, "scenes/%s.scene" % name, w, h)
    assert ref[:8] == b"RTXD0001" and len(ref) > 200
    assert ref == mine, first_difference(ref, mine)


def test_binding_source_is_what_integration_md_quotes():
    """INTEGRATION.md shows the binding: it must be the code that is compiled and tested here, not a sketch."""
    src = open(os.path.join(ROOT, "oracle", "ref_binding.cpp")).read()
    body = src[src.index("// ---- BINDING (begin)"):src.index("// ---- BINDING (end)")]
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for fn in ("static void flattenAC(", "static void fillView(", "static void fillDesc(", "static rtx_scene* uploadScene(",
               "static void rtxLaunchWorkers(", "static void rtxLaunchSSAA(", "static void rtxRender("):
        i = body.index(fn)
        # the function's text up to its closing brace at column 0
        j = body.index("\n}\n", i) + 3
        assert body[i:j] in md, "INTEGRATION.md does not quote %s as compiled" % fn
