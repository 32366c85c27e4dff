"""GPU: the shipped binding for the reference's decoder object (js/reference_binding.js) driving the REAL addon --
binding -> napi/leon_napi.node -> libleon_hip.so -> HIP -> planes -- on the box where the kernels run.  The decoder object
is a stand-in that carries the reference's field names (tests/binding_addon_check.js says which, with the jsv.js lines);
tensors and expected planes are those of tests/golden/glsl_idct_cases.json: what the UNMODIFIED reference's IDCT_GL was
handed and what its shaders produced on tools/softgl.  (tests/test_reference_binding.py runs the same binding under the
reference's own decodeFrame loop with a recording stand-in for the addon -- CPU, build container.)"""
import json
import os
import shutil
import subprocess

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(shutil.which("node") is None, reason="no node")
def test_binding_drives_the_addon_to_the_executed_references_planes():
    addon = os.path.join(ROOT, "mpeg1video-decoder-webgl_amd", "napi", "leon_napi.node")
    assert os.path.exists(addon), "leon_napi.node is not built (__graft_entry__.build())"
    out = subprocess.run(["node", os.path.join(ROOT, "tests", "binding_addon_check.js"), os.path.join(ROOT, "tests", "golden", "glsl_idct_cases.json")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout)
    assert d["abi"] == 3
    assert [c["name"] for c in d["cases"]] == ["synthetic_ippp_default_matrices", "full_int16_garbage_vectors_leave_picture",
                                               "single_coefficients_and_dc", "custom_intra_matrix_clamps_zero_to_one"]
    n = 0
    for c in d["cases"]:
        for i, p in enumerate(c["pictures"]):
            assert p["y"] and p["cb"] and p["cr"], "%s picture %d (type %d): planes differ from the executed reference's" % (c["name"], i, p["type"])
            assert p["oneTexturePerSlot"] and p["inuse"] == 1
            n += 1
        # displayed frames are released through `texture.inuse = false` (player.js:2820), a forward reference only after
        # the picture that predicts from it has been submitted: the loop lives on two slots of the 13
        assert c["distinctSlots"] == 2 and set(c["slots"]) == {0, 1}, c["slots"]
        # nobody releases: 13 pictures fit, the 14th throws what the reference throws (jsv.js:1175)
        assert c["exhaustion"]["taken"] == 13 and "no free render buffers" in c["exhaustion"]["thrown"]
        assert c["afterFreeDecodedBuffers"] == 0            # GLfreeDecodedBuffers (seek): every slot free again
        assert c["rgbaBytes"] > 0
    assert n == 16
