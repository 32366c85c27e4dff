"""CPU: the C-ABI library is built, loads (no GPU needed for dlopen) and exports every symbol
include/leon.h declares; the ctypes structs match the header's layout."""
import ctypes as C
import os
import re

from helpers import ROOT


def _declared(header="leon.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(leon_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported():
    import leon_ctypes as L
    lib = L.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libleon_hip.so does not export %s" % n
    assert set(names) == set(L.SYMBOLS)
    assert lib.leon_abi_version() == 2
    pipe = [n for n in _declared("leon_pipeline.h") if n.startswith("leon_pipeline_") and n not in ("leon_pipeline_callback",)]
    assert set(pipe) == set(L.PIPELINE_SYMBOLS), set(pipe) ^ set(L.PIPELINE_SYMBOLS)
    for n in pipe:
        assert hasattr(lib, n), "libleon_hip.so does not export %s" % n


def test_struct_layouts():
    import leon_ctypes as L
    assert C.sizeof(L.Config) == 40 and L.Config.stream.offset == 24 and L.Config.alpha.offset == 32
    assert C.sizeof(L.Picture) == 16 + 9 * 8 + 16 + 8 and L.Picture.coef_y.offset == 16 and L.Picture.rgba_out.offset == 88 and L.Picture.coef_a.offset == 104
    assert C.sizeof(L.KernelStats) == 32
    assert C.sizeof(L.LaunchTime) == 32 and L.LaunchTime.ms.offset == 8


def test_no_cpu_fallback_without_device():
    """Without a usable gfx950 device leon_create must fail loudly (this container has no GPU)."""
    import torch
    import leon_ctypes as L
    if torch.cuda.is_available():
        return
    try:
        L.Decoder(64, 48)
    except L.LeonError as e:
        assert e.code == L.ERR_NO_DEVICE and "no CPU fallback" in str(e)
    else:
        raise AssertionError("leon_create succeeded without a GPU")


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package or under tools/ may reference it
    (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do)."""
    for top in ("mpeg1video-decoder-webgl_amd", "tools", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, top)):
            for f in fs:
                if f.endswith((".py", ".cpp", ".h", ".js", ".cc", ".sh")):
                    src = open(os.path.join(dp, f), errors="replace").read()
                    assert "oracle_py" not in src and "leon_oracle" not in src and "libleon_oracle" not in src \
                        and "from oracle" not in src and "import oracle" not in src, os.path.join(dp, f)
