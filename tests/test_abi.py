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
    assert lib.leon_abi_version() == 3 == L.ABI_VERSION
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


def test_structs_have_the_size_the_c_compiler_gives_them(tmp_path):
    """every struct that crosses the ABI by pointer: sizeof in C == sizeof of the ctypes mirror (a binding built
    against an older header hands the library structs that are too small); both libraries report the header's version"""
    import subprocess
    import leon_ctypes as L
    import leon_vlc_ctypes as V
    names = {"leon_config": L.Config, "leon_picture": L.Picture, "leon_sparse_picture": L.SparsePicture, "leon_kernel_stats": L.KernelStats,
             "leon_launch_time": L.LaunchTime, "leon_pipeline_config": L.PipelineConfig, "leon_pipeline_frame": L.PipelineFrame,
             "leon_pipeline_info": L.PipelineInfo, "leon_pipeline_stats": L.PipelineStats,
             "leon_vlc_info": V.Info, "leon_vlc_picture": V.Picture, "leon_vlc_picture_scan": V.PictureScan, "leon_vlc_gpu_tables": V.GpuTables}
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "leon.h"\n#include "leon_pipeline.h"\n#include "leon_vlc.h"\nint main(void){\n' +
                   "".join('printf("%s %%zu\\n", sizeof(%s));\n' % (n, n) for n in names) +
                   'printf("LEON_ABI_VERSION %d\\nLEON_VLC_ABI_VERSION %d\\n", LEON_ABI_VERSION, LEON_VLC_ABI_VERSION);return 0;}\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for n, t in names.items():
        assert int(got[n]) == C.sizeof(t), "%s: C says %s bytes, the ctypes mirror %d" % (n, got[n], C.sizeof(t))
    assert int(got["LEON_ABI_VERSION"]) == L.ABI_VERSION == L.load().leon_abi_version()
    assert int(got["LEON_VLC_ABI_VERSION"]) == V.ABI_VERSION == V.load().leon_vlc_abi_version()


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
