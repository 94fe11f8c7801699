"""The C-ABI library loads and exports every symbol include/gencore_amd.h declares; struct layouts agree between
the header (compiled with gcc) and the ctypes mirror.  No compute call is made (works without a GPU)."""
import ctypes as C
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "gencore_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gce_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_exported(built):
    from gencore_amd import capi
    lib = capi.load_library()
    names = declared_functions()
    assert set(names) == set(capi.EXPORTED_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.gce_abi_version() == capi.GCE_ABI_VERSION


def test_struct_layouts_match_header(built, tmp_path):
    from gencore_amd import capi
    src = tmp_path / "probe.c"
    src.write_text('#include <stdio.h>\n#include "gencore_amd.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",sizeof(gce_core),sizeof(gce_params),'
                   'sizeof(gce_batch),sizeof(gce_stats),sizeof(gce_result),sizeof(gce_timing),sizeof(gce_depth),sizeof(gce_bam_info),'
                   'sizeof(gce_bam_run),sizeof(gce_payload_layout),sizeof(gce_depth_run));return 0;}\n')
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [capi.CORE_DTYPE.itemsize, C.sizeof(capi.GceParams), C.sizeof(capi.GceBatch), C.sizeof(capi.GceStats),
                     C.sizeof(capi.GceResult), C.sizeof(capi.GceTiming), C.sizeof(capi.GceDepth), C.sizeof(capi.GceBamInfo),
                     C.sizeof(capi.GceBamRun), C.sizeof(capi.GcePayloadLayout), C.sizeof(capi.GceDepthRun)]


def test_defaults_match_reference_options(built):
    from gencore_amd import capi
    lib = capi.load_library()
    p = capi.GceParams()
    lib.gce_params_default(C.byref(p))
    q = capi.default_params()
    for name, _ in capi.GceParams._fields_:
        if name in ("target_len", "umi_prefix"):
            continue
        assert getattr(p, name) == getattr(q, name), name
    # src/options.cpp:4-40
    assert (p.proper_umi_diff_threshold, p.unproper_umi_diff_threshold, p.duplex_mismatch_threshold) == (1, 0, 2)
    assert (p.high_quality, p.moderate_quality, p.low_quality) == (30, 20, 15)
    assert (p.score_high, p.score_moderate, p.score_low, p.score_bad, p.base_score_req) == (8, 6, 4, 2, 6)
    assert p.score_percent_req == 0.8 and p.skip_low_complexity_cluster_threshold == 1000 and p.flush_period == 10000


def test_umi_prefix_autodetect(built):
    from gencore_amd import capi
    lib = capi.load_library()
    for name, want in ((b"A:1:UMI_ACGT", b"UMI"), (b"A:1:umi_ACGT", b"umi"), (b"A:1:ACGT", b"")):   # src/gencore.cpp:207-216
        buf = (C.c_char * 32)()
        lib.gce_detect_umi_prefix(name, buf)
        assert buf.value == want


def test_no_device_is_a_loud_error_not_a_fallback(built):
    """Without a GPU gce_create must refuse (GCE_ERR_NO_DEVICE): the product has no CPU path."""
    import torch
    if torch.cuda.is_available():
        return
    from gencore_amd import capi
    lib = capi.load_library()
    h = C.c_void_p()
    p = capi.default_params()
    assert lib.gce_create(C.byref(p), C.byref(h)) == -2


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "gencore_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle_py" not in txt and "gencore_oracle" not in txt and "libgencore_oracle" not in txt, f
