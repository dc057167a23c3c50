"""The C-ABI library loads on a CPU-only box and exports every symbol include/trhip.h declares.
No compute calls here (no GPU)."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "trhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(trhip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_lists_agree():
    from toyrenderer_amd import rhi
    assert _declared() == sorted(rhi.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    from toyrenderer_amd import rhi
    lib = rhi.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.trhip_abi_version() == 1


def test_shader_registry_uses_reference_names():
    from toyrenderer_amd import rhi
    names = set(rhi.shader_names())
    for n in ("gpuculling_CS_GPUCulling LATE_CULL=0", "gpuculling_CS_GPUCulling LATE_CULL=1",
              "gpuculling_CS_BuildLateCullIndirectArgs", "minmaxdownsample_CS_Main",
              "ffx_spd_downsample_pass_CS FFX_SPD_OPTION_DOWNSAMPLE_FILTER=1",
              "updateinstanceconsts_CS_UpdateInstanceConstsAndBuildTLAS",
              "basepass_AS_Main LATE_CULL=0", "basepass_AS_Main LATE_CULL=1"):
        assert n in names, n


def test_no_device_is_an_error_not_a_fallback():
    """On a box without a GPU device creation must fail loudly (no CPU fallback)."""
    import torch
    from toyrenderer_amd import rhi
    if torch.cuda.device_count() > 0:
        return
    try:
        rhi.Device(0)
    except rhi.TrhipError as e:
        assert "no HIP device" in str(e) or "HIP error" in str(e)
    else:
        raise AssertionError("Device(0) succeeded without a GPU")


def test_product_does_not_import_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "toyrenderer_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)
                assert not re.search(r"#\s*include\s*[<\"][^>\"]*oracle", src), os.path.join(dp, f)
                assert "libtr_oracle" not in src and "pyoracle" not in src, os.path.join(dp, f)


def test_host_library_exports_every_declared_symbol():
    from toyrenderer_amd import host
    text = open(os.path.join(ROOT, "include", "trhost.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(trhost_[a-z0-9_]+)\s*\(", text)))
    assert declared == sorted(host.HOST_SYMBOLS)
    lib = host.load()
    for name in declared:
        assert hasattr(lib, name), name


def test_render_graph_heap_allocator_best_fit_split_merge():
    """RenderGraph::Heap (reference RenderGraph.cpp:443-580): best fit, split, merge on free; a block
    whose left-over would be >= 16 MB is not considered (reference behaviour kept)."""
    from toyrenderer_amd import host
    K = 65536
    MB = 1 << 20
    res, used, peak, blocks = host.heap_sim(16 * MB, [K, 2 * K, K, -2, K, -1, -3, -5])
    assert list(res[:5]) == [0, K, 3 * K, K, K]          # the freed 2K hole is the best fit for K
    assert used == 0 and peak == 4 * K and blocks == 1   # everything merged back into one free block
    # no fit -> UINT64_MAX
    res, *_ = host.heap_sim(4 * K, [4 * K, K])
    assert res[0] == 0 and res[1] == np.iinfo(np.uint64).max
    # a small request does not carve up a fresh big heap (left-over >= 16 MB)
    res, *_ = host.heap_sim(64 * MB, [K])
    assert res[0] == np.iinfo(np.uint64).max
    res, *_ = host.heap_sim(64 * MB, [64 * MB - 15 * MB])
    assert res[0] == 0
