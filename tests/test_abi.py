"""CPU-side checks of the C-ABI library: it builds for gfx950, loads without a GPU, exports every symbol
include/atlasfit.h declares, and fails loudly (no CPU fallback) when no device is present."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    import aiod_amd
    return aiod_amd.load_library()


def test_exports_every_declared_symbol(lib):
    import aiod_amd
    hdr = open(os.path.join(ROOT, "include", "atlasfit.h")).read()
    declared = set(re.findall(r"\b(af_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"af_handle", "af_config", "af_status", "af_net"}
    assert declared == set(aiod_amd.atlasfit.ABI_SYMBOLS), declared ^ set(aiod_amd.atlasfit.ABI_SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def test_config_struct_layout(lib):
    import ctypes
    import aiod_amd
    c = aiod_amd.default_config(768, 432, 80)
    assert ctypes.sizeof(aiod_amd.AfConfig) == 4 * (15 + 7 + 1 + 9 + 4 + 4) == lib.af_config_size()
    c2 = aiod_amd.default_config(768, 432, 80, two_layer=True)
    assert c.two_layer == 0 and c2.two_layer == 1 and c2.number_of_layers_mapping2 == 4 and c2.positional_encoding_num_alpha == 5
    assert abs(c2.alpha_flow_factor - 4900.0) < 1e-3 and c2.stop_bootstrapping_iteration == 10000
    assert c.samples_batch == 10000 and c.stop_global_rigidity == 5000 and abs(c.uv_mapping_scale - 0.8) < 1e-7


def test_no_cpu_fallback():
    import torch
    import aiod_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(aiod_amd.AtlasFitError):
        aiod_amd.AtlasFit(aiod_amd.default_config(32, 16, 4, samples_batch=64))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "all-in-one-deflicker_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("CPU oracle", ""), os.path.join(dp, f)


def test_launch_plan_arithmetic(lib):
    """The packed-launch split of the mapping batch (host.hip plan_mapping_split) for a 256-CU chip."""
    import ctypes

    def plan(ncu, rows_map, rows_atlas, dep):
        out = (ctypes.c_int * 3)()
        assert lib.af_debug_plan(ncu, rows_map, rows_atlas, dep, ctypes.byref(out)) == 0
        return tuple(out)
    N = 10000
    # 7 segments: 2188 tiles = 2 whole rounds (2048) + 35 workgroups; atlas 235 WGs -> 14 beyond one round lead the grid
    assert plan(256, 7 * N, 3 * N, 3 * N) == (2048, 2048 + 4 * 14, 2188)
    # 9 segments: 2813 tiles = 2 whole rounds + 192 workgroups (171 beyond one round)
    assert plan(256, 9 * N, 3 * N, 3 * N) == (2048, 2048 + 4 * 171, 2813)
    # small batches (tests, pre-train): no whole round holds the rows the atlas depends on -> no split, nothing rides along
    assert plan(256, 9 * 256, 3 * 256, 3 * 256) == (72, 72, 72)
    # the whole rounds must cover the dependent rows: 1024 tiles hold 32768 rows < 3N = 36000 -> unsplit
    assert plan(256, 5 * 12000 // 4, 36000, 36000)[0] == plan(256, 5 * 12000 // 4, 36000, 36000)[2]
    # remainder fits beside the atlas workgroups in one round: nothing needs to lead
    t1, t2, nt = plan(256, 7 * 9700, 3 * 9700, 3 * 9700)
    assert (t1, t2, nt) == (2048, 2048, 2122)
