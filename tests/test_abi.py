"""CPU suite: the C-ABI library loads without a GPU and exports every symbol
include/hyas.h declares; entry points that need no device behave; the ones that
do fail loudly instead of computing anything on the CPU."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "hyas.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(re.findall(r"\b(hyas_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from path_planner_amd import _lib as hl

    if not hl.lib_path().exists():
        import __graft_entry__ as g

        g.build()
    return hl.load_library()


def test_header_symbols_are_exported(lib):
    names = declared_symbols()
    assert len(names) >= 28
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/hyas.h but not exported by libhyas.so"


def test_python_binding_lists_every_symbol():
    from path_planner_amd import _lib as hl

    assert sorted(hl.EXPORTS) == declared_symbols()


def test_no_torch_or_stl_types_in_header():
    text = (ROOT / "include" / "hyas.h").read_text()
    assert 'extern "C"' in text
    for bad in ("std::", "torch", "at::Tensor", "template"):
        assert bad not in text


def test_struct_layouts_match_header(lib):
    """ctypes mirrors and numpy record types have the sizes the header promises."""
    import orc
    from path_planner_amd import _lib as hl

    assert C.sizeof(hl.Params) == C.sizeof(orc.Params)
    assert [f[0] for f in hl.Params._fields_] == [f[0] for f in orc.Params._fields_]
    assert hl.SUCC_DT.itemsize == 28 and hl.NODE_DT.itemsize == 20 and hl.GOAL_DT.itemsize == 12 and hl.RESULT_DT.itemsize == 32


def test_default_params_match_reference_defaults(lib):
    """hyas_default_params = struct defaults of include/parameter_manager.h:10-42, :124-175."""
    import orc
    from path_planner_amd import _lib as hl

    p = hl.default_params()
    q = orc.default_params()
    for name, _ in hl.Params._fields_:
        assert getattr(p, name) == getattr(q, name), name
    h = hl.hot_path_params()
    assert h.curve_type == 1 and h.number_of_successors == 5 and h.reverse == 1 and h.analytical_expansion_every_point == 1
    assert h.adaptive_steering_angle_and_step_size == 0 and h.adaptive_step_size == 0


def test_null_context_calls_return_errors(lib):
    """Every entry point rejects a null context instead of crashing."""
    buf = np.zeros(64, np.uint8)
    vp = buf.ctypes.data_as(C.c_void_p)
    assert lib.hyas_set_costmap(None, vp, 4, 4, 1.0, 0.0, 0.0, 0) < 0
    assert lib.hyas_collision_batch(None, 0, vp, None, 1, vp) < 0
    assert lib.hyas_expand_batch(None, vp, vp, None, 1, vp, vp) < 0
    assert lib.hyas_shot_batch(None, vp, vp, 1, vp, vp) < 0
    assert lib.hyas_plan_batch(None, vp, vp, 1, vp) < 0
    assert lib.hyas_table_lookup(None, vp, 1, vp) < 0
    assert lib.hyas_twod_size(None) < 0
    assert lib.hyas_last_kernel_ms(None, 0, None) < 0
    assert lib.hyas_last_error(None) == b"null context"
    lib.hyas_destroy(None)


def test_no_cpu_fallback_without_gpu(lib):
    """Without a HIP device hyas_create returns NULL and the Python wrapper raises: nothing is computed on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from path_planner_amd import Hyas, HyasError, hot_path_params

    p = hot_path_params()
    assert lib.hyas_create(C.byref(p), 0) is None
    with pytest.raises(HyasError):
        Hyas(p)
    from path_planner_amd.planner import Costmap2D, HybridAStarPlanner

    pl = HybridAStarPlanner()
    with pytest.raises(HyasError):
        pl.initialize("planner", Costmap2D(np.zeros((8, 8), np.uint8)))
    assert pl.initialized_ is False


def test_product_does_not_touch_the_oracle():
    """Nothing under path_planner_amd/ imports, links or opens the oracle."""
    for f in (ROOT / "path_planner_amd").rglob("*"):
        if f.suffix in (".py", ".h", ".hip", ".cpp") or f.name == "Makefile":
            t = f.read_text()
            assert "oracle" not in t.lower(), f
