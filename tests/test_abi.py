"""The C-ABI library loads and exports every symbol include/gsplat.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from gaussiansplat_amd import build
    build.build()
    from gaussiansplat_amd import backend
    return backend.load()


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "gsplat.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    from gaussiansplat_amd import backend
    names = _declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gsplat.h but not exported"
    assert sorted(backend.SYMBOLS) == names, "backend.SYMBOLS out of sync with include/gsplat.h"


def test_abi_version_and_config_layout(lib):
    from gaussiansplat_amd import backend
    assert lib.gs_abi_version() == backend.GS_ABI_VERSION == 3
    cfg = backend.default_config()
    assert cfg.struct_size == C.sizeof(backend.GsConfig) == 96 and cfg.abi_version == 3
    assert cfg.schedule == 3 and cfg.slab_mode == 1 and cfg.debug_flags == 0 and cfg.slab_max_ratio == 0.0 and cfg.list_cap == 0
    assert cfg.tile_parts == 0 and list(cfg.reserved) == [0, 0, 0]              # reserved words, and the one that got a name, default to 0
    hdr = open(os.path.join(ROOT, "include", "gsplat.h")).read()
    assert re.search(r"#define GS_ABI_VERSION\s+3\b", hdr)
    assert cfg.tile_size == 16 and cfg.order == backend.ORDER_DEPTH_DESC and abs(cfg.t_min - 1e-5) < 1e-12


def test_no_cpu_fallback(lib):
    """Without a HIP device the product path must fail loudly (never route through a CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gaussiansplat_amd import backend, renderer
    with pytest.raises(backend.GsError) as e:
        backend.Context()
    assert e.value.code == -2
    with pytest.raises(RuntimeError):
        renderer.getRenderer("GAUSSIAN_3D", (64, 64, 3), (16, 16), (4, 4), 100)


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "gaussiansplat_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "gs_oracle" not in txt, os.path.join(dp, f)
