"""CPU: the C-ABI library loads and exports every symbol include/dga_hip.h declares (no compute calls)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    text = (ROOT / "include" / "dga_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dga_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(dga):
    from deepgemm_ascend_amd import _lib
    L = ctypes.CDLL(str(_lib.LIB_PATH))
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/dga_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)


def test_struct_layouts_match_header(dga):
    from deepgemm_ascend_amd import _lib
    assert ctypes.sizeof(_lib.Tiling) == 72 and ctypes.sizeof(_lib.Platform) == 56 and ctypes.sizeof(_lib.Problem) == 28
    assert _lib.lib().dga_abi_version() == 7


def test_status_strings_and_null_checks(dga):
    from deepgemm_ascend_amd import _lib
    L = _lib.lib()
    assert L.dga_status_string(0) == b"ok" and b"HIP" in L.dga_status_string(-5)
    assert L.dga_get_best_config(1, 1, 1, 1, None) == -1
    assert L.dga_tiling(None, None) == -1
    assert L.dga_infer_shape(None, 2, None, 2, None) == -1


def test_ops_refuse_cpu_tensors(dga):
    """The product has no CPU path: host tensors are rejected before any launch."""
    import torch
    a = torch.zeros((16, 128), dtype=torch.uint8); b = torch.zeros((128, 128), dtype=torch.uint8)
    sfa = torch.ones((16, 1)); sfb = torch.ones((1, 1)); out = torch.zeros((16, 128), dtype=torch.bfloat16)
    with pytest.raises(dga.DGAError):
        dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out)


def test_shape_checks_raise(dga):
    import torch
    a = torch.zeros((16, 128), dtype=torch.uint8); b = torch.zeros((128, 256), dtype=torch.uint8)
    with pytest.raises(dga.DGAError):
        dga.gemm_fp8_fp8_bf16_nt((a, torch.ones((16, 1))), (b, torch.ones((1, 2))), torch.zeros((16, 128), dtype=torch.bfloat16))


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the product package may import, link or run it."""
    for p in (ROOT / "deepgemm_ascend_amd").rglob("*.py"):
        text = p.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), p
        assert "libdga_oracle" not in text, p
    for p in (ROOT / "deepgemm_ascend_amd" / "csrc").iterdir():
        if p.is_file():
            assert "dga_oracle" not in p.read_text(errors="ignore"), p
