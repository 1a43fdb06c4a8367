"""CPU: the C-ABI library loads and exports every symbol include/svpc_hip.h declares (no compute without a GPU),
and the product refuses to run without the GPU (no silent fallback)."""
import pytest
import torch

from svpc_amd import _lib, ops


def test_library_exports_every_declared_symbol():
    decls = _lib.declarations()
    assert len(decls) >= 55
    lib = _lib.load()
    for name in decls:
        assert hasattr(lib, name), name
    assert lib.svpc_abi_version() == 2
    assert lib.svpc_opt_chunk() > 0 and lib.svpc_opt_meta_bytes() == 72
    assert lib.svpc_ln_bwd_groups(19200) >= 1 and lib.svpc_colsum_chunks(19200) >= 1


def test_ops_fail_loudly_on_cpu_tensors():
    x, w = torch.randn(4, 8), torch.randn(3, 8)
    with pytest.raises(_lib.SvpcKernelError):
        ops.linear(x, w, None)
    with pytest.raises(_lib.SvpcKernelError):
        ops.layernorm(x, torch.ones(8), torch.zeros(8), 1e-12)


def test_product_package_never_imports_the_oracle():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "svpc_amd")):
        for f in files:
            if f.endswith(".py"):
                code = [l for l in open(os.path.join(dirpath, f)).read().splitlines() if "import" in l]
                assert not any("oracle" in l or "emul_ops" in l for l in code), f
