"""GPU: the reference-side ctypes stub printed in INTEGRATION.md is executed as written (only the library path is made
absolute) and checked against the oracle's correlation, so the document cannot drift from include/nvq.h."""
import os
import re

import pytest
import torch

from oracle import sr_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_stub_runs_and_matches_the_oracle():
    from nerve_cl import _nvq
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "nvq_correlation_forward" in b and "def correlation_nhwc" in b)
    lib_path = _nvq.lib()._name
    assert '"libnvq.so"' in stub
    ns: dict = {}
    exec(stub.replace('"libnvq.so"', repr(lib_path)), ns)
    g = torch.Generator().manual_seed(3)
    x1, x2 = torch.randn(2, 16, 9, 14, generator=g), torch.randn(2, 16, 9, 14, generator=g)
    got = ns["correlation_nhwc"](x1.permute(0, 2, 3, 1).contiguous().cuda(), x2.permute(0, 2, 3, 1).contiguous().cuda())
    assert got.shape == (2, 9, 14, 96) and got[..., 81:].abs().max().item() == 0
    want = sr_oracle.correlation(x1, x2)
    err = (got[..., :81].permute(0, 3, 1, 2).cpu() - want).abs().max().item() / want.abs().max().item()
    assert err < 2e-5
