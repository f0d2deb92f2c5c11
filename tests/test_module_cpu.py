"""CPU: host logic of the drop-in module surface (no kernels run here)."""
import copy
import os

import numpy as np
import pytest
import torch

from nerve_cl.models import EnhancementConfig, EnhancementEngine, SuperResolutionNet
from oracle import sr_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("kw", [
    dict(num_features=64, num_residual_blocks=8),
    dict(num_features=32, num_residual_blocks=4),
    dict(num_features=64, num_residual_blocks=8, temporal_window=2, scale_factor=4),
    dict(num_features=16, num_residual_blocks=1, scale_factor=3),
])
def test_state_dict_inventory(kw):
    m = SuperResolutionNet(**kw)
    want = sr_oracle.param_shapes(**kw)
    got = {n: tuple(p.shape) for n, p in m.named_parameters()}
    assert list(got) == list(want) and got == want
    assert {n: tuple(b.shape) for n, b in m.named_buffers()} == sr_oracle.buffer_shapes(kw["num_features"])
    assert all(p.dtype == torch.float32 for p in m.parameters())
    assert m.state_dict()["feature_extractor.body.0.bn.num_batches_tracked"].dtype == torch.int64
    assert m.num_frames == 2 * kw.get("temporal_window", 1) + 1
    assert m.scale_factor == kw.get("scale_factor", 2)


@pytest.mark.parametrize("tag,kw", [
    ("cfg1", dict(scale_factor=2, num_features=32, num_residual_blocks=4, temporal_window=1)),
    ("cfg2", dict(scale_factor=2, num_features=64, num_residual_blocks=8, temporal_window=1)),
])
def test_default_init_equals_reference(tag, kw):
    """Same construction order => same RNG stream => the reference's default weights."""
    g = np.load(os.path.join(GOLD, "default_init_seed0.npz"))
    torch.manual_seed(0)
    m = SuperResolutionNet(**kw)
    sd = m.state_dict()
    keys = [k[len(tag) + 1:] for k in g.files if k.startswith(tag + "/")]
    assert sorted(keys) == sorted(sd.keys())
    for n in keys:
        f = sd[n].double().reshape(-1)
        head = np.zeros(4)
        head[:min(4, f.numel())] = f[:4].numpy()
        got = np.concatenate([[f.sum().item(), f.norm().item()], head])
        assert np.allclose(got, g[f"{tag}/{n}"], rtol=1e-12, atol=1e-12), n


def test_param_counts_and_flops_formula():
    m = SuperResolutionNet()
    assert m.get_num_parameters() == 1987283
    # reference formula with its hard-coded F=64 / 8 blocks (super_resolution.py:411-431)
    H = W = 128
    want = H * W * 3 * 64 * 9 + H * W * 64 * 81 * 2 + H * W * 64 * 64 * 9 * 8 + H * W * 64 * 12 * 9
    assert m.get_flops() == want
    assert SuperResolutionNet(num_features=32, num_residual_blocks=4).get_num_parameters() == 820339


def test_module_duck_typing():
    m = SuperResolutionNet(num_features=16, num_residual_blocks=1)
    m2 = copy.deepcopy(m)
    m2.load_state_dict(m.state_dict(), strict=True)
    m.eval(); m.train(); m.zero_grad()
    assert next(m.parameters()).device.type == "cpu"
    lay, total = m._bucket_layout()
    assert all(o % 4 == 0 for o, _ in lay.values())
    assert total >= m.get_num_parameters()


def test_cpu_forward_is_refused_loudly():
    m = SuperResolutionNet(num_features=16, num_residual_blocks=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(1, 3, 3, 8, 8))
    with pytest.raises(ValueError):
        m(torch.rand(1, 3, 8, 8))          # 4-D input: tuple-unpack error like the reference
    with pytest.raises(RuntimeError, match="no CPU fallback"):     # the sub-modules compute on their own - on HIP tensors
        m.feature_extractor(torch.rand(1, 3, 8, 8))
    with pytest.raises(RuntimeError, match="only holds parameters"):
        m.residual_blocks(torch.rand(1, 16, 8, 8))               # a plain container of the reference's nn.Sequential shape


def test_engine_surface():
    e = EnhancementEngine(EnhancementConfig(frame_recovery_enabled=False, super_resolution_enabled=True))
    names = [n for n, _ in e.named_parameters()]
    assert names[-1] == "enhancement_strength" or "enhancement_strength" in names
    assert sum(n.startswith("super_resolution.") for n in names) == 131
    info = e.get_model_info()
    assert info["parameters"]["super_resolution"] == 1987283
    assert info["parameters"]["total"] == 1987284
    cfg = EnhancementConfig()
    assert (cfg.frame_recovery_enabled, cfg.recovery_base_channels, cfg.recovery_temporal_window,
            cfg.super_resolution_enabled, cfg.scale_factor, cfg.sr_num_features, cfg.sr_num_residual_blocks,
            cfg.sr_temporal_window, cfg.use_lightweight_sr, cfg.enhancement_mode, cfg.upscale_first) == \
        (True, 64, 2, True, 2, 64, 8, 1, False, "sequential", False)


def _checksums(sd):
    out = {}
    for n, t in sd.items():
        f = t.double().reshape(-1)
        head = np.zeros(4)
        head[:min(4, f.numel())] = f[:4].numpy()
        out[n] = np.concatenate([[f.sum().item(), f.norm().item()], head])
    return out


@pytest.mark.parametrize("tag", ["fr16", "fr64", "engine"])
def test_frame_recovery_and_default_engine_state_dict_and_default_init_equal_the_reference(tag):
    """FrameRecoveryNet (reference frame_recovery.py:361-395) and EnhancementEngine() (enhancement_engine.py:62-93): the
    reference's state_dict keys / shapes, and - same construction order, same RNG stream - its default weights."""
    from nerve_cl.models import EnhancementEngine, FrameRecoveryNet
    g = np.load(os.path.join(GOLD, "fr_default_init_seed0.npz"))
    torch.manual_seed(0)
    m = FrameRecoveryNet(3, 16, 2) if tag == "fr16" else FrameRecoveryNet() if tag == "fr64" else EnhancementEngine()
    got = _checksums(m.state_dict())
    keys = [k[len(tag) + 1:] for k in g.files if k.startswith(tag + "/")]
    assert sorted(keys) == sorted(got)
    for n in keys:
        assert np.allclose(got[n], g[f"{tag}/{n}"], rtol=1e-12, atol=1e-12), n
    if tag == "fr64":
        assert len(list(m.parameters())) == 117
        with pytest.raises(RuntimeError):
            m(torch.zeros(1, 3, 64, 64), torch.zeros(1, 2, 3, 64, 64))      # CPU tensors: no fallback


def test_cached_parameter_slots_follow_the_module():
    """BucketedNet._slots (the per-step replacement of named_parameters / named_buffers walks): same names, same order, the same
    tensor objects as the module tree - also after a parameter or buffer has been REPLACED (attribute assignment,
    load_state_dict(assign=True)), after .to(dtype) and in a deepcopy (whose slots must point into the copy)."""
    from nerve_cl.models import FrameRecoveryNet, LightweightSuperResolution
    for m in (SuperResolutionNet(num_features=32, num_residual_blocks=2), LightweightSuperResolution(scale_factor=2), FrameRecoveryNet()):
        def same(net):
            got, want = net._named_params(), list(net.named_parameters())
            assert [n for n, _ in got] == [n for n, _ in want] and all(a is b for (_, a), (_, b) in zip(got, want))
            td = net._tensor_dict()
            assert list(td) == [n for n, _ in want] + [n for n, _ in net.named_buffers()]
            for n, b in net.named_buffers():
                assert td[n] is b
            for n, p in want:
                assert td[n].data_ptr() == p.data_ptr()
        same(m)
        name, old = next(iter(m.named_parameters()))
        owner = m.get_submodule(name.rpartition(".")[0])
        setattr(owner, name.rpartition(".")[2], torch.nn.Parameter(torch.zeros_like(old)))     # a replaced Parameter object
        same(m)
        assert m._named_params()[0][1] is not old
        sd = {k: v.clone() + 1 for k, v in m.state_dict().items()}
        m.load_state_dict(sd, assign=True)                                                   # every tensor replaced
        same(m)
        m.double()
        same(m)
        c = copy.deepcopy(m)
        same(c)
        assert all(a is not b for (_, a), (_, b) in zip(c._named_params(), m._named_params()))
