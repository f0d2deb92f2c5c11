"""GPU: nerve_cl.continual.EWC (flat-bucket HIP kernels) against the two-task fixture captured from the
reference's EWC + SuperResolutionNet (tests/golden/ewc_two_tasks.npz) and against the oracle's formulas."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import sr_oracle, synth
from oracle.make_goldens import grad_summary

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


class Adapter(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, x):
        return self.net(x.unsqueeze(1).expand(-1, 3, -1, -1, -1))


def test_two_task_sequence_matches_reference_fixture():
    from nerve_cl.continual import EWC
    from nerve_cl.models import SuperResolutionNet
    g = np.load(os.path.join(GOLD, "ewc_two_tasks.npz"))
    Fc, N, win, s, nS, H, W = [int(v) for v in g["cfg"]]
    sd = synth.formula_state(3, s, Fc, N, win, gain=synth.GOLDEN_GAIN)
    net = SuperResolutionNet(3, s, Fc, N, win)
    net.load_state_dict(sd)
    model = Adapter(net).cuda()
    ewc = EWC(model, ewc_lambda=5000.0)
    assert ewc.penalty() == 0.0                       # python float before the first task
    for k in range(2):
        lr = synth.formula_clip(4, 1, H, W, seed=31 + k)[:, 0]
        hr = synth.formula_target(4, H * s, W * s, seed=41 + k)
        batches = [(lr[0:2], hr[0:2]), (lr[2:4], hr[2:4])]
        ewc.register_task(k, batches)
        assert ewc.num_tasks == k + 1
        assert not model.training                     # compute_fisher leaves the model in eval()
        with torch.no_grad():
            for n, p in model.named_parameters():
                d = torch.from_numpy((synth.hash01(p.numel(), synth.name_seed(n) + k + 1).reshape(p.shape) * 2 - 1)
                                     .astype(np.float32)) * 0.01
                p.add_(d.cuda())
        model.zero_grad()
        pen = ewc.penalty()
        pen.backward()
        want = float(g[f"penalty{k}"])
        assert abs(pen.item() - want) <= 1e-3 * abs(want), (pen.item(), want)
        for n, p in model.named_parameters():
            key = n[len("net."):]
            f_ref, f_got = g[f"fisher{k}/{key}"], grad_summary(ewc.fisher_dict[n].cpu())
            assert abs(f_got[1] - f_ref[1]) <= 2e-3 * max(f_ref[1], 1e-20), ("fisher", n)
            p_ref, p_got = g[f"pgrad{k}/{key}"], grad_summary(p.grad.cpu())
            assert abs(p_got[1] - p_ref[1]) <= 2e-3 * max(p_ref[1], 1e-20), ("penalty grad", n)
            assert np.abs(p_got[2:] - p_ref[2:]).max() <= 2e-3 * max(np.abs(p_ref[2:]).max(), p_ref[1] * 1e-2), n
    sdict = ewc.state_dict()
    assert set(sdict) == {"ewc_lambda", "mode", "decay", "num_tasks", "fisher_dict", "optpar_dict", "task_fisher",
                          "task_optpar"}
    e2 = EWC(model, ewc_lambda=1.0)
    e2.load_state_dict(sdict)
    assert abs(e2.penalty().item() - ewc.penalty().item()) <= 1e-5 * abs(pen.item())


def test_penalty_gradient_is_lambda_fisher_delta_and_scales_with_upstream():
    from nerve_cl.continual import EWC
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(10, 32), nn.ReLU(), nn.Linear(32, 10)).cuda()
    ewc = EWC(model, ewc_lambda=1000)
    data = [(torch.randn(32, 10), torch.randn(32, 10)) for _ in range(3)]
    ewc.register_task(0, data)
    assert ewc.num_tasks == 1
    before = ewc.penalty().item()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn_like(p) * 0.1)
    model.zero_grad()
    pen = ewc.penalty()
    assert pen.item() > before                        # reference tests/test_continual.py:71-89
    (0.5 * pen).backward()
    named = list(model.named_parameters())
    want_pen = sr_oracle.ewc_penalty([(n, p.detach().cpu()) for n, p in named],
                                     {n: v.cpu() for n, v in ewc.fisher_dict.items()},
                                     {n: v.cpu() for n, v in ewc.optpar_dict.items()}, 1000.0)
    assert abs(pen.item() - want_pen.item()) <= 1e-5 * abs(want_pen.item())
    for n, p in named:
        want = 0.5 * 1000 * ewc.fisher_dict[n] * (p.detach() - ewc.optpar_dict[n])
        assert torch.allclose(p.grad, want, rtol=1e-5, atol=1e-8), n


def test_separate_mode_and_cpu_refusal():
    from nerve_cl.continual import EWC
    torch.manual_seed(1)
    model = nn.Linear(6, 4).cuda()
    ewc = EWC(model, ewc_lambda=10.0, mode="separate")
    for t in range(2):
        ewc.register_task(t, [(torch.randn(8, 6), torch.randn(8, 4))])
        with torch.no_grad():
            model.weight.add_(0.05)
    pen = ewc.penalty()
    want = 0.0
    for t in range(2):
        for n, p in model.named_parameters():
            want = want + (ewc.task_fisher[t][n] * (p.detach() - ewc.task_optpar[t][n]) ** 2).sum()
    assert abs(pen.item() - 5.0 * want.item()) <= 1e-5 * abs(5.0 * want.item())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        EWC(nn.Linear(3, 3)).register_task(0, [(torch.randn(2, 3), torch.randn(2, 3))])


def test_engine_dict_surface_on_gpu():
    """reference tests/test_models.py:82-98: SR-only engine returns 'enhanced' of shape (1,3,128,128)."""
    from nerve_cl.models import EnhancementConfig, EnhancementEngine
    eng = EnhancementEngine(EnhancementConfig(frame_recovery_enabled=False, super_resolution_enabled=True,
                                              sr_num_features=16, sr_num_residual_blocks=1)).cuda()
    frames = torch.rand(1, 3, 3, 64, 64, device="cuda")
    res = eng(frames)
    assert set(res) == {"super_resolved", "enhanced"} and res["enhanced"].shape == (1, 3, 128, 128)
    assert torch.equal(res["enhanced"], res["super_resolved"])
    res["enhanced"].mean().backward()
    assert eng.enhancement_strength.grad is None      # read through .item(): never trains
    assert all(p.grad is not None for p in eng.super_resolution.parameters())
    short = eng(frames[:, :2], center_idx=1)          # right-padded by repeating the last frame
    assert short["enhanced"].shape == (1, 3, 128, 128)


def test_engine_strength_blend_and_enhance_video():
    """strength < 1 blends with the bicubic-upsampled centre frame (reference enhancement_engine.py:172-180);
    enhance_video slides the window over a clip (:186-248)."""
    import torch.nn.functional as F
    from nerve_cl.models import EnhancementConfig, EnhancementEngine
    torch.manual_seed(0)
    eng = EnhancementEngine(EnhancementConfig(frame_recovery_enabled=False, super_resolution_enabled=True,
                                              sr_num_features=16, sr_num_residual_blocks=1)).cuda().eval()
    frames = torch.rand(2, 3, 3, 20, 28, device="cuda")
    full = eng(frames)["enhanced"]
    half = eng(frames, enhancement_strength=0.25)
    want = 0.25 * half["super_resolved"] + 0.75 * F.interpolate(frames[:, 1].cpu(), size=(40, 56), mode="bicubic",
                                                                align_corners=False).cuda()
    assert torch.equal(half["super_resolved"], full)
    assert (half["enhanced"] - want).abs().max().item() < 2e-6
    half["enhanced"].sum().backward()
    g = next(eng.super_resolution.parameters()).grad
    assert g is not None and torch.isfinite(g).all()
    with torch.no_grad():
        vid = eng.enhance_video(torch.rand(5, 3, 16, 16, device="cuda"))
    assert vid.shape == (5, 3, 32, 32)


def test_adaptive_engine_strength_mode_and_names():
    """AdaptiveEnhancementEngine.adaptive_forward (reference enhancement_engine.py:336-381): strength =
    clamp(0.3*budget + 0.3*pref + 0.4*mean(complexity), 0.3, 1), mode from the budget, extra result keys."""
    import nerve_cl
    from nerve_cl.models import AdaptiveEnhancementEngine, EnhancementConfig, FrameRecoveryNet
    assert {"FrameRecoveryNet", "SuperResolutionNet", "EnhancementEngine", "EpisodicMemory", "EWC", "MAML"} <= set(nerve_cl.__all__)
    assert FrameRecoveryNet(base_channels=16).get_num_parameters() > 0      # built (tests/test_frame_recovery_gpu.py)
    torch.manual_seed(1)
    eng = AdaptiveEnhancementEngine(EnhancementConfig(frame_recovery_enabled=False, sr_num_features=16,
                                                      sr_num_residual_blocks=1)).cuda().eval()
    frames = torch.rand(2, 3, 3, 24, 24, device="cuda")
    with torch.no_grad():
        res = eng.adaptive_forward(frames, resource_budget=0.5, user_quality_preference=0.8)
        cx = eng.estimate_complexity(frames[:, 1])
        plain = eng(frames, enhancement_strength=res["enhancement_strength"])["enhanced"]
    assert res["complexity"].shape == (2, 1) and torch.equal(res["complexity"], cx)
    want = min(1.0, max(0.3, 0.3 * 0.5 + 0.3 * 0.8 + 0.4 * cx.mean().item()))
    assert abs(res["enhancement_strength"] - want) < 1e-7 and want < 1.0
    assert torch.equal(res["enhanced"], plain) and not torch.equal(res["enhanced"], res["super_resolved"])
    assert (eng.config.frame_recovery_enabled, eng.config.super_resolution_enabled) == (False, True)   # 'sr_only'
    with torch.no_grad():
        eng.adaptive_forward(frames, resource_budget=0.1)
    assert eng.config.use_lightweight_sr        # flag only; the modules are not rebuilt (as in the reference)


@pytest.mark.parametrize("bf16", [False, True])
def test_enhance_video_feature_cache_matches_uncached(bf16):
    """enhance_video(cache_features=True) extracts every frame's features once for the whole clip (11 frames > NVQ_MAX_T, so
    in two launches) and must give the results of the per-window path, clip boundaries and strength blend included."""
    from nerve_cl import _nvq
    from nerve_cl.models import EnhancementConfig, EnhancementEngine
    torch.manual_seed(1)
    eng = EnhancementEngine(EnhancementConfig(frame_recovery_enabled=False, super_resolution_enabled=True,
                                              sr_num_features=64 if bf16 else 16, sr_num_residual_blocks=1)).cuda().eval()
    if bf16:
        eng.super_resolution.math_mode, eng.super_resolution.bf16_activations = _nvq.MATH_BF16, True
    # non-trivial running statistics
    for n, b in eng.super_resolution.named_buffers():
        if n.endswith("running_mean"):
            b.copy_(torch.randn_like(b) * 0.1)
        elif n.endswith("running_var"):
            b.copy_(torch.rand_like(b) + 0.5)
    vid = torch.rand(11, 3, 24, 40, device="cuda")
    with torch.no_grad():
        a = eng.enhance_video(vid, cache_features=False)
        b = eng.enhance_video(vid, cache_features=True)
        eng.enhancement_strength.fill_(0.6)
        c = eng.enhance_video(vid.unsqueeze(0).repeat(2, 1, 1, 1, 1), cache_features=False)
        d = eng.enhance_video(vid.unsqueeze(0).repeat(2, 1, 1, 1, 1), cache_features=True)
    assert a.shape == (11, 3, 48, 80) and torch.equal(a, b)
    assert c.shape == (2, 11, 3, 48, 80) and torch.equal(c, d)


def test_synaptic_intelligence_matches_reference_formulas():
    """SynapticIntelligence (reference ewc.py:306-379) on flat buckets + the HIP penalty kernels against a direct
    per-tensor evaluation of the reference's formulas on the same parameter / gradient sequence."""
    from nerve_cl.continual import SynapticIntelligence
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3)).cuda()
    si = SynapticIntelligence(model, si_lambda=0.7, damping=0.1)
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    # reference bookkeeping, per tensor
    W = {n: torch.zeros_like(p) for n, p in model.named_parameters()}
    p_old = {n: p.detach().clone() for n, p in model.named_parameters()}
    omega = {n: torch.zeros_like(p) for n, p in model.named_parameters()}
    x, y = torch.randn(16, 7, device="cuda"), torch.randn(16, 3, device="cuda")
    for task in range(2):
        for _ in range(3):
            opt.zero_grad()
            (torch.nn.functional.mse_loss(model(x), y) + si.penalty()).backward()
            opt.step()
            si.update_importance()
            for n, p in model.named_parameters():
                W[n] += -p.grad * (p.detach() - p_old[n])
                p_old[n] = p.detach().clone()
        si.register_task()
        for n, p in model.named_parameters():
            delta = p.detach() - p_old[n]
            omega[n] += W[n] / (delta ** 2 + 0.1)
            W[n] = torch.zeros_like(p)
            p_old[n] = p.detach().clone()
        with torch.no_grad():                      # move the weights so that the penalty is non-zero
            for p in model.parameters():
                p.add_(0.05 * torch.randn_like(p))
        want = 0.7 * sum((omega[n] * (p - p_old[n]) ** 2).sum() for n, p in model.named_parameters())
        got = si.penalty()
        assert abs(got.item() - want.item()) <= 1e-5 * max(1.0, abs(want.item()))
        model.zero_grad()
        got.backward()
        for n, p in model.named_parameters():
            ref = 2 * 0.7 * omega[n] * (p.detach() - p_old[n])
            assert (p.grad - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item()), n
            assert torch.allclose(si.omega[n], omega[n], rtol=1e-5, atol=1e-7)
