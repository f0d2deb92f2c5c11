"""CPU: host-side continual-learning helpers (replay memory, FOMAML, distillation) stay importable and
behave as the reference's own tests expect (tests/test_continual.py:14-55,92-123)."""
import pytest
import torch
import torch.nn as nn

from nerve_cl.continual import ContinualDistillation, EpisodicMemory, EWC, FOMAML


def test_memory_store_sample_capacity_and_stratification():
    m = EpisodicMemory(capacity=100)
    for _ in range(50):
        m.store(torch.randn(3, 32, 32), torch.randn(3, 64, 64), {"content_type": "test"})
    assert len(m) == 50
    lr, hr, meta = m.sample(batch_size=16)
    assert lr.shape == (16, 3, 32, 32) and hr.shape == (16, 3, 64, 64) and len(meta) == 16
    m2 = EpisodicMemory(capacity=20)
    for _ in range(50):
        m2.store(torch.randn(3, 8, 8), torch.randn(3, 16, 16))
    assert len(m2) == 20
    m3 = EpisodicMemory(capacity=30, strategy="stratified", seed=0)
    for ct in ("sports", "animation", "movie"):
        for _ in range(20):
            m3.store(torch.randn(3, 8, 8), torch.randn(3, 16, 16), {"content_type": ct})
    dist = m3.get_stats()["content_distribution"]
    # reference semantics (memory.py:150-169): an under-represented type takes slots from the largest one, a type that
    # has caught up falls back to reservoir replacement - balanced up to the reservoir's noise, never starved
    assert len(dist) == 3 and min(dist.values()) >= 6 and sum(dist.values()) == 30
    # on-disk format of the reference (memory.py:325-349) and draws without replacement, capped at what is stored
    import os, tempfile
    with tempfile.TemporaryDirectory() as d:
        m3.save(os.path.join(d, "mem.pt"))
        blob = torch.load(os.path.join(d, "mem.pt"), weights_only=True)
        assert set(blob) == {"buffer", "total_seen", "strategy", "capacity"} and len(blob["buffer"][0]) == 4
        m4 = EpisodicMemory(capacity=30, strategy="stratified")
        m4.load(os.path.join(d, "mem.pt"))
        assert len(m4) == 30 and m4.get_stats()["content_distribution"] == dist and m4.total_seen == 60
    lr, hr, meta = m4.sample(64)
    assert 24 <= lr.shape[0] <= 30            # per-type quotas are capped by what each type holds (reference :287-305)
    lr, hr, meta = m4.sample(4, "sports")
    assert all(mm["content_type"] == "sports" for mm in meta)
    assert set(m4.get_stats()) == {"size", "capacity", "utilization", "total_seen", "content_distribution", "strategy"}


def test_fomaml_adapt_and_distillation_on_plain_modules():
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(10, 32), nn.ReLU(), nn.Linear(32, 10))
    maml = FOMAML(model, inner_lr=0.01, inner_steps=5)
    data = (torch.randn(16, 10), torch.randn(16, 10))
    adapted = maml.adapt(data, nn.MSELoss())
    assert adapted is not model and isinstance(adapted, nn.Module)
    assert nn.MSELoss()(adapted(data[0]), data[1]) < nn.MSELoss()(model(data[0]), data[1])
    cd = ContinualDistillation(model)
    losses = cd.compute_loss(data[0], data[1], nn.MSELoss())
    assert float(losses["distill"]) == 0.0
    cd.register_task()
    losses = cd.compute_loss(data[0], data[1], nn.MSELoss())
    assert set(losses) == {"task", "distill", "total"} and cd.task_count == 1


def test_ewc_constructs_on_cpu_but_refuses_to_compute_there():
    ewc = EWC(nn.Linear(4, 4), ewc_lambda=10)
    assert ewc.penalty() == 0.0 and ewc.num_tasks == 0 and ewc.mode == "online" and ewc.decay == 0.999


def test_maml_train_step_reptile_and_streaming_memory_on_plain_modules():
    """Reference package surface (nerve_cl/continual/__init__.py): MAML / Reptile / ContentAdaptiveMAML /
    StreamingEpisodicMemory run on any nn.Module (these helpers are host-side loops)."""
    from nerve_cl.continual import MAML, Reptile, ContentAdaptiveMAML, StreamingEpisodicMemory
    torch.manual_seed(0)
    loss = nn.MSELoss()

    def make_tasks():
        return [{"support": (torch.randn(8, 6), torch.randn(8, 3)), "query": (torch.randn(8, 6), torch.randn(8, 3))}
                for _ in range(3)]
    model = nn.Sequential(nn.Linear(6, 16), nn.Tanh(), nn.Linear(16, 3))
    maml = MAML(model, inner_lr=0.05, outer_lr=1e-2, inner_steps=2)
    before = [p.detach().clone() for p in model.parameters()]
    v = maml.train_step(make_tasks(), loss)
    assert isinstance(v, float) and any(not torch.equal(a, b) for a, b in zip(before, model.parameters()))
    state = maml.state_dict()
    assert set(state) == {"model", "meta_optimizer", "inner_lr", "outer_lr", "inner_steps", "first_order"}
    maml.load_state_dict(state)
    with pytest.raises(NotImplementedError):
        MAML(model, first_order=False).meta_step(make_tasks(), loss)
    # Reptile: theta <- theta + outer_lr * (mean adapted - theta); with outer_lr = 0 nothing moves
    rep = Reptile(model, inner_lr=0.05, outer_lr=0.0, inner_steps=2)
    before = [p.detach().clone() for p in model.parameters()]
    rep.train_step(make_tasks(), loss)
    assert all(torch.equal(a, b) for a, b in zip(before, model.parameters()))
    Reptile(model, outer_lr=0.5, inner_steps=2).train_step(make_tasks(), loss)
    assert any(not torch.equal(a, b) for a, b in zip(before, model.parameters()))
    cam = ContentAdaptiveMAML(model, ["sports", "animation"], inner_lr=0.02)
    assert isinstance(cam.adapt_to_content(make_tasks()[0]["support"], "sports", loss, steps=1), nn.Module)
    mem = StreamingEpisodicMemory(capacity=16, recency_weight=0.9, seed=1)
    for i in range(40):
        mem.store(torch.full((3, 4, 4), float(i)), torch.full((3, 8, 8), float(i)), {"content_type": "a", "i": i})
    lr, hr, meta = mem.sample(8)
    assert lr.shape == (8, 3, 4, 4) and hr.shape == (8, 3, 8, 8) and len({m["i"] for m in meta}) == 8
    assert all("_time" not in m for m in meta)
