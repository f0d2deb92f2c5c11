"""Host-side replay buffer used by the 'replay' strategy of train_continual.py.

Not on the GPU hot path: a bounded list of (lr, hr, metadata) samples kept
on the CPU with reservoir / FIFO / per-content-type stratified replacement, and batch sampling.
Interface follows the calls the reference scripts and tests make
(experiments/train_continual.py:93-108, tests/test_continual.py:17-55)."""
from __future__ import annotations

import random
from collections import defaultdict
from typing import Any, Dict, List, Optional

import torch


class EpisodicMemory:
    def __init__(self, capacity: int = 1000, strategy: str = "reservoir", seed: Optional[int] = None):
        if strategy not in ("reservoir", "stratified", "fifo", "importance", "diversity"):
            raise ValueError(f"unknown strategy {strategy!r}")
        self.capacity, self.strategy = capacity, strategy
        self._items: List[dict] = []
        self._seen = 0
        self._rng = random.Random(seed)

    def __len__(self) -> int:
        return len(self._items)

    def _by_type(self) -> Dict[str, List[int]]:
        groups: Dict[str, List[int]] = defaultdict(list)
        for i, it in enumerate(self._items):
            groups[it["meta"].get("content_type", "unknown")].append(i)
        return groups

    def store(self, lr: torch.Tensor, hr: torch.Tensor, metadata: Optional[Dict[str, Any]] = None,
              importance: float = 1.0) -> bool:
        item = {"lr": lr.detach().cpu(), "hr": hr.detach().cpu(), "meta": dict(metadata or {}),
                "importance": float(importance)}
        self._seen += 1
        if len(self._items) < self.capacity:
            self._items.append(item)
            return True
        if self.strategy == "fifo":
            self._items.pop(0)
            self._items.append(item)
            return True
        if self.strategy == "stratified":
            # evict from the most populous content type so that types stay balanced
            groups = self._by_type()
            biggest = max(groups.values(), key=len)
            own = groups.get(item["meta"].get("content_type", "unknown"), [])
            victims = own if len(own) >= len(biggest) else biggest   # never grow the largest class
            self._items[self._rng.choice(victims)] = item
            return True
        if self.strategy == "importance":
            j = min(range(len(self._items)), key=lambda k: self._items[k]["importance"])
            if self._items[j]["importance"] <= item["importance"]:
                self._items[j] = item
                return True
            return False
        # reservoir (also the fallback for 'diversity')
        j = self._rng.randrange(self._seen)
        if j < self.capacity:
            self._items[j] = item
            return True
        return False

    def sample(self, batch_size: int, device: Optional[torch.device] = None, content_type: Optional[str] = None):
        if not self._items:
            raise ValueError("memory is empty")
        pool = list(range(len(self._items)))
        if content_type is not None:
            pool = [i for i in pool if self._items[i]["meta"].get("content_type") == content_type] or pool
        if self.strategy == "stratified" and content_type is None:
            groups = list(self._by_type().values())
            idx = [self._rng.choice(groups[k % len(groups)]) for k in range(batch_size)]
        else:
            idx = [self._rng.choice(pool) for _ in range(batch_size)]
        lr = torch.stack([self._items[i]["lr"] for i in idx])
        hr = torch.stack([self._items[i]["hr"] for i in idx])
        if device is not None:
            lr, hr = lr.to(device), hr.to(device)
        return lr, hr, [self._items[i]["meta"] for i in idx]

    def get_stats(self) -> Dict[str, Any]:
        return {"size": len(self), "capacity": self.capacity, "total_seen": self._seen,
                "content_distribution": {k: len(v) for k, v in self._by_type().items()}}

    def clear(self) -> None:
        self._items, self._seen = [], 0

    def save(self, path: str) -> None:
        torch.save({"capacity": self.capacity, "strategy": self.strategy, "seen": self._seen,
                    "items": self._items}, path)

    def load(self, path: str) -> None:
        blob = torch.load(path, weights_only=False)   # a file this class wrote itself
        self.capacity, self.strategy, self._seen, self._items = blob["capacity"], blob["strategy"], blob["seen"], blob["items"]



class StreamingEpisodicMemory(EpisodicMemory):
    """Reservoir memory with a recency-biased sampler (reference memory.py:352-440): sampling weight of an item =
    (1 - recency_weight) * importance + recency_weight / (1 + now - time stored)."""

    def __init__(self, capacity: int = 1000, recency_weight: float = 0.2, compress_old: bool = True, seed: Optional[int] = None):
        super().__init__(capacity, strategy="reservoir", seed=seed)
        self.recency_weight, self.compress_old = recency_weight, compress_old
        self.current_time = 0

    def store(self, lr: torch.Tensor, hr: torch.Tensor, metadata: Optional[Dict[str, Any]] = None,
              importance: float = 1.0) -> bool:
        self.current_time += 1
        meta = dict(metadata or {})
        meta["_time"] = self.current_time
        return super().store(lr, hr, meta, importance)

    def sample(self, batch_size: int = 32, content_type: Optional[str] = None, device: Optional[torch.device] = None,
               use_recency: bool = True):
        if not use_recency:
            return super().sample(batch_size, device=device, content_type=content_type)
        if not self._items:
            raise ValueError("memory is empty")
        batch_size = min(batch_size, len(self._items))
        w = [(1 - self.recency_weight) * it["importance"] + self.recency_weight / (1 + self.current_time - it["meta"]["_time"])
             for it in self._items]
        idx: List[int] = []
        pool = list(range(len(self._items)))
        for _ in range(batch_size):                           # weighted sampling without replacement
            j = self._rng.choices(range(len(pool)), weights=[w[i] for i in pool])[0]
            idx.append(pool.pop(j))
        lr = torch.stack([self._items[i]["lr"] for i in idx])
        hr = torch.stack([self._items[i]["hr"] for i in idx])
        if device is not None:
            lr, hr = lr.to(device), hr.to(device)
        return lr, hr, [{k: v for k, v in self._items[i]["meta"].items() if k != "_time"} for i in idx]
