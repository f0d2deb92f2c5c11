"""Host-side replay buffer used by the 'replay' strategy of train_continual.py.

Not on the GPU hot path: a bounded list of (lr, hr, metadata) samples kept
on the CPU with reservoir / FIFO / per-content-type stratified replacement, and batch sampling.
Interface follows the calls the reference scripts and tests make
(experiments/train_continual.py:93-108, tests/test_continual.py:17-55)."""
from __future__ import annotations

import random
from collections import defaultdict
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import torch


@dataclass
class MemorySample:
    """One stored sample as the reference exposes it (memory.py:16-34: ``memory.buffer`` is a list of these, and it is the
    store itself: appending / removing samples or changing ``importance`` / ``access_count`` through it takes effect)."""
    frame_lr: torch.Tensor
    frame_hr: torch.Tensor
    metadata: Dict[str, Any] = field(default_factory=dict)
    importance: float = 1.0
    access_count: int = 0

    def to(self, device: torch.device) -> "MemorySample":
        return MemorySample(self.frame_lr.to(device), self.frame_hr.to(device), self.metadata, self.importance,
                            self.access_count)


class EpisodicMemory:
    """Bounded CPU-side sample store.  Semantics follow the reference (memory.py:38-349): eviction per strategy once full,
    ``sample(batch_size, content_type, device)`` draws WITHOUT replacement and never more than is stored, and
    ``save`` / ``load`` use the reference's on-disk dictionary {'buffer': [(lr, hr, metadata, importance)], 'total_seen',
    'strategy', 'capacity'} so files written by either side load on the other (tensors, dicts and floats only:
    ``weights_only=True`` suffices)."""

    STRATEGIES = ("uniform", "reservoir", "stratified", "fifo", "importance", "diversity")

    def __init__(self, capacity: int = 1000, strategy: str = "reservoir", diversity_weight: float = 0.3,
                 seed: Optional[int] = None):
        if strategy not in self.STRATEGIES:
            raise ValueError(f"unknown strategy {strategy!r}")
        self.capacity, self.strategy, self.diversity_weight = capacity, strategy, diversity_weight
        self._items: List[MemorySample] = []
        self._seen = 0
        self._rng = random.Random(seed)

    def __len__(self) -> int:
        return len(self._items)

    @property
    def total_seen(self) -> int:
        return self._seen

    @property
    def buffer(self) -> List[MemorySample]:
        """The stored samples: the LIVE list, as in the reference (memory.py:52)."""
        return self._items

    @buffer.setter
    def buffer(self, items: List[MemorySample]) -> None:
        self._items = list(items)

    @staticmethod
    def _ctype(item: MemorySample) -> str:
        return item.metadata.get("content_type", "unknown")

    def _by_type(self) -> Dict[str, List[int]]:
        groups: Dict[str, List[int]] = defaultdict(list)
        for i, it in enumerate(self._items):
            groups[self._ctype(it)].append(i)
        return groups

    def _reservoir(self, item: dict) -> bool:
        """Keep the newcomer with probability capacity / seen, in a uniformly chosen slot (reference :133-148)."""
        if self._rng.random() < self.capacity / self._seen:
            self._items[self._rng.randrange(self.capacity)] = item
            return True
        return False

    def store(self, frame_lr: torch.Tensor, frame_hr: torch.Tensor, metadata: Optional[Dict[str, Any]] = None,
              importance: float = 1.0) -> bool:
        """Reference memory.py:84-130 (same argument names: the scripts pass them positionally, callers may use keywords)."""
        item = MemorySample(frame_lr.detach().cpu(), frame_hr.detach().cpu(), metadata or {}, float(importance), 0)
        self._seen += 1
        if len(self._items) < self.capacity:
            self._items.append(item)
            return True
        if self.strategy == "reservoir":
            return self._reservoir(item)
        if self.strategy == "stratified":
            # a type that is not (yet) the largest one takes a slot from the largest; otherwise reservoir (reference :150-169)
            groups = self._by_type()
            biggest = max(groups, key=lambda k: len(groups[k]))
            if len(groups.get(self._ctype(item), [])) < len(groups[biggest]):
                self._items[self._rng.choice(groups[biggest])] = item
                return True
            return self._reservoir(item)
        if self.strategy == "importance":
            j = min(range(len(self._items)), key=lambda k: self._items[k].importance)
            if item.importance > self._items[j].importance:
                self._items[j] = item
                return True
            return False
        if self.strategy == "diversity":
            # replace the stored sample closest in mean colour if the newcomer is further than 0.1 from it (reference :186-211)
            feats = torch.stack([it.frame_lr.mean(dim=(1, 2)) for it in self._items])
            dist = torch.norm(feats - item.frame_lr.mean(dim=(1, 2)), dim=1)
            j = int(dist.argmin())
            if dist[j] > 0.1:
                self._items[j] = item
                return True
            return False
        self._items.pop(0)                                        # 'uniform' / 'fifo': first in, first out
        self._items.append(item)
        return True

    def _spread(self, batch_size: int) -> List[int]:
        """batch_size indices split evenly over the content types, the first types taking the remainder (reference :287-305)."""
        groups = self._by_type()
        per, rem = divmod(batch_size, len(groups))
        idx: List[int] = []
        for members in groups.values():
            n = min(per + (1 if rem > 0 else 0), len(members))
            rem -= 1
            idx.extend(self._rng.sample(members, n))
        return idx[:batch_size]

    def sample(self, batch_size: int = 32, content_type: Optional[str] = None, device: Optional[torch.device] = None):
        if not self._items:
            raise ValueError("Memory buffer is empty")
        batch_size = min(batch_size, len(self._items))
        groups = self._by_type()
        if content_type is not None and content_type in groups:
            idx = self._rng.sample(groups[content_type], min(batch_size, len(groups[content_type])))
        else:
            idx = self._spread(batch_size)
        for i in idx:
            self._items[i].access_count += 1
        lr = torch.stack([self._items[i].frame_lr for i in idx])
        hr = torch.stack([self._items[i].frame_hr for i in idx])
        if device is not None:
            lr, hr = lr.to(device), hr.to(device)
        return lr, hr, [self._items[i].metadata for i in idx]

    def get_stats(self) -> Dict[str, Any]:
        return {"size": len(self), "capacity": self.capacity, "utilization": len(self) / self.capacity,
                "total_seen": self._seen, "content_distribution": {k: len(v) for k, v in self._by_type().items()},
                "strategy": self.strategy}

    def clear(self) -> None:
        self._items, self._seen = [], 0

    def save(self, path: str) -> None:
        torch.save({"buffer": [(it.frame_lr, it.frame_hr, it.metadata, it.importance) for it in self._items],
                    "total_seen": self._seen, "strategy": self.strategy, "capacity": self.capacity}, path)

    def load(self, path: str) -> None:
        blob = torch.load(path, weights_only=True)
        self._items = [MemorySample(lr, hr, meta, float(imp), 0) for lr, hr, meta, imp in blob["buffer"]]
        self._seen = blob["total_seen"]


class StreamingEpisodicMemory(EpisodicMemory):
    """Reservoir memory with a recency-biased sampler (reference memory.py:352-440): sampling weight of an item =
    (1 - recency_weight) * importance + recency_weight / (1 + now - time stored)."""

    def __init__(self, capacity: int = 1000, recency_weight: float = 0.2, compress_old: bool = True, seed: Optional[int] = None):
        super().__init__(capacity, strategy="reservoir", seed=seed)
        self.recency_weight, self.compress_old = recency_weight, compress_old
        self.current_time = 0

    def store(self, frame_lr: torch.Tensor, frame_hr: torch.Tensor, metadata: Optional[Dict[str, Any]] = None,
              importance: float = 1.0) -> bool:
        self.current_time += 1
        meta = dict(metadata or {})
        meta["_time"] = self.current_time
        return super().store(frame_lr, frame_hr, meta, importance)

    def sample(self, batch_size: int = 32, content_type: Optional[str] = None, device: Optional[torch.device] = None,
               use_recency: bool = True):
        if not use_recency:
            return super().sample(batch_size, content_type=content_type, device=device)
        if not self._items:
            raise ValueError("memory is empty")
        batch_size = min(batch_size, len(self._items))
        w = [(1 - self.recency_weight) * it.importance + self.recency_weight / (1 + self.current_time - it.metadata["_time"])
             for it in self._items]
        idx: List[int] = []
        pool = list(range(len(self._items)))
        for _ in range(batch_size):                           # weighted sampling without replacement
            j = self._rng.choices(range(len(pool)), weights=[w[i] for i in pool])[0]
            idx.append(pool.pop(j))
        lr = torch.stack([self._items[i].frame_lr for i in idx])
        hr = torch.stack([self._items[i].frame_hr for i in idx])
        if device is not None:
            lr, hr = lr.to(device), hr.to(device)
        return lr, hr, [{k: v for k, v in self._items[i].metadata.items() if k != "_time"} for i in idx]
