"""Radix tree over token ids whose values are KV slot ids: the prefix cache.

Behavioural counterpart of the reference's ``RadixCache`` (reference:
radix_cache.py:22-272, radix_tree.py:6-18): a compressed trie keyed by token ids, each
node holding the KV slots of its key; matching splits a node where a key diverges
inside it, inserting reports how much of the key was already present, leaves are
evicted least-recently-used first, and a lock count per node (held by running
requests, propagated to the ancestors) keeps their slots from eviction.  The order in
which nodes are created, touched and evicted is the reference's, so the same request
stream yields the same slot ids.

Differences that do not change decisions: the tree lives entirely on the host -- slot
ids are CPU int32 tensors, the driver keeps a host copy of every request's page-table
row, so no device read is needed to cache a request -- and recency is a counter
instead of wall-clock time (no ties, reproducible).
"""

from __future__ import annotations

import heapq
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

_NO_SLOTS = torch.empty(0, dtype=torch.int32)


class PrefixNode:
    __slots__ = ("parent", "children", "key", "slots", "lock_ref", "touched")

    def __init__(self, touched: int):
        self.parent: Optional["PrefixNode"] = None
        self.children: Dict[int, "PrefixNode"] = {}  # first token of the child's key -> child
        self.key: List[int] = []
        self.slots: torch.Tensor = _NO_SLOTS
        self.lock_ref = 0
        self.touched = touched

    def __lt__(self, other: "PrefixNode") -> bool:  # heap order: least recently touched first
        return self.touched < other.touched


def _common_prefix(a: Sequence[int], b: Sequence[int]) -> int:
    n = 0
    for x, y in zip(a, b):
        if x != y:
            break
        n += 1
    return n


class PrefixCache:
    def __init__(self):
        self.reset()

    def reset(self) -> None:
        self._clock = 0
        self.root = PrefixNode(self._tick())
        self.root.lock_ref = 1  # never evicted
        self._evictable = 0

    def _tick(self) -> int:
        self._clock += 1
        return self._clock

    # ------------------------------------------------------------------ lookup / insert
    def match_prefix(self, key: Sequence[int]) -> Tuple[torch.Tensor, PrefixNode]:
        """Slots of the longest cached prefix of ``key`` and the node it ends in.  A
        match that stops inside a node splits that node (radix_cache.py:39-85)."""
        found: List[torch.Tensor] = []
        node, key = self.root, list(key)
        last = self.root
        while True:
            node.touched = self._tick()
            if not key or key[0] not in node.children:
                break
            child = node.children[key[0]]
            n = _common_prefix(child.key, key)
            if n < len(child.key):
                child = self._split(child, n)
                found.append(child.slots)
                last = child
                break
            found.append(child.slots)
            last = child
            node, key = child, key[n:]
        return (torch.cat(found) if found else _NO_SLOTS.clone()), last

    def _split(self, child: PrefixNode, n: int) -> PrefixNode:
        """New node with the first ``n`` tokens of ``child``, which keeps the rest
        (radix_cache.py:87-99)."""
        head = PrefixNode(self._tick())
        head.children = {child.key[n]: child}
        head.parent = child.parent
        head.lock_ref = child.lock_ref
        head.key, head.slots = child.key[:n], child.slots[:n]
        child.parent = head
        child.key, child.slots = child.key[n:], child.slots[n:]
        head.parent.children[head.key[0]] = head
        return head

    def insert(self, key: Sequence[int], slots: torch.Tensor) -> int:
        """Add ``key`` -> ``slots``; returns how many leading tokens were already
        cached (their slots in ``slots`` are duplicates the caller frees)
        (radix_cache.py:101-139)."""
        node, key, matched = self.root, list(key), 0
        while True:
            node.touched = self._tick()
            if not key:
                return matched
            if key[0] not in node.children:
                leaf = PrefixNode(self._tick())
                leaf.key, leaf.slots = key, slots
                leaf.parent = node
                node.children[key[0]] = leaf
                self._evictable += len(slots)
                return matched
            child = node.children[key[0]]
            n = _common_prefix(child.key, key)
            if n < len(child.key):
                child = self._split(child, n)
            elif n == len(key):
                return matched + n
            matched += n
            node, key, slots = child, key[n:], slots[n:]

    # ------------------------------------------------------------------ locks
    def inc_lock_ref(self, node: Optional[PrefixNode]) -> None:
        while node is not None and node is not self.root:
            if node.lock_ref == 0:
                self._evictable -= len(node.slots)
            node.lock_ref += 1
            node = node.parent

    def dec_lock_ref(self, node: Optional[PrefixNode]) -> None:
        while node is not None and node is not self.root:
            if node.lock_ref == 1:
                self._evictable += len(node.slots)
            node.lock_ref -= 1
            node = node.parent

    def evictable_size(self) -> int:
        return self._evictable

    # ------------------------------------------------------------------ eviction
    def evict(self, num_tokens: int, free_slots: Callable[[torch.Tensor], None]) -> None:
        """Free at least ``num_tokens`` slots of unlocked leaves, least recently
        touched first; a parent left without children becomes a candidate
        (radix_cache.py:209-229)."""
        leaves, stack = [], [self.root]
        while stack:
            n = stack.pop()
            if n.children:
                stack.extend(n.children.values())
            else:
                leaves.append(n)
        heapq.heapify(leaves)
        freed = 0
        while freed < num_tokens and leaves:
            n = heapq.heappop(leaves)
            if n is self.root:
                break
            if n.lock_ref > 0:
                continue
            free_slots(n.slots)
            freed += len(n.slots)
            del n.parent.children[n.key[0]]
            self._evictable -= len(n.slots)
            if not n.parent.children:
                heapq.heappush(leaves, n.parent)
