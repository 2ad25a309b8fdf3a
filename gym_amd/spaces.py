"""Box / Discrete / MultiDiscrete — the three space types the classic-control path uses.

Host-side mirror of gym/spaces/{space,box,discrete,multi_discrete}.py and of
gym/vector/utils/spaces.py:batch_space, restricted to what the path needs (same attribute
names, same sample()/contains()/seed() meaning, same PCG64 seeding as gym/utils/seeding.py:9-27
so that `space.seed(s); space.sample()` yields the reference's numbers).  These are plain host
objects: device-side sampling is the Philox action stream of the engine, not these.
"""
from __future__ import annotations

from copy import deepcopy
from typing import Optional, Sequence

import numpy as np

from . import error


def np_random(seed: Optional[int] = None):
    """gym/utils/seeding.py:9-27."""
    if seed is not None and not (isinstance(seed, (int, np.integer)) and 0 <= seed):
        raise error.Error(f"Seed must be a non-negative integer or omitted, not {seed}")
    seed_seq = np.random.SeedSequence(None if seed is None else int(seed))
    return np.random.Generator(np.random.PCG64(seed_seq)), seed_seq.entropy


class Space:
    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._np_random = None
        if seed is not None:
            if isinstance(seed, np.random.Generator):
                self._np_random = seed
            else:
                self.seed(seed)

    @property
    def np_random(self) -> np.random.Generator:
        if self._np_random is None:
            self.seed()
        return self._np_random

    @property
    def shape(self):
        return self._shape

    def seed(self, seed: Optional[int] = None) -> list:
        self._np_random, s = np_random(seed)
        return [s]

    def sample(self, mask=None):
        raise NotImplementedError

    def contains(self, x) -> bool:
        raise NotImplementedError

    def __contains__(self, x) -> bool:
        return self.contains(x)


class Box(Space):
    """Bounded/unbounded box in R^n (gym/spaces/box.py:53-278)."""

    def __init__(self, low, high, shape: Optional[Sequence[int]] = None, dtype=np.float32, seed=None):
        dtype = np.dtype(dtype)
        if shape is not None:
            shape = tuple(int(d) for d in shape)
        elif isinstance(low, np.ndarray):
            shape = low.shape
        elif isinstance(high, np.ndarray):
            shape = high.shape
        elif np.isscalar(low) and np.isscalar(high):
            shape = (1,)
        else:
            raise ValueError("Box shape is inferred from low and high, expect their types to be np.ndarray, an integer or a float")
        low = np.full(shape, low, dtype=float) if np.isscalar(low) else np.asarray(low)
        high = np.full(shape, high, dtype=float) if np.isscalar(high) else np.asarray(high)
        assert low.shape == shape and high.shape == shape, "low/high shape mismatch"
        self.bounded_below = -np.inf < low
        self.bounded_above = np.inf > high
        self.low = low.astype(dtype)
        self.high = high.astype(dtype)
        super().__init__(shape, dtype, seed)

    def is_bounded(self, manner: str = "both") -> bool:
        below, above = bool(np.all(self.bounded_below)), bool(np.all(self.bounded_above))
        if manner == "both":
            return below and above
        if manner == "below":
            return below
        if manner == "above":
            return above
        raise ValueError(f"manner is not in {{'below', 'above', 'both'}}, actual value: {manner}")

    def sample(self, mask=None) -> np.ndarray:
        """Same distribution per coordinate and — because seeded streams must reproduce the reference's numbers — the same
        ORDER of generator calls as gym/spaces/box.py:171-222: normal for the coordinates without any bound, shifted
        exponential for those bounded on one side (lower-bounded first), uniform for the two-sided ones."""
        if mask is not None:
            raise error.Error(f"Box.sample cannot be provided a mask, actual value: {mask}")
        rng = self.np_random
        lo_ok, hi_ok = self.bounded_below, self.bounded_above
        top = self.high if self.dtype.kind == "f" else self.high.astype("int64") + 1   # integer boxes include `high`
        out = np.empty(self.shape)
        draws = (
            (~lo_ok & ~hi_ok, lambda m, k: rng.normal(size=k)),
            (lo_ok & ~hi_ok, lambda m, k: rng.exponential(size=k) + self.low[m]),
            (~lo_ok & hi_ok, lambda m, k: -rng.exponential(size=k) + self.high[m]),
            (lo_ok & hi_ok, lambda m, k: rng.uniform(low=self.low[m], high=top[m], size=k)),
        )
        for m, draw in draws:   # one generator call per group, even for an empty group (the reference draws size-0 arrays too)
            out[m] = draw(m, m[m].shape)
        if self.dtype.kind == "i":
            out = np.floor(out)
        return out.astype(self.dtype)

    def contains(self, x) -> bool:
        """Membership as in gym/spaces/box.py:224-238: castable dtype, exact shape, inside the closed bounds."""
        if not isinstance(x, np.ndarray):
            try:
                x = np.asarray(x, dtype=self.dtype)
            except (ValueError, TypeError):
                return False
        if x.shape != self.shape or not np.can_cast(x.dtype, self.dtype):
            return False
        return bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self) -> str:
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    def __eq__(self, other) -> bool:
        return (isinstance(other, Box) and self.shape == other.shape and self.dtype == other.dtype
                and np.allclose(self.low, other.low) and np.allclose(self.high, other.high))


class Discrete(Space):
    """{start, ..., start+n-1} (gym/spaces/discrete.py:9-126)."""

    def __init__(self, n: int, seed=None, start: int = 0):
        assert isinstance(n, (int, np.integer)) and n > 0, "n (counts) have to be positive"
        self.n = int(n)
        self.start = int(start)
        super().__init__((), np.int64, seed)

    def sample(self, mask=None) -> int:  # gym/spaces/discrete.py:47-81
        if mask is not None:
            assert isinstance(mask, np.ndarray) and mask.dtype == np.int8 and mask.shape == (self.n,)
            valid = mask == 1
            if np.any(valid):
                return int(self.start + self.np_random.choice(np.where(valid)[0]))
            return self.start
        return int(self.start + self.np_random.integers(self.n))

    def contains(self, x) -> bool:  # gym/spaces/discrete.py:83-94
        if isinstance(x, int):
            as_int = x
        elif isinstance(x, (np.generic, np.ndarray)) and (np.issubdtype(x.dtype, np.integer) and x.shape == ()):
            as_int = int(x)
        else:
            return False
        return self.start <= as_int < self.start + self.n

    def __repr__(self) -> str:
        return f"Discrete({self.n})" if self.start == 0 else f"Discrete({self.n}, start={self.start})"

    def __eq__(self, other) -> bool:
        return isinstance(other, Discrete) and self.n == other.n and self.start == other.start


class MultiDiscrete(Space):
    """Cartesian product of Discrete spaces (gym/spaces/multi_discrete.py:12-175)."""

    def __init__(self, nvec, dtype=np.int64, seed=None):
        self.nvec = np.array(nvec, dtype=dtype, copy=True)
        assert (self.nvec > 0).all(), "nvec (counts) have to be positive"
        super().__init__(self.nvec.shape, dtype, seed)

    def sample(self, mask=None) -> np.ndarray:  # gym/spaces/multi_discrete.py:69-123
        if mask is not None:
            raise error.Error("MultiDiscrete.sample masks are not supported by this adapter")
        return (self.np_random.random(self.nvec.shape) * self.nvec).astype(self.dtype)

    def contains(self, x) -> bool:  # gym/spaces/multi_discrete.py:125-138
        if isinstance(x, (list, tuple)):
            x = np.array(x)
        return bool(isinstance(x, np.ndarray) and x.shape == self.shape and x.dtype != object and np.all(0 <= x)
                    and np.all(x < self.nvec))

    def __len__(self):
        return len(self.nvec)

    def __repr__(self):
        return f"MultiDiscrete({self.nvec})"

    def __eq__(self, other):
        return isinstance(other, MultiDiscrete) and np.all(self.nvec == other.nvec)


class Tuple(Space):
    """Product of spaces (gym/spaces/tuple.py:12-160), as far as Blackjack's observation space needs it."""

    def __init__(self, spaces, seed=None):
        self.spaces = tuple(spaces)
        super().__init__(None, None, seed)

    def seed(self, seed=None) -> list:
        out = super().seed(seed)
        sub = self.np_random.integers(np.iinfo(np.int32).max, size=len(self.spaces))   # tuple.py:67-79: one sub-seed per space
        for sp, sd in zip(self.spaces, sub):
            out += sp.seed(int(sd))
        return out

    def sample(self, mask=None):
        return tuple(sp.sample() for sp in self.spaces)

    def contains(self, x) -> bool:
        if isinstance(x, (list, np.ndarray)):
            x = tuple(x)
        return isinstance(x, tuple) and len(x) == len(self.spaces) and all(sp.contains(v) for sp, v in zip(self.spaces, x))

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def __repr__(self):
        return "Tuple(" + ", ".join(str(s) for s in self.spaces) + ")"

    def __eq__(self, other):
        return isinstance(other, Tuple) and self.spaces == other.spaces


def batch_space(space: Space, n: int = 1) -> Space:
    """gym/vector/utils/spaces.py:17-125 for Box, Discrete, MultiDiscrete and Tuple."""
    if isinstance(space, Tuple):
        return Tuple(tuple(batch_space(sp, n) for sp in space.spaces), seed=deepcopy(space.np_random))
    if isinstance(space, Box):
        repeats = tuple([n] + [1] * space.low.ndim)
        low, high = np.tile(space.low, repeats), np.tile(space.high, repeats)
        return Box(low=low, high=high, dtype=space.dtype, seed=deepcopy(space.np_random))
    if isinstance(space, MultiDiscrete):                # :71-81: a Box of integer vectors, one row per sub-env
        high = np.tile(space.nvec, tuple([n] + [1] * space.nvec.ndim)) - 1
        return Box(low=np.zeros_like(high), high=high, dtype=space.dtype, seed=deepcopy(space.np_random))
    if isinstance(space, Discrete):
        if space.start == 0:
            return MultiDiscrete(np.full((n,), space.n, dtype=space.dtype), dtype=space.dtype,
                                 seed=deepcopy(space.np_random))
        return Box(low=space.start, high=space.start + space.n - 1, shape=(n,), dtype=space.dtype,
                   seed=deepcopy(space.np_random))
    raise ValueError(f"Cannot batch space with type `{type(space)}`.")
