"""`gym.vector.utils` for the spaces of this engine (gym/vector/utils/__init__.py): the helpers SyncVectorEnv builds its batches with.

    batch_space(space, n)               gym/vector/utils/spaces.py:17-125   (gym_amd.spaces.batch_space)
    iterate(space, items)               gym/vector/utils/spaces.py:128-212  what step_async does with the `actions` argument
    create_empty_array(space, n, fn)    gym/vector/utils/numpy_utils.py:77-150  the preallocated observation buffer (sync_vector_env.py:62-64)
    concatenate(space, items, out)      gym/vector/utils/numpy_utils.py:14-74   np.stack of the sub-envs' observations (:159-161)

The device engine needs none of them on its own hot path — its batches are born batched — but code written against SyncVectorEnv uses
them around it (iterating a batched action, preallocating a rollout buffer), so they are here with the reference's semantics for Box,
Discrete, MultiDiscrete and Tuple, the spaces this engine has; a space of another kind is a `CustomSpaceError` / `ValueError` exactly
where the reference raises one.  Compared with the live reference in tests/test_host_logic.py."""
from __future__ import annotations

from typing import Callable, Iterable, Iterator

import numpy as np

from .. import error
from ..spaces import Box, Discrete, MultiDiscrete, Space, Tuple, batch_space

__all__ = ["batch_space", "iterate", "create_empty_array", "concatenate"]

_ARRAY_SPACES = (Box, Discrete, MultiDiscrete)     # one ndarray per batch


def _not_a_space(space):
    return ValueError(f"Space of type `{type(space)}` is not a valid `gym.Space` instance.")


def iterate(space: Space, items) -> Iterator:
    """The elements of a batch `items` of `batch_space(space, n)`, one per sub-env (spaces.py:128-212).  A Discrete space is not batched
    into something iterable by itself (its batch is a MultiDiscrete): TypeError, as in the reference (:163-165)."""
    if isinstance(space, Discrete):
        raise TypeError("Unable to iterate over a space of type `Discrete`.")
    if isinstance(space, (Box, MultiDiscrete)):
        try:
            return iter(items)
        except TypeError as e:
            raise TypeError(f"Unable to iterate over the following elements: {items}") from e
    if isinstance(space, Tuple):
        # a tuple of batches -> a sequence of tuples (:178-190)
        return zip(*(iterate(sub, items[i]) for i, sub in enumerate(space.spaces)))
    if isinstance(space, Space):
        raise error.CustomSpaceError(f"Unable to iterate over {items}, since {space} is a custom `gym.Space` instance "
                               "(i.e. not one of `Box`, `Dict`, etc...).")
    raise ValueError(f"Space of type `{type(space)}` is not a valid `gym.Space` instance.")


def create_empty_array(space: Space, n: int = 1, fn: Callable = np.zeros):
    """An array (tuple of arrays for Tuple) shaped like a batch of n elements of `space`; n = None: like one element (numpy_utils.py:77-150)."""
    if isinstance(space, _ARRAY_SPACES):
        shape = space.shape if n is None else (n,) + tuple(space.shape)
        return fn(shape, dtype=space.dtype)
    if isinstance(space, Tuple):
        return tuple(create_empty_array(sub, n=n, fn=fn) for sub in space.spaces)
    if isinstance(space, Space):
        return None                                    # custom spaces: nothing to preallocate (:148-150)
    raise _not_a_space(space)


def concatenate(space: Space, items: Iterable, out):
    """Stack the sub-envs' elements into `out` (from create_empty_array) and return it (numpy_utils.py:14-74)."""
    if isinstance(space, _ARRAY_SPACES):
        return np.stack(items, axis=0, out=out)
    if isinstance(space, Tuple):
        items = list(items)
        return tuple(concatenate(sub, [item[i] for item in items], out[i]) for i, sub in enumerate(space.spaces))
    if isinstance(space, Space):
        return tuple(items)                            # custom spaces are not stacked (:71-73)
    raise _not_a_space(space)
