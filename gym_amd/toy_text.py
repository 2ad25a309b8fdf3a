"""Tabular toy_text environments on the HIP engine: FrozenLake-v1, FrozenLake8x8-v1, Taxi-v3,
CliffWalking-v0.

The reference classes (gym/envs/toy_text/{frozen_lake,taxi,cliffwalking}.py) are table lookups: `__init__` enumerates the
MDP into `P[s][a] = [(prob, next_state, reward, terminated), ...]` and `step()` is `categorical_sample` over that list
(utils.py:4-8).  Here the host states the three MDPs (`frozen_lake_mdp`, `taxi_mdp`, `cliff_walking_mdp`: grid rules
written out from the reference's semantics, checked entry by entry against the reference's own `env.P` in the tests),
flattens them to dense arrays and hands them to the table-driven kernels behind mxv_tab_* (gym_amd/csrc/mxv_tab.hip).
`HipTabularVectorEnv` mirrors what gym.vector.SyncVectorEnv returns for these ids — int64 observations, float64 rewards,
`infos["prob"]` (and Taxi's `action_mask`), `final_observation` as an int64 array — including the dtype quirk of
VectorEnv._add_info (vector_env.py:208-258: an info array takes the type of the FIRST value stored in it).
`TabularRollout` is the device-resident front-end ([K, N] trajectory tensors from one fused launch).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Union

import numpy as np

from . import _native, error
from .spaces import Discrete
from .vector_env import LazyInfos, VectorEnv, _Pending

__all__ = ["TabularMDP", "generate_random_map", "frozen_lake_mdp", "taxi_mdp", "cliff_walking_mdp", "HipTabularVectorEnv", "TabularRollout",
           "HipBlackjackVectorEnv", "BlackjackRollout", "TOY_TEXT_REGISTRY", "taxi_encode", "taxi_decode"]


@dataclass
class TabularMDP:
    """Dense form of the reference's P dict: arrays [S][A][M] (M = longest transition list, padding: cum_prob = -1)."""
    num_states: int
    num_actions: int
    cum_prob: np.ndarray      # float64: np.cumsum of each list's probabilities (what categorical_sample compares with)
    prob: np.ndarray          # float64
    next_state: np.ndarray    # int32
    reward: np.ndarray        # float64
    terminated: np.ndarray    # uint8
    initial_distrib: np.ndarray  # float64 [S] (initial_state_distrib)
    reset_prob_is_int: bool   # reset() returns {"prob": 1} (int; FrozenLake, CliffWalking) or {"prob": 1.0} (Taxi)
    action_mask: Optional[np.ndarray] = None  # int8 [S][A] (Taxi's info["action_mask"]) or None

    @property
    def max_transitions(self) -> int:
        return self.cum_prob.shape[2]

    @property
    def initial_cum(self) -> np.ndarray:
        return np.cumsum(self.initial_distrib)

    def transitions(self, s: int, a: int):
        """The reference's P[s][a] list of (prob, next_state, reward, terminated)."""
        k = int((self.cum_prob[s, a] >= 0).sum())
        return [(float(self.prob[s, a, i]), int(self.next_state[s, a, i]), float(self.reward[s, a, i]),
                 bool(self.terminated[s, a, i])) for i in range(k)]


def _densify(P: List[List[list]], initial, reset_prob_is_int, action_mask=None) -> TabularMDP:
    S, A = len(P), len(P[0])
    M = max(len(P[s][a]) for s in range(S) for a in range(A))
    cum = np.full((S, A, M), -1.0)
    prob = np.zeros((S, A, M))
    nxt = np.zeros((S, A, M), np.int32)
    rew = np.zeros((S, A, M))
    term = np.zeros((S, A, M), np.uint8)
    for s in range(S):
        for a in range(A):
            tr = P[s][a]
            k = len(tr)
            cum[s, a, :k] = np.cumsum(np.asarray([t[0] for t in tr]))   # utils.py:6-7
            prob[s, a, :k] = [t[0] for t in tr]
            nxt[s, a, :k] = [t[1] for t in tr]
            rew[s, a, :k] = [t[2] for t in tr]
            term[s, a, :k] = [t[3] for t in tr]
    return TabularMDP(S, A, cum, prob, nxt, rew, term, np.asarray(initial, np.float64), reset_prob_is_int, action_mask)


# ---- FrozenLake (gym/envs/toy_text/frozen_lake.py:17-29,160-224) -------------------------------------------------------
FROZEN_LAKE_MAPS = {
    "4x4": ["SFFF", "FHFH", "FFFH", "HFFG"],
    "8x8": ["SFFFFFFF", "FFFFFFFF", "FFFHFFFF", "FFFFFHFF", "FFFHFFFF", "FHHFFFHF", "FHFFHFHF", "FFFHFFFG"],
}


def _has_path(board, size: int) -> bool:
    """Depth-first search from (0, 0): can G be reached without stepping on H? (frozen_lake.py:33-49)"""
    frontier, seen = [(0, 0)], set()
    while frontier:
        r, c = frontier.pop()
        if (r, c) in seen:
            continue
        seen.add((r, c))
        for dr, dc in ((1, 0), (0, 1), (-1, 0), (0, -1)):
            nr, nc = r + dr, c + dc
            if 0 <= nr < size and 0 <= nc < size:
                if board[nr][nc] == "G":
                    return True
                if board[nr][nc] != "H":
                    frontier.append((nr, nc))
    return False


def generate_random_map(size: int = 8, p: float = 0.8) -> List[str]:
    """frozen_lake.py:52-72: i.i.d. tiles (frozen with probability p) from NumPy's GLOBAL generator, S and G pinned to the
    corners, redrawn until a path exists."""
    while True:
        p = min(1, p)
        board = np.random.choice(["F", "H"], (size, size), p=[p, 1 - p])
        board[0][0] = "S"
        board[-1][-1] = "G"
        if _has_path(board, size):
            return ["".join(row) for row in board]


def frozen_lake_mdp(desc: Optional[Sequence[str]] = None, map_name: Optional[str] = "4x4", is_slippery: bool = True) -> TabularMDP:
    """Actions LEFT=0, DOWN=1, RIGHT=2, UP=3 (:11-14).  On ice the agent moves in the chosen direction or, when slippery,
    in one of the two perpendicular ones, each with probability 1/3 in the order (a-1)%4, a, (a+1)%4 (:213-221); moves are
    clamped to the grid (:180-189); stepping on G pays 1.0, G and H end the episode (:191-197) and are absorbing with a
    single (1.0, s, 0, True) transition (:208-209)."""
    if desc is None:
        desc = generate_random_map() if map_name is None else FROZEN_LAKE_MAPS[map_name]   # :168-171
    grid = [[(c.decode() if isinstance(c, bytes) else str(c)) for c in row] for row in desc]
    nrow, ncol = len(grid), len(grid[0])
    moves = {0: (0, -1), 1: (1, 0), 2: (0, 1), 3: (-1, 0)}

    def go(r, c, b):
        dr, dc = moves[b]
        return min(max(r + dr, 0), nrow - 1), min(max(c + dc, 0), ncol - 1)

    P = []
    for r in range(nrow):
        for c in range(ncol):
            s = r * ncol + c
            per_action = []
            for a in range(4):
                if grid[r][c] in "GH":
                    per_action.append([(1.0, s, 0, True)])
                    continue
                outcomes = []
                for b in ([(a - 1) % 4, a, (a + 1) % 4] if is_slippery else [a]):
                    nr, nc = go(r, c, b)
                    letter = grid[nr][nc]
                    outcomes.append((1.0 / 3.0 if is_slippery else 1.0, nr * ncol + nc, float(letter == "G"), letter in "GH"))
                per_action.append(outcomes)
            P.append(per_action)
    start = np.array([[ch == "S" for ch in row] for row in grid], dtype=np.float64).ravel()
    start /= start.sum()                                                                 # :169-170
    return _densify(P, start, reset_prob_is_int=True)


# ---- Taxi (gym/envs/toy_text/taxi.py:14-24,127-252) ---------------------------------------------------------------------
_TAXI_LOCS = [(0, 0), (0, 4), (4, 0), (4, 3)]   # R, G, Y, B
# walls of the 5x5 grid (the '|' characters of taxi.py's MAP): moving east from (row, col) is blocked
_TAXI_EAST_WALLS = {(0, 1), (1, 1), (3, 0), (3, 2), (4, 0), (4, 2)}


def _taxi_encode(row, col, pass_loc, dest):
    return ((row * 5 + col) * 5 + pass_loc) * 4 + dest   # taxi.py:208-217


def taxi_encode(taxi_row, taxi_col, pass_loc, dest_idx) -> int:
    """TaxiEnv.encode (taxi.py:210-219)."""
    return _taxi_encode(taxi_row, taxi_col, pass_loc, dest_idx)


def taxi_decode(i: int):
    """TaxiEnv.decode (taxi.py:221-231): (taxi_row, taxi_col, pass_loc, dest_idx); like the reference, an iterator."""
    out = [i % 4]
    i //= 4
    out.append(i % 5)
    i //= 5
    out.append(i % 5)
    i //= 5
    out.append(i)
    assert 0 <= i < 5
    return reversed(out)


def taxi_mdp() -> TabularMDP:
    """500 states (taxi row, col, passenger location 0-3 or 4 = in taxi, destination), 6 actions: south, north, east, west,
    pickup, dropoff (:127-193).  Every step costs -1; an illegal pickup/dropoff -10; delivering pays 20 and ends the episode.
    Episodes start uniformly in the 300 states whose passenger is waiting somewhere other than the destination (:148-149)."""
    P = [None] * 500
    initial = np.zeros(500)
    mask = np.zeros((500, 6), np.int8)
    for row in range(5):
        for col in range(5):
            can_east = col < 4 and (row, col) not in _TAXI_EAST_WALLS
            can_west = col > 0 and (row, col - 1) not in _TAXI_EAST_WALLS
            for pas in range(5):
                for dest in range(4):
                    s = _taxi_encode(row, col, pas, dest)
                    if pas < 4 and pas != dest:
                        initial[s] += 1
                    here = (row, col)
                    per_action = []
                    for a in range(6):
                        nrow, ncol, npas, reward, term = row, col, pas, -1, False
                        if a == 0:
                            nrow = min(row + 1, 4)
                        elif a == 1:
                            nrow = max(row - 1, 0)
                        elif a == 2:
                            ncol = col + 1 if can_east else col
                        elif a == 3:
                            ncol = col - 1 if can_west else col
                        elif a == 4:
                            if pas < 4 and here == _TAXI_LOCS[pas]:
                                npas = 4
                            else:
                                reward = -10
                        else:
                            if pas == 4 and here == _TAXI_LOCS[dest]:
                                npas, reward, term = dest, 20, True
                            elif pas == 4 and here in _TAXI_LOCS:
                                npas = _TAXI_LOCS.index(here)
                            else:
                                reward = -10
                        per_action.append([(1.0, _taxi_encode(nrow, ncol, npas, dest), reward, term)])
                    P[s] = per_action
                    # action_mask (:231-252)
                    mask[s] = [row < 4, row > 0, can_east, can_west, pas < 4 and here == _TAXI_LOCS[pas],
                               pas == 4 and here in _TAXI_LOCS]
    initial /= initial.sum()
    return _densify(P, initial, reset_prob_is_int=False, action_mask=mask)


# ---- CliffWalking (gym/envs/toy_text/cliffwalking.py:69-146) ---------------------------------------------------------------
def cliff_walking_mdp() -> TabularMDP:
    """4x12 grid, actions UP=0, RIGHT=1, DOWN=2, LEFT=3 (:11-14); start (3,0), goal (3,11), the cliff is row 3 between
    them: stepping on it costs -100 and teleports to the start (no termination), every other move costs -1 (:118-138)."""
    nrow, ncol = 4, 12
    deltas = [(-1, 0), (0, 1), (1, 0), (0, -1)]
    start = 3 * ncol
    P = []
    for s in range(nrow * ncol):
        r, c = divmod(s, ncol)
        per_action = []
        for dr, dc in deltas:
            nr, nc = min(max(r + dr, 0), nrow - 1), min(max(c + dc, 0), ncol - 1)
            if nr == 3 and 1 <= nc <= ncol - 2:
                per_action.append([(1.0, start, -100, False)])
            else:
                per_action.append([(1.0, nr * ncol + nc, -1, (nr, nc) == (nrow - 1, ncol - 1))])
        P.append(per_action)
    initial = np.zeros(nrow * ncol)
    initial[start] = 1.0
    return _densify(P, initial, reset_prob_is_int=True)


@dataclass
class ToyTextSpec:
    id: str
    build: callable
    max_episode_steps: Optional[int]
    reward_threshold: Optional[float] = None
    kwargs: tuple = ()


# gym/envs/__init__.py:101-127
TOY_TEXT_REGISTRY: Dict[str, ToyTextSpec] = {
    "FrozenLake-v1": ToyTextSpec("FrozenLake-v1", lambda **kw: frozen_lake_mdp(**{"map_name": "4x4", **kw}), 100, 0.70,
                                 ("desc", "map_name", "is_slippery")),
    "FrozenLake8x8-v1": ToyTextSpec("FrozenLake8x8-v1", lambda **kw: frozen_lake_mdp(**{"map_name": "8x8", **kw}), 200, 0.85,
                                    ("desc", "map_name", "is_slippery")),
    "CliffWalking-v0": ToyTextSpec("CliffWalking-v0", lambda **kw: cliff_walking_mdp(), None),
    "Taxi-v3": ToyTextSpec("Taxi-v3", lambda **kw: taxi_mdp(), 200, 8.0),
}


def _spec(id: str) -> ToyTextSpec:
    try:
        return TOY_TEXT_REGISTRY[id]
    except KeyError:
        raise error.UnregisteredEnv(f"No HIP tabular engine for id {id!r}; supported: {sorted(TOY_TEXT_REGISTRY)}") from None


def _make_handle(mdp: TabularMDP, num_envs: int, limit: Optional[int], device: int, env_offset: int, seed: int, action_seed: int,
                 compact: bool = False, general_kernel: bool = False):
    return _native.Tab(mdp.num_states, mdp.num_actions, mdp.cum_prob, mdp.prob, mdp.next_state, mdp.reward, mdp.terminated,
                       mdp.initial_cum, num_envs, -1 if limit is None else int(limit), device=device,
                       env_offset=env_offset, seed=seed, action_seed=action_seed, compact=compact, general_kernel=general_kernel)


class HipTabularVectorEnv(VectorEnv):
    """`num_envs` copies of one tabular toy_text env on one MI355X, with SyncVectorEnv's call surface and return contract."""

    metadata = {"render_modes": []}
    render_mode = None

    def __init__(self, id: str, num_envs: int = 1, *, device: int = 0, max_episode_steps: Optional[int] = None,
                 env_offset: int = 0, **kwargs):
        # `spec` is public and gets replaced: gym.make("hip/<id>") runs `env.unwrapped.spec = <gym EnvSpec "hip/<id>">`
        # (gym/envs/registration.py:657).  The engine's own registry entry therefore lives in a private attribute and is what
        # call() / pickling use.
        self.spec = self._tt_spec = _spec(id)
        if kwargs.pop("render_mode", None) is not None:
            raise TypeError(f"{id}: the device engine does not render (render_mode must be None)")
        for k in kwargs:
            if k not in self.spec.kwargs:
                raise TypeError(f"{id} got an unexpected keyword argument {k!r}")
        self.mdp = self.spec.build(**kwargs)
        super().__init__(num_envs, Discrete(self.mdp.num_states), Discrete(self.mdp.num_actions))
        limit = self.spec.max_episode_steps if max_episode_steps is None else max_episode_steps
        entropy = int.from_bytes(os.urandom(8), "little")
        self._handle = _make_handle(self.mdp, num_envs, limit, device, env_offset, entropy, entropy ^ 0x9E3779B97F4A7C15)
        self._limit, self._env_offset = limit, env_offset
        self._actions = None
        self._was_reset = False

    # -- pickling (the reference's checkpoint: tests/envs/test_envs.py:192-200) ----------------------------------------
    def __getstate__(self):
        self._assert_is_running()
        d = {k: v for k, v in self.__dict__.items() if k not in ("_handle", "spec", "_tt_spec")}
        d["_spec_id"] = self._tt_spec.id           # the registry entry holds the (unpicklable) builder
        if self.spec is not self._tt_spec:
            d["_outer_spec"] = self.spec           # e.g. gym's EnvSpec after gym.make("hip/<id>")
        d["_snapshot"] = self._handle.snapshot()
        d["_ctor"] = (self._handle.device, self._limit, self._env_offset)
        return d

    def __setstate__(self, d):
        d = dict(d)
        snap, (device, limit, env_offset) = d.pop("_snapshot"), d.pop("_ctor")
        self._tt_spec = _spec(d.pop("_spec_id"))
        self.spec = d.pop("_outer_spec", self._tt_spec)
        self.__dict__.update(d)
        self._handle = _make_handle(self.mdp, self.num_envs, limit, device, env_offset, snap["base_seed"], snap["action_seed"])
        self._handle.restore(snap)

    # -- infos ---------------------------------------------------------------------------------------------------
    def _mask_info(self, infos, states):
        if self.mdp.action_mask is None:
            return
        table, n = self.mdp.action_mask, self.num_envs

        def build():
            arr = np.full(n, None, dtype=object)   # type(np.ndarray) is not numeric -> object array (vector_env.py:248-253)
            for i in range(n):
                arr[i] = table[states[i]].copy()
            return arr

        dict.__setitem__(infos, "action_mask", _Pending(build))
        dict.__setitem__(infos, "_action_mask", np.ones(n, dtype=bool))

    def reset_wait(self, seed: Optional[Union[int, List[int]]] = None, options: Optional[dict] = None):
        self._assert_is_running()
        if seed is not None:
            if isinstance(seed, (int, np.integer)):
                if seed < 0:
                    raise error.Error(f"Seed must be a non-negative integer or omitted, not {seed}")
                self._handle.seed(int(seed), None)
            else:
                seeds = list(seed)
                assert len(seeds) == self.num_envs
                for s in seeds:
                    if not (isinstance(s, (int, np.integer)) and s >= 0):
                        raise error.Error(f"Seed must be a non-negative integer or omitted, not {s}")
                self._handle.seed(0, np.array(seeds, dtype=np.uint64))
        obs = self._handle.reset_host()
        self._was_reset = True
        self._actions = None
        infos = LazyInfos()
        one = np.int64 if self.mdp.reset_prob_is_int else np.float64
        dict.__setitem__(infos, "prob", np.ones(self.num_envs, dtype=one))
        dict.__setitem__(infos, "_prob", np.ones(self.num_envs, dtype=bool))
        self._mask_info(infos, obs)
        return obs, infos

    # -- step ------------------------------------------------------------------------------------------------------
    def step_async(self, actions):
        self._assert_is_running()
        if self._actions is not None:
            raise error.AlreadyPendingCallError("Calling `step_async` while waiting for a pending call to `step` to "
                                                "complete.", "step")
        a = np.asarray(actions)
        if a.shape != (self.num_envs,) or not np.issubdtype(a.dtype, np.integer):
            raise KeyError(f"{actions!r} is not a batch of {self.num_envs} integer actions")
        self._actions = np.ascontiguousarray(a, dtype=np.int64)

    def step_wait(self):
        self._assert_is_running()
        if self._actions is None:
            raise error.NoAsyncCallError("Calling `step_wait` without any prior call to `step_async`.", "step")
        actions, self._actions = self._actions, None
        if not self._was_reset:
            raise error.ResetNeeded("Cannot call env.step() before calling env.reset()")
        try:
            obs, rew, term, trunc, prob, fin, fprob = self._handle.step_host(actions, pooled=True)
        except _native.MxvError as e:
            if e.code == _native.ERR_INVALID_ACTION:
                bad = actions[(actions < 0) | (actions >= self.mdp.num_actions)]
                raise KeyError(int(bad[0]) if bad.size else actions) from None   # the reference: P[s][a] -> KeyError
            raise
        n = self.num_envs
        done = term | trunc
        infos = LazyInfos()
        # VectorEnv._add_info: the "prob" array takes the type of the first value stored, i.e. sub-env 0's: the int 1 of
        # FrozenLake/CliffWalking's reset() when sub-env 0 just finished (float probabilities then truncate), else float
        if done[0] and self.mdp.reset_prob_is_int:
            prob = prob.astype(np.int64)
        dict.__setitem__(infos, "prob", prob)
        dict.__setitem__(infos, "_prob", np.ones(n, dtype=bool))
        self._mask_info(infos, obs)
        if done.any():
            idx = np.flatnonzero(done)
            mask_table = self.mdp.action_mask
            dict.__setitem__(infos, "final_observation", np.where(done, fin, 0).astype(np.int64))  # python ints -> int array
            dict.__setitem__(infos, "_final_observation", done.copy())

            def build_final_info():
                arr = np.full(n, None, dtype=object)
                for i in idx:
                    d = {"prob": float(fprob[i])}
                    if mask_table is not None:
                        d["action_mask"] = mask_table[fin[i]].copy()
                    arr[i] = d
                return arr

            dict.__setitem__(infos, "final_info", _Pending(build_final_info))
            dict.__setitem__(infos, "_final_info", done.copy())
        return obs, rew, term, trunc, infos

    # -- attribute access of the sub-envs (P, desc-free: the MDP tables) ---------------------------------------------
    def call(self, name: str, *args, **kwargs) -> tuple:
        self._assert_is_running()
        if name == "s":
            return tuple(int(v) for v in self._handle.get_state()[0])
        if name == "_max_episode_steps":               # TimeLimit's attributes (time_limit.py:43-44)
            return (self._limit if self._limit and self._limit > 0 else None,) * self.num_envs
        if name == "_elapsed_steps":
            return tuple(int(v) for v in self._handle.get_state()[1])
        if name == "P":
            P = {s: {a: self.mdp.transitions(s, a) for a in range(self.mdp.num_actions)} for s in range(self.mdp.num_states)}
            return (P,) * self.num_envs
        if name == "initial_state_distrib":
            return (self.mdp.initial_distrib.copy(),) * self.num_envs
        if self._tt_spec.id == "Taxi-v3" and name in ("encode", "decode", "action_mask"):  # TaxiEnv.s pure helpers (taxi.py:210-252)
            if name == "encode":
                return (taxi_encode(*args, **kwargs),) * self.num_envs
            if name == "decode":
                return tuple(taxi_decode(*args, **kwargs) for _ in range(self.num_envs))
            (state,) = args or (kwargs["state"],)
            return tuple(self.mdp.action_mask[int(state)].copy() for _ in range(self.num_envs))
        raise AttributeError(f"{self._tt_spec.id} sub-environments have no attribute {name!r}")

    def close_extras(self, **kwargs):
        h = getattr(self, "_handle", None)
        if h is not None:
            h.close()

    def _assert_is_running(self):
        if self.closed:
            raise error.ClosedEnvironmentError(f"Trying to operate on `{type(self).__name__}`, after a call to `close()`.")

    @property
    def unwrapped(self):
        return self

    @property
    def handle(self) -> "_native.Tab":
        return self._handle


class TabularRollout:
    """Device-resident front-end: K sampled steps per launch into [K, N] torch tensors (obs / actions int64, reward / prob
    float64, terminated / truncated uint8), state resident on the device between calls.  compact=True: the trajectory tensors hold the
    contract dtypes of SURVEY.md §8(d) — int32 obs / actions, float32 reward / prob: 18 instead of 34 bytes per env-step, same values."""

    def __init__(self, id: str, num_envs: int, *, device: int = 0, env_offset: int = 0, seed: int = 0, action_seed: int = 0,
                 max_episode_steps: Optional[int] = None, compact: bool = False, general_kernel: bool = False, **kwargs):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("TabularRollout needs a HIP device; gym_amd has no CPU fallback")
        self._torch = torch
        self.spec = _spec(id)
        self.mdp = self.spec.build(**kwargs)
        self.num_envs = int(num_envs)
        self.device = torch.device("cuda", device)
        limit = self.spec.max_episode_steps if max_episode_steps is None else max_episode_steps
        self.compact = bool(compact)
        self.int_dtype, self.real_dtype = (torch.int32, torch.float32) if compact else (torch.int64, torch.float64)
        self.handle = _make_handle(self.mdp, num_envs, limit, device, env_offset, seed, action_seed, compact=compact,
                                   general_kernel=general_kernel)
        self.stream = torch.cuda.Stream(device=self.device)
        self.handle.set_stream(self.stream.cuda_stream)
        with torch.cuda.stream(self.stream):
            self.obs = torch.zeros(self.num_envs, dtype=torch.int64, device=self.device)
        self.stream.synchronize()

    def reset(self, seed: Optional[int] = None):
        if seed is not None:
            self.handle.seed(seed)
        self.handle.reset(self.obs)
        self.ready()        # the returned tensor is safe to read on the caller's current stream (GPU-side ordering, no host wait)
        return self.obs

    def trajectory_buffers(self, K: int, layout: str = "auto"):
        """[K, N] output tensors of rollout_per_step.  Sets of 2 GiB and more ("auto") are sorted by HBM class (gym_amd/placement.py):
        the launch writes four 8-byte streams, and it runs 5.7 / 6.1 / 7.1 us per 2^20-env step with them split 2 + 2 / 1 + 3 / 4 + 0
        over two classes (profiles/r3/r3g_tab_class_ab.jsonl) — obs + reward on one, actions + prob on another.  layout="separate":
        ordinary allocations.  The report is left in self.last_placement."""
        t, n, dev = self._torch, self.num_envs, self.device
        it, rt = self.int_dtype, self.real_dtype
        specs = [("obs", (K, n), it, False), ("reward", (K, n), rt, False), ("actions", (K, n), it, False),
                 ("prob", (K, n), rt, False), ("terminated", (K, n), t.uint8, False), ("truncated", (K, n), t.uint8, False)]
        if layout == "auto":
            from . import placement

            stored = 18 if self.compact else 34
            layout = "sorted" if stored * K * n >= (2 << 30) and placement.enabled() else "separate"     # MXV_PLACEMENT=off: never sort
        if layout == "sorted":
            from .placement import sorted_tensors

            out, self.last_placement = sorted_tensors(specs, {"obs": 0, "reward": 0, "actions": 1, "prob": 1}, dev, self.stream)
            return out
        if layout != "separate":
            raise ValueError(f"layout must be 'auto', 'sorted' or 'separate', got {layout!r}")
        with t.cuda.stream(self.stream):
            return {name: t.empty(shape, dtype=dt, device=dev) for name, shape, dt, _ in specs}

    def rollout_per_step(self, K: int, out: Optional[dict] = None):
        out = self.trajectory_buffers(K) if out is None else out
        self.handle.rollout(K, out["obs"], out["reward"], out["terminated"], out["truncated"], out["prob"],
                            actions_out_dev=out["actions"], per_step=True)
        return out

    def rollout_tape(self, actions, out: Optional[dict] = None):
        K = actions.shape[0]
        assert actions.is_cuda and actions.is_contiguous() and actions.dtype == self.int_dtype
        out = self.trajectory_buffers(K) if out is None else out
        self.stream.wait_stream(self._torch.cuda.current_stream(self.device))   # the tape was produced on the caller's stream
        self.handle.rollout_tape(K, actions, out["obs"], out["reward"], out["terminated"], out["truncated"], out["prob"],
                                 per_step=True)
        out["actions"] = actions
        return out

    def ready(self):
        """The caller's current torch stream waits (on the GPU) for everything launched on the engine's stream."""
        self._torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def synchronize(self):
        self.handle.sync()

    def close(self):
        self.handle.close()


class BlackjackRollout:
    """Device-resident front-end of the Blackjack engine (mxv_bj_*): K sampled steps per launch into [K, ...] torch tensors — obs
    [K, 3, N] (player total, dealer's first card, usable ace), actions [K, N], reward [K, N], terminated / truncated uint8 [K, N] — with
    the hands resident on the device between calls.  compact=True: int32 observations / actions and float32 rewards (the contract dtypes
    of SURVEY.md §8d: 22 instead of 42 bytes per env-step, same values)."""

    def __init__(self, num_envs: int, *, device: int = 0, env_offset: int = 0, seed: int = 0, action_seed: int = 0, natural: bool = False,
                 sab: bool = True, max_episode_steps: Optional[int] = None, compact: bool = False):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("BlackjackRollout needs a HIP device; gym_amd has no CPU fallback")
        self._torch = torch
        self.num_envs = int(num_envs)
        self.device = torch.device("cuda", device)
        self.compact = bool(compact)
        self.int_dtype, self.real_dtype = (torch.int32, torch.float32) if compact else (torch.int64, torch.float64)
        self.handle = _native.Blackjack(self.num_envs, natural=natural, sab=sab, device=device, env_offset=env_offset, seed=seed,
                                        action_seed=action_seed, max_episode_steps=-1 if max_episode_steps is None else max_episode_steps)
        self.stream = torch.cuda.Stream(device=self.device)
        self.handle.set_stream(self.stream.cuda_stream)
        with torch.cuda.stream(self.stream):
            self.obs = torch.zeros((3, self.num_envs), dtype=torch.int64, device=self.device)
        self.stream.synchronize()
        self.last_placement = None

    def reset(self, seed: Optional[int] = None):
        if seed is not None:
            self.handle.seed(seed, action_seed=self.handle._action_seed)
        self.handle.reset(self.obs)     # (mxv_bj_reset writes int64: the reference's observation dtype)
        self.ready()        # the returned tensor is safe to read on the caller's current stream (GPU-side ordering, no host wait)
        return self.obs.to(self.int_dtype) if self.compact else self.obs      # compact: the trajectories' dtype (int32), like every other tensor it hands out

    def trajectory_buffers(self, K: int, layout: str = "auto", want_final: bool = False):
        """Output tensors of rollout_per_step.  Sets of 2 GiB and more ("auto") are sorted by HBM class (gym_amd/placement.py): the launch
        writes five 8-byte (4-byte) streams — the three observation columns on one class, rewards + actions on another.  The report is left
        in self.last_placement."""
        t, n, dev = self._torch, self.num_envs, self.device
        it, rt = self.int_dtype, self.real_dtype
        specs = [("obs", (K, 3, n), it, False), ("reward", (K, n), rt, False), ("actions", (K, n), it, False),
                 ("terminated", (K, n), t.uint8, False), ("truncated", (K, n), t.uint8, False)]
        if want_final:
            specs.append(("final_obs", (K, 3, n), it, True))
        if layout == "auto":
            from . import placement

            layout = "sorted" if (22 if self.compact else 42) * K * n >= (2 << 30) and placement.enabled() else "separate"
        if layout == "sorted":
            from .placement import sorted_tensors

            out, self.last_placement = sorted_tensors(specs, {"obs": 0, "reward": 1, "actions": 1}, dev, self.stream)
            return out
        if layout != "separate":
            raise ValueError(f"layout must be 'auto', 'sorted' or 'separate', got {layout!r}")
        with t.cuda.stream(self.stream):
            return {name: (t.zeros if zero else t.empty)(shape, dtype=dt, device=dev) for name, shape, dt, zero in specs}

    def rollout_per_step(self, K: int, out: Optional[dict] = None):
        out = self.trajectory_buffers(K) if out is None else out
        self.handle.rollout(K, out["obs"], out["reward"], out["terminated"], out["truncated"], out.get("final_obs"),
                            actions_out_dev=out["actions"], per_step=True, compact=self.compact)
        return out

    def rollout_tape(self, actions, out: Optional[dict] = None):
        """K = actions.shape[0] steps with the caller's actions (int64 [K, N] on the device)."""
        K = actions.shape[0]
        assert actions.is_cuda and actions.is_contiguous() and actions.dtype == self._torch.int64
        out = self.trajectory_buffers(K) if out is None else out
        self.stream.wait_stream(self._torch.cuda.current_stream(self.device))   # the tape was produced on the caller's stream
        self.handle.rollout(K, out["obs"], out["reward"], out["terminated"], out["truncated"], out.get("final_obs"),
                            actions_tape_dev=actions, per_step=True, compact=self.compact)
        with self._torch.cuda.stream(self.stream):
            out["actions"][:K].copy_(actions)      # a tape-driven launch records no actions: the returned set still holds the ones that were played
        return out

    def ready(self):
        """The caller's current torch stream waits (on the GPU) for everything launched on the engine's stream."""
        self._torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def synchronize(self):
        self.handle.sync()

    def close(self):
        self.handle.close()


# ---- Blackjack-v1 (gym/envs/toy_text/blackjack.py): not a P table — its own kernel (gym_amd/csrc/mxv_bj.hip) ---------------
class HipBlackjackVectorEnv(VectorEnv):
    """`num_envs` Blackjack tables on one MI355X with SyncVectorEnv's contract: observations are a tuple of three int64
    arrays (player total, dealer's showing card, usable ace) — batch_space(Tuple(Discrete(32), Discrete(11), Discrete(2)));
    rewards float64; infos empty except `final_observation` (object array of (int, int, bool) tuples) / `final_info`."""

    metadata = {"render_modes": []}
    render_mode = None

    def __init__(self, id: str = "Blackjack-v1", num_envs: int = 1, *, device: int = 0, natural: bool = False, sab: bool = True,
                 max_episode_steps: Optional[int] = None, env_offset: int = 0, **kwargs):
        from .spaces import Tuple

        if kwargs.pop("render_mode", None) is not None:
            raise TypeError(f"{id}: the device engine does not render (render_mode must be None)")
        if kwargs:
            raise TypeError(f"{id} got an unexpected keyword argument {next(iter(kwargs))!r}")
        self.spec = ToyTextSpec(id, None, None)
        self.natural, self.sab = bool(natural), bool(sab)     # gym/envs/__init__.py:95-99 registers sab=True, natural=False
        super().__init__(num_envs, Tuple((Discrete(32), Discrete(11), Discrete(2))), Discrete(2))
        entropy = int.from_bytes(os.urandom(8), "little")
        self._handle = _native.Blackjack(num_envs, natural=natural, sab=sab, device=device, env_offset=env_offset,
                                         max_episode_steps=-1 if max_episode_steps is None else int(max_episode_steps),
                                         seed=entropy, action_seed=entropy ^ 0x9E3779B97F4A7C15)
        self._ctor = (device, env_offset, max_episode_steps)
        self._actions = None
        self._was_reset = False

    def __getstate__(self):
        self._assert_is_running()
        d = {k: v for k, v in self.__dict__.items() if k != "_handle"}
        d["_snapshot"] = self._handle.snapshot()
        return d

    def __setstate__(self, d):
        d = dict(d)
        snap = d.pop("_snapshot")
        self.__dict__.update(d)
        device, env_offset, limit = self._ctor
        self._handle = _native.Blackjack(self.num_envs, natural=self.natural, sab=self.sab, device=device, env_offset=env_offset,
                                         max_episode_steps=-1 if limit is None else int(limit),
                                         seed=snap["base_seed"], action_seed=snap["action_seed"])
        self._handle.restore(snap)

    @staticmethod
    def _obs(cols):
        return (cols[0].copy(), cols[1].copy(), cols[2].copy())

    def reset_wait(self, seed: Optional[Union[int, List[int]]] = None, options: Optional[dict] = None):
        self._assert_is_running()
        if seed is not None:
            if isinstance(seed, (int, np.integer)):
                if seed < 0:
                    raise error.Error(f"Seed must be a non-negative integer or omitted, not {seed}")
                self._handle.seed(int(seed), None, int(seed) ^ 0x9E3779B97F4A7C15)
            else:
                seeds = list(seed)
                assert len(seeds) == self.num_envs
                self._handle.seed(0, np.array(seeds, dtype=np.uint64), int(seeds[0]) ^ 0x9E3779B97F4A7C15)
        cols = self._handle.reset_host()
        self._was_reset = True
        self._actions = None
        return self._obs(cols), {}

    def step_async(self, actions):
        self._assert_is_running()
        if self._actions is not None:
            raise error.AlreadyPendingCallError("Calling `step_async` while waiting for a pending call to `step` to "
                                                "complete.", "step")
        a = np.asarray(actions)
        if a.shape != (self.num_envs,) or not np.issubdtype(a.dtype, np.integer):
            raise AssertionError(f"{actions!r} ({type(actions)}) invalid")
        self._actions = np.ascontiguousarray(a, dtype=np.int64)

    def step_wait(self):
        self._assert_is_running()
        if self._actions is None:
            raise error.NoAsyncCallError("Calling `step_wait` without any prior call to `step_async`.", "step")
        actions, self._actions = self._actions, None
        if not self._was_reset:
            raise error.ResetNeeded("Cannot call env.step() before calling env.reset()")
        try:
            cols, rew, term, trunc, fin = self._handle.step_host(actions, pooled=True)
        except _native.MxvError as e:
            if e.code == _native.ERR_INVALID_ACTION:
                raise AssertionError(f"{actions!r} ({type(actions)}) invalid") from None   # blackjack.py:122
            raise
        n = self.num_envs
        done = term | trunc
        infos = LazyInfos()
        if done.any():
            idx = np.flatnonzero(done)

            def build_final_obs():
                arr = np.full(n, None, dtype=object)
                for i in idx:
                    arr[i] = (int(fin[0, i]), int(fin[1, i]), bool(fin[2, i]))   # _get_obs(): (int, int, bool)
                return arr

            def build_final_info():
                arr = np.full(n, None, dtype=object)
                for i in idx:
                    arr[i] = {}
                return arr

            dict.__setitem__(infos, "final_observation", _Pending(build_final_obs))
            dict.__setitem__(infos, "_final_observation", done.copy())
            dict.__setitem__(infos, "final_info", _Pending(build_final_info))
            dict.__setitem__(infos, "_final_info", done.copy())
        return self._obs(cols), rew, term, trunc, infos

    def close_extras(self, **kwargs):
        h = getattr(self, "_handle", None)
        if h is not None:
            h.close()

    def _assert_is_running(self):
        if self.closed:
            raise error.ClosedEnvironmentError(f"Trying to operate on `{type(self).__name__}`, after a call to `close()`.")

    @property
    def unwrapped(self):
        return self

    @property
    def handle(self) -> "_native.Blackjack":
        return self._handle
