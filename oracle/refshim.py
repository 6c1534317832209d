"""TEST INFRASTRUCTURE ONLY -- loader for the read-only reference (this container only).

Imports `/root/reference/graph_envs` by registering two in-memory stand-ins for packages
that are absent from the image:

* ``gymnasium``: exactly the names the reference touches -- ``Env`` with a no-op
  ``reset(seed=, options=)`` (reference: shortest_path.py:14,50), ``spaces.Discrete`` (:40),
  ``spaces.Box`` (:42), ``spaces.GraphInstance`` (:86), ``envs.registration.register``
  (__init__.py:3) and ``make``.  The no-op ``Env.reset`` is faithful because real gymnasium
  only seeds its private ``self._np_random`` there, which no reference env ever reads.
* ``torch_geometric``: empty module (only dereferenced inside utils.to_pyg_graph).

Nothing from the reference is copied; nothing is written next to it
(``sys.dont_write_bytecode``).  The reference never travels to the GPU box, so everything
that uses this module must skip when ``/root/reference`` is absent.
"""
import importlib
import os
import sys
import types
from collections import namedtuple

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "graph_envs"))


def _install_standins():
    if "gymnasium" in sys.modules and not getattr(sys.modules["gymnasium"], "_ge_standin", False):
        return  # a real gymnasium is present: use it
    gym = types.ModuleType("gymnasium")
    gym._ge_standin = True

    class Env:
        def reset(self, seed=None, options=None):
            return None

    class Discrete:
        def __init__(self, n):
            self.n = int(n)

    class Box:
        def __init__(self, low, high, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Discrete = Discrete
    spaces.Box = Box
    spaces.GraphInstance = namedtuple("GraphInstance", ["nodes", "edges", "edge_links"])

    registry = {}

    def register(id, entry_point, **kw):
        registry[id] = entry_point

    def make(id, **kwargs):
        mod, cls = registry[id].split(":")
        return getattr(importlib.import_module(mod), cls)(**kwargs)

    envs = types.ModuleType("gymnasium.envs")
    registration = types.ModuleType("gymnasium.envs.registration")
    registration.register = register
    registration.registry = registry
    envs.registration = registration
    gym.Env, gym.spaces, gym.envs, gym.make, gym.register = Env, spaces, envs, make, register
    sys.modules["gymnasium"] = gym
    sys.modules["gymnasium.spaces"] = spaces
    sys.modules["gymnasium.envs"] = envs
    sys.modules["gymnasium.envs.registration"] = registration
    if "torch_geometric" not in sys.modules:
        try:
            importlib.import_module("torch_geometric")
        except Exception:
            sys.modules["torch_geometric"] = types.ModuleType("torch_geometric")


def load_reference():
    """Return (gymnasium-like module, graph_envs package) for the reference."""
    if not reference_available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.dont_write_bytecode = True
    _install_standins()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import matplotlib
    matplotlib.use("Agg")
    graph_envs = importlib.import_module("graph_envs")
    return sys.modules["gymnasium"], graph_envs
