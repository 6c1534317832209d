"""graphenvs_amd -- MI355X-native batched graph-RL environments (hot path of teshnizi/GraphEnvs).

    import graphenvs_amd as ge
    env = ge.make_vec("ShortestPath-v0", num_envs=65536, n_nodes=64, n_edges=192)
    obs, info = env.reset(seed=0)
    obs, reward, terminated, truncated, info = env.step(actions)   # actions: int64 [B] on the GPU

``make(id, **kwargs)`` is the single-env, numpy-returning facade with the reference's surface.
"""
from . import utils  # noqa: F401
from .envs import GraphEnv, make  # noqa: F401
from .ragged import MixedVectorEnv, RaggedVectorEnv  # noqa: F401
from .sharded import ShardedVectorEnv  # noqa: F401
from .vector_env import ENV_IDS, GraphBatch, VectorGraphEnv, make_vec  # noqa: F401

name = "graphenvs_amd"
__all__ = ["make", "make_vec", "GraphEnv", "VectorGraphEnv", "GraphBatch", "ENV_IDS", "utils",
           "RaggedVectorEnv", "MixedVectorEnv", "ShardedVectorEnv", "register_with_gymnasium"]


def register_with_gymnasium():
    """Register the nine ids of the reference with gymnasium when it is installed (graph_envs/__init__.py:9-56): gymnasium.make(id,
    **kwargs) then returns the single-env facade (GraphEnv), whose reset/step return what the reference's classes return."""
    from gymnasium.envs.registration import register
    for env_id in ENV_IDS:
        register(id=env_id, entry_point=lambda _id=env_id, **kw: make(_id, **kw))
