"""Task registry with the reference's surface: `TaskSpec`, `TASKS`, `list_tasks`, `list_task_cards`, `get_task`, `make_env`
(/root/reference/backend/mlagents/registry.py:18-49,52-337,340-370).

The five north-star tasks (basic, gridworld, ball3d, push, ant/crawler) plus walljump get an `env_factory`
backed by the HIP engine; the
other registry ids are kept as catalogue entries so `list`/alias resolution behave the same, but they are not trainable
here (their dynamics are outside the hot-path scope).
"""
from __future__ import annotations

from collections.abc import Callable
from dataclasses import asdict, dataclass, field
from typing import Any, Literal

from . import envs

Interface = Literal["gymnasium", "pettingzoo", "mlagents-llapi", "external"]
ResearchTier = Literal["foundation", "benchmark", "frontier", "roadmap"]


@dataclass(frozen=True)
class TaskSpec:
    id: str
    title: str
    family: str
    interface: Interface
    research_tier: ResearchTier
    default_algorithm: str
    policy_prefix: str
    total_timesteps: int
    eval_episodes: int = 20
    n_envs: int = 1
    reward_threshold: float | None = None
    tags: tuple[str, ...] = ()
    observation: str = "vector"
    action: str = "discrete"
    publication_role: str = "supporting"
    status: str = "standardized"
    notes: str = ""
    env_factory: Callable[[], Any] | None = field(default=None, repr=False, compare=False)

    @property
    def trainable(self) -> bool:
        return self.interface == "gymnasium" and self.env_factory is not None

    def card(self) -> dict[str, Any]:
        data = asdict(self)
        data.pop("env_factory", None)
        data["trainable"] = self.trainable
        return data


_OUT_OF_SCOPE = "catalogue entry only: dynamics not implemented by the MI355X engine (outside the hot-path scope)"

TASKS: dict[str, TaskSpec] = {
    # ---- the five hot-path tasks (numbers as in registry.py:53-116,225-240) ----
    "basic": TaskSpec("basic", "Basic Move-To-Goal", "control", "gymnasium", "foundation", "dqn", "basic_policy", 25_000, eval_episodes=50,
                      n_envs=1, reward_threshold=0.85, tags=("sparse-reward", "tabular-state", "unity-ml-agents"),
                      publication_role="unit sanity check for action/observation plumbing", env_factory=envs.make_basic_env),
    "ball3d": TaskSpec("ball3d", "3D Ball Balance", "continuous-control", "gymnasium", "foundation", "ppo", "ball3d_policy", 150_000,
                       eval_episodes=30, n_envs=8, reward_threshold=150.0, tags=("physics", "stability", "unity-ml-agents"),
                       publication_role="browser/Unity parity smoke benchmark", env_factory=envs.make_ball3d_env),
    "gridworld": TaskSpec("gridworld", "GridWorld Goal-Conditioned Navigation", "navigation", "gymnasium", "foundation", "dqn", "gridworld_policy",
                          100_000, eval_episodes=100, n_envs=1, reward_threshold=0.75,
                          tags=("goal-conditioned", "procedural-layout", "discrete-control"),
                          publication_role="generalization and seed-control baseline", env_factory=envs.make_gridworld_env),
    "push": TaskSpec("push", "Push Block", "navigation", "gymnasium", "benchmark", "dqn", "push_policy", 200_000, eval_episodes=100, n_envs=1,
                     reward_threshold=0.65, tags=("object-manipulation", "sparse-reward", "planning"),
                     publication_role="single-agent manipulation transfer task", env_factory=envs.make_push_env),
    "walljump": TaskSpec("walljump", "Wall Jump", "navigation", "gymnasium", "benchmark", "dqn", "walljump_policy", 150_000, eval_episodes=100,
                         n_envs=1, reward_threshold=0.7, tags=("conditional-skill", "exploration", "procedural-wall"),
                         publication_role="conditional-control benchmark", env_factory=envs.make_walljump_env),
    "ant": TaskSpec("ant", "Crawler (synthetic 172/20 articulated chain)", "continuous-control", "gymnasium", "benchmark", "ppo", "ant_policy",
                    3_000_000, eval_episodes=20, n_envs=8, tags=("locomotion", "articulated"), action="continuous",
                    publication_role="locomotion throughput/scaling shape",
                    notes="The reference delegates to gymnasium Ant-v5 (MuJoCo, envs.py:274-277); this engine provides a build-defined "
                          "172-dim-obs / 20-dim-action stand-in of the BASELINE shape (physics parity unpinned).",
                    env_factory=envs.make_ant_env),
}

# remaining registry ids: (title, family, interface, tier, algorithm, prefix, timesteps, eval_episodes, n_envs, observation, action)
_CATALOGUE = {
    "brickbreak": ("Brick Break", "arcade", "gymnasium", "benchmark", "ppo", "brickbreak_policy", 500_000, 50, 8, "vector", "discrete"),
    "bicycle": ("Bicycle", "continuous-control", "gymnasium", "benchmark", "ppo", "bicycle_policy", 500_000, 50, 8, "vector", "discrete"),
    "glider": ("Glider", "aerospace", "gymnasium", "frontier", "ppo", "glider_policy", 1_000_000, 50, 8, "vector", "discrete"),
    "labyrinth": ("Labyrinth", "games", "gymnasium", "frontier", "ppo", "labyrinth_policy", 2_000_000, 100, 8, "image", "discrete"),
    "astrodynamics": ("Astrodynamics", "aerospace", "gymnasium", "frontier", "ppo", "astrodynamics_policy", 2_000_000, 50, 8, "vector", "discrete"),
    "kraken": ("Kraken", "games", "gymnasium", "benchmark", "ppo", "kraken_policy", 1_000_000, 50, 8, "vector", "multi-discrete"),
    "worm": ("Worm", "continuous-control", "gymnasium", "benchmark", "ppo", "worm_policy", 2_000_000, 20, 8, "vector", "continuous"),
    "foodcollector": ("Food Collector", "multi-agent", "pettingzoo", "roadmap", "ippo", "foodcollector_policy", 2_000_000, 20, 1, "vector", "hybrid"),
    "intersection": ("Intersection", "multi-agent", "pettingzoo", "frontier", "mappo", "intersection_policy", 5_000_000, 20, 1, "vector", "discrete"),
    "minecraft": ("Minecraft", "open-ended-games", "pettingzoo", "frontier", "hierarchical-rl-plus-llm", "minecraft_policy", 10_000_000, 20, 1, "vector", "discrete"),
    "simcity": ("SimCity", "open-ended-games", "pettingzoo", "frontier", "hierarchical-rl-plus-llm", "simcity_policy", 10_000_000, 20, 1, "vector", "discrete"),
    "fish": ("Fish", "multi-agent", "pettingzoo", "roadmap", "ippo", "fish_policy", 3_000_000, 20, 1, "vector", "discrete"),
    "self-driving-car": ("Self-Driving Car", "safety", "pettingzoo", "frontier", "mappo", "self_driving_car_policy", 5_000_000, 20, 1, "vector", "discrete"),
}
for _id, (_title, _fam, _iface, _tier, _algo, _prefix, _steps, _eval, _n, _obs, _act) in _CATALOGUE.items():
    TASKS[_id] = TaskSpec(_id, _title, _fam, _iface, _tier, _algo, _prefix, _steps, eval_episodes=_eval, n_envs=_n, observation=_obs, action=_act,
                          status=_OUT_OF_SCOPE)


def list_tasks(*, include_roadmap: bool = True) -> list[TaskSpec]:
    tasks = list(TASKS.values())
    if not include_roadmap:
        tasks = [task for task in tasks if task.trainable]
    return sorted(tasks, key=lambda task: (task.family, task.id))


def list_task_cards(*, include_roadmap: bool = True) -> list[dict[str, Any]]:
    return [task.card() for task in list_tasks(include_roadmap=include_roadmap)]


def get_task(task_id: str) -> TaskSpec:
    normalized = task_id.lower().replace("_", "-")
    aliases = {"brick-break": "brickbreak", "food-collector": "foodcollector", "self_driving_car": "self-driving-car", "crawler": "ant"}
    key = aliases.get(normalized, normalized)
    if key not in TASKS:
        raise KeyError(f"Unknown task '{task_id}'. Available: {', '.join(sorted(TASKS))}")
    return TASKS[key]


def make_env(task_id: str):
    task = get_task(task_id)
    if not task.trainable or task.env_factory is None:
        raise ValueError(f"Task '{task_id}' is not a Gymnasium/SB3 trainable task yet.")
    return task.env_factory()
