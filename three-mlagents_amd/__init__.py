"""three-mlagents_amd: MI355X-native vectorized-env + PPO engine behind the three-mlagents research API.

Mirrors /root/reference/backend/mlagents/__init__.py:8-10 (`TaskSpec, get_task, list_tasks, make_env`).
Import as `three_mlagents_amd` (alias package at the repo root; the directory name has a hyphen).
"""
__version__ = "0.1.0"

from .registry import TaskSpec, get_task, list_tasks, make_env  # noqa: E402

__all__ = ["TaskSpec", "get_task", "list_tasks", "make_env", "__version__"]
