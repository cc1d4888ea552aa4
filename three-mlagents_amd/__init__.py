"""three-mlagents_amd: MI355X-native vectorized-env + PPO engine behind the three-mlagents research API.

Mirrors /root/reference/backend/mlagents/__init__.py:8-10 (`TaskSpec, get_task, list_tasks, make_env`) and adds the
device-side entry points.  Import as `three_mlagents_amd` (alias package at the repo root).
"""
__version__ = "0.1.0"
__all__ = ["__version__"]
