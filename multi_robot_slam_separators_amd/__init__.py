"""MI355X-native inter-robot separator finder (hot path of multi_robot_SLAM_separators)."""
from . import _abi  # noqa: F401
