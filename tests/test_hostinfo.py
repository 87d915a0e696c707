"""Host-side helper: the CPU share used to size thread pools (torch intra-op, oracle OpenMP)."""
import os

from multi_robot_slam_separators_amd.hostinfo import cpu_share


def test_cpu_share_is_positive_and_within_the_affinity_mask():
    n = cpu_share()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def test_oracle_team_matches_the_cpu_share(oracle):
    assert oracle.num_threads() <= cpu_share()
