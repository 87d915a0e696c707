"""The section 8(e) partition (row-sharded NN + all-gather of minima + replicated walk, candidates p mod G over a
replicated store, flag + record exchange, interleave) run by 2 and 3 gloo ranks on CPU must deliver exactly what one
rank delivers: same candidate list, same flags, same accepted records in the same order, on EVERY rank.  The
orchestration is multi_robot_slam_separators_amd/sharded.py -- the code path `bench.py --partition 8e` runs on the GPUs
-- with the oracle standing in for the compute."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(world, prefix, *extra):
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_worker.py"), str(r), str(world), port, prefix, *extra],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    for r, p in enumerate(procs):
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        assert p.returncode == 0 and ("rank %d ok" % r) in o.decode(), o.decode()[-3000:]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_step_equals_single_rank(tmp_path, oracle, world):
    ref = str(tmp_path / "ref")
    _run(1, ref)
    m0 = np.load(ref + "_m_0.npy")
    f0 = np.load(ref + "_flags_0.npy")
    a0 = np.load(ref + "_acc_0.npy")
    assert len(m0) >= 20 and 5 <= f0.sum() < len(m0)           # a real mix of accepted and rejected candidates
    out = str(tmp_path / ("w%d" % world))
    _run(world, out)
    for r in range(world):
        assert np.load(out + "_m_%d.npy" % r).tobytes() == m0.tobytes(), "candidate list differs on rank %d" % r
        assert np.array_equal(np.load(out + "_flags_%d.npy" % r), f0)
        assert np.load(out + "_acc_%d.npy" % r).tobytes() == a0.tobytes(), "accepted records differ on rank %d" % r


def test_sharded_step_dense_fallback(tmp_path, oracle):
    """One rank reports "candidate set too dense for the device path": every rank sees it in the gathered block, that
    rank recomputes synchronously, the minima are gathered once more (the step's one wait + the three of the fallback,
    asserted in the worker) and the result is the same."""
    ref = str(tmp_path / "ref")
    _run(1, ref)
    out = str(tmp_path / "dense")
    _run(2, out, "dense")
    for r in range(2):
        for part in ("m", "flags", "acc"):
            assert np.load(out + "_%s_%d.npy" % (part, r)).tobytes() == np.load(ref + "_%s_0.npy" % part).tobytes()


def test_row_blocks_and_flatten():
    from multi_robot_slam_separators_amd import _abi, sharded
    assert sharded.row_blocks(10, 3) == [(0, 3), (3, 6), (6, 10)]
    assert sharded.row_blocks(2, 4) == [(0, 0), (0, 1), (1, 1), (1, 2)]
    m1 = np.zeros(3, dtype=_abi.MATCH_DTYPE); m1["idx_local"] = [4, 5, 6]; m1["idx_other"] = [1, 2, 3]
    m2 = np.zeros(2, dtype=_abi.MATCH_DTYPE); m2["idx_local"] = [9, 8]; m2["idx_other"] = [7, 7]
    ids, il, io = sharded.flatten_candidates([(0, m1), (3, m2)])
    assert ids.tolist() == [0, 0, 0, 3, 3] and il.tolist() == [4, 5, 6, 9, 8] and io.tolist() == [1, 2, 3, 7, 7]
