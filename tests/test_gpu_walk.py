"""The argsort + walk of DataHandler.find_matches (PKG/scripts/data_handler.py:191-205) ON THE DEVICE (sf_nn_walk_device:
k_walk_tile_sort / k_walk_rank / k_walk_emit, csrc/k_nn.hip) against a restatement of the reference's loop and the library's host walk
(sf_nn_walk) on the same per-row minima: bit-exact match lists, in walk order -- ties between equal minima (lowest row
first), columns claimed by several rows (the first in sorted order wins, the others still consume their slot, :199-200),
rows at / over the threshold, +inf rows, max_matches_nb below and above the row count, `cap` truncation, row counts on
both sides of the 2048-row sort tile."""
import numpy as np
import pytest
import torch

from multi_robot_slam_separators_amd import _abi, lib, synth

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")


def _minima(seed, n_l, n_r, frac_under=0.6, ties=True):
    rng = np.random.default_rng(seed)
    rm = rng.uniform(0.0, 0.13 / frac_under, size=n_l)
    arg = rng.integers(0, max(1, n_r // 2), size=n_l).astype(np.int32)        # many rows share a column
    if ties and n_l >= 8:
        k = max(2, n_l // 7)
        src = rng.integers(0, n_l, size=k)
        dst = rng.integers(0, n_l, size=k)
        rm[dst] = rm[src]                                                     # exact ties, arbitrary row order
        rm[rng.integers(0, n_l, size=max(1, n_l // 50))] = 0.0
        rm[rng.integers(0, n_l, size=max(1, n_l // 50))] = -0.0
    rm[rng.integers(0, n_l, size=max(1, n_l // 20))] = np.inf                 # rows without a candidate
    if n_l > 3:
        rm[3] = 0.13                                                          # exactly the threshold: not under it
    return rm, arg


def _ref_walk(rm, arg, n_r, thr, max_nb, cap):
    """data_handler.py:191-205 restated on explicit row minima: stable argsort (lowest index wins
    ties), the first min(N_l, max_matches_nb) rows, skip a taken idx_other (the slot is consumed), stop at the first row
    that is not under the threshold."""
    order = np.argsort(rm, kind="stable")
    out, taken = [], set()
    for s in range(min(len(rm), max_nb)):
        il = int(order[s])
        if not rm[il] < thr:
            break
        io = int(arg[il])
        if io < 0 or io >= n_r or io in taken:
            continue
        out.append((il, io, rm[il]))
        taken.add(io)
        if len(out) >= cap:
            break
    return np.array(out, dtype=_abi.MATCH_DTYPE) if out else np.zeros(0, dtype=_abi.MATCH_DTYPE)


def _device_walk(f, rm, arg, n_r, cap, status=0):
    n_l = len(rm)
    d_rm = torch.from_numpy(rm).to(DEV)
    d_arg = torch.from_numpy(arg).to(DEV)
    d_st = torch.tensor([status], dtype=torch.int32, device=DEV)
    d_m = torch.zeros((max(cap, 1), _abi.MATCH_DTYPE.itemsize), dtype=torch.uint8, device=DEV)
    d_n = torch.full((1,), -7, dtype=torch.int32, device=DEV)
    f.nn_walk_device(d_rm.data_ptr(), d_arg.data_ptr(), d_st.data_ptr(), n_l, n_r, d_m.data_ptr(), cap, d_n.data_ptr())
    f.synchronize()
    n = int(d_n.item())
    return np.frombuffer(d_m.cpu().numpy().tobytes(), dtype=_abi.MATCH_DTYPE)[:n].copy()


@pytest.mark.parametrize("n_l,n_r,max_nb", [(1, 1, 20), (5, 3, 20), (64, 64, 20), (300, 500, 300), (2047, 700, 5000),
                                            (2048, 2048, 2048), (2049, 100, 100000), (10000, 10000, 10000),
                                            (10000, 10000, 20), (50000, 20000, 50000)])
def test_device_walk_equals_host_walk(n_l, n_r, max_nb):
    p = synth.camera_params()
    p.netvlad_max_matches_nb = max_nb
    p.netvlad_distance = 0.13
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        for seed in range(3):
            rm, arg = _minima(1000 * seed + n_l, n_l, n_r)
            want = f.nn_walk(rm, arg, n_r, cap=n_l)
            ref = _ref_walk(rm, arg, n_r, 0.13, max_nb, n_l)
            assert np.array_equal(want["idx_local"], ref["idx_local"]) and np.array_equal(want["idx_other"], ref["idx_other"])
            assert np.array_equal(want["distance"], ref["distance"])
            got = _device_walk(f, rm, arg, n_r, n_l)
            assert got.tobytes() == want.tobytes(), (n_l, n_r, max_nb, seed, len(got), len(want))


def test_device_walk_cap_status_and_column_range():
    """`cap` ends the walk like the host's `if n >= cap: break`; a non-zero status word voids it; a column outside the
    received range never matches but still consumes its slot (sf_nn_walk's guard for caller-provided minima)."""
    p = synth.camera_params()
    p.netvlad_max_matches_nb = 4000
    n_l, n_r = 3000, 800
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        rm, arg = _minima(42, n_l, n_r)
        arg[::17] = n_r + 5
        arg[5::29] = -1
        full = f.nn_walk(rm, arg, n_r, cap=n_l)
        assert len(full) > 100
        for cap in (1, 7, 64, len(full) - 1, len(full), len(full) + 10):
            want = f.nn_walk(rm, arg, n_r, cap=cap)
            got = _device_walk(f, rm, arg, n_r, cap)
            assert got.tobytes() == want.tobytes() and len(got) == min(cap, len(full))
        assert len(_device_walk(f, rm, arg, n_r, n_l, status=1)) == 0
        # nothing under the threshold
        none = np.full(n_l, 0.5)
        assert len(_device_walk(f, none, arg, n_r, n_l)) == 0


def test_device_walk_all_rows_tied():
    """Every minimum equal (one key for 6000 rows across three sort tiles): the order is the row order."""
    p = synth.camera_params()
    p.netvlad_max_matches_nb = 10000
    n_l = 6000
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        rm = np.full(n_l, 0.05)
        arg = np.arange(n_l, dtype=np.int32)[::-1].copy()
        got = _device_walk(f, rm, arg, n_l, n_l)
        assert np.array_equal(got["idx_local"], np.arange(n_l)) and np.array_equal(got["idx_other"], arg)
        assert got.tobytes() == f.nn_walk(rm, arg, n_l, cap=n_l).tobytes()
