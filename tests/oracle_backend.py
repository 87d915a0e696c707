"""Oracle-backed twin of data_handler.FinderBackend (TEST INFRASTRUCTURE): same interface, CPU
restatement underneath, so the host-side mirrors can be replayed against the oracle."""
import numpy as np

from oracle import pyoracle


class OracleBackend:
    def __init__(self, params):
        self.p = params
        self.local, self.received = [], []
        self.local_used, self.other_used, self.ignored = [], [], []

    def nn_append_local(self, rows):
        self.local.extend(np.asarray(rows, dtype=np.float64).reshape(-1, np.asarray(rows).shape[-1]))

    def nn_append_received(self, rows):
        self.received.extend(np.asarray(rows, dtype=np.float64).reshape(-1, np.asarray(rows).shape[-1]))

    def mark_local_used(self, i):
        self.local_used.append(int(i))

    def mark_other_used(self, j):
        self.other_used.append(int(j))

    def ignore_pair(self, i, j):
        self.ignored.append((int(i), int(j)))

    def find_matches(self):
        m, _, _ = pyoracle.find_matches(np.array(self.local), np.array(self.received), self.local_used,
                                        self.other_used, self.ignored, self.p.netvlad_distance,
                                        self.p.netvlad_max_matches_nb)
        return [(int(r["idx_local"]), int(r["idx_other"])) for r in m]

    def estimate_transform(self, f_from, f_to):
        return pyoracle.estimate_transform(self.p, f_from, f_to)
