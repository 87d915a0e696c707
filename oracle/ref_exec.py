"""Runs the reference's OWN NN-stage statements (build container only; test infrastructure).

`PKG/scripts/data_handler.py` cannot be imported: it is Python 2 (`except X, e`, `print "..."`) and imports rospy /
cv2 / tensorflow (SyntaxError / ModuleNotFoundError -- ordinary errors, nothing was denied).
But the functions on the hot path are Python-3-clean once the Python-2-only tails are cut off, so this module reads
the file AS TEXT from /root/reference at run time, slices those functions by their `def` lines, and `exec`s them with

  * `rospy` = a logging stub (loginfo / logwarn / logerr do nothing),
  * `np`, `cdist`, `collections` = the real numpy / scipy / collections (the reference's arithmetic libraries),
  * `self` = a plain object carrying the DataHandler attributes the statements touch.

What is executed (reference text, never stored in this repository and never shipped to the GPU box):
  find_matches                       data_handler.py:166-209  whole function
  add_frames_kept_pairs_to_ignore    :437-438                 whole function
  get_kf_ids_from_frames_kept_ids    :440-441                 whole function
  get_geom_features                  :421-422                 whole function
  receive_separators_service         :373-408  up to (not including) the "# Add the separator to the factor graph"
                                     block, whose try/except is Python-2 syntax: the mask bookkeeping :387-408
  find_matches_service               :297-328  up to (not including) the pose-estimate block (:330-335, Python-2
                                     try/except); ONE line written here is appended: `return (kf_matched_ids,
                                     matches_computing_robot_resp, matches_querying_robot_resp)`

Only `oracle/gen_golden.py` imports this module; the fixtures it writes (`tests/golden/*.npz`) are data.
"""
import collections
import os
import textwrap
import types

import numpy as np
from scipy.spatial.distance import cdist

REF_FILE = "/root/reference/ros_ws/src/multi_robot_separators/scripts/data_handler.py"


class _Log:
    """rospy stand-in: log calls do nothing, except that the LAST ndarray handed to logwarn is kept -- find_matches
    logs its masked distance matrix (data_handler.py:206), which lets the fixtures record the reference's own
    per-row minima."""
    last_array = None

    def logwarn(self, *a, **k):
        if a and isinstance(a[0], np.ndarray):
            _Log.last_array = a[0].copy()

    def __getattr__(self, name):
        if name.startswith("log"):
            return lambda *a, **k: None
        if name == "ServiceException":
            return Exception
        raise AttributeError(name)


def available():
    return os.path.isfile(REF_FILE)


def _method_source(lines, name, stop_before=None):
    """Text of `def name(self...)` inside class DataHandler: from its def line to the next def of the same
    indentation, or to the first line containing `stop_before`."""
    start = None
    for i, l in enumerate(lines):
        if l.startswith("    def %s(" % name):
            start = i
            break
    if start is None:
        raise RuntimeError("reference has no method %s" % name)
    end = len(lines)
    for j in range(start + 1, len(lines)):
        if lines[j].startswith("    def ") or (stop_before is not None and stop_before in lines[j]):
            end = j
            break
    return textwrap.dedent("".join(lines[start:end])), (start + 1, end)


def load():
    """-> (namespace of executed reference functions, {name: (first line, last line)})"""
    with open(REF_FILE, "r") as fh:
        lines = fh.readlines()
    g = {"np": np, "cdist": cdist, "collections": collections, "rospy": _Log(),
         # response constructors of the ROS services: plain tuples here
         "FindMatchesResponse": lambda *a: tuple(a), "ReceiveSeparatorsResponse": lambda *a: tuple(a)}
    spans = {}
    for name, stop in (("find_matches", None), ("add_frames_kept_pairs_to_ignore", None),
                       ("get_kf_ids_from_frames_kept_ids", None), ("get_geom_features", None),
                       ("receive_separators_service", "# Add the separator to the factor graph"),
                       ("find_matches_service", "pose_estimates = collections.deque()")):
        src, span = _method_source(lines, name, stop)
        if name == "find_matches_service":
            src += "    return (kf_matched_ids, matches_computing_robot_resp, matches_querying_robot_resp)\n"
        exec(compile(src, "<reference %s:%d-%d>" % (os.path.basename(REF_FILE), span[0], span[1]), "exec"), g)
        spans[name] = span
    return g, spans


class RefDataHandler:
    """The attributes DataHandler.__init__ creates that the executed statements touch (data_handler.py:96-141,268),
    with the reference's functions bound as methods."""

    def __init__(self, netvlad_distance, netvlad_dimensions, netvlad_max_matches_nb):
        g, self.spans = load()
        self.netvlad_distance = netvlad_distance
        self.netvlad_dimensions = netvlad_dimensions
        self.netvlad_max_matches_nb = netvlad_max_matches_nb
        self.local_descriptors = []            # lists of python floats, as .tolist() leaves them (:157-158)
        self.received_descriptors = []
        self.local_kf_already_used = []
        self.other_kf_already_used = []
        self.frames_kept_pairs_ignored = []
        self.separators_found = []
        self.kf_ids_of_frames_kept = []
        self.geometric_feats = []
        self.send_estimates_of_poses = False
        for name in ("find_matches", "add_frames_kept_pairs_to_ignore", "get_kf_ids_from_frames_kept_ids",
                     "get_geom_features", "receive_separators_service", "find_matches_service"):
            setattr(self, name, types.MethodType(g[name], self))


def find_matches(local, received, local_used, other_used, pairs_ignored, netvlad_distance, max_matches_nb):
    """One call of the reference's find_matches on the given state -> list of (idx_local, idx_other)."""
    h = RefDataHandler(netvlad_distance, int(np.asarray(local).shape[1]), max_matches_nb)
    h.local_descriptors = np.asarray(local, dtype=np.float64).tolist()
    h.received_descriptors = np.asarray(received, dtype=np.float64).tolist()
    h.local_kf_already_used = [int(v) for v in local_used]
    h.other_kf_already_used = [int(v) for v in other_used]
    h.frames_kept_pairs_ignored = [[int(a), int(b)] for a, b in pairs_ignored]
    _Log.last_array = None
    matches = [(int(a), int(b)) for a, b in h.find_matches()]
    return matches, _Log.last_array          # (matches, the masked distance matrix the reference logged)
