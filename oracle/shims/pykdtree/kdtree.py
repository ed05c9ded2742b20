"""pykdtree stand-in (TEST INFRASTRUCTURE ONLY): exact Euclidean kNN through scipy.spatial.cKDTree.

pykdtree (storpipfugl/pykdtree, C + OpenMP, un-pinned in the reference's requirements.txt:5) is not installable here.
Semantics relied on by slam/odometry/local_map.py:369,385,405: exact kNN sorted by distance, query(x) -> (dist[N], idx[N]),
query(x, k) -> (N, k).  cKDTree computes in float64; the answer is the same mathematical nearest neighbour except on
exact distance ties.
"""
import os
import numpy as np
from scipy.spatial import cKDTree


class KDTree:
    def __init__(self, data, leafsize=16):
        self.data = np.asarray(data)
        self._tree = cKDTree(self.data.astype(np.float64), leafsize=leafsize)
        self._workers = int(os.environ.get("ORACLE_KDTREE_WORKERS", "-1"))

    def query(self, query_pts, k=1, **kwargs):
        d, i = self._tree.query(np.asarray(query_pts, dtype=np.float64), k=k, workers=self._workers)
        return d.astype(self.data.dtype, copy=False), i.astype(np.int64, copy=False)
