"""N4 -- user similarity on the device (Recommender.compute_userSimilarities, recommender.py:216-290).

cluster_labels(): the reference's scikit-learn pipeline (StandardScaler -> PCA -> BIRCH, :226-261) on the
host, exactly as the reference calls it -- the library IS the reference's algorithm for that step.
user_similarities(): everything after it (:263-288) on the device, with the hot path's own kernels: rows
centred with the reference's integer truncation (qrlsh_center_rows), the pairs of every cluster from the
bucket machinery (labels = a one-band key), cosine by qrlsh_score_pairs on the integer rows, negatives
dropped, per-user top-K by qrlsh_topk_select_*.
"""
import math

import numpy as np
import torch

from . import _lib, ops
from .ops import _ptr, _stream


def max_candidates(nu):
    """K = round(log_1.5 nu)  (recommender.py:220)"""
    return round(math.log(nu, 1.5))


def cluster_labels(ratings):
    """recommender.py:226-261 -> int64 labels [nu]; clusters of a single user share the label n_clusters"""
    from sklearn.cluster import Birch
    from sklearn.decomposition import PCA
    from sklearn.preprocessing import StandardScaler
    ratings = np.asarray(ratings)
    nu = ratings.shape[0]
    n_clusters = round(nu ** (1 / 1.3))
    feats = StandardScaler().fit_transform(ratings)
    feats = PCA(n_components=min(feats.shape[0], feats.shape[1], 200)).fit(feats).transform(feats)
    label = Birch(n_clusters=n_clusters).fit(feats).predict(feats).astype(np.int64)
    sizes = np.bincount(label)
    label[np.isin(label, np.flatnonzero(sizes == 1))] = n_clusters     # pool the singletons (:259-261)
    return label


def center_rows(ratings):
    """int32 [nu][nq] device tensor -> the truncated centred rows, stride padded to a multiple of 4"""
    lib = _lib.load()
    nu, nq = ratings.shape
    stride = (nq + 3) // 4 * 4
    out = torch.empty((nu, stride), dtype=torch.int32, device=ratings.device)
    _lib.check(lib.qrlsh_center_rows(_ptr(ratings), nu, nq, stride, _ptr(out), _stream()))
    return out


def user_similarities(ratings, labels, K=None, device="cuda"):
    """ratings (nu, nq) integers (0 = missing), labels (nu,) cluster ids ->
    (src, dst, milli) int32 device tensors: for every user its at most K most similar users of the same cluster
    with POSITIVE rounded similarity, sorted by (user, value descending, neighbour id ascending).
    The reference's lists can also hold zero-valued entries (the user itself, negative cosines set to 0) when a
    cluster has fewer than K positive neighbours; they weigh nothing in weighted_average and are left out."""
    r = ratings if isinstance(ratings, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(ratings), dtype=np.int32))
    r = r.to(device=device, dtype=torch.int32).contiguous()
    nu = r.shape[0]
    if K is None:
        K = max_candidates(nu)
    lab = labels if isinstance(labels, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(labels), dtype=np.int64))
    keys = lab.to(device=device, dtype=torch.int64).reshape(1, nu).contiguous()
    rows = center_rows(r)
    norms = ops.row_norms(rows)
    pairs = ops.candidate_pairs(keys, 4)                    # users sharing a label, u < v, sorted
    z = torch.empty((0,), dtype=torch.int32, device=device)
    if pairs.numel() == 0:
        return z, z.clone(), z.clone()
    milli = ops.score_pairs(rows, norms, pairs)[0]
    keep = milli > 0                                         # :276 negatives -> 0; zero weights are dropped
    pairs, milli = pairs[keep], milli[keep].contiguous()
    if pairs.numel() == 0:
        return z, z.clone(), z.clone()
    ib = ops.id_bits_for(nu)
    inv = (1000 - milli).to(torch.int64)
    pi, pj = pairs >> 32, pairs & 0xFFFFFFFF
    if ops.wide_ids(ib):
        rev = ((pj << 11) | inv, pi.to(torch.int32))
    else:
        rev = (pj << (ib + 11)) | (inv << ib) | pi
    return ops.topk_select(pairs.contiguous(), milli, rev, K, ib, nu)


def sims_to_dict(src, dst, val, nu):
    """COO -> {u: {'indexes': int64[], 'values': float64[]}} for EVERY user (recommender.py:278-288 builds an
    entry per user; users without a positive neighbour get empty arrays)"""
    src, dst, val = (t.cpu().numpy() for t in (src, dst, val))
    out = {u: {"indexes": np.zeros(0, np.int64), "values": np.zeros(0, np.float64)} for u in range(nu)}
    if len(src):
        cut = np.flatnonzero(np.diff(src)) + 1
        for s, e in zip(np.concatenate(([0], cut)), np.concatenate((cut, [len(src)]))):
            out[int(src[s])] = {"indexes": dst[s:e].astype(np.int64), "values": val[s:e].astype(np.float64) / 1000.0}
    return out
