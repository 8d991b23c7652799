"""N4 -- user similarity on the device (Recommender.compute_userSimilarities, recommender.py:216-290).

cluster_labels(): the reference's scikit-learn pipeline (StandardScaler -> PCA -> BIRCH, :226-261): on the host
exactly as the reference calls it, or (device=...) with StandardScaler + PCA on the device -- the Gram matrix of the
standardized ratings on the matrix cores (qrlsh_user_gram), its eigen-decomposition, scores = U sqrt(lambda) -- and
only BIRCH (2000 points of 200 dimensions) left to the reference's scikit-learn call.
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


def pca_features_host(ratings):
    """recommender.py:226-234 with the reference's own scikit-learn calls on the host -> float64 [nu, n_comps]"""
    from sklearn.decomposition import PCA
    from sklearn.preprocessing import StandardScaler
    ratings = np.asarray(ratings)
    feats = StandardScaler().fit_transform(ratings)
    return PCA(n_components=min(feats.shape[0], feats.shape[1], 200)).fit(feats).transform(feats)


def standardized_gram(ratings):
    """int32 [nu][nq] device tensor -> (mean [nq], 1 / scale [nq], G [nu][nu]) float64 on the device: the column
    statistics of StandardScaler and G = Z Z^T of the standardized matrix Z (qrlsh_user_gram: matrix cores, the
    standardization fused into the operand staging; Z is never materialised)"""
    lib = _lib.load()
    nu, nq = ratings.shape
    dev = ratings.device
    mean = torch.empty((nq,), dtype=torch.float64, device=dev)
    inv = torch.empty((nq,), dtype=torch.float64, device=dev)
    gram = torch.empty((nu, nu), dtype=torch.float64, device=dev)
    ws = ops._ws(lib.qrlsh_user_gram_workspace_bytes(nu, nq), dev)
    _lib.check(lib.qrlsh_user_gram(_ptr(ratings), nu, nq, _ptr(mean), _ptr(inv), _ptr(gram), _ptr(ws), ws.numel(), _stream()))
    return mean, inv, gram


def pca_features(ratings, device="cuda"):
    """The same features on the device: the PCA scores of a (users x queries) matrix with users << queries are
    U_k sqrt(lambda_k) of the eigen-decomposition of its Gram matrix (standardized_gram); the small symmetric
    eigenproblem is a library call (torch.linalg.eigh on the device).  Equal to scikit-learn's transform(X) up to the
    sign of each component and float64 rounding -- BIRCH (Euclidean distances) does not see either.
    -> float64 device tensor [nu, n_comps], components by descending variance"""
    r = ratings if isinstance(ratings, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(ratings), dtype=np.int32))
    r = r.to(device=device, dtype=torch.int32).contiguous()
    nu, nq = r.shape
    k = min(nu, nq, 200)
    _, _, gram = standardized_gram(r)
    lam, vec = torch.linalg.eigh(gram)               # ascending eigenvalues
    lam_k = lam[nu - k:].flip(0).clamp_min(0.0)
    return vec[:, nu - k:].flip(1) * lam_k.sqrt()


def cluster_labels(ratings, device=None):
    """recommender.py:226-261 -> int64 labels [nu]; clusters of a single user share the label n_clusters.
    device=None: the whole step with the reference's scikit-learn calls on the host.  device="cuda": StandardScaler +
    PCA on the device (pca_features), BIRCH -- 200-dimensional points, milliseconds -- with the reference's own
    scikit-learn call on the host."""
    from sklearn.cluster import Birch
    nu = ratings.shape[0]
    n_clusters = round(nu ** (1 / 1.3))
    feats = pca_features_host(ratings) if device is None else ops.to_host(pca_features(ratings, device))
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


def cluster_pair_scores(ratings, labels, device="cuda"):
    """(pairs int64 [n] = u << 32 | v with u < v and labels[u] == labels[v], sorted; milli int32 [n] =
    rint(1000 * cosine of the two truncated centred rows)) on the device -- the O(sum of cluster_size^2 * nq)
    part of recommender.py:263-276, every entry of every cluster's similarity matrix above its diagonal"""
    r = ratings if isinstance(ratings, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(ratings), dtype=np.int32))
    r = r.to(device=device, dtype=torch.int32).contiguous()
    nu = r.shape[0]
    lab = labels if isinstance(labels, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(labels), dtype=np.int64))
    keys = lab.to(device=device, dtype=torch.int64).reshape(1, nu).contiguous()
    rows = center_rows(r)
    norms = ops.row_norms(rows)
    pairs = ops.candidate_pairs(keys, 4)                    # users sharing a label, u < v, sorted
    if pairs.numel() == 0:
        return pairs, torch.empty((0,), dtype=torch.int32, device=device)
    return pairs, ops.score_pairs(rows, norms, pairs)[0]


def user_similarities(ratings, labels, K=None, device="cuda"):
    """ratings (nu, nq) integers (0 = missing), labels (nu,) cluster ids ->
    (src, dst, milli) int32 device tensors: for every user its at most K most similar users of the same cluster
    with POSITIVE rounded similarity, sorted by (user, value descending, neighbour id ascending).
    The reference's lists can also hold zero-valued entries (the user itself, negative cosines set to 0) when a
    cluster has fewer than K positive neighbours -- they weigh nothing in weighted_average and are left out --
    and it orders equal values arbitrarily (np.argsort); when a tie straddles the K-th place the two cuts keep
    different (equally similar) neighbours.  reference_cut() reproduces the reference's own order on the host."""
    nu = ratings.shape[0]
    if K is None:
        K = max_candidates(nu)
    pairs, milli = cluster_pair_scores(ratings, labels, device)
    z = torch.empty((0,), dtype=torch.int32, device=device)
    keep = milli > 0                                         # :276 negatives -> 0; zero weights are dropped
    pairs, milli = pairs[keep].contiguous(), milli[keep].contiguous()
    if pairs.numel() == 0:
        return z, z.clone(), z.clone()
    ib = ops.id_bits_for(nu)
    inv = (1000 - milli).to(torch.int64)
    pi, pj = pairs >> 32, pairs & 0xFFFFFFFF
    if ops.wide_ids(ib):
        rev = ((pj << 11) | inv, pi.to(torch.int32))
    else:
        rev = (pj << (ib + 11)) | (inv << ib) | pi
    return ops.topk_select(pairs, milli, rev, K, ib, nu)


def reference_cut(pairs, milli, labels, K):
    """The reference's own per-user cut (recommender.py:273-288) applied on the host to the device's scores:
    per cluster the dense similarity matrix (milli / 1000, diagonal and negatives 0), per user
    np.argsort(row)[::-1][:K] -- the same numpy call on the same values, hence the same (arbitrary) order among
    ties and the same zero-valued entries as the reference.  -> {u: {'indexes', 'values'}}"""
    labels = np.asarray(labels)
    p = pairs.cpu().numpy()
    m = milli.cpu().numpy()
    pu, pv = (p >> 32).astype(np.int64), (p & 0xFFFFFFFF).astype(np.int64)
    order = np.argsort(labels[pu], kind="stable")
    pu, pv, m = pu[order], pv[order], m[order]
    lab_of_pair = labels[pu]
    cuts = np.flatnonzero(np.diff(lab_of_pair)) + 1
    seg = dict(zip(lab_of_pair[np.concatenate(([0], cuts))].tolist() if len(pu) else [],
                   zip(np.concatenate(([0], cuts)).tolist(), np.concatenate((cuts, [len(pu)])).tolist())))
    out = {}
    for c in np.unique(labels):
        members = np.flatnonzero(labels == c)
        local = {int(u): k for k, u in enumerate(members)}
        sim = np.zeros((len(members), len(members)), dtype=np.float64)
        if int(c) in seg:
            a, b = seg[int(c)]
            li = np.fromiter((local[int(u)] for u in pu[a:b]), dtype=np.int64, count=b - a)
            lj = np.fromiter((local[int(v)] for v in pv[a:b]), dtype=np.int64, count=b - a)
            val = np.maximum(m[a:b], 0).astype(np.float64) / 1000.0
            sim[li, lj] = val
            sim[lj, li] = val
        for k, u in enumerate(members):
            best = np.argsort(sim[k])[::-1][:K]
            out[int(u)] = {"indexes": members[best], "values": sim[k][best]}
    return out


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
