"""Drop-in for the hot-path half of the reference's recommender.py (lines 105-214).

Same names and return types as the reference's Recommender for the query-similarity path:

    compute_shingles()            -> dict row -> [queries]            (recommender.py:68-103)
    compute_signatures()          -> int64 ndarray (nq, PERM)         (recommender.py:105-143)
    compute_querySimilarities()   -> {q: {'indexes', 'values'}}       (recommender.py:145-214)

The MinHash / LSH / scoring / top-K work runs in libqrlsh (HIP, gfx950).  The module constant
PERM and the global legacy numpy RNG are honoured exactly like the reference: under
np.random.seed(s) the signatures are bit-identical to the reference's.  Tie order inside a
query's top-K list is defined here (value descending, then neighbour id ascending); the
reference's is arbitrary.

Widening beyond the hot path: answer sets (compute_shingles, N2) and the
hybrid prediction loop (compute_scores / weighted_average, N1) run on the device; CSV ingest
(init / parse_queries, N3) takes datatable Frames (what main.py passes) or pandas frames; user similarity (N4) is the same
sklearn pipeline the reference calls, on the host.
"""
import math
import time

import numpy as np
import pandas as pd
import torch

import qrlsh
from qrlsh import ops, pipeline
from lsh import LSH  # noqa: F401  (same import the reference has)

# constants (recommender.py:30-34)
PERM = 180  # number of independent hash functions

QUERY_WEIGHT = 0.6
USER_WEIGHT = 0.4
DEFAULT_MEAN = 60

LSH_THRESH = 0.2  # recommender.py:153


def _as_pandas(x):
    """pandas view of a frame-like argument of init(): a pandas DataFrame as is, a datatable Frame (or any object
    with .to_pandas()) through its own conversion, anything else through numpy"""
    if isinstance(x, pd.DataFrame):
        return x
    if hasattr(x, "to_pandas"):
        return x.to_pandas()
    names = getattr(x, "names", None)
    return pd.DataFrame(np.asarray(x), columns=list(names) if names is not None else None)


def _as_numpy(x):
    return x.to_numpy() if hasattr(x, "to_numpy") else np.asarray(x)


class Recommender:

    device = "cuda"
    verbose = True
    bands = None            # override the band rule (BASELINE shapes 128/32, 256/64 need it)
    max_candidates = None   # override K

    def _log(self, *a):
        if self.verbose:
            print(*a)

    cluster_on_device = True   # StandardScaler + PCA of compute_userSimilarities on the device (False: scikit-learn on the host)
    sum_order = "sequential"   # order of weighted_average's two np.sum calls, see compute_scores

    def init(self, users, queries, queriesIDs, dataset, ratings):
        """recommender.py:51-64.  Takes what main.py:23-85 passes -- `datatable` Frames (anything Frame-like:
        `.to_pandas()` / `.to_numpy()` / `.names`) -- and pandas frames / arrays alike (N3): users = one column
        of user ids, queries = the frame parse_queries returned, dataset = the table (every column becomes
        str, :57), ratings = the utility matrix with its leading 'user' column (dropped, :61; missing ratings
        become 0, :62).  The caller's frames are not modified (the reference edits `dataset` and `ratings` in
        place; nothing reads them afterwards)."""
        self.usersIDs = _as_numpy(users).T[0]
        self.queries = _as_numpy(queries)
        self.queriesIDs = np.array(queriesIDs)
        self.dataset = _as_pandas(dataset).astype(str)
        self.tupleCount = {}
        r = _as_pandas(ratings)
        if "user" in list(r.columns):
            r = r.drop(columns=["user"])
        r = r.apply(pd.to_numeric, errors="coerce") if any(dt == object for dt in r.dtypes) else r
        self.ratings = np.nan_to_num(r.to_numpy(dtype=np.float64, na_value=np.nan), nan=0.0).astype(np.int64)

    def parse_queries(self, path: str):
        """recommender.py:386-416: one query per line, `id,attr=value,attr=value,...`;
        -> (DataFrame with one column per dataset feature, "" where unconstrained; list of ids)"""
        ids, data = [], []
        with open(path) as fh:
            for line in fh:
                parts = line.rstrip("\n").split(",")
                ids.append(parts[0])
                row = ["" for _ in self.datasetFeatures]
                for item in parts[1:]:
                    name, value = item.split("=")
                    row[self.datasetFeatures.index(name)] = value
                data.append(row)
        return pd.DataFrame(data, columns=list(self.datasetFeatures), dtype=object), ids

    # ---- producer of the hot path's input (row N2: answer sets on the device) -----
    def answer_sets_device(self):
        """CSR answer sets on the device: (offsets int64 [nq+1], rows int32 [nnz]); the table is
        dictionary-encoded into per-(feature, value) bitmaps once and cached."""
        from qrlsh import answers
        idx = getattr(self, "_answer_index", None)
        if idx is None or getattr(self, "_answer_index_key", None) != id(self.dataset):
            cols = [self.dataset[f].to_numpy() for f in self.datasetFeatures]
            idx = answers.build_answer_index(cols, self.device)
            self._answer_index, self._answer_index_key = idx, id(self.dataset)
        qrows = answers.encode_queries(idx, self.queries)
        return answers.answer_sets(idx, qrows)

    def answer_sets(self):
        """host copy of answer_sets_device(): (offsets int64 [nq+1], rows int32 [nnz])"""
        off, rows = self.answer_sets_device()
        off, rows = ops.to_host(off), ops.to_host(rows)
        for q, n in enumerate(np.diff(off)):
            self.tupleCount[q] = int(n)      # recommender.py:93
        return off, rows

    def compute_shingles(self):
        """recommender.py:68-103: inverted index row -> [queries containing it]."""
        drows = self.dataset.shape[0]
        self._log("\nDataset : {}, Total queries: {}".format(drows, self.queriesIDs.size))
        initial = time.time()
        offsets, rows = self.answer_sets()
        shingles_dict = {d: [] for d in range(drows)}
        for q in range(self.queriesIDs.size):
            for ind in rows[offsets[q]:offsets[q + 1]]:
                shingles_dict[int(ind)].append(q)
        self._log(str(round(time.time() - initial, 3)) + "s for shingles_dict")
        return shingles_dict

    # ---- hot path ------------------------------------------------------------------
    def _device_inputs(self):
        offsets, rows = self.answer_sets_device()
        drows = self.dataset.shape[0]
        self._log("\nPermutations: {}".format(PERM))
        # PERM consecutive draws from the global legacy RNG, exactly as recommender.py:120
        perms = ops.legacy_permutations(PERM, drows, rng=np.random)
        table = ops.perm_table(perms, self.device)
        return offsets, rows, table

    def compute_signatures(self):
        """(nq, PERM) int64 signature matrix (recommender.py:105-143)."""
        initial = time.time()
        offsets, rows, table = self._device_inputs()
        sig, _, _ = ops.minhash(offsets, rows, table, b=None, want_norm=False)
        out = ops.to_host(sig).astype(np.int64)
        self._log(str(round(time.time() - initial, 3)) + "s for signature_matrix")
        return out

    def _band_rule(self):
        if self.bands is not None:
            return self.bands
        for b in list(range(1, PERM + 1))[::-1]:
            if PERM % b == 0 and b % 10 == 0:
                r = PERM / b
                thresh = round((1 / b) ** (1 / r), 2)
                if thresh >= LSH_THRESH:
                    return b
        raise ValueError("no band count satisfies the rule of recommender.py:156-163 for PERM=%d "
                         "(the reference raises UnboundLocalError here); set Recommender.bands" % PERM)

    def compute_querySimilarities(self):
        """{q: {'indexes': int64[<=K], 'values': float64[<=K]}} (recommender.py:145-214)."""
        queryTime = time.time()
        nq = self.queriesIDs.size
        MAX_CANDIDATES = self.max_candidates if self.max_candidates is not None else round(math.log(nq, 1.5))
        self._log("\nQuery Thresh: " + str(LSH_THRESH))
        band = self._band_rule()
        offsets, rows, table = self._device_inputs()
        self._log("\nMax query candidates: {}, Max bands: {}, Band size: {}, Total queries: {}".format(
            MAX_CANDIDATES, band, PERM / band, nq))
        initial = time.time()
        res = pipeline.query_similarities(offsets, rows, table, band, MAX_CANDIDATES)
        torch.cuda.synchronize()
        self._log("Candidate pairs [{}s]: {}".format(round(time.time() - initial, 3), res.pairs.numel()))
        self.last_result = res
        query_sim = pipeline.sims_to_dict(res.src, res.dst, res.val)
        self._log("\n" + str(round(time.time() - queryTime, 3)) + "s for overall queries_similarity scores")
        return query_sim

    # ---- N4: user similarity (clustering = the reference's sklearn call on the host; the rest on the device) ----
    def compute_userSimilarities(self):
        """{u: {'indexes', 'values'}} (recommender.py:216-290): StandardScaler -> PCA -> BIRCH (scikit-learn, as
        in the reference), then on the device the centred cosine inside each cluster (with the reference's
        integer truncation of the centred rows); the per-user cut is the reference's own numpy call
        (np.argsort(row)[::-1][:K], :282) on those scores, so the lists -- tie order and zero-valued entries
        included -- are the reference's.  qrlsh.users.user_similarities is the all-device variant with a
        defined tie order (value descending, id ascending) for sizes where a host loop per user is not wanted."""
        from qrlsh import users
        t0 = time.time()
        nu = self.usersIDs.size
        top = round(math.log(nu, 1.5))
        n_clusters = round(nu ** (1 / 1.3))
        self._log("\nMax user candidates: {}, Total users: {}".format(top, nu))
        self._log("\nCluster count: {}, Total users: {}".format(n_clusters, nu))
        # StandardScaler + PCA on the device (Gram matrix on the matrix cores), BIRCH by the reference's own
        # scikit-learn call; cluster_on_device = False: the whole clustering on the host, as the reference runs it
        label = users.cluster_labels(self.ratings, device=self.device if self.cluster_on_device else None)
        pairs, milli = users.cluster_pair_scores(self.ratings, label, self.device)
        user_sim = users.reference_cut(pairs, milli, label, top)
        self._log("\n" + str(round(time.time() - t0, 3)) + "s for overall users_similarity scores")
        return user_sim

    # ---- N1: hybrid prediction (device) ------------------------------------------------
    def compute_scores(self):
        """(scores_to_predict, finalPredictions DataFrame, scores_missed), recommender.py:292-343."""
        from qrlsh import predict
        self._log("\n========== QUERY SIMILARITY ==========")
        self.compute_querySimilarities()
        res = self.last_result
        self._log("\n========== USER SIMILARITY ==========")
        user_sim = self.compute_userSimilarities()
        self._log("\n========== WEIGHTED AVERAGES ==========")
        t0 = time.time()
        scores_to_predict = np.array(np.where(self.ratings == 0)).T
        # weighted_average (recommender.py:36) is @jit(nopython=True): where the reference runs as shipped
        # (requirements.txt: numba) its np.sum is ONE accumulator in index order -> sum_order "sequential" (default).
        # "pairwise" is numpy's own np.sum order, what the reference does with @jit removed -- the run the committed
        # fixtures were captured from (numba is not installable here), on which both orders give the same matrices;
        # parity with the numba build itself stays unpinned.
        final = predict.fill_predictions(self.ratings, res.src, res.dst, res.val, user_sim, QUERY_WEIGHT, USER_WEIGHT,
                                         DEFAULT_MEAN, self.device, sum_order=self.sum_order)
        final = ops.to_host(final)
        self._log(str(round(time.time() - t0, 3)) + "s for weighted averages")
        finalPredictions = pd.DataFrame(final, columns=self.queriesIDs, index=self.usersIDs).astype(int)
        scores_missed = np.array(np.where(finalPredictions == 0)).T
        return scores_to_predict, finalPredictions, scores_missed

    def top_k_queries(self, to_predict, predictions, missed, ask=input):
        """Interactive top-k prompt of recommender.py:345-381 (`ask` is injectable for tests)."""
        pred = predictions.to_numpy()
        nu = self.usersIDs.size
        again = ""
        while again.lower() != "no":
            user = -1
            while not (0 <= user < nu):
                txt = ask("Enter user ID: [int][Max: " + str(nu) + "] ")
                user = int(txt) - 1 if txt.isdigit() else -1
            fresh = [j for i, j in to_predict if i == user and pred[i][j] != 0]
            k = 0
            while not (0 < k <= len(fresh)):
                txt = ask("Enter number of recommendations: [int][Max: " + str(len(fresh)) + "] ")
                k = int(txt) if txt.isdigit() else 0
            order = np.argsort(pred[user][fresh])[::-1][:k]
            print("\nTop {} unrated query recommendations for U{}: ".format(k, user + 1))
            for rank, pos in enumerate(order):
                print("{}. Q{} - {}".format(rank + 1, fresh[pos] + 1, pred[user][fresh][pos]))
            print()
            again = ""
            while again.lower() not in ("yes", "no"):
                again = ask("Do you want more suggestions? [Yes-No][Default: Yes] ") or "yes"
