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

Answer sets (compute_shingles) are built on the device too (qrlsh.answers).
User similarity, the prediction loop, CSV ingest and the interactive prompt are outside this
round's scope.
"""
import math
import time

import numpy as np
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


class Recommender:

    device = "cuda"
    verbose = True
    bands = None            # override the band rule (BASELINE shapes 128/32, 256/64 need it)
    max_candidates = None   # override K

    def _log(self, *a):
        if self.verbose:
            print(*a)

    def init(self, users, queries, queriesIDs, dataset, ratings):
        """recommender.py:51-64 for pandas / numpy inputs (the datatable ingest is row N3)."""
        def to_np(x):
            return x.to_numpy() if hasattr(x, "to_numpy") else np.asarray(x)
        self.usersIDs = to_np(users).T[0]
        self.queries = to_np(queries)
        self.queriesIDs = np.array(queriesIDs)
        self.dataset = dataset.astype(str) if hasattr(dataset, "astype") else dataset
        self.tupleCount = {}
        r = ratings.drop(columns=["user"]) if hasattr(ratings, "drop") and "user" in getattr(ratings, "columns", []) else ratings
        self.ratings = np.nan_to_num(to_np(r).astype(float), nan=0.0)

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
        off, rows = off.cpu().numpy(), rows.cpu().numpy()
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
        out = sig.cpu().numpy().astype(np.int64)
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
