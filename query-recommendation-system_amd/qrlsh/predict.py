"""N1 -- the hybrid prediction loop on the device (recommender.py:301-331).

fill_predictions() takes the utility matrix, the query top-K lists in the COO form the hot path
returns, and the user top-K lists, and returns the completed matrix (qrlsh_predict)."""
import numpy as np
import torch

from . import _lib
from .ops import _ptr, _stream

QUERY_WEIGHT = 0.6   # recommender.py:32-34
USER_WEIGHT = 0.4
DEFAULT_MEAN = 60


MAX_NEIGHBOURS = 64   # PRED_MAXK in csrc/predict.hip


def fill_predictions(ratings, q_src, q_dst, q_milli, user_sims, query_weight=QUERY_WEIGHT,
                     user_weight=USER_WEIGHT, default_mean=DEFAULT_MEAN, device="cuda", sum_order="pairwise",
                     transpose_lists=True):
    """ratings: (nu, nq) integer array / tensor, 0 = missing.
    q_src/q_dst/q_milli: the hot path's top-K COO (sorted by src; value = milli / 1000).
    user_sims: {u: {'indexes', 'values'}} as compute_userSimilarities returns it.
    sum_order: "pairwise" (numpy's np.sum order: the reference as plain Python, what the fixtures pin) or
    "sequential" (numba's nopython np.sum: the reference where numba is installed; unpinned).
    transpose_lists: give the library a workspace and the longest list (one extra read-back), so that it runs its
    tile form (64 users x 16 queries per workgroup over a byte copy of the matrix) or, for ratings outside 0 .. 255,
    its row form; False: one thread per cell over the CSR lists.  Same results.
    -> int32 device tensor (nu, nq): finalPredictions of recommender.py:301-331.
    Raises ValueError when a neighbour list is longer than 64 (K = round(log_1.5 n) stays below 52 for any
    n < 1e9; only an overridden max_candidates gets there)."""
    if sum_order not in ("pairwise", "sequential"):
        raise ValueError("sum_order must be 'pairwise' or 'sequential'")
    lib = _lib.load()
    r = torch.as_tensor(np.ascontiguousarray(np.asarray(ratings), dtype=np.int32)) if not isinstance(ratings, torch.Tensor) else ratings.to(torch.int32)
    r = r.to(device).contiguous()
    nu, nq = r.shape
    q_src = q_src.to(device)
    counts = torch.bincount(q_src.to(torch.int64), minlength=nq)
    q_off = torch.zeros((nq + 1,), dtype=torch.int64, device=device)
    torch.cumsum(counts, dim=0, out=q_off[1:])
    q_idx = q_dst.to(device).to(torch.int32).contiguous()
    q_val = (q_milli.to(device).to(torch.float64) / 1000.0).contiguous()
    ku = max((len(user_sims[u]["indexes"]) for u in user_sims), default=0)
    if ku > MAX_NEIGHBOURS:
        raise ValueError("a user has %d neighbours; the prediction kernel handles at most %d" % (ku, MAX_NEIGHBOURS))
    ui = np.full((nu, max(ku, 1)), -1, dtype=np.int32)
    uv = np.zeros((nu, max(ku, 1)), dtype=np.float64)
    for u in range(nu):
        if u in user_sims:
            n = len(user_sims[u]["indexes"])
            ui[u, :n] = user_sims[u]["indexes"]
            uv[u, :n] = user_sims[u]["values"]
    u_idx = torch.from_numpy(ui).to(device)
    u_val = torch.from_numpy(uv).to(device)
    out = torch.empty((nu, nq), dtype=torch.int32, device=device)
    too_long = torch.zeros((1,), dtype=torch.int32, device=device)
    # the longest query list (one read-back): the kernel sweeps the lists transposed to [kq][nq]
    kq = int(counts.max().item()) if (nq and transpose_lists) else 0
    if kq > MAX_NEIGHBOURS:
        raise ValueError("a query has more than %d neighbours (%d; max_candidates overridden?); the prediction kernel "
                         "handles at most %d" % (MAX_NEIGHBOURS, kq, MAX_NEIGHBOURS))
    ws = torch.empty((max(int(lib.qrlsh_predict_workspace_bytes(nu, nq, kq)), 16),), dtype=torch.uint8, device=device)
    _lib.check(lib.qrlsh_predict(_ptr(r), nu, nq, _ptr(q_off), _ptr(q_idx), _ptr(q_val), _ptr(u_idx), _ptr(u_val),
                                 ui.shape[1] if ku else 0, float(query_weight), float(user_weight), float(default_mean),
                                 _lib.SUM_SEQUENTIAL if sum_order == "sequential" else _lib.SUM_PAIRWISE,
                                 _ptr(out), _ptr(too_long), kq, _ptr(ws) if kq else None, ws.numel(), _stream()))
    if int(too_long.item()):
        raise ValueError("a query has more than %d neighbours (max_candidates overridden?); the prediction kernel "
                         "handles at most %d" % (MAX_NEIGHBOURS, MAX_NEIGHBOURS))
    return out
