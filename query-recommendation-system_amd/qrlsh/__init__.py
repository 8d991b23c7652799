"""qrlsh -- MI355X-native MinHash-LSH candidate generation and pair scoring.

Host layer above the C ABI of include/qrlsh.h: torch tensors are used only as device
buffers (and torch.distributed for the multi-GPU exchange); all compute is in
libqrlsh.so (hand-written HIP for gfx950).
"""
from . import _lib  # noqa: F401
from .ops import (  # noqa: F401
    perm_table,
    legacy_permutations,
    minhash,
    can_compact,
    sig_to_int32,
    band_keys,
    row_norms,
    sort_u64,
    bucket_sort,
    emit_pairs,
    emit_pairs_fast,
    emit_pairs_any,
    unique_sorted,
    row_unique,
    unique_pairs,
    candidate_pairs,
    score_pairs,
    topk_edges,
    id_bits_for,
)
from .pipeline import (  # noqa: F401
    select_bands,
    max_candidates,
    query_similarities,
    sims_to_dict,
    HotPathResult,
)
from .synth import synth_csr, poisson_cdf_u32  # noqa: F401
from .answers import build_answer_index, encode_queries, answer_sets, AnswerIndex  # noqa: F401

__all__ = [n for n in dir() if not n.startswith("_")]
