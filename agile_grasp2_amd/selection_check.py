"""Harness-side checker (tests, bench.py's N > 1 self-validation) -- not part of the product path.

Comparison of a selection (score threshold + top-k, grasp_detector.cpp:200-207,239-252) made from LeNet
scores that agree only within a tolerance.

The GPU's scores g_i and the oracle's o_i differ by at most `tol` (fp32, different summation order), so a
record whose score lies within 2 tol of the threshold, or within 2 tol of the k-th best score, may
legitimately be selected by one side and not by the other.  Instead of skipping the whole comparison when
such a record exists (what the tests did up to round 3), the records are sorted into three classes from
the ORACLE's scores alone:

  must_in    o_i >= thr + 2 tol, and fewer than k other records could outrank it on the GPU
             (#{j != i : o_j >= thr - 2 tol and o_j > o_i - 2 tol} <= k - 1)
  must_out   o_i < thr - 2 tol, or at least k records certainly outrank it
             (#{j : o_j >= thr + 2 tol and o_j > o_i + 2 tol} >= k)
  uncertain  the rest -- at most `max_uncertain` of them, asserted

and the GPU's selection has to contain every must_in record, no must_out record, have the length the
classes allow, and list its records in an order the oracle's scores allow (descending up to 2 tol).
"""
from __future__ import annotations

import numpy as np


def _key(h):
    return (int(h["sample_slot"]), int(h["orientation"]))


def check_selection(got_sel, oracle_all, thr: float, k: int, tol: float, max_uncertain: int = 12,
                    tag="") -> dict:
    """got_sel: the GPU's selected records (in its order); oracle_all: every scored record of the oracle."""
    o = np.asarray(oracle_all["score"], dtype=np.float64)
    keys = [_key(h) for h in oracle_all]
    assert len(set(keys)) == len(keys), tag
    n = len(o)
    kk = n if k < 0 else k
    maybe = o >= thr - 2 * tol            # could pass the threshold on the GPU
    sure = o >= thr + 2 * tol             # certainly passes
    srt_maybe = np.sort(o[maybe])
    srt_sure = np.sort(o[sure])
    # records that could outrank i: o_j > o_i - 2 tol among `maybe` (minus i itself when it is one of them)
    could = len(srt_maybe) - np.searchsorted(srt_maybe, o - 2 * tol, side="right")
    if tol > 0:
        could = could - maybe.astype(int)   # (i itself is among them)
    # records that certainly outrank i: o_j > o_i + 2 tol among `sure`
    certainly = len(srt_sure) - np.searchsorted(srt_sure, o + 2 * tol, side="right")
    must_in = sure & (could <= kk - 1)
    must_out = (~maybe) | (certainly >= kk)
    uncertain = ~(must_in | must_out)
    assert not (must_in & must_out).any(), tag
    assert int(uncertain.sum()) <= max_uncertain, (tag, int(uncertain.sum()), "records within tolerance of the threshold / cut")
    idx_of = {key: i for i, key in enumerate(keys)}
    got_keys = [_key(h) for h in got_sel]
    assert len(set(got_keys)) == len(got_keys), (tag, "a record was selected twice")
    got_idx = []
    for key in got_keys:
        assert key in idx_of, (tag, key, "selected record is not among the oracle's scored records")
        got_idx.append(idx_of[key])
    got_idx = np.asarray(got_idx, dtype=np.int64)
    got_mask = np.zeros(n, dtype=bool)
    got_mask[got_idx] = True
    assert not (got_mask & must_out).any(), (tag, "selected a record the oracle's scores rule out",
                                             [keys[i] for i in np.flatnonzero(got_mask & must_out)][:5])
    assert (got_mask | ~must_in).all(), (tag, "missed a record the oracle's scores demand",
                                         [keys[i] for i in np.flatnonzero(must_in & ~got_mask)][:5])
    lo = int(must_in.sum())
    hi = min(kk, int((must_in | uncertain).sum()))
    assert lo <= len(got_idx) <= hi, (tag, len(got_idx), lo, hi)
    # order: by score, descending -- as far as the oracle's scores can tell
    if len(got_idx) > 1:
        og = o[got_idx]
        assert (og[:-1] >= og[1:] - 2 * tol).all(), (tag, "selection not in descending score order")
    return {"must_in": lo, "uncertain": int(uncertain.sum()), "selected": len(got_idx)}
