"""oracle/toppin.py -- TEST INFRASTRUCTURE ONLY.

The metric's own clause, "top-10 score bit-exact", pinned to the oracle at any database size: a top list is r
(score, position in the length-sorted database) pairs per query; the scalar restatement (sw_oracle.c::osw_oracle_sw_scalar,
the exact unbounded score every host path of the reference reports, HybridSearch.c:1618-1874 / FPGAsearch.c:377-507)
is run on exactly those pairs and on every planted homolog of the synthetic database (positions known from
synth.DatabasePlan), and a planted copy that outscores a list's last entry must be in the list
(order of the reference's sort_scores, utils.c:3-86: descending score, ties by descending index).

Used by tests/ (-m gpu: the device lists and tables; CPU: the committed single-GPU reference runs) and by bench.py's
cpu_baseline leg (`top10_equals_oracle`).  Never by anything under oswald_amd/.
"""
from __future__ import annotations

import numpy as np


def pin_top_list(pyoracle, plan, order, queries, sm, go, ge, top_scores, top_pos, score_at=None, max_planted=480, max_planted_cells=4.0e9):
    """plan: synth.DatabasePlan; order: sorted position -> generation id (stable argsort of plan.lengths);
    top_scores / top_pos: [nq][r] (pos < 0 or score < 0: empty slot); score_at(q, pos): the search's score of
    (query q, sorted position pos), or None when only the list is at hand.
    Returns a dict of counts and the first few mismatches; ["ok"] says whether everything held."""
    top_scores = np.asarray(top_scores).astype(np.int64)
    top_pos = np.asarray(top_pos).astype(np.int64)
    order = np.asarray(order, dtype=np.int64)
    nq, r = top_scores.shape
    bad = []
    pairs = 0
    for q in range(nq):
        key_prev = None
        for j in range(r):
            pos, sc = int(top_pos[q, j]), int(top_scores[q, j])
            if pos < 0 or sc < 0:
                continue
            want = pyoracle.sw_scalar(queries[q], plan.residues_of([order[pos]]), sm, go, ge)
            pairs += 1
            if want != sc:
                bad.append(("list", q, j, pos, sc, want))
            key = (sc << 32) | pos
            if key_prev is not None and key >= key_prev:
                bad.append(("order", q, j, pos, sc, None))
            key_prev = key
    planted = absent = 0
    if plan.planted_query:
        inv = np.empty(len(order), dtype=np.int64)
        inv[order] = np.arange(len(order), dtype=np.int64)
        per_query = {}
        for idx, qi in plan.planted_query.items():
            per_query.setdefault(qi, []).append(idx)
        budget = max(1, max_planted // max(1, len(per_query)))
        # (a bounded check also for sets of very long queries: a copy costs about m^2 cells of scalar work)
        cells_per_round = sum(float(len(queries[qi])) ** 2 for qi in per_query)
        budget = max(1, min(budget, int(max_planted_cells / max(cells_per_round, 1.0))))
        for qi, ids in sorted(per_query.items()):
            for idx in ids[:budget]:           # (in copy order: 5 %, 10 %, ... substitutions -- the best scorers first)
                pos = int(inv[idx])
                want = pyoracle.sw_scalar(queries[qi], plan.planted[idx], sm, go, ge)
                planted += 1
                if score_at is not None:
                    got = int(score_at(qi, pos))
                    if got != want:
                        bad.append(("planted", qi, None, pos, got, want))
                listed = bool((top_pos[qi] == pos).any())
                last_valid = np.flatnonzero(top_pos[qi] >= 0)
                full = len(last_valid) == r
                must = not full or ((want << 32) | pos) > ((int(top_scores[qi, last_valid[-1]]) << 32) | int(top_pos[qi, last_valid[-1]]))
                if must and not listed:
                    absent += 1
                    bad.append(("absent", qi, None, pos, None, want))
    return {"ok": not bad, "list_pairs": pairs, "planted": planted, "planted_absent": absent, "mismatches": len(bad), "first": bad[:5]}
