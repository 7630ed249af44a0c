# coding=utf-8
"""VERDICT r1 item 9 — is there anything for an LDS-staged duplicate-index reduction to merge in the DCCF gradient scatter?

The backward's dx role (dccf_kernels.hip::bwd_dx_role) emits ONE 256-B atomic row add per (batch row n, candidate s) into
gV[cand[n][s]] (the A noise copies are already summed in registers), role 0 one per batch row into gU[u(n)].  A workgroup walks
the batch rows n = 4 x + wave + 4 gx k (x = blockIdx.x).  Rows that could be merged before the atomic are equal ids INSIDE one
workgroup's rows.  This script counts them on the bench's data shape (Electronics-sized tables, Zipf(0.8) positives, uniform
negatives and candidates as torch.randint draws them, src/models/DCCF.py:72) — CPU only, no GPU needed.

    python scripts/dup_analysis.py            -> one JSON line (committed as profiles/r02_dup_analysis.json)
"""
import json

import numpy as np

U, I, S = 192403, 63001, 10


def batch(rng, B):
    w = 1.0 / np.power(np.arange(1, I + 1, dtype=np.float64), 0.8)
    w /= w.sum()
    u = rng.randint(0, U, B)
    pos = rng.choice(I, size=B, p=w)
    neg = rng.randint(0, I, B)
    uid = np.concatenate([u, u])
    it0 = np.concatenate([pos, neg])
    cand = np.concatenate([it0[:, None], rng.randint(0, I, (2 * B, S))], 1)       # [N, S+1]
    return uid, cand


def analyse(B, roles=8, trials=20):
    rng = np.random.RandomState(B)
    N = 2 * B
    gx = max(1, min((N + 3) // 4, max(1, (1024 if N >= 2048 else 256) // roles)))
    out = {'item_rows_total': 0, 'item_dups_in_workgroup': 0, 'item_dups_in_batch': 0, 'user_rows_total': 0,
           'user_dups_in_workgroup': 0, 'user_dups_in_batch': 0}
    for _ in range(trials):
        uid, cand = batch(rng, B)
        n = np.arange(N)
        wg = (n // 4) % gx                       # n = 4 x + wave + 4 gx k
        out['item_rows_total'] += cand.size
        out['user_rows_total'] += N
        out['item_dups_in_batch'] += cand.size - len(np.unique(cand))
        out['user_dups_in_batch'] += N - len(np.unique(uid))
        for x in range(gx):
            c = cand[wg == x].reshape(-1)
            out['item_dups_in_workgroup'] += c.size - len(np.unique(c))
            uu = uid[wg == x]
            out['user_dups_in_workgroup'] += uu.size - len(np.unique(uu))
    r = {'batch_size': B, 'row_splits': gx, 'rows_per_workgroup': N / gx}
    r['item_atomics_saved_by_workgroup_reduction'] = round(out['item_dups_in_workgroup'] / out['item_rows_total'], 5)
    r['item_atomics_saved_by_whole_batch_reduction'] = round(out['item_dups_in_batch'] / out['item_rows_total'], 5)
    r['user_atomics_saved_by_workgroup_reduction'] = round(out['user_dups_in_workgroup'] / out['user_rows_total'], 5)
    r['user_atomics_saved_by_whole_batch_reduction'] = round(out['user_dups_in_batch'] / out['user_rows_total'], 5)
    # share of ALL gradient-row atomics of the step (items: N (S+1), users: N) a workgroup-level reduction would remove
    tot = out['item_rows_total'] + out['user_rows_total']
    r['all_row_atomics_saved_by_workgroup_reduction'] = round((out['item_dups_in_workgroup'] + out['user_dups_in_workgroup']) / tot, 5)
    return r


if __name__ == '__main__':
    print(json.dumps({'what': 'duplicate gradient-row targets inside one backward workgroup (DCCF, Electronics shape)',
                      'results': [analyse(B) for B in (128, 512, 4096)]}))
