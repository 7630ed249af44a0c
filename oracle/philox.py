# coding=utf-8
"""ORACLE — test infrastructure, NOT product code.

numpy restatement of the counter-based RNG the HIP path uses for its *fused* (on-device) random draws, so the
device streams can be checked bit-for-bit (integers) / to fp32 rounding (normals).

The reference draws its randomness from torch's global generators (models/DCCF.py:72 ``torch.randint`` on the CPU
generator; models/DCCF.py:87 ``normal_`` on the CUDA generator; data_processor/DataProcessor.py:497-504
``np.random.randint``) — streams that no other implementation can reproduce.  Parity with the reference is
therefore established with *injected* draws (tests/golden); this module pins the replacement streams:

  Philox4x32-10 (Salmon et al., SC'11; the same generator family torch's CUDA backend uses), key = (seed_lo,
  seed_hi ^ STREAM), counter = (c0, c1, step_lo, step_hi):

    STREAM_CAND   c0 = batch row n, c1 = g        -> 4 candidates s = 4g..4g+3, item = mulhi(u32, item_num)
    STREAM_NOISE  c0 = flat row l,  c1 = 32*(f//128) + f%32 -> the 4 normals of f%128//32 = 0..3 (Box-Muller on
                  (x0,x1) -> o=0 (cos), o=1 (sin); (x2,x3) -> o=2, o=3), multiplied by std
    STREAM_DROP   c0 = l//4, c1 = d             -> row l%4 of the group, column d, is dropped iff u32 < floor(p * 2^32)
    STREAM_NEG    c0 = uid, c1 = draw//4, c2 = epoch -> draw%4-th u32, item = mulhi(u32, item_num)
    STREAM_EVALNEG  same counters with c2 = tag (1 validation, 2 test)
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

STREAM_CAND, STREAM_NOISE, STREAM_DROP, STREAM_NEG, STREAM_INIT, STREAM_EVALNEG = 1, 2, 3, 4, 5, 6


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    """Vectorised Philox4x32-R.  Counters are broadcastable integer arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & MASK for c in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(rounds):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return [c.astype(np.uint32) for c in (c0, c1, c2, c3)]


def _key(seed, stream):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return seed & 0xFFFFFFFF, ((seed >> 32) ^ stream) & 0xFFFFFFFF


def mulhi(x, n):
    return ((x.astype(np.uint64) * np.uint64(n)) >> np.uint64(32)).astype(np.int64)


def candidates(seed, step, N, S, item_num):
    """Device restatement of models/DCCF.py:72 (uniform candidates, no filtering).  -> int64 [N, S]"""
    k0, k1 = _key(seed, STREAM_CAND)
    G = (S + 3) // 4
    n = np.arange(N)[:, None]
    g = np.arange(G)[None, :]
    xs = philox4x32(n, g, step & 0xFFFFFFFF, (step >> 32) & 0xFFFFFFFF, k0, k1)
    out = np.stack([mulhi(x, item_num) for x in xs], axis=2).reshape(N, G * 4)
    return out[:, :S]


def _u01(x):
    """23-bit uniform strictly inside (0,1): ((x >> 9) + 0.5) * 2^-23 — exact in fp32."""
    return ((x >> np.uint32(9)).astype(np.float64) + 0.5) * (2.0 ** -23)


def noise(seed, step, L, F, std):
    """Device restatement of models/DCCF.py:87: iid N(0, std^2) per (l, f).  -> float32 [L, F]"""
    k0, k1 = _key(seed, STREAM_NOISE)
    f = np.arange(F)
    c1 = 32 * (f // 128) + (f % 32)
    o = (f % 128) // 32
    uc1, inv = np.unique(c1, return_inverse=True)
    l = np.arange(L)[:, None]
    xs = philox4x32(l, uc1[None, :], step & 0xFFFFFFFF, (step >> 32) & 0xFFFFFFFF, k0, k1)
    u = [_u01(x) for x in xs]
    rA = np.sqrt(-2.0 * np.log(u[0]))
    rB = np.sqrt(-2.0 * np.log(u[2]))
    z = np.stack([rA * np.cos(2 * np.pi * u[1]), rA * np.sin(2 * np.pi * u[1]),
                  rB * np.cos(2 * np.pi * u[3]), rB * np.sin(2 * np.pi * u[3])], axis=0)     # [4, L, n_c1]
    out = z[o[None, :], np.arange(L)[:, None], inv[None, :]]
    return (out * std).astype(np.float32)


def drop_threshold(p):
    return int(min(max(p, 0.0) * 4294967296.0, 4294967295.0))


def dropout_keep(seed, step, L, D, p, layer=0):
    """Device restatement of models/DCCF.py:94 (Bernoulli keep mask) of mlp layer `layer` (the layer index sits in bits
    16.. of the column counter: mlp.0 draws as it always did).  -> uint8 [L, D] (1 = kept)"""
    if p <= 0.0:
        return np.ones((L, D), dtype=np.uint8)
    k0, k1 = _key(seed, STREAM_DROP)
    G = (L + 3) // 4
    g = np.arange(G)[:, None]
    d = np.arange(D)[None, :] | (int(layer) << 16)
    xs = philox4x32(g, d, step & 0xFFFFFFFF, (step >> 32) & 0xFFFFFFFF, k0, k1)      # 4 x [G, D]: word w = row 4g+w
    x = np.stack(xs, axis=1).reshape(G * 4, D)[:L]
    return (x >= np.uint32(drop_threshold(p))).astype(np.uint8)


def train_negatives(seed, epoch, uids, item_num, hist_indptr, hist_items):
    """Device restatement of data_processor/DataProcessor.py:446-524 for train=True, neg_n=1: one negative per train
    row, uniform over items, rejected while in the user's train history or among the negatives already drawn for
    that user this epoch (``tmp_history_dict`` persists over the epoch, :516-517).  Rows of one user are served in
    row order; draw j of user u is the (j%4)-th word of Philox(c0=u, c1=j//4, c2=epoch).  When fewer than 20 % of
    the items remain the reference switches to ``np.random.choice(range(1, item_num))`` (:490-493) — item 0 is then
    never drawn; mirrored here by rejecting item 0 in that regime.  -> int64 [len(uids)]"""
    k0, k1 = _key(seed, STREAM_NEG)
    uids = np.asarray(uids, dtype=np.int64)
    out = np.empty(len(uids), dtype=np.int64)
    order = np.argsort(uids, kind='stable')
    su = uids[order]
    starts = np.flatnonzero(np.r_[True, su[1:] != su[:-1]])
    ends = np.r_[starts[1:], len(su)]
    for s, e in zip(starts, ends):
        u = int(su[s])
        hist = set(hist_items[hist_indptr[u]:hist_indptr[u + 1]].tolist())
        drawn = []
        j = 0
        for r in range(s, e):
            remain = item_num - len(hist) - len(drawn)
            low = (1.0 * remain / item_num) < 0.2
            # low regime: the reference samples from range(1, item_num), so item 0 only counts if it is excluded already
            admissible = remain - (1 if low and 0 not in hist and 0 not in drawn else 0)
            if admissible < 1:          # the reference asserts / np.random.choice raises; the device writes -1
                out[order[r]] = -1
                continue
            while True:
                xs = philox4x32(u, j // 4, epoch, 0, k0, k1)
                it = int(mulhi(np.asarray(xs[j % 4]).reshape(1), item_num)[0])
                j += 1
                if it in hist or it in drawn or (low and it == 0):
                    continue
                break
            drawn.append(it)
            out[order[r]] = it
    return out


def eval_negatives(seed, tag, users, item_num, hist_indptr, hist_items, neg_n):
    """Device restatement of data_processor/DataProcessor.py:408-444,446-524 for train=False: neg_n negatives per distinct
    user, uniform over items, outside the user's train + validation/test history and distinct.  Draws of user u are the
    words of Philox(c0=u, c1=j//4, c2=tag) on STREAM_EVALNEG, consumed in order: draw j is accepted iff its item is
    admissible and not accepted before; the first neg_n accepted draws, in draw order, are the result.  When fewer than
    20 % of the items remain the reference never draws item 0 (:490-493), mirrored by rejecting item 0 in that regime.
    -> int64 [len(users), neg_n] (-1 where fewer than neg_n admissible items exist)"""
    k0, k1 = _key(seed, STREAM_EVALNEG)
    out = np.full((len(users), neg_n), -1, dtype=np.int64)
    for w, u in enumerate(np.asarray(users, dtype=np.int64)):
        hist = set(hist_items[hist_indptr[u]:hist_indptr[u + 1]].tolist())
        remain = item_num - len(hist)
        low = 5 * remain < item_num
        if low and 0 not in hist:
            remain -= 1
        if remain < neg_n:
            continue
        acc, seen, j = [], set(), 0
        while len(acc) < neg_n:
            xs = philox4x32(int(u), np.arange(j // 4, j // 4 + 64), tag, 0, k0, k1)       # 256 draws at a time
            cand = np.stack([mulhi(x, item_num) for x in xs], axis=1).reshape(-1)
            for c in cand.tolist():
                j += 1
                if (low and c == 0) or c in hist or c in seen:
                    continue
                seen.add(c)
                acc.append(c)
                if len(acc) == neg_n:
                    break
        out[w] = acc
    return out


# ------------------------------------------------------------------------------------------------ the epoch's permutation
EPOCH_PERM_ROUNDS = 12


def _mix64(x):
    """splitmix64's finalizer (exact 64-bit wrap)."""
    M = (1 << 64) - 1
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
    return x ^ (x >> 31)


def epoch_perm_keys(seed, epoch):
    """The round keys dccf_build_epoch_batches derives from (seed, epoch) (dccf_amd/csrc/mf_kernels.hip): splitmix64 steps."""
    M = (1 << 64) - 1
    z = (int(seed) * 0x9E3779B97F4A7C15 + int(epoch) * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) & M
    ks = []
    for _ in range(EPOCH_PERM_ROUNDS):
        z = (z + 0x9E3779B97F4A7C15) & M
        ks.append(_mix64(z))
    return ks


def epoch_perm(seed, epoch, n):
    """perm[i] for i < n: the keyed bijection k_epoch_batches evaluates per index instead of torch.randperm — plays
    shuffle_in_unison_scary's role (src/utils/utils.py:82-92).  A 12-round alternating Feistel network on b-bit integers
    (b = bits of n - 1, at least 2; halves of ceil(b/2) and floor(b/2) bits; round function = splitmix64's finalizer of the
    right half + the round key), cycle-walked until the value is below n.  (Round 2's four rounds of multiply-add-xorshift were
    measurably non-uniform on small domains — tests/test_oracle_golden.py holds this one to a chi-square.)"""
    M = (1 << 64) - 1
    ks = epoch_perm_keys(seed, epoch)
    b = 2
    while (1 << b) < n:
        b += 1
    lb = b // 2
    hb = b - lb
    out = np.empty(n, dtype=np.int64)
    for i in range(n):
        x = i
        while True:
            wl, wr = hb, lb
            L, R = x >> wr, x & ((1 << wr) - 1)
            for r in range(EPOCH_PERM_ROUNDS):
                F = _mix64((R + ks[r]) & M)
                L, R, wl, wr = R, L ^ (F & ((1 << wl) - 1)), wr, wl
            x = (L << wr) | R
            if x < n:
                break
        out[i] = x
    return out
