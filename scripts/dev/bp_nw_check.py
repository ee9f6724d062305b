"""dev: emulation of the bit-parallel Needleman-Wunsch decisions of k_seg_vote_bp (dentist_amd/csrc/dh_consensus.hip) against a
full-matrix NW with the traceback rule of util/string.d:775-831 (smallest neighbour; diagonal > insertion > deletion)."""
import random
M64 = (1 << 64) - 1
def scalar_ops(ref, qry):
    rl, ql = len(ref), len(qry)
    F = [[0] * (ql + 1) for _ in range(rl + 1)]
    for j in range(ql + 1): F[0][j] = j
    for i in range(rl + 1): F[i][0] = i
    for i in range(1, rl + 1):
        for j in range(1, ql + 1):
            F[i][j] = min(F[i-1][j-1] + (ref[i-1] != qry[j-1]), F[i-1][j] + 1, F[i][j-1] + 1)
    i, j, ops = rl, ql, []
    while i > 0 and j > 0:
        diag, up, left = F[i-1][j-1], F[i-1][j], F[i][j-1]
        op = 0 if (diag <= left and diag <= up) else (2 if left <= up else 1)
        ops.append(op)
        if op == 0: i -= 1; j -= 1
        elif op == 2: j -= 1
        else: i -= 1
    ops += [1] * i + [2] * j
    return ops, F[rl][ql]

def bp_ops(ref, qry, NW):
    W = 64 * NW; MASK = (1 << W) - 1; HALF = 32 * NW
    rl, ql = len(ref), len(qry)
    planes = [0, 0, 0]
    for j0, c in enumerate(qry):
        for b in range(3):
            planes[b] |= ((c >> b) & 1) << (HALF + j0)
    Pv = (MASK >> HALF) << HALF; Mv = ~Pv & MASK
    lv = (MASK >> (HALF + 1)) << (HALF + 1)
    dec = {}
    for i in range(1, rl + 1):
        rc = ref[i-1]
        top = lv >> (W - 1)
        lv = (lv >> 1) | (top << (W - 1))
        win = [(planes[b] >> (i - 1)) & MASK for b in range(3)]
        x = [MASK if (rc >> b) & 1 else 0 for b in range(3)]
        Eq = ~((win[0] ^ x[0]) | (win[1] ^ x[1]) | (win[2] ^ x[2])) & MASK & lv
        D0 = ((((Eq & Pv) + Pv) & MASK) ^ Pv) | Eq | Mv
        HP = (Mv | ~(D0 | Pv)) & MASK; HN = Pv & D0
        HPs = (HP << 1) & MASK; HNs = (HN << 1) & MASK
        A = (~HNs & MASK) | 1
        B = ~(D0 & HP) & MASK
        t2 = HPs & HP
        t1 = (HPs & ~HP & ~HN) | (~HPs & ~HNs & HP)
        L = (~t2 & ~(t1 & D0)) & MASK & ~1
        dec[i] = (A & B, L)
        Xv = D0 >> 1
        Pv = (HN | ~(Xv | HP)) & MASK; Mv = HP & Xv
    i, j, ops = rl, ql, []
    while i > 0 and j > 0:
        R = j - i + HALF
        assert 0 <= R < W
        z, l = dec[i]
        op = 0 if (z >> R) & 1 else (2 if (l >> R) & 1 else 1)
        ops.append(op)
        if op == 0: i -= 1; j -= 1
        elif op == 2: j -= 1
        else: i -= 1
    ops += [1] * i + [2] * j
    return ops

random.seed(1)
bad = 0; n = 0
for trial in range(3000):
    rl = random.randint(1, 126)
    ref = [random.randint(0, 3) for _ in range(rl)]
    err = random.choice([0.02, 0.1, 0.2, 0.3])
    qry = []
    for c in ref:
        r = random.random()
        if r < err / 3: continue
        if r < 2 * err / 3: qry.append(random.randint(0, 3)); continue
        if r < err: qry.append(c); qry.append(random.randint(0, 3)); continue
        qry.append(c)
    if random.random() < 0.05: qry = []
    if random.random() < 0.05: qry[random.randrange(len(qry) + 1):0] = [4]
    ops, d = scalar_ops(ref, qry)
    for NW in (1, 2):
        if d + 1 <= 32 * NW - 1 and abs(len(ref) - len(qry)) < d + 1:
            n += 1
            o2 = bp_ops(ref, qry, NW)
            if o2 != ops:
                bad += 1
                if bad < 4: print("MISMATCH NW", NW, "rl", rl, "ql", len(qry), "d", d)
print("checked", n, "bad", bad)
