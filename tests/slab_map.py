"""An independent, pure-numpy restatement of ONE block's witness slab, written
straight from the slab-map table of SURVEY.md 8(a) / DESIGN.md (not from the
oracle's layouter).  Tests use it as a second opinion on the row order the
oracle derives by running the reference's region call order."""
import numpy as np

MIX = [[2, 3, 1, 1], [1, 2, 3, 1], [1, 1, 2, 3], [3, 1, 1, 2]]  # src/aes128.rs:228-233
RCON = [1, 2, 4, 8, 16, 32, 64, 128, 27, 54]                     # src/utils.rs:28


def key_schedule(key, sbox):
    """Returns (rk[11][16], W[96], kx[400], ky[400], kz[400], ymask, zmask) for one key."""
    rk = [list(int(v) for v in key)]
    W = list(rk[0])
    kx, ky, kz = [0] * 400, [0] * 400, [0] * 400
    ym, zm = [0] * 400, [0] * 400
    row = 0

    def put(x, y=None, z=None):
        nonlocal row
        kx[row] = x
        if y is not None:
            ky[row], ym[row] = y, 1
        if z is not None:
            kz[row], zm[row] = z, 1
        row += 1

    for rho in range(1, 11):
        p = rk[-1]
        shifted = [p[13], p[14], p[15], p[12]]
        W += shifted
        subbed = [int(sbox[v]) for v in shifted]
        for s_, b_ in zip(shifted, subbed):
            put(s_, b_)
        rc = [RCON[rho - 1], 0, 0, 0]
        W += rc
        rconned = [a ^ b for a, b in zip(subbed, rc)]
        for a, b, c in zip(subbed, rc, rconned):
            put(a, b, c)
        words = []
        nw = [a ^ b for a, b in zip(p[0:4], rconned)]
        for a, b, c in zip(p[0:4], rconned, nw):
            put(a, b, c)
        words += nw
        for i in range(1, 4):
            nn = [a ^ b for a, b in zip(p[4 * i:4 * i + 4], nw)]
            for a, b, c in zip(p[4 * i:4 * i + 4], nw, nn):
                put(a, b, c)
            nw = nn
            words += nw
        for v in words:
            put(v)
        rk.append(words)
    assert row == 400 and len(W) == 96
    return rk, W, kx, ky, kz, ym, zm


def encrypt_slab(pt, rk, sbox, mul2, mul3):
    """Returns (x[1360], y[1360], z[1360], ymask, zmask, ct[16]) for one block."""
    x, y, z = [0] * 1360, [0] * 1360, [0] * 1360
    ym, zm = [0] * 1360, [0] * 1360
    row = 0

    def put(a, b=None, c=None):
        nonlocal row
        x[row] = a
        if b is not None:
            y[row], ym[row] = b, 1
        if c is not None:
            z[row], zm[row] = c, 1
        row += 1

    pt = [int(v) for v in pt]
    for v in pt:
        put(v)
    s = []
    for i in range(16):
        s.append(pt[i] ^ rk[0][i])
        put(pt[i], rk[0][i], s[i])
    for R in range(1, 11):
        sub = [int(sbox[v]) for v in s]
        for i in range(16):
            put(s[i], sub[i])
        sh = [[sub[4 * ((w + j) % 4) + j] for j in range(4)] for w in range(4)]
        if R < 10:
            mixed = []
            for w in range(4):
                for m in range(4):
                    tmp = []
                    for t in range(4):
                        c = MIX[m][t]
                        a = sh[w][t]
                        if c == 1:
                            put(a)
                            tmp.append(a)
                        else:
                            v = int(mul2[a]) if c == 2 else int(mul3[a])
                            put(a, v)
                            tmp.append(v)
                    i1, i2 = tmp[0] ^ tmp[1], tmp[2] ^ tmp[3]
                    put(tmp[0], tmp[1], i1)
                    put(tmp[2], tmp[3], i2)
                    put(i1, i2, i1 ^ i2)
                    mixed.append(i1 ^ i2)
        else:
            mixed = [sh[w][j] for w in range(4) for j in range(4)]
        ns = []
        for i in range(16):
            ns.append(mixed[i] ^ rk[R][i])
            put(mixed[i], rk[R][i], ns[i])
        s = ns
    assert row == 1360
    return (np.array(x, np.uint8), np.array(y, np.uint8), np.array(z, np.uint8), np.array(ym, np.uint8),
            np.array(zm, np.uint8), np.array(s, np.uint8))
