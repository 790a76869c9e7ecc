"""LDS bank conflicts of the dense conv kernels' access patterns by the banking rules of MI355X_MICROARCH.md (LDS table): a wave64
access is served in fixed lane groups, one LDS cycle per group when conflict-free, each extra distinct address on a busy bank adds a cycle.
CPU only. Prints extra cycles / lane-group cycles per pattern:

  * conv2d_i8r_kernel: code stores (ds_write_b32) with and without the row rotation rho, B-fragment reads, scale reads as ds_read_b128
    and as the ds_read2_b32 pairs the compiler made of them through a float pointer (the 20 % of profiles/r04);
  * conv2d_f16s_kernel: fragment reads / split stores on 144-byte padded rows against 128-byte XOR-swizzled rows.
"""
R128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
        list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
H32 = [list(range(0, 32)), list(range(32, 64))]
C16 = [list(range(16 * i, 16 * i + 16)) for i in range(4)]
C8 = [list(range(8 * i, 8 * i + 8)) for i in range(8)]


def conflicts(addr_of_lane, groups, nbanks, width):
    extra = 0
    for grp in groups:
        bank_addrs = {}
        for l in grp:
            a = addr_of_lane(l)
            for d in range(width // 4):
                bank_addrs.setdefault(((a // 4) + d) % nbanks, set()).add(a + 4 * d)
        extra += max(len(v) for v in bank_addrs.values()) - 1
    return extra, len(groups)


def f(r):
    return (0x1320 >> (((r >> 2) & 3) * 4)) & 3


def i8r():
    rho = lambda p: (p & ~3) | ((p + (p >> 2)) & 3)
    ident = lambda p: p
    for TJ in (2, 1):
        PQ, BN = 2 * TJ, 32 * TJ
        for name, m in (("rho", rho), ("identity", ident)):
            e = c = 0
            for wave in range(4):
                for i in range(PQ):
                    def a(l):
                        tid = wave * 64 + l
                        cq, pq = tid & 15, tid >> 4
                        row = m(PQ * pq + i)
                        return row * 64 + ((cq >> 2) ^ f(row)) * 16 + (cq & 3) * 4
                    x, y = conflicts(a, H32, 32, 4)
                    e += x; c += y
            print("i8r PQ=%d code stores (ds_write_b32), rows by %-8s: %d extra / %d cycles" % (PQ, name, e, c))
        e = c = 0
        for wn in range(2):
            for j in range(TJ):
                def a(l):
                    row = rho(wn * (BN // 2) + j * 16 + (l & 15))
                    return row * 64 + ((l >> 4) ^ f(row)) * 16
                x, y = conflicts(a, R128, 64, 16)
                e += x; c += y
        print("i8r PQ=%d B-fragment reads (ds_read_b128): %d extra / %d cycles" % (PQ, e, c))
    print("i8r scale reads as ds_read_b128: %d extra / %d cycles" % conflicts(lambda l: (l & 15) * 16, R128, 64, 16))
    e = c = 0
    for half in range(2):           # ds_read2_b32 with offsets (0, 1): two dwords per lane, banks mod 32 over 32 lanes
        for d in range(2):
            x, y = conflicts(lambda l: (l & 15) * 16 + half * 8 + d * 4, H32, 32, 4)
            e += x; c += y
    print("i8r scale reads as two ds_read2_b32: %d extra / %d cycles" % (e, c))


def f16s():
    for name, off in (("144-byte padded rows", lambda row, ch: row * 144 + ch * 16),
                      ("128-byte rows, chunk ^ ((row >> 1) & 7)", lambda row, ch: row * 128 + ((ch ^ ((row >> 1) & 7)) * 16))):
        er = cr = ew = cw = 0
        for lo in (0, 4):
            x, y = conflicts(lambda l: off(l & 15, lo + (l >> 4)), R128, 64, 16)
            er += x; cr += y
            for wave in range(4):
                def a(l):
                    item = wave * 64 + l
                    return off(item // 8, lo + ((item % 8) >> 1)) + (item & 1) * 8
                x, y = conflicts(a, C16, 32, 8)
                ew += x; cw += y
        print("f16s %-42s fragment reads %d extra / %d, split stores (ds_write_b64) %d extra / %d" % (name + ":", er, cr, ew, cw))


if __name__ == "__main__":
    i8r()
    f16s()
