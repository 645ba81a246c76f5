"""Dev check: LDS layout of conv_patch.hip.h (round 3: dz-pure fragments, row pitch 4128 / 2080 bytes) is bank-conflict free
for every ds_read_b128 of its K loop.  Fragment = column pair f of the tile's two pooled rows: rows 0-3 = (row 0, col 2f),
4-7 = (row 1, col 2f), 8-11 = (row 1, col 2f+1), 12-15 = (row 0, col 2f+1); row e of a window = (dy, dx); LDS byte address
= (2 ypl + dy + ky) LP + (2 xp + dx + kx) 64 + 16 fk."""
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
COMP = [(0, 0), (1, 0), (1, 1), (0, 1)]


def main():
    for name, lp, pairs in (('56 x 56 planes (conv2a)', 4128, 14), ('28 x 28 planes (conv3a, conv3b)', 2080, 7)):
        extra, reads = 0, 0
        for f in range(pairs):
            for ky in range(3):
                for kx in range(3):
                    for g in GROUPS:
                        slots = {}
                        for lane in g:
                            frow, fk = lane & 15, lane >> 4
                            ypl, xo = COMP[frow >> 2]
                            dy, dx = (frow >> 1) & 1, frow & 1
                            addr = (2 * ypl + dy + ky) * lp + (2 * (2 * f + xo) + dx + kx) * 64 + fk * 16
                            slots.setdefault((addr % 256) // 16, set()).add(addr)
                        reads += 1
                        extra += sum(len(v) - 1 for v in slots.values())
        print('conv_patch, %s: %d group reads, %d extra LDS cycles' % (name, reads, extra))
        assert extra == 0
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
