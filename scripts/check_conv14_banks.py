"""Dev check: LDS layout of conv_patch14.hip.h (dz-pure fragments) is bank-conflict free for every ds_read_b128.

Fragment f (0..6) of a block tile = column xp = f of its four slots: rows 0-3 / 4-7 = the windows of the even slots u = 0, 2,
rows 8-11 / 12-15 = those of the odd slots u = 1, 3; row e of a window = (dy, dx) = (e >> 1, e & 1); all 16 rows read the
same plane.  LDS byte address = (4 u + dy + ky) 1152 + 32 (u & 1) + (2 xp + dx + kx) 64 + 16 fk."""
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
LROW = 1152


def window(f, w):
    return 2 * (w & 1) + (w >> 1), f             # (slot u, xp): w = 0, 1 -> even slots 0, 2; w = 2, 3 -> odd slots 1, 3


def main():
    extra, worst, reads = 0, 1, 0
    for f in range(7):
        for ky in range(3):
            for kx in range(3):
                for g in GROUPS:
                    slots = {}
                    for lane in g:
                        frow, fk = lane & 15, lane >> 4
                        u, xp = window(f, frow >> 2)
                        dy, dx = (frow >> 1) & 1, frow & 1
                        addr = (4 * u + dy + ky) * LROW + 32 * (u & 1) + (2 * xp + dx + kx) * 64 + 16 * fk
                        slots.setdefault((addr % 256) // 16, set()).add(addr)
                    reads += 1
                    extra += sum(len(v) - 1 for v in slots.values())
                    worst = max(worst, max(len(v) for v in slots.values()))
    print('conv_patch14: %d group reads, %d extra LDS cycles, worst %d-way' % (reads, extra, worst))
    assert extra == 0
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
