"""Dev check (CPU, no GPU): the address maps of the 14 x 14 window-pair form of wgrad_patch.hip.h.

Re-states the kernel's formulas -- LDS-DMA destinations and source-side swizzles (fetch), per-group read bases (bases),
per-step immediates (read1) -- fills a model LDS with (window, plane, row, x, 8-byte piece) tags the way the DMA would, and
checks that every fragment read of every (group, step, unit, lane) lands on the tag the filter gradient needs:
X[window, z + kz, y + ky, x + kx][channel half ct, piece pp] and dY[window, z, y, x][tile j, piece pp]; also that the 32
lanes of a half of every ds_read_b64_tr_b16 cover the 64 banks once.   python scripts/check_wgrad14_maps.py"""
import itertools

PLANE = 16 * 64
WPITCH = 6 * PLANE + 128
XBUF = 2 * WPITCH
DY_X, DY_Z = 256, 14 * 256
DY_Y = 4 * DY_Z
DYBUF = 2 * DY_Y
DY_OFF = 8 * XBUF
NG = 7


def fetch(lds, kc, gq, n_windows, col0=0):
    """model of fetch(): lds maps 8-byte-aligned address -> tag"""
    c = col0 + kc
    win0 = 2 * c
    two = win0 + 1 < n_windows
    r0, nr = (0, 4) if gq == 0 else (2 * gq + 2, 2)
    nx = nr * 8
    for t in range(nx + DYBUF // 1024):
        for lane in range(64):
            if t < nx:
                k, w, zp = t >> 3, (t >> 2) & 1, (t & 3) + 1
                r = r0 + k
                lx, lc = lane >> 2, lane & 3
                src_win = win0 + (w if two else 0)
                src_chunk = lc ^ (2 * (r & 1))                     # 16-byte chunk of the pixel's 64-byte slice
                dst = ((kc * 16 + r) & 7) * XBUF + w * WPITCH + zp * PLANE + lane * 16
                for half8 in range(2):                             # two 8-byte pieces per 16-byte chunk
                    piece = src_chunk * 2 + half8                  # 8-byte piece index 0..7 of the 64-byte slice
                    lds[dst + 8 * half8] = ('x', src_win, zp, r, lx, piece, two or w == 0)
            else:
                jj = t - nx
                ly, ychk = lane >> 3, lane & 7
                pi = jj * 8 + ly
                w, t2 = pi & 1, pi >> 1
                t3 = t2 // 14
                x, z, yp = t2 - t3 * 14, t3 & 3, t3 >> 2
                sw = (x & 1) | (yp << 1)
                src_chunk = ychk ^ (2 * sw)
                dst = DY_OFF + ((kc * NG + gq) & 1) * DYBUF + jj * 1024 + lane * 16
                valid = two or w == 0
                for half8 in range(2):
                    piece = src_chunk * 2 + half8                  # 8-byte piece 0..15 of the pixel's 128-byte slice
                    lds[dst + 8 * half8] = ('y', win0 + w if valid else None, z, 2 * gq + yp, x, piece, valid)


def check(n_windows):
    ncols = (n_windows + 1) // 2
    lds = {}
    groups = [(kc, gq) for kc in range(ncols) for gq in range(NG)]
    fetch(lds, 0, 0, n_windows)
    if len(groups) > 1:
        fetch(lds, *groups[1], n_windows)
    banks_ok = True
    for G, (kc, gq) in enumerate(groups):
        if G + 2 < len(groups):
            pass
        sa = [((kc * 16 + 2 * gq + k) & 7) * XBUF for k in range(4)]
        yb = DY_OFF + ((kc * NG + gq) & 1) * DYBUF
        for wave in range(8):
            ct, tap0 = wave & 1, wave >> 1
            for i in range(7):
                tap = tap0 + 4 * i
                if tap > 26:
                    continue
                kz, ky, kx = tap // 9, (tap // 3) % 3, tap % 3
                for s in range(7):
                    for h in range(2):
                        seen = {}
                        for lane in range(64):
                            fcol, g = lane & 15, lane >> 4
                            q, pp = fcol >> 2, fcol & 3
                            l_x, l_y, l_w, l_z = q & 1, q >> 1, g & 1, g >> 1
                            xl0 = l_w * WPITCH + l_z * PLANE + l_x * 64 + pp * 8 + 32 * ((ct ^ l_y) & 1)
                            lanepart = xl0 ^ ((ky & 1) << 5)
                            rowaddr = sa[ky + 1] if l_y else sa[ky]
                            addr = lanepart + rowaddr + kz * PLANE + kx * 64 + s * 128 + h * 2 * PLANE
                            tag = lds.get(addr)
                            win = 2 * kc + l_w
                            z, y, x = l_z + 2 * h, 2 * gq + l_y, 2 * s + l_x
                            want = ('x', win if win < n_windows else 2 * kc, z + kz, y + ky, x + kx, ct * 4 + pp, win < n_windows)
                            if z + kz in (0, 5):
                                assert tag is None, 'the halo planes of a row slot are zeroed once and never written'
                            elif win < n_windows:
                                assert tag == want, ('X', (kc, gq, wave, i, s, h, lane), tag, want)
                            else:
                                assert tag is not None and tag[0] == 'x', 'missing window must still read finite input'
                            seen.setdefault(lane >> 5, set()).add((addr % 256) // 4)
                            seen.setdefault(('b', lane >> 5), set()).add((addr % 256) // 4 + 1)
                        for half in (0, 1):
                            banks = seen[half] | seen[('b', half)]
                            banks_ok &= len(banks) == 64
            # dY reads (same for every wave)
        for j in range(4):
            for s in range(7):
                for h in range(2):
                    seen = {0: set(), 1: set()}
                    for lane in range(64):
                        fcol, g = lane & 15, lane >> 4
                        q, pp = fcol >> 2, fcol & 3
                        l_x, l_y, l_w, l_z = q & 1, q >> 1, g & 1, g >> 1
                        swl = l_x | (l_y << 1)
                        addr = yb + l_y * DY_Y + l_z * DY_Z + l_x * DY_X + l_w * 128 + pp * 8 + ((j ^ swl) << 5) + s * 2 * DY_X + h * 2 * DY_Z
                        tag = lds.get(addr)
                        win = 2 * kc + l_w
                        want = ('y', win if win < n_windows else None, l_z + 2 * h, 2 * gq + l_y, 2 * s + l_x, j * 4 + pp, win < n_windows)
                        assert tag == want, ('dY', (kc, gq, j, s, h, lane), tag, want)
                        seen[lane >> 5] |= {(addr % 256) // 4, (addr % 256) // 4 + 1}
                    banks_ok &= len(seen[0]) == 64 and len(seen[1]) == 64
        # the kernel fetches group G + 2 after group G's barrier
        if G + 2 < len(groups):
            fetch(lds, *groups[G + 2], n_windows)
    assert banks_ok, 'a 32-lane half of a transposing read does not cover the 64 banks exactly once'
    return len(groups)


if __name__ == '__main__':
    for n in (1, 2, 5):
        print('n_windows = %d: %d groups, every fragment read on its tag, banks covered once' % (n, check(n)))
