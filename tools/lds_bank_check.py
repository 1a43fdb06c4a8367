# LDS bank-conflict check of the swizzled [rows][128 B] image for the 16x16x32 operand reads
def f(r):
    u0, u1, u2 = (r >> 1) & 1, (r >> 2) & 1, (r >> 3) & 1
    return (u0 << 2) | ((u1 ^ u2) << 1) | u1
def off(r, ch, w=0): return r * 128 + ((ch ^ f(r)) << 4) + w
# ds_read_b128 groups
G128 = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128 += [[l + 32 for l in g] for g in G128]
def conflicts_b128(addr):  # addr[lane] byte address; 16-byte accesses, bank = (a/4)%64
    worst = 1
    for g in G128:
        slots = {}
        for l in g:
            s = (addr[l] // 16) % 16
            slots.setdefault(s, set()).add(addr[l])
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst
def conflicts_tr(addr):  # ds_read_b64_tr_b16: 2 x 32 lanes, 8-byte accesses, bank=(a/4)%64
    worst = 1
    for half in (range(32), range(32, 64)):
        banks = {}
        for l in half:
            b = (addr[l] // 8) % 32
            banks.setdefault(b, set()).add(addr[l])
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst
w = 1
for t in range(7):
    for ks in range(2):
        addr = [off(min(16 * t + (l & 15), 103), 4 * ks + (l >> 4)) for l in range(64)]
        kb = [(l & 15) * 128 + (((l >> 4) ^ f(l & 15)) << 4) for l in range(64)]
        if t < 6:
            assert addr == [(kb[l] ^ (ks << 6)) + 2048 * t for l in range(64)]
        w = max(w, conflicts_b128(addr))
print("K b128 worst way:", w)
w = 1
for u in range(4):
    for ab in range(2):
        if 32 * u + 16 * ab >= 104: continue
        for dt in range(4):
            addr = []
            for l in range(64):
                g, li = l >> 4, l & 15; q, p = li >> 2, li & 3
                r = 32 * u + 16 * ab + 4 * g + q
                rc = min(r, 103)
                addr.append(off(rc, 2 * dt + (p >> 1), 8 * (p & 1)))
                vb = (4 * g + q) * 128 + (((p >> 1) ^ f(4 * g + q)) << 4) + 8 * (p & 1)
                if r < 104: assert addr[-1] == (vb ^ (dt << 5)) + (32 * u + 16 * ab) * 128, (u, ab, dt, l)
            c = conflicts_tr(addr)
            if c > 1: print("tr conflict", u, ab, dt, c)
            w = max(w, c)
print("V tr worst way:", w)
# DMA source map: block b, lane l -> LDS byte b*1024 + 16 l holds chunk (l&7)^f(row) of row 8b + (l>>3)
for b in range(13):
    for l in range(64):
        row, sl = 8 * b + (l >> 3), l & 7
        ch = sl ^ f(row)
        assert off(row, ch) == b * 1024 + 16 * l
        ch0 = (l & 7) ^ f(l >> 3)
        # f(row) for row = 8b + lrow: u2 = b&1 toggles bits: (u1^u2)<<1 -> ^2
        assert ch == ch0 ^ ((b & 1) << 1)
print("DMA map ok")
