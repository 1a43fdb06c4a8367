import numpy as np
def mix32(x):
    x = x.astype(np.uint32)
    x ^= x >> 16; x = (x * np.uint32(0x7feb352d)).astype(np.uint32)
    x ^= x >> 15; x = (x * np.uint32(0x846ca68b)).astype(np.uint32)
    x ^= x >> 16
    return x
def draw_old(row, k, key, max_k):
    idx = (row[:, None].astype(np.uint64) * max_k + k[None, :]).astype(np.uint32)
    return mix32(idx ^ np.uint32(key)) >> 16
PHI = np.uint32(0x9E3779B1); M24 = np.uint32(0xB5297B)
def draw_new(row, k, key, max_k):
    A = mix32(row.astype(np.uint32) ^ np.uint32(key))
    x = (A[:, None] + (k[None, :].astype(np.uint32) * PHI)).astype(np.uint32)
    y = x ^ (x >> 16)
    z = ((y & np.uint32(0xFFFFFF)).astype(np.uint64) * np.uint64(M24)).astype(np.uint64) & np.uint64(0xFFFFFFFF)
    return (z >> np.uint64(16)).astype(np.uint32)
def stats(name, d, thr):
    keep = (d >= thr).astype(np.float64)
    R, K = keep.shape
    p = 1 - keep.mean()
    rs = keep.sum(1); cs = keep.sum(0)
    var_ratio_r = rs.var() / (K * p * (1 - p)); var_ratio_c = cs.var() / (R * p * (1 - p))
    c = keep - keep.mean()
    def corr(a, b): return (a * b).mean() / c.var()
    lag_k = [corr(c[:, :-l], c[:, l:]) for l in (1, 2, 3, 4, 8, 16, 32)]
    lag_r = [corr(c[:-l, :], c[l:, :]) for l in (1, 2, 3, 100)]
    x22 = corr(c[:-1, :-1] * c[1:, 1:], c[:-1, 1:] * c[1:, :-1])   # 2x2 interaction
    # uniformity of the 16-bit draw: chi2 over 256 bins of top byte and low byte
    def chi(v):
        h = np.bincount(v, minlength=256).astype(np.float64); e = h.sum() / 256
        return ((h - e) ** 2 / e).sum() / 255
    print("%-6s drop %.4f  var ratio rows %.3f cols %.3f  lagk %s  lagr %s  x22 %.4f  chi hi %.2f lo %.2f" %
          (name, p, var_ratio_r, var_ratio_c, np.round(lag_k, 4), np.round(lag_r, 4), x22, chi((d >> 8).ravel()), chi((d & 255).ravel())))
R = 2304 * 100; K = 100
row = np.arange(R, dtype=np.uint64); k = np.arange(K, dtype=np.uint64)
for key in (0x1234567, 0x9E3779B9 * 8 + 2019, 77):
    for thr in (6553, 26214):
        stats("old", draw_old(row, k, key, 100), thr)
        stats("new", draw_new(row, k, key, 100), thr)
