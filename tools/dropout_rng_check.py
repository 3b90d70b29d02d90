"""statistics of the dropout bit generator (csrc/vb_rt.h: vb_dropout_bits8), old four-mixer-word form against the current
two-mixer-words + two-24-bit-multiplies form, in numpy: keep rates, pair / lag / stream / seed correlations, chi-squares, bit
balance.  python tools/dropout_rng_check.py"""
import numpy as np
M32 = np.uint64(0xFFFFFFFF)
def u32(x): return x & M32
def mix32(x, k):
    x = u32(x); x ^= x >> np.uint64(16); x = u32(x * np.uint64(0x7feb352d)); x ^= k
    x ^= x >> np.uint64(15); x = u32(x * np.uint64(0x846ca68b)); x ^= x >> np.uint64(16)
    return x
def mul24(a, b): return u32((a & np.uint64(0xFFFFFF)) * (np.uint64(b) & np.uint64(0xFFFFFF)))
def keys(seed, stream):
    s0, s1 = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    st = np.uint64(stream)
    k1 = mix32(s0 ^ u32(st * np.uint64(0x9E3779B9)), s1 ^ np.uint64(0x5ca1ab1e))
    k2b = mix32(u32(s1 + u32(st * np.uint64(0x85EBCA6B))), k1)
    return k1, k2b
def gen_old(seed, stream, groups):
    k1, k2b = keys(seed, stream)
    g = groups.astype(np.uint64)
    k2 = k2b ^ u32((g >> np.uint64(30)) * np.uint64(0xC2B2AE35))
    c = u32(g * np.uint64(4) + k1)
    return [mix32(u32(c + np.uint64(i)), k2) for i in range(4)]
def gen_new(seed, stream, groups):
    k1, k2b = keys(seed, stream)
    g = groups.astype(np.uint64)
    k2 = k2b ^ u32((g >> np.uint64(31)) * np.uint64(0xC2B2AE35))
    c = u32(g * np.uint64(2) + k1)
    w0 = mix32(c, k2); w1 = mix32(u32(c + np.uint64(1)), k2)
    w2 = mul24(w0 >> np.uint64(8), 0x9E3779) ^ w1
    w3 = mul24(w1 >> np.uint64(8), 0x85EBCB) ^ w0
    return [w0, w1, w2, w3]
def u16s(ws):
    out = []
    for w in ws:
        out.append((w & np.uint64(0xFFFF)).astype(np.int64)); out.append((w >> np.uint64(16)).astype(np.int64))
    return np.stack(out, 1)           # [groups, 8]
def report(name, gen):
    n = 1 << 21
    groups = np.arange(n, dtype=np.uint64) + np.uint64(12345)
    t = int(round(0.1 * 65536))
    u = u16s(gen(1234567891234, 7, groups))
    keep = (u >= t).astype(np.float64)
    p = keep.mean(0); sig = np.sqrt(0.1 * 0.9 / n)
    print(name, "keep-rate z per element:", np.round((p - (1 - t / 65536)) / sig, 2))
    kc = keep - keep.mean(0)
    C = (kc.T @ kc) / n / (0.09)
    off = C[np.triu_indices(8, 1)]
    print("   within-group pair corr z (28 pairs): max |z| %.2f" % (np.abs(off).max() * np.sqrt(n)))
    # neighbouring groups, same element and all cross pairs
    z = []
    for lag in (1, 2, 41, 96):
        Cl = (kc[:-lag].T @ kc[lag:]) / (n - lag) / 0.09
        z.append(np.abs(Cl).max() * np.sqrt(n - lag))
    print("   lagged-group corr max |z| (lags 1,2,41,96):", np.round(z, 2))
    # another stream / seed
    u2 = u16s(gen(1234567891234, 8, groups)); k2 = (u2 >= t).astype(np.float64); k2c = k2 - k2.mean(0)
    Cx = (kc.T @ k2c) / n / 0.09
    print("   cross-stream corr max |z| %.2f" % (np.abs(Cx).max() * np.sqrt(n)))
    u3 = u16s(gen(1234567891235, 7, groups)); k3 = (u3 >= t).astype(np.float64); k3c = k3 - k3.mean(0)
    print("   cross-seed corr max |z| %.2f" % (np.abs((kc.T @ k3c) / n / 0.09).max() * np.sqrt(n)))
    # uniformity of the 16-bit values: chi-square on 256 buckets per element
    chi = []
    for e in range(8):
        h = np.bincount(u[:, e] >> 8, minlength=256); ex = n / 256
        chi.append(((h - ex) ** 2 / ex).sum())
    print("   chi2(255 dof, mean 255, sd 22.6) per element:", np.round(chi, 0))
    # low-order bit balance
    bal = [abs(((u[:, e] >> b) & 1).mean() - 0.5) * 2 * np.sqrt(n) for e in range(8) for b in range(16)]
    print("   bit balance max |z| %.2f" % max(bal))
report("OLD", gen_old)
report("NEW", gen_new)

def lagscan(name, gen, seeds):
    n = 1 << 20
    groups = np.arange(n, dtype=np.uint64) * np.uint64(1) + np.uint64(999)
    t = int(round(0.1 * 65536))
    allz = []
    for sd in seeds:
        u = u16s(gen(sd, 3, groups)); keep = (u >= t).astype(np.float64); kc = keep - keep.mean(0)
        for lag in (1, 2, 3, 4, 8, 41, 96, 164, 768 // 8):
            Cl = (kc[:-lag].T @ kc[lag:]) / (n - lag) / 0.09
            allz.extend((np.abs(Cl) * np.sqrt(n - lag)).ravel().tolist())
        C = (kc.T @ kc) / n / 0.09
        allz.extend((np.abs(C[np.triu_indices(8, 1)]) * np.sqrt(n)).tolist())
    a = np.array(allz)
    print(name, "n stats", a.size, "max |z| %.2f" % a.max(), "frac>3: %.4f (expect 0.0027)" % (a > 3).mean(), "frac>2: %.4f (expect 0.0455)" % (a > 2).mean())
lagscan("OLD", gen_old, [11, 222, 3333, 44444, 555555])
lagscan("NEW", gen_new, [11, 222, 3333, 44444, 555555])
