"""The generator behind the stochastic rounding of the fp8 tiles (csrc/mfma4.hip.h, sr_next): s <- (s & 0xffffff) * 0x6C8E95 + 0x3C6EF35F
(one v_mad_u32_u24); an entry's seed is bits [31:25] of s, the next entry's bits [24:18].  Checked here: both 7-bit fields are uniform
(chi^2 / 127 = 0.9 / 1.0 over 200 000 steps), serial correlations at lags 1 and 2 below 0.006, the two fields of one step uncorrelated
(0.001), and the 24-bit state has the full period 2^24 (a lane takes 1024 steps per launch at f = 4096).   python3 experiments/sr_probe/lcg_check.py"""
import numpy as np
A, C = 0x6C8E95, 0x3C6EF35F


def stream(s0, n):
    s, out = int(s0), np.empty(n, dtype=np.uint32)
    for i in range(n):
        s = ((s & 0xffffff) * A + C) & 0xffffffff
        out[i] = s
    return out


x = stream(12345, 200000)
for name, f in (('[31:25]', (x >> 25) & 127), ('[24:18]', (x >> 18) & 127)):
    h = np.bincount(f, minlength=128)
    u = (f.astype(float) + 0.5) / 128
    print(name, 'chi2/127 = %.2f' % (((h - len(f) / 128) ** 2 / (len(f) / 128)).sum() / 127), 'mean %.4f' % u.mean(),
          'lag-1 correlation %.4f' % np.corrcoef(u[:-1], u[1:])[0, 1], 'lag-2 %.4f' % np.corrcoef(u[:-2], u[2:])[0, 1])
print('correlation between the two fields of one step %.4f' % np.corrcoef(((x >> 25) & 127).astype(float), ((x >> 18) & 127).astype(float))[0, 1])
s = s0 = 12345 & 0xffffff
n = 0
while True:
    s = (s * A + C) & 0xffffff
    n += 1
    if s == s0 or n > 2 ** 24 + 5:
        break
print('period of the 24-bit state', n)
