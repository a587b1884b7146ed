"""Bug hunt, part 10: klnmf_upload_V in pieces -- the same matrix uploaded whole, and as a random tiling of rectangular blocks (random
order, strided sources: views into a larger array, fp32 / fp64 sources, per-block scale folded back), must give the same context:
same loss of (W0, H0), same W0, same two updates.  f16 / f64 / f32.

    python3 scripts/upload_fuzz.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def tiling(rs, n, f):
    rows = sorted(set([0, n] + [int(v) for v in rs.randint(1, max(2, n), size=rs.randint(0, 4))]))
    cols = sorted(set([0, f] + [int(v) for v in rs.randint(1, max(2, f), size=rs.randint(0, 4))]))
    blocks = [(r0, r1, c0, c1) for r0, r1 in zip(rows[:-1], rows[1:]) for c0, c1 in zip(cols[:-1], cols[1:])]
    rs.shuffle(blocks)
    return blocks


def main():
    from multimodal_amd import _native
    from oracle import klnmf_oracle as orc
    rs = np.random.RandomState(0)
    bad = 0
    for (n, f, k) in [(37, 53, 7), (300, 257, 33), (1000, 64, 8), (33000, 40, 5), (70001, 33, 12), (65, 4100, 20)]:
        X = orc.synthetic_V(n + f, n, f, k)
        H0 = orc.synthetic_H0(n + f, f, k)
        for prec in ('f16', 'f64', 'f32'):
            out = []
            for mode in ('whole', 'tiled'):
                with _native.Context(prec) as c:
                    c.set_problem(n, f, k, 4)
                    c.set_v_max(float(X.max()))
                    if mode == 'whole':
                        c.upload_V(X, row0=0, col0=0, scale=1.0)
                    else:
                        for (r0, r1, c0, c1) in tiling(rs, n, f):
                            big = np.zeros((r1 - r0 + 3, c1 - c0 + 5), dtype=[np.float64, np.float32][rs.randint(2)])
                            sc = [1.0, 0.5, 4.0][rs.randint(3)]
                            big[1:1 + r1 - r0, 2:2 + c1 - c0] = X[r0:r1, c0:c1] / sc
                            c.upload_V(big[1:1 + r1 - r0, 2:2 + c1 - c0], row0=r0, col0=c0, scale=sc)      # a strided view
                    c.set_H(H0)
                    c.init_W()
                    e0 = c.error()
                    W0 = c.get_W()
                    c.update(True)
                    c.update(True)
                    out.append((e0, W0, c.get_W(), c.get_H(), c.error()))
            (e0a, W0a, Wa, Ha, ea), (e0b, W0b, Wb, Hb, eb) = out
            # fp32 sources of the tiled upload round the data once more (2^-24 relative): exact modes see that
            # ... and the fp16 storage rounds an fp32-rounded source: on the rare ties one half ulp differs (5e-4 of ONE entry)
            tol = {'f64': 3e-7, 'f32': 3e-6, 'f16': 3e-4}.get(prec, 2e-6)
            rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
            devs = (abs(e0a - e0b) / abs(e0a), rel(W0b, W0a), rel(Wb, Wa), rel(Hb, Ha), abs(ea - eb) / abs(ea))
            ok = all(d <= tol for d in devs)
            print('%-7s %6d x %4d k=%2d  %s  loss0 %.1e W0 %.1e W %.1e H %.1e loss %.1e' % ((prec, n, f, k, 'ok  ' if ok else 'FAIL') + devs), flush=True)
            bad += 0 if ok else 1
    print('%d case(s) differ' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
