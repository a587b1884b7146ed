"""Per-iteration launch timeline of bench.py from a rocprofv3 kernel trace: for every kernel of the steady-state iteration
its average duration AND the average gap to the kernel before it (the dependent boundary a stream pays between launches),
so that the fixed cost of an iteration can be read off: sum(durations) + sum(gaps) = the iteration.

Usage (GPU box):  python3 scripts/timeline.py OUTDIR -- <bench.py arguments>
Runs `rocprofv3 --kernel-trace` on `python3 bench.py <arguments>` itself, then reads the *_kernel_trace.csv it leaves."""
import csv
import glob
import os
import subprocess
import sys


def short(name):
    name = name.split('(')[0]
    name = name.replace('void ', '').replace('klnmf::', '')
    return name[:70]


def main():
    out = os.path.abspath(sys.argv[1])
    args = sys.argv[sys.argv.index('--') + 1:]
    root = os.environ.get('GRAFT_REPO_ROOT', os.getcwd())
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR='/tmp')
    cmd = ['rocprofv3', '--kernel-trace', '--output-format', 'csv', '-d', out, '-o', 'tl', '--',
           'python3', os.path.join(root, 'bench.py')] + args
    with open(os.path.join(out, 'bench.log'), 'w') as log:
        rc = subprocess.call(cmd, cwd='/tmp', env=env, stdout=log, stderr=subprocess.STDOUT)
    files = glob.glob(os.path.join(out, '**', '*kernel_trace.csv'), recursive=True)
    if rc != 0 or not files:
        print('rocprofv3 rc=%d, no trace' % rc)
        sys.exit(1)
    rows = []
    for fn in files:
        with open(fn) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    rows.sort()
    # the steady state: the last third of the dispatches; an iteration = from one row pass (the longest kernel) to the next
    tail = rows[len(rows) * 2 // 3:]
    longest = max(set(n for _, _, n in tail), key=lambda n: sum(e - s for s, e, m in tail if m == n))
    starts = [i for i, (_, _, n) in enumerate(tail) if n == longest]
    iters = [tail[a:b] for a, b in zip(starts[:-1], starts[1:])]
    # keep the iterations with the most common launch sequence
    seqs = {}
    for it in iters:
        seqs.setdefault(tuple(n for _, _, n in it), []).append(it)
    seq, group = max(seqs.items(), key=lambda kv: len(kv[1]))
    print('%d dispatches, %d steady-state iterations of %d launches each (anchor: %s)' % (len(rows), len(group), len(seq), longest))
    tot_d = tot_g = 0.0
    prev_end = {id(it): None for it in group}
    print('%-72s %10s %10s' % ('kernel', 'dur us', 'gap-before us'))
    for j, name in enumerate(seq):
        d = sum(it[j][1] - it[j][0] for it in group) / len(group) / 1e3
        if j == 0:
            # gap before the anchor = from the end of the previous iteration's last kernel
            gaps = []
            for a, b in zip(starts[:-1], starts[1:]):
                if tuple(n for _, _, n in tail[a:b]) == seq and a > 0:
                    gaps.append(tail[a][0] - tail[a - 1][1])
            g = sum(gaps) / max(1, len(gaps)) / 1e3
        else:
            g = sum(it[j][0] - it[j - 1][1] for it in group) / len(group) / 1e3
        tot_d += d
        tot_g += g
        print('%-72s %10.2f %10.2f' % (name, d, g))
    span = sum(it[-1][1] - it[0][0] for it in group) / len(group) / 1e3
    print('sum of durations %.2f us, sum of gaps %.2f us, iteration (start of anchor to end of last kernel) %.2f us' % (tot_d, tot_g, span))


if __name__ == '__main__':
    main()
