#!/bin/bash
# Samples the card's power and clocks (rocm-smi, every 0.25 s) while bench.py runs its default segments: evidence for "the part
# is power-managed" (DESIGN.md sections 6-8).  Output: the samples and their summary.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 bench.py --no-cpu-baseline --no-16bit-segment --data device --repeats 30 --steps 60 > /tmp/pc_bench.json 2>/tmp/pc_bench.err &
BP=$!
sleep 4
for i in $(seq 1 60); do
  rocm-smi --showpower --showclocks --showuse --json 2>/dev/null | head -c 2000; echo
  sleep 0.25
  kill -0 $BP 2>/dev/null || break
done > /tmp/pc_samples.txt
wait $BP
grep '^{' /tmp/pc_bench.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], 'it/s', d['ms_per_step'], 'ms')"
python3 - <<'PY'
import json
rows = []
for ln in open('/tmp/pc_samples.txt'):
    ln = ln.strip()
    if not ln.startswith('{'):
        continue
    try:
        d = json.loads(ln)
    except Exception:
        continue
    c = d.get('card0', {})
    rows.append(c)
import statistics
def num(v):
    try:
        return float(str(v).strip('()').replace('Mhz', ''))
    except Exception:
        return None
print(len(rows), 'samples (rocm-smi --showpower --showclocks --showuse, 0.25 s apart, while the timed segments run)')
for key in ('Current Socket Graphics Package Power (W)', 'sclk clock speed:', 'mclk clock speed:', 'fclk clock speed:', 'GPU use (%)'):
    vals = [num(r.get(key)) for r in rows if num(r.get(key)) is not None]
    busy = [v for v, r in zip(vals, rows) if num(r.get('GPU use (%)')) and num(r.get('GPU use (%)')) >= 95]
    print('%-45s all: %s' % (key, ' '.join('%g' % v for v in vals)))
    if busy:
        print('%-45s samples at >= 95 %% use: median %g  min %g  max %g' % ('', statistics.median(busy), min(busy), max(busy)))
PY
