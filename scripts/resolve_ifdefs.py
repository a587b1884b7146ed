#!/usr/bin/env python3
"""A small `unifdef`: resolves the preprocessor conditionals of a source file whose condition mentions only symbols
given on the command line, leaves every other conditional alone.

    python scripts/resolve_ifdefs.py FILE -USYM ... -DSYM=VALUE ...   (rewrites FILE in place)

Round 5 used it to take the timing-only ablations and experiment switches out of the product kernels (the removed
branches live on in the git history and under experiments/)."""
import re
import sys


def evaluate(expr, table):
    """True / False if `expr` only needs symbols of `table`, else None."""
    e = re.sub(r'//.*$', '', expr).strip()
    e = re.sub(r'/\*.*?\*/', '', e)
    names = set(re.findall(r'[A-Za-z_]\w*', e)) - {'defined'}
    if not names or not names <= set(table):
        return None

    def defined(m):
        return '1' if table[m.group(1)] is not None else '0'
    e = re.sub(r'defined\s*\(\s*(\w+)\s*\)', defined, e)
    e = re.sub(r'defined\s+(\w+)', defined, e)
    for n in names:
        v = table[n]
        e = re.sub(r'\b%s\b' % n, '0' if v is None else str(v), e)
    e = e.replace('&&', ' and ').replace('||', ' or ').replace('!', ' not ').replace(' not =', '!=')
    return bool(eval(e, {'__builtins__': {}}))


def resolve(lines, table):
    out = []
    # stack entries: [mode, taken, keep_directives]; mode: 'known' (resolved) or 'unknown' (left alone)
    stack = []

    def emitting():
        return all(s[1] for s in stack if s[0] == 'known')
    i = 0
    while i < len(lines):
        line = lines[i]
        full = line
        while full.rstrip().endswith('\\') and i + 1 < len(lines) and re.match(r'\s*#\s*(if|elif)', line):
            i += 1
            full = full.rstrip()[:-1] + lines[i]
        m = re.match(r'\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)', full, re.S)
        if not m:
            if emitting():
                out.append(line)
            i += 1
            continue
        kind, rest = m.group(1), m.group(2).strip()
        if kind in ('ifdef', 'ifndef', 'if'):
            if kind == 'ifdef':
                sym = rest.split()[0]
                val = (table[sym] is not None) if sym in table else None
            elif kind == 'ifndef':
                sym = rest.split()[0]
                val = (table[sym] is None) if sym in table else None
            else:
                val = evaluate(rest, table)
            if val is None:
                stack.append(['unknown', True, True])
                if emitting():
                    out.append(full)
            else:
                stack.append(['known', val, False, val])      # [.., .., .., any branch taken so far]
        elif kind == 'elif':
            top = stack[-1]
            if top[0] == 'unknown':
                if emitting():
                    out.append(full)
            else:
                if top[3]:
                    top[1] = False
                else:
                    val = evaluate(rest, table)
                    if val is None:
                        # every earlier branch was resolved as false: this #elif is the first live condition of the chain
                        stack[-1] = ['unknown', True, True]
                        if emitting():
                            out.append('#if ' + rest)
                    else:
                        top[1] = val
                        top[3] = val
        elif kind == 'else':
            top = stack[-1]
            if top[0] == 'unknown':
                if emitting():
                    out.append(full)
            else:
                top[1] = not top[3]
                top[3] = True
        else:
            top = stack.pop()
            if top[0] == 'unknown' and emitting():
                out.append(full)
        i += 1
    if stack:
        raise SystemExit('unbalanced conditionals')
    return out


def main():
    path = sys.argv[1]
    table = {}
    for a in sys.argv[2:]:
        if a.startswith('-U'):
            table[a[2:]] = None
        elif a.startswith('-D'):
            k, _, v = a[2:].partition('=')
            table[k] = v if v else '1'
    lines = open(path).read().split('\n')
    open(path, 'w').write('\n'.join(resolve(lines, table)))


if __name__ == '__main__':
    main()
