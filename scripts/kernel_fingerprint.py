#!/usr/bin/env python3
"""Fingerprint of the device code in a built library: per kernel, the instruction count, a hash of its instruction stream
(mnemonics + operands, addresses stripped) and the counts of the instruction kinds the reviews quote.

    python scripts/kernel_fingerprint.py LIB.so [regex] > fingerprint.txt

Two builds whose fingerprints agree for a kernel execute the same instructions for it: how a source clean-up is shown to
have left the product kernels as they were (profiles/r05_fingerprint_*.txt)."""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
BUNDLER = '/opt/rocm/lib/llvm/bin/clang-offload-bundler'
KINDS = ['v_mfma', 'ds_read_b64_tr_b16', 'ds_read_b64_tr_b8', 'ds_read_b128', 'global_load_lds_dwordx4', 'v_rcp_f32',
         'v_log_f32', 'v_cvt_scalef32_pk_fp8_f16', 'v_pk_mul_f16', 's_barrier', 's_nop', 'scratch_']


def device_code(lib, tmp):
    """The gfx950 code objects embedded in `lib` (section .hip_fatbin: one clang offload bundle per translation unit)."""
    fat = os.path.join(tmp, 'fatbin')
    subprocess.check_call([OBJDUMP.replace('objdump', 'objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', lib, fat])
    blob = open(fat, 'rb').read()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    outs = []
    for i, s0 in enumerate(starts):
        part = os.path.join(tmp, 'bundle%d' % i)
        open(part, 'wb').write(blob[s0:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        targets = subprocess.run([BUNDLER, '--list', '--type=o', '--input=' + part], capture_output=True, text=True).stdout.split()
        tgt = [t for t in targets if 'gfx950' in t]
        if not tgt:
            continue
        out = os.path.join(tmp, 'dev%d.co' % i)
        subprocess.check_call([BUNDLER, '--unbundle', '--type=o', '--input=' + part, '--targets=' + tgt[0], '--output=' + out])
        outs.append(out)
    return outs


def main():
    lib = sys.argv[1]
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    with tempfile.TemporaryDirectory() as tmp:
        txt = ''.join(subprocess.run([OBJDUMP, '-d', '--no-show-raw-insn', '-C', co], capture_output=True, text=True).stdout
                      for co in device_code(lib, tmp))
    cur, body = None, {}
    for line in txt.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.*)>:$', line)
        if m:
            cur = m.group(1)
            body[cur] = []
            continue
        if cur is None or not line.startswith('\t'):
            continue
        ins = line.strip().split('//')[0].strip()
        if ins:
            body[cur].append(ins)
    for name in sorted(body):
        short = re.sub(r'\(.*$', '', name).replace('void klnmf::', '').replace('klnmf::', '')
        if pat and not pat.search(short):
            continue
        ins = body[name]
        # branch targets are absolute addresses: strip them so that moving a kernel inside the object does not change its hash
        norm = [re.sub(r'<[^>]*>', '', re.sub(r'\b[0-9a-f]{4,}\b(?= <)', '', i)) for i in ins]
        h = hashlib.sha1('\n'.join(norm).encode()).hexdigest()[:12]
        kinds = ' '.join('%s=%d' % (k, sum(1 for i in ins if i.startswith(k))) for k in KINDS)
        print('%-56s n=%-6d sha=%s %s' % (short, len(ins), h, kinds))


if __name__ == '__main__':
    main()
