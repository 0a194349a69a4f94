#!/usr/bin/env python3
"""Linear scan of a kernel's gfx950 assembly (hipcc -S --cuda-device-only) for vector / accumulator registers that are READ before any
write to them in program order (entry state: v0 = work-item id; kernel arguments arrive in SGPRs).  Program order over-approximates
dominance for the forward-structured kernels here (a loop body is preceded by its preheader), so an empty report means that no
v_mfma / VALU source can observe a register the wave did not write — the 'consumes state it did not write' hypothesis of VERDICT r3 item 2.
    python profiles/scripts/isa_uninit_scan.py engine_bf16.s mlp_kernelINS_6bf16_tELi4ELb0ELb0
"""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(pat) + r'\w*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
reg = re.compile(r'\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]')


def regs(tok):
    out = []
    for m in reg.finditer(tok):
        if m.group(1):
            out.append((m.group(1), int(m.group(2))))
        else:
            out += [(m.group(3), k) for k in range(int(m.group(4)), int(m.group(5)) + 1)]
    return out


written = {('v', 0)}
bad = []
n_mfma = 0
for i in range(start + 1, end):
    l = lines[i].split(';')[0].strip()
    if not l or l.startswith('.') or l.endswith(':') or l.startswith('s_') and not l.startswith('s_') :
        continue
    op, _, rest = l.partition(' ')
    if not (op.startswith('v_') or op.startswith('global_') or op.startswith('buffer_') or op.startswith('ds_') or op.startswith('scratch_') or op.startswith('flat_')):
        continue
    ops = [o.strip() for o in rest.split(',')]
    is_store = ('store' in op or op.startswith('ds_write') or op.startswith('global_atomic')) and 'load' not in op
    no_dst = is_store or op.startswith('v_cmp') and not op.startswith('v_cmpx') and False
    if op.startswith('v_cmp'):
        dst, srcs = [], ops[1:] if len(ops) > 2 or ops[0].startswith('s') or ops[0] == 'vcc' else ops
    elif is_store:
        dst, srcs = [], ops
    else:
        dst, srcs = ops[:1], ops[1:]
        if op in ('v_readlane_b32', 'v_readfirstlane_b32'):
            dst = []
            srcs = ops[1:]
    if op.startswith('v_mfma'):
        n_mfma += 1
    for s in srcs:
        for r in regs(s):
            if r not in written:
                bad.append((i + 1, op, r))
    # v_fmac / v_mac / dpp-with-old / mfma-in-place read their destination too
    if op.startswith(('v_fmac', 'v_mac', 'v_pk_fmac', 'v_dot2c', 'v_writelane')):
        for r in regs(dst[0]):
            if r not in written:
                bad.append((i + 1, op + ' (dst read)', r))
    for d in dst:
        for r in regs(d):
            written.add(r)
print(f'{pat}: {end - start} lines, {n_mfma} v_mfma, {len(written)} registers written; reads before any write: {len(bad)}')
for b in bad[:40]:
    print('  line', b[0], b[1], b[2][0] + str(b[2][1]))
