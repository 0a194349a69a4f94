#!/usr/bin/env python3
"""ISA check behind the rule of DESIGN 4.17: NO INLINE-ASM RESULT MAY BE A MATRIX-INSTRUCTION OPERAND.

A VALU write needs wait states before a v_mfma reads the register; the compiler's hazard recogniser inserts them for instructions it can see and cannot look
into an `asm` statement.  Round 4's packed relu (`v_pk_max_i16` as inline assembly feeding the decoder head's MFMA) was right alone on the chip and wrong whenever
another kernel shared the CU.  This scan walks the gfx950 assembly of a translation unit (`hipcc -S --cuda-device-only`, `make -C achelous_amd/csrc isa`):
for every `;;#ASMSTART .. ;;#ASMEND` block it collects the vector registers the block's instructions write, then follows the straight-line code behind the block
until each of those registers is overwritten by a compiler-visible instruction (or the function ends / `window` instructions have passed) and reports every
v_mfma that reads one of them as srcA / srcB / srcC in between.

    python profiles/scripts/isa_asm_mfma_scan.py achelous_amd/csrc/build/engine_f16.s [...]      exit code 1 = findings"""
import re
import sys

REG = re.compile(r'\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]')


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out |= {(m.group(3), k) for k in range(int(m.group(4)), int(m.group(5)) + 1)}
    return out


def split_ops(line):
    op, _, rest = line.partition(' ')
    return op, [o.strip() for o in rest.split(',')] if rest.strip() else []


def writes_vector(op):
    if op.startswith(('v_cmp', 'v_nop')) and not op.startswith('v_cmpx'):
        return False
    if 'store' in op or op.startswith(('ds_write', 'buffer_atomic', 'global_atomic', 's_')):
        return False
    return op.startswith(('v_', 'global_load', 'buffer_load', 'ds_read', 'scratch_load', 'flat_load', 'ds_bpermute', 'ds_swizzle', 'ds_permute'))


def scan(path, window=400):
    lines = open(path).read().split('\n')
    findings, blocks, func = [], 0, '?'
    i = 0
    while i < len(lines):
        raw = lines[i]
        if raw.startswith('_Z') and raw.rstrip().endswith((':',)) or re.match(r'^_Z\w+:', raw):
            func = raw.split(':')[0]
        if ';;#ASMSTART' in raw:
            j = i + 1
            tainted = set()
            while j < len(lines) and ';;#ASMEND' not in lines[j]:
                l = lines[j].split(';')[0].strip()
                if l and not l.startswith('.'):
                    op, ops = split_ops(l)
                    if writes_vector(op) and ops:
                        tainted |= regs(ops[0])
                j += 1
            if tainted:
                blocks += 1
                k, seen = j + 1, 0
                while k < len(lines) and tainted and seen < window:
                    l = lines[k].split(';')[0].strip()
                    if lines[k].startswith('.Lfunc_end'):
                        break
                    if ';;#ASMSTART' in lines[k]:
                        # the next asm block: its writes are handled by its own scan; registers it overwrites stop being this block's results
                        kk = k + 1
                        while kk < len(lines) and ';;#ASMEND' not in lines[kk]:
                            l2 = lines[kk].split(';')[0].strip()
                            if l2 and not l2.startswith('.'):
                                op2, ops2 = split_ops(l2)
                                if writes_vector(op2) and ops2:
                                    tainted -= regs(ops2[0])
                            kk += 1
                        k = kk + 1
                        continue
                    if l and not l.startswith('.') and not l.endswith(':'):
                        seen += 1
                        op, ops = split_ops(l)
                        if op.startswith('v_mfma') or op.startswith('v_smfmac'):
                            src = set()
                            for o in ops[1:]:
                                src |= regs(o)
                            hit = src & tainted
                            if hit:
                                findings.append((path, func, k + 1, l, sorted(hit)[:4], i + 1))
                        if writes_vector(op) and ops:
                            tainted -= regs(ops[0])
                    k += 1
            i = j
        i += 1
    return blocks, findings


def main():
    bad = 0
    for p in sys.argv[1:]:
        blocks, findings = scan(p)
        print(f'{p}: {blocks} inline-asm blocks with vector results, {len(findings)} consumed by a matrix instruction')
        for path, func, ln, text, hit, asm_ln in findings:
            print(f'  {func} line {ln}: {text}   <- reads {hit} written by the asm block at line {asm_ln}')
        bad += len(findings)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
