#!/usr/bin/env python3
"""Static audit of the inline-asm statements in imm_amd/csrc/*.hip against the wait-state rules hipcc does not apply to them.

hipcc treats an `asm volatile` statement as one opaque instruction: it neither counts its memory operations nor pads the
hazards of the instructions inside (cdna_hip_programming.md §5.7).  Round 5 found one such hazard the hard way (a 16-byte
`buffer_store` without trailing wait states in conv_halo2.hip: 0.3 % corrupted stores whenever a second process shared the GPU).
This tool compiles every source that contains inline asm to gfx950 assembly (`hipcc --cuda-device-only -S`, no GPU needed)
and checks, on the instruction stream the compiler actually emitted around each `;;#ASMSTART … ;;#ASMEND` block:

  S  a VMEM store of more than 64 bits inside an asm block is followed, inside the block, by `s_nop 1` or more;
  G  no SGPR that an asm VMEM instruction reads (descriptor, soffset, base) was written by a VALU instruction
     (`v_readfirstlane`, `v_readlane` = the compiler's SGPR spill reloads, `v_cmp`, carry-outs) fewer than 5 wait states earlier
     with no SALU write of that SGPR in between;
  D  no `v_cmpx` (VALU write of EXEC) fewer than 5 wait states ahead of an asm DPP instruction, and every asm DPP instruction is
     preceded inside its block by `s_nop 1` or more (VALU write of its source -> DPP read: 2 wait states);
  M  an `s_mov_b32 m0` inside an asm block is followed by at least one wait state before the LDS-DMA load that reads M0;
  L  between an asm VMEM load into VGPRs (which hipcc believes written at ;;#ASMEND) and the first asm `s_waitcnt vmcnt` behind it, no
     instruction — compiler copy, spill, reuse — touches the destination registers.

    python tools/asm_hazard_audit.py            # all sources with inline asm, 8 compiles in parallel; exit code 1 on a finding
    python tools/asm_hazard_audit.py file.s …   # audit assembly files that already exist
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VMEM = ('buffer_', 'global_', 'scratch_', 'flat_')


def vregs(tok):
    out = set()
    for m in re.finditer(r'(?<![\w])v\[(\d+):(\d+)\]', tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'(?<![\w\[])v(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


def sregs(tok):
    out = set()
    for m in re.finditer(r's\[(\d+):(\d+)\]', tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'(?<![\w\[])s(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


def valu_sgpr_dsts(mn, ops):
    """SGPRs a VALU instruction writes (lane reads, compares into an SGPR pair, carry-outs)."""
    d = set()
    if not ops:
        return d
    if mn.startswith(('v_readfirstlane', 'v_readlane', 'v_cmp')):
        d |= sregs(ops[0])
    if '_co_' in mn or mn.startswith(('v_addc', 'v_subb', 'v_div_scale', 'v_mad_u64', 'v_mad_i64')):
        if len(ops) > 1:
            d |= sregs(ops[1])
    return d


def parse(text):
    """-> list of (line number, mnemonic, operands, inside an asm block, index of that block)"""
    ins, in_asm, blk = [], False, 0
    for n, line in enumerate(text.split('\n'), 1):
        t = line.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm, blk = True, blk + 1
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        t = t.split(';')[0].strip()
        if not t or t.startswith('//'):
            continue
        if t.endswith(':'):
            ins.append((n, 'LABEL', [t], in_asm, blk))
            continue
        if t.startswith('.'):
            continue
        parts = t.split(None, 1)
        ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
        ins.append((n, parts[0], ops, in_asm, blk))
    return ins


def states(mn, ops):
    return int(ops[0], 0) + 1 if mn == 's_nop' else 1


def audit(name, text):
    ins = parse(text)
    found = []
    for i, (n, mn, ops, ia, blk) in enumerate(ins):
        if not ia:
            continue
        nxt = ins[i + 1] if i + 1 < len(ins) else None
        same_blk_next = nxt is not None and nxt[3] and nxt[4] == blk
        if mn.startswith(VMEM) and '_store_' in mn and mn.endswith(('x3', 'x4')):
            if not (same_blk_next and nxt[1] == 's_nop' and int(nxt[2][0], 0) >= 1):
                found.append('%s:%d S: %s without a trailing s_nop 1 inside the asm block' % (name, n, mn))
        if mn.startswith(VMEM):
            need = set()
            for o in ops:
                need |= sregs(o)
            ws, j = 0, i - 1
            while j >= 0 and ws < 5 and need:
                n2, mn2, ops2 = ins[j][:3]
                if mn2 == 'LABEL':
                    j -= 1
                    continue
                if mn2.startswith('s_') and ops2 and not mn2.startswith(('s_nop', 's_waitcnt', 's_barrier', 's_cmp', 's_cbranch',
                                                                          's_branch', 's_bitcmp')):
                    need -= sregs(ops2[0])          # the SALU is the most recent writer: no hazard through it
                if mn2.startswith('v_'):
                    hit = valu_sgpr_dsts(mn2, ops2) & need
                    if hit:
                        found.append('%s:%d G: %s reads s%s written by %s at line %d only %d wait states earlier'
                                     % (name, n, mn, sorted(hit), mn2, n2, ws))
                ws += states(mn2, ops2)
                j -= 1
            if ' lds' in (' ' + ' '.join(ops)) or mn.endswith('_lds') or '_lds_' in mn:
                prev = ins[i - 1] if i else None
                prev2 = ins[i - 2] if i > 1 else None
                if prev and prev[1] == 's_mov_b32' and prev[2] and prev[2][0] == 'm0':
                    found.append('%s:%d M: LDS-DMA directly behind the s_mov_b32 m0 it depends on' % (name, n))
                elif not (prev and prev2 and prev[1] == 's_nop' and prev2[1] == 's_mov_b32' and prev2[2][0] == 'm0'):
                    found.append('%s:%d M: LDS-DMA in an asm block that does not set M0 itself' % (name, n))
        if mn.startswith(VMEM) and '_load_' in mn and ' lds' not in (' ' + ' '.join(ops)) and ops:
            dst = vregs(ops[0])
            j, hops, forked = i + 1, 0, False
            while j < len(ins) and dst:
                n2, mn2, ops2, ia2 = ins[j][:4]
                if mn2 == 's_branch' and hops < 8:      # unconditional: the scan continues at the target
                    tgt = [k for k, x in enumerate(ins) if x[1] == 'LABEL' and x[2][0] == ops2[0] + ':']
                    if tgt:
                        j, hops = tgt[0] + 1, hops + 1
                        continue
                if mn2.startswith('s_cbranch'):
                    # behind a conditional branch the linear scan may walk a path the loads were not on (the zero-initialisation of
                    # an `if (bias) load; else zero` diamond): writes are no longer attributed, reads still are
                    forked = True
                if ia2 and mn2 == 's_waitcnt' and 'vmcnt' in ' '.join(ops2):
                    break
                if mn2 == 's_endpgm':
                    found.append('%s:%d L: %s is never waited for' % (name, n, mn))
                    break
                if mn2 != 'LABEL' and ops2:
                    writes_first = (mn2.startswith(('v_', 'ds_read', 'ds_load')) and not mn2.startswith(('v_cmp', 'v_readfirstlane', 'v_readlane'))) \
                        or (mn2.startswith(VMEM) and '_load_' in mn2 and ' lds' not in (' ' + ' '.join(ops2)))
                    wr = vregs(ops2[0]) if writes_first else set()
                    rd = set()
                    for o in (ops2[1:] if writes_first else ops2):
                        rd |= vregs(o)
                    if rd & dst:
                        found.append('%s:%d L: v%s of %s read by %s at line %d before the asm wait'
                                     % (name, n, sorted(rd & dst), mn, mn2, n2))
                        break
                    if wr & dst:
                        if not forked:
                            found.append('%s:%d L: v%s of %s overwritten by %s at line %d before the asm wait'
                                         % (name, n, sorted(wr & dst), mn, mn2, n2))
                            break
                        dst = dst - wr
                j += 1
        if '_dpp' in mn:
            prev = ins[i - 1] if i else None
            if not (prev and prev[3] and prev[4] == blk and prev[1] == 's_nop' and int(prev[2][0], 0) >= 1):
                found.append('%s:%d D: %s without s_nop 1 in front of it inside the asm block' % (name, n, mn))
            ws, j = 0, i - 1
            while j >= 0 and ws < 5:
                n2, mn2, ops2 = ins[j][:3]
                if mn2.startswith('v_cmpx'):
                    found.append('%s:%d D: %s %d wait states behind %s (line %d)' % (name, n, mn, ws, mn2, n2))
                if mn2 != 'LABEL':
                    ws += states(mn2, ops2)
                j -= 1
    n_asm = len({b for (_n, _m, _o, ia, b) in ins if ia})
    n_vmem = sum(1 for (_n, m, _o, ia, _b) in ins if ia and m.startswith(VMEM))
    return found, n_asm, n_vmem


def compile_all(outdir):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    srcs = [p for p in sorted(glob.glob(os.path.join(ROOT, 'imm_amd', 'csrc', '*.hip')))
            if re.search(r'asm volatile\("[^"]', open(p).read())]
    procs, outs = [], []
    for src in srcs:
        out = os.path.join(outdir, os.path.basename(src)[:-4] + '.s')
        outs.append(out)
        procs.append(subprocess.Popen([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DIMM_SOURCE_DIGEST="audit"',
                                       '--cuda-device-only', '-S', src, '-o', out], stderr=subprocess.DEVNULL))
        while sum(p.poll() is None for p in procs) >= (os.cpu_count() or 4):
            procs[0].wait() if procs[0].poll() is None else next(p for p in procs if p.poll() is None).wait()
    for p, src in zip(procs, srcs):
        if p.wait() != 0:
            raise RuntimeError('hipcc -S failed for ' + src)
    return outs


def main(argv):
    with tempfile.TemporaryDirectory() as tmp:
        files = argv if argv else compile_all(tmp)
        bad = 0
        for f in files:
            found, n_asm, n_vmem = audit(os.path.basename(f), open(f).read())
            print('%-24s %5d asm blocks, %5d VMEM instructions inside them, %d findings' % (os.path.basename(f), n_asm, n_vmem, len(found)))
            for line in found:
                print('  ' + line)
            bad += len(found)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
