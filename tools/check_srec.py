"""Checks the hand-issued scalar record loads of the fast kernels in the compiler's output (mil_device.hpp: srec_load / srec_wait).

srec_load issues `s_load_dwordx8` by inline asm, so the compiler does not know that its destination registers are still
being written; srec_wait (`s_waitcnt lgkmcnt(0)`) comes before the source touches them.  What the source cannot promise is
that the COMPILER leaves them alone in between (a copy, a spill into a VGPR lane): this script proves it on the assembly.
Per kernel it builds the control-flow graph of the listing and runs a forward data-flow: SREC_LOAD adds its destination
SGPRs to the pending set, SREC_WAIT empties it, a block starts with the union over its predecessors; an instruction that
names a pending SGPR (read or write) is an error, and so is a load whose destination overlaps its own address pair (a
replayed load would read a clobbered address).

usage: python tools/check_srec.py [listing.s]     (without an argument: compiles mil_fast.hip with -S first)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'renderer_amd', 'csrc')
SREG = re.compile(r'\bs(\d+)\b|\bs\[(\d+):(\d+)\]')


def listing():
    out = os.path.join(tempfile.mkdtemp(prefix='mil_srec_'), 'fast.s')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-fPIC', '-std=c++17', '--offload-arch=gfx950', '-Wno-unused-function',
                           '-ffp-contract=on', '-S', '--cuda-device-only', '-I', CSRC, os.path.join(CSRC, 'mil_fast.hip'), '-o', out],
                          stderr=subprocess.DEVNULL)
    return out


def sregs(text):
    regs = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            regs.add(int(m.group(1)))
        else:
            regs.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return regs


def kernels(path):
    name, body = None, []
    for ln in open(path):
        m = re.match(r'^(_Z\w+):\s', ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(ln.rstrip('\n'))
            if ln.startswith('.Lfunc_end'):
                yield name, body
                name = None


def check(name, body):
    # basic blocks: a label starts one, a branch or s_endpgm ends one
    blocks, cur, label_of = [], None, {}
    def start(label=None):
        nonlocal cur
        cur = dict(label=label, ins=[], succ=[], fall=True)
        blocks.append(cur)
        if label:
            label_of[label] = len(blocks) - 1
    start()
    for ln in body:
        t = ln.split(';')[0].strip() if 'SREC_' not in ln else ln.strip()
        m = re.match(r'^(\.LBB\w+):', ln)
        if m:
            start(m.group(1))
            continue
        if not t or t.startswith('.') or t.startswith(';'):
            continue
        cur['ins'].append(t)
        op = t.split()[0]
        if op == 's_branch':
            cur['succ'].append(t.split()[1]); cur['fall'] = False; start()
        elif op.startswith('s_cbranch'):
            cur['succ'].append(t.split()[1]); start()
        elif op in ('s_endpgm', 's_setpc_b64'):
            cur['fall'] = False; start()
    succ = []
    for i, b in enumerate(blocks):
        out = [label_of[l] for l in b['succ'] if l in label_of]
        if b['fall'] and i + 1 < len(blocks):
            out.append(i + 1)
        succ.append(out)
    pend_in = [set() for _ in blocks]
    errors, loads = set(), set()
    work = list(range(len(blocks)))
    while work:
        i = work.pop()
        pend = set(pend_in[i])
        for t in blocks[i]['ins']:
            if 'SREC_LOAD' in t:
                m = re.match(r's_load_dwordx8\s+s\[(\d+):(\d+)\],\s*(\S+),', t)
                dst = set(range(int(m.group(1)), int(m.group(2)) + 1))
                hit = (sregs(m.group(3)) & pend) | (dst & pend) | (sregs(m.group(3)) & dst)     # the last: destination over its own address
                if hit:
                    errors.add((blocks[i]['label'], t, tuple(sorted(hit))))
                pend |= dst
                loads.add((i, t))
            elif 'SREC_WAIT' in t:
                pend = set()
            elif pend:
                hit = sregs(t) & pend
                if hit:
                    errors.add((blocks[i]['label'], t, tuple(sorted(hit))))
        for j in succ[i]:
            if not pend <= pend_in[j]:
                pend_in[j] |= pend
                work.append(j)
    return len(loads), sorted(errors, key=str)


def main(path=None):
    path = path or listing()
    total, bad = 0, 0
    for name, body in kernels(path):
        loads, errors = check(name, body)
        total += loads
        if loads:
            print(f'{name}: {loads} hand-issued loads, {len(errors)} uses of a pending register')
        for lab, t, hit in errors:
            bad += 1
            print(f'    {lab}: `{t}` touches pending s{list(hit)}')
    print(f'{total} loads checked, {bad} errors')
    return total, bad


if __name__ == '__main__':
    total, bad = main(sys.argv[1] if len(sys.argv) > 1 else None)
    sys.exit(1 if bad or not total else 0)
