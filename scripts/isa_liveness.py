#!/usr/bin/env python3
"""VGPR liveness of one kernel in a hipcc -S listing: where the register peak of a kernel is.

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only file.hip -o file.s
    python3 scripts/isa_liveness.py file.s <mangled kernel name> [top]

Basic blocks from the labels and the s_branch / s_cbranch instructions, defs and uses from the operand positions (approximate: an
instruction's first vector operand is its definition unless it is a store, a compare or a lane read; v_fmac / writelane / DPP moves /
permlane swaps also read it), backward dataflow to a fixpoint.  Prints the `top` program points with the most live VGPRs."""
import re, sys

def vregs(tok):
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1): out.append(set(range(int(m.group(1)), int(m.group(2)) + 1)))
        else: out.append({int(m.group(3))})
    return out

def main():
    fn, name = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    lines = open(fn).read().split('\n')
    start = [i for i, l in enumerate(lines) if l.startswith(name + ':')][0]
    end = [i for i in range(start, len(lines)) if lines[i].strip().startswith('.end_amdhsa_kernel')][0]
    ins = []            # (text, label or None)
    for l in lines[start + 1:end]:
        t = l.split(';')[0].rstrip()
        if not t.strip(): continue
        if re.match(r'^\.LBB\d+_\d+:', t): ins.append(('', t.split(':')[0])); continue
        if t.startswith('\t') and not t.strip().startswith('.'): ins.append((t.strip(), None))
    label_at = {lab: i for i, (t, lab) in enumerate(ins) if lab}
    n = len(ins)
    succ = [[] for _ in range(n)]
    for i, (t, lab) in enumerate(ins):
        op = t.split()[0] if t else ''
        if op == 's_branch': succ[i] = [label_at[t.split()[1]]]
        elif op.startswith('s_cbranch'): succ[i] = [label_at[t.split()[1]]] + ([i + 1] if i + 1 < n else [])
        elif op in ('s_endpgm',): succ[i] = []
        else: succ[i] = [i + 1] if i + 1 < n else []
    defs, uses = [set() for _ in range(n)], [set() for _ in range(n)]
    for i, (t, lab) in enumerate(ins):
        if not t: continue
        op = t.split()[0]
        ops = t[len(op):].split(',')
        groups = [vregs(o) for o in ops]
        first = set().union(*groups[0]) if groups and groups[0] else set()
        rest = set()
        for g in groups[1:]:
            for r in g: rest |= r
        nodef = op.startswith(('global_store', 'scratch_store', 'ds_write', 'ds_store', 'buffer_store', 'flat_store', 'v_cmp', 'v_readlane', 'v_readfirstlane', 'global_atomic', 's_'))
        if nodef: uses[i] = first | rest
        else:
            defs[i] = first; uses[i] = set(rest)
            if op.startswith(('v_fmac', 'v_mac', 'v_writelane', 'v_permlane')) or 'dpp' in t or 'row_' in t or 'quad_perm' in t: uses[i] |= first
            if op.startswith('v_permlane') and len(groups) > 1:
                for r in groups[1]: defs[i] |= r
    live_in = [set() for _ in range(n)]
    changed = True
    while changed:
        changed = False
        for i in range(n - 1, -1, -1):
            out = set()
            for j in succ[i]: out |= live_in[j]
            new = uses[i] | (out - defs[i])
            if new != live_in[i]: live_in[i] = new; changed = True
    order = sorted(range(n), key=lambda i: -len(live_in[i]))
    print('instructions', n, ' max live', len(live_in[order[0]]))
    shown = []
    for i in order:
        if all(abs(i - j) > 40 for j in shown):
            shown.append(i)
            print(f'--- at {i}: {len(live_in[i])} live')
            for j in range(max(0, i - 3), min(n, i + 4)): print('   ', j, ins[j][0] or ins[j][1])
        if len(shown) >= top: break
    hist = [len(live_in[i]) for i in range(n)]
    print('profile (every 100 instructions: max live):', [max(hist[i:i + 100]) for i in range(0, n, 100)])

if __name__ == '__main__':
    main()
