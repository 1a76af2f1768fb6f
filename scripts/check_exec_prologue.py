"""Scan AMDGPU assembly (hipcc -S --cuda-device-only) for register copies the allocator placed at the top of a join block BEFORE the
`s_or_b64 exec, exec, ...` that restores the lanes of the enclosing region: such a copy (v_accvgpr_write/read, v_mov) runs for the
lanes of the inner region only (or for none, when the block is reached with exec = 0), and the value it was meant to carry is lost
in the other lanes -- seen on the box as a lane pair kernel whose hand-off pointer was valid in even lanes only.
usage: check_exec_prologue.py file.s [...]   exit code 1 if anything is found."""
import re, sys

VALU = re.compile(r'^(v_|global_|buffer_|ds_|flat_|scratch_)')
COPIES = ('v_accvgpr_write_b32', 'v_accvgpr_read_b32', 'v_accvgpr_mov_b32')      # (v_mov copies in this position are also the program's own: ends of `then` blocks)
LANEOPS = ('v_writelane_b32', 'v_readlane_b32', 'v_readfirstlane_b32')
import os
ALLOW = ()          # (kernel-name substring, block label or '*'): blocks reviewed by hand and found benign -- none today
# (PDMP_EXEC_PROLOGUE_ALLOW="kernel-substring:label,..." adds entries for one build: a reviewed block while the list above is being updated)
ALLOW += tuple(tuple(e.split(':', 1)) for e in os.environ.get('PDMP_EXEC_PROLOGUE_ALLOW', '').split(',') if ':' in e)
bad = 0
for fn in sys.argv[1:]:
    kernel, block, pending = None, None, []
    for ln, raw in enumerate(open(fn), 1):
        code = raw.split(';')[0].strip()
        if not code:
            continue
        if code.endswith(':'):
            if code.startswith('_Z'):
                kernel = code[:-1]
            block, pending = code[:-1], []
            continue
        if code.startswith('.'):
            continue
        if pending is None:
            continue
        op = code.split()[0]
        if op in ('s_or_b64',) and re.match(r's_or_b64\s+exec,\s*exec,', code):
            # ANY VGPR <-> AGPR copy in front of the restore is flagged (fail closed): a join block's prologue runs with the inner region's
            # exec mask, so a copy there reaches only some lanes whatever else the scheduler hoisted next to it (round 3's version required
            # the prologue to hold copies ONLY and would have missed a prologue that mixes them with a v_mov or a load).  A block that is
            # known to be benign goes on ALLOW below, by kernel-name substring and block label.
            copies = [pc for pc in pending if pc[1].split()[0] in COPIES]
            if copies and not any(k in (kernel or '') and (b == '*' or b == block) for k, b in ALLOW):
                try:
                    for (l, c) in copies:
                        print(f"{fn}:{l}: [{kernel}] {block}: `{c}` before `{code}`")
                except BrokenPipeError:
                    pass
                bad += len(copies)
            pending = None            # only the block's prologue matters
            continue
        if op.startswith('s_') and 'exec' in code:
            pending = None            # another exec write / read (s_and_saveexec, s_mov exec, ...): the prologue of the join ends here;
            continue                  # what follows runs under a mask the program itself has just set
        if op.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_setpc')):
            pending = None
            continue
        if VALU.match(op) and op not in LANEOPS:
            pending.append((ln, code))
sys.exit(1 if bad else 0)
