"""Static check of device assembly for LDS results consumed before they arrived.

The kernels read LDS through inline assembly in places (ds_read_b64_tr_b16, ds_read_b128 with immediate offsets): the
compiler does not know those results are asynchronous, so the `s_waitcnt lgkmcnt(N)` in front of their first use is written
by hand.  A missing or too-lax wait reads a stale register -- silently, and only sometimes.  This walks the control-flow
graph of every kernel in a `.s` file with the queue of outstanding LGKM operations as the dataflow state and reports any
instruction that touches the destination of an LDS read that no wait has retired yet (compiler-generated reads go through the
same check).  Rules encoded (gfx9 family): LDS operations retire in issue order, `lgkmcnt(N)` leaves at most N outstanding;
scalar memory loads share the counter and may return out of order, so with one of them in flight only `lgkmcnt(0)` retires
anything; LDS stores and no-return atomics count too.  Join points take, per outstanding read, the smaller number of later operations of the two paths (a finite, monotone lattice).

usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S x.hip -o x.s; python tools/isa_hazard_check.py x.s [name filter]
exit status 1 if a hazard was found.
"""
import re
import subprocess
import sys

RET = re.compile(r"^ds_(read|bpermute|permute|swizzle|consume|append|ordered|\w+_rtn)")
VREG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            for k in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), k))
    return out


def kernels(path):
    lines = open(path).read().split("\n")
    out, cur, name = [], None, None
    for l in lines:
        m = re.match(r"^(_Z\S+):", l)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
                out.append((name, cur))
                cur = None
            else:
                cur.append(l)
    return out


def blocks_of(body):
    """-> list of blocks [(label or None, [(line_no, text)])], successors by index"""
    blocks, cur, label = [], [], None
    for i, l in enumerate(body):
        t = l.split(";")[0].strip()
        if not t:
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            if cur or label is not None:
                blocks.append((label, cur))
            cur, label = [], m.group(1)
            continue
        if t.startswith(".") or t.endswith(":"):
            continue
        cur.append((i, t))
        if t.startswith("s_branch") or t.startswith("s_cbranch") or t.startswith("s_endpgm"):
            blocks.append((label, cur))
            cur, label = [], None
    if cur or label is not None:
        blocks.append((label, cur))
    index = {lab: k for k, (lab, _) in enumerate(blocks) if lab}
    succ = []
    for k, (_, ins) in enumerate(blocks):
        last = ins[-1][1] if ins else ""
        s = []
        if last.startswith("s_endpgm"):
            pass
        elif last.startswith("s_branch"):
            s.append(index[last.split()[1]])
        elif last.startswith("s_cbranch"):
            s.append(index[last.split()[1]])
            if k + 1 < len(blocks):
                s.append(k + 1)
        elif k + 1 < len(blocks):
            s.append(k + 1)
        succ.append(s)
    return blocks, succ


def step(state, t, line_no, report):
    """state: {issue line: (frozenset dests, age = LGKM operations issued after it (capped), unordered)} -> new state"""
    op = t.split()[0]
    touched = regs(t[len(op):])
    if op.startswith("ds_") and RET.match(op):
        # another LDS read into the same register is no hazard (LDS returns in issue order: the later value lands later);
        # its ADDRESS operands are checked like any other use, and the older read no longer owns the register
        mine = regs(t[len(op):].split(",")[0])
        touched = regs(",".join(t[len(op):].split(",")[1:]))
        state = {at: (d - mine, a, u) for at, (d, a, u) in state.items()}
    for at, (dests, _, _) in state.items():
        hit = dests & touched
        if hit:
            report(line_no, t, at, hit)
    if op.startswith("s_waitcnt"):
        m = re.search(r"lgkmcnt\((\d+)\)", t)
        if m is None and re.match(r"s_waitcnt\s+(0x[0-9a-f]+|\d+)\s*$", t):       # raw immediate
            n = (int(t.split()[1], 0) >> 8) & 0xf
        elif m is None:
            return state
        else:
            n = int(m.group(1))
        if n == 0:
            return {}
        return {at: e for at, e in state.items() if e[2] or e[1] < n}
    lds = op.startswith("ds_")
    smem = op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_scratch_load")
    if not (lds or smem):
        return state
    out = {at: (d, min(a + 1, 15), u or smem) for at, (d, a, u) in state.items()}
    if lds and RET.match(op):
        out[line_no] = (frozenset(regs(t[len(op):].split(",")[0])), 0, False)
    return out


def merge(a, b):
    """join of two states (None = unreached); returns (state, changed relative to a)"""
    if a is None:
        return dict(b), True
    out, changed = dict(a), False
    for at, (d, age, u) in b.items():
        if at not in out:
            out[at] = (d, age, u)
            changed = True
        else:
            d0, a0, u0 = out[at]
            n = (d0, min(a0, age), u0 or u)
            if n != out[at]:
                out[at] = n
                changed = True
    return out, changed


def check(name, body):
    blocks, succ = blocks_of(body)
    found = {}

    def report(line_no, t, at, hit):
        found[(line_no, at)] = (t, sorted(hit))

    states = [None] * len(blocks)
    states[0] = {}
    work = [0]
    while work:
        k = work.pop()
        q = states[k]
        for line_no, t in blocks[k][1]:
            q = step(q, t, line_no, report)
        for s in succ[k]:
            states[s], changed = merge(states[s], q)
            if changed and s not in work:
                work.append(s)
    return found


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    bad = 0
    for name, body in kernels(path):
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if flt not in dem:
            continue
        found = check(name, body)
        reads = sum(1 for l in body if RET.match(l.strip()))
        print(f"{'HAZARD' if found else 'ok    '}  {dem[:140]}  ({reads} LDS reads)")
        for (line_no, at), (t, hit) in sorted(found.items()):
            bad += 1
            print(f"        line {line_no}: `{t[:80]}` touches {hit[:4]} of the LDS read issued at line {at}: `{body[at].strip()[:60]}`")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
