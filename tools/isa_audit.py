"""Which kernels' device code changed between two states of the tree?  Compiles every csrc/*.hip of a git revision (checked out
into a scratch directory) and of the working tree to gfx950 assembly with the Makefile's flags, splits the assembly per kernel,
drops what is not an instruction (directives, comments, label numbering) and hashes the rest.  A kernel whose hash equals the one
of a revision that passed the GPU tests is the code that ran there, instruction for instruction; a kernel whose hash differs
has not run on hardware, whatever the source diff looks like.
usage: python tools/isa_audit.py <revision> [<revision-or-WORKTREE>] [--filter substring] [--defines "-DX ..."]
       python tools/isa_audit.py --manifest out.json label1=rev1 label2=rev2 ...   (tests/golden/isa_manifest.json)
"""
import argparse
import hashlib
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "--cuda-device-only", "-S", "-w"]


def checkout(rev, dst):
    subprocess.run(f"git -C {ROOT} archive {rev} rlipv2_amd/csrc include | tar -x -C {dst}", shell=True, check=True)
    return dst


def compile_tree(tree, defines):
    src = os.path.join(tree, "rlipv2_amd", "csrc")
    files = sorted(f for f in os.listdir(src) if f.endswith(".hip"))
    out = {}

    def one(f):
        s = os.path.join(tree, f + ".s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *defines, os.path.join(src, f), "-o", s],
                           capture_output=True, text=True, cwd=src)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-2000:])
            raise SystemExit(f"hipcc failed on {f} in {tree}")
        return f, s

    with ThreadPoolExecutor(4) as ex:
        for f, s in ex.map(one, files):
            out[f] = s
    return out


def kernels(path):
    """{mangled name: (instruction count, hash)} of one assembly file."""
    res, cur, name = {}, None, None
    for l in open(path):
        l = l.rstrip("\n")
        m = re.match(r"^(_Z\S+):", l)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is None:
            continue
        if l.startswith(".Lfunc_end"):
            res[name] = (len(cur), hashlib.sha1("\n".join(cur).encode()).hexdigest()[:12])
            cur = None
            continue
        t = l.split(";")[0].strip()
        if not t or t.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", t):
                cur.append("L:")                           # a branch target, whatever its number
            continue
        t = re.sub(r"\.LBB\d+_(\d+)", r".LBB_\1", t)       # the function index in a label changes when a file gains a kernel
        if t.startswith("s_load_dword"):                   # kernarg offsets (a by-value struct argument that grew a field)
            t = re.sub(r"(0x[0-9a-f]+|offset:0x[0-9a-f]+)\s*$", "OFF", t)
        cur.append(t)
    return res


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, r))


def audit(tree, defines):
    out = {}
    for f, s in compile_tree(tree, defines).items():
        for k, v in kernels(s).items():
            out[(f, k)] = v
    return out


def tree_of(rev, dst):
    """a revision (or WORKTREE) unpacked into `dst`"""
    if rev == "WORKTREE":
        subprocess.run(f"mkdir -p {dst}/rlipv2_amd && cp -r {ROOT}/rlipv2_amd/csrc {dst}/rlipv2_amd/ && rm -rf {dst}/rlipv2_amd/csrc/_obj "
                       f"&& cp -r {ROOT}/include {dst}/", shell=True, check=True)
    else:
        checkout(rev, dst)
    return dst


def hashes_of(rev, defines=()):
    """{(file, mangled kernel name): (instruction count, hash)} of a revision or of the working tree"""
    with tempfile.TemporaryDirectory() as t:
        return audit(tree_of(rev, t), list(defines))


def manifest(revisions):
    """Hardware status of every kernel of the working tree: for each kernel its hash and the EARLIEST-listed revision among
    `revisions` = [(label, revision), ...] whose build contains a kernel with exactly that instruction stream (under any
    name: template arguments get renamed), else "never_run"."""
    known = [(label, {v[1] for v in hashes_of(rev).values()}) for label, rev in revisions]
    cur = hashes_of("WORKTREE")
    dm = demangle(sorted({k for _, k in cur}))
    out = {}
    for (f, k), (n, h) in sorted(cur.items()):
        status = next((label for label, hs in known if h in hs), "never_run")
        out[f"{f}::{dm[k]}"] = {"instructions": n, "hash": h, "hardware": status}
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--manifest":
        import json
        revs = [tuple(a.split("=", 1)) for a in sys.argv[3:]]
        m = manifest(revs)
        json.dump({"revisions": dict(revs), "kernels": m}, open(sys.argv[2], "w"), indent=1, sort_keys=True)
        counts = {}
        for v in m.values():
            counts[v["hardware"]] = counts.get(v["hardware"], 0) + 1
        print(counts)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("a")
    ap.add_argument("b", nargs="?", default="WORKTREE")
    ap.add_argument("--filter", default="")
    ap.add_argument("--defines", default="")
    a = ap.parse_args()
    defines = a.defines.split()
    trees = []
    with tempfile.TemporaryDirectory() as ta, tempfile.TemporaryDirectory() as tb:
        for rev, t in ((a.a, ta), (a.b, tb)):
            if rev == "WORKTREE":
                subprocess.run(f"mkdir -p {t}/rlipv2_amd && cp -r {ROOT}/rlipv2_amd/csrc {t}/rlipv2_amd/ && rm -rf {t}/rlipv2_amd/csrc/_obj "
                               f"&& cp -r {ROOT}/include {t}/", shell=True, check=True)
            else:
                checkout(rev, t)
            trees.append(audit(t, defines))
    A, B = trees
    dm = demangle(sorted({k for _, k in list(A) + list(B)}))
    same = changed = 0
    rows = []
    hashes_a = {v[1]: k for k, v in A.items()}
    hashes_b = {v[1] for v in B.values()}
    for key in sorted(set(A) | set(B)):
        f, k = key
        name = dm[k]
        if a.filter not in name:
            continue
        if key not in A:
            if B[key][1] in hashes_a:                      # same instructions under another name (template arguments changed)
                same += 1
                rows.append(f"RENAMED  {f:24s} {B[key][0]:6d}          {name[:110]}  ==  {dm[hashes_a[B[key][1]][1]][:90]}")
            else:
                rows.append(f"NEW      {f:24s} {B[key][0]:6d}          {name[:150]}")
        elif key not in B:
            if A[key][1] not in hashes_b:
                rows.append(f"GONE     {f:24s} {A[key][0]:6d}          {name[:150]}")
        elif A[key][1] != B[key][1]:
            changed += 1
            rows.append(f"CHANGED  {f:24s} {A[key][0]:6d} -> {B[key][0]:6d} {name[:150]}")
        else:
            same += 1
    print(f"# device code of {a.a} vs {a.b}: {same} kernels identical, {changed} changed, "
          f"{sum(r.startswith('NEW') for r in rows)} new, {sum(r.startswith('GONE') for r in rows)} gone")
    print("\n".join(rows))


if __name__ == "__main__":
    main()
