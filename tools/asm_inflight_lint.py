#!/usr/bin/env python3
"""tools/asm_inflight_lint.py [file.hip ...] -- looks for the hazard behind round 3's three-plane bug in the ISA of every
kernel of a HIP file: a register that an inline-asm `global_load_*` has written (the compiler does not see that the
value is still in flight) must not be READ by a compiler-generated instruction before the next `s_waitcnt vmcnt(0)`
the kernel executes -- a register copy at a loop's latch, a spill, an operand reuse.  A forward data flow over the kernel's basic blocks
(`s_waitcnt vmcnt(N)` lands all but the N youngest asm loads -- youngest = latest in the text, a heuristic that fits
the stagers' straight-line load sequences; compiler-visible loads between them are not counted).  Exit code 1 if there are hits.
usage: python tools/asm_inflight_lint.py [resampler_amd/csrc/fir_split.hip]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "--cuda-device-only", "-S"]
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def transfer(state, ins, report=None):
    """One instruction on the set of in-flight registers (register -> line of its asm load)."""
    no, t, in_asm = ins
    op, _, rest = t.partition(" ")
    if op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", rest)
        if m:   # in-order counter: all but the N youngest loads have landed (youngest = latest in the text: a heuristic)
            n = int(m.group(1))
            keep = set(sorted(set(state.values()))[-n:]) if n else set()
            return {r: ln for r, ln in state.items() if ln in keep}
        return dict(state)
    ops = [o.strip() for o in rest.split(",")]
    st = dict(state)
    if in_asm and op.startswith("global_load"):
        for r in regs(ops[0]):
            st[r] = no
        return st
    if op.startswith(("global_store", "scratch_store", "ds_write", "buffer_store", "flat_store", "global_atomic", "ds_add", "ds_max")):
        srcs = set().union(*[regs(o) for o in ops]) if ops else set()
        dst = set()
    else:
        src_ops = ops[1:]
        if op.startswith("v_pk_") and op.endswith("_f32"):
            # a packed-f32 source is a register pair of which op_sel / op_sel_hi say which halves are read: [0, 0] = the low
            # register twice (a scalar broadcast), [1, 1] = the high one twice
            sel = re.search(r"op_sel:\[([01,]+)\]", rest)
            hi = re.search(r"op_sel_hi:\[([01,]+)\]", rest)
            sel = [int(v) for v in sel.group(1).split(",")] if sel else [0, 0, 0]
            hi = [int(v) for v in hi.group(1).split(",")] if hi else [1, 1, 1]
            src_ops = [re.sub(r"\s*op_sel.*$", "", o) for o in src_ops]
            picked = []
            for k, o in enumerate(src_ops[:3]):
                m = re.match(r"^-?\|?v\[(\d+):(\d+)\]\|?$", o.strip())
                if m and k < len(sel) and k < len(hi) and sel[k] == hi[k]:
                    picked.append("v%d" % (int(m.group(1)) + sel[k]))
                else:
                    picked.append(o)
            src_ops = picked
        srcs = set().union(*[regs(o) for o in src_ops]) if src_ops else set()
        dst = regs(ops[0]) if ops else set()
    bad = sorted(r for r in srcs if r in st)
    if bad and report is not None:
        report.append((no, t, [(r, st[r]) for r in bad[:4]]))
    for r in dst - srcs:   # another write ends the loaded value's life in that register
        st.pop(r, None)
    return st


def lint_kernel(body):
    """body: [(line number, text, inside an asm statement)], labels as ('label', name).  Forward data flow over the basic
    blocks (in-flight set at a block's entry = union over its predecessors), then one reporting pass."""
    blocks, cur, name = {}, [], "entry"
    order = []
    for item in body:
        if item[0] == "label":
            blocks[name] = cur
            order.append(name)
            name, cur = item[1], []
        else:
            cur.append(item)
            if item[1].startswith(("s_cbranch", "s_branch")):   # a branch ends its block: what follows is a block of its own
                blocks[name] = cur
                order.append(name)
                name, cur = "@%d" % item[0], []
    blocks[name] = cur
    order.append(name)
    succ = {}
    for i, n in enumerate(order):
        ins = blocks[n]
        out = []
        fall = True
        for (_, t, _) in ins:
            op, _, rest = t.partition(" ")
            if op == "s_branch":
                out.append(rest.strip())
                fall = False
            elif op.startswith("s_cbranch"):
                out.append(rest.strip())
            elif op == "s_endpgm":
                fall = False
        if fall and i + 1 < len(order):
            out.append(order[i + 1])
        succ[n] = [o for o in out if o in blocks]
    entry = {n: {} for n in order}
    work = list(order)
    while work:
        n = work.pop(0)
        st = entry[n]
        for ins in blocks[n]:
            st = transfer(st, ins)
        for m in succ[n]:
            merged = dict(entry[m])
            changed = False
            for r, ln in st.items():
                if r not in merged:
                    merged[r] = ln
                    changed = True
            if changed:
                entry[m] = merged
                if m not in work:
                    work.append(m)
    hits = []
    for n in order:
        st = entry[n]
        for ins in blocks[n]:
            st = transfer(st, ins, hits)
    return hits


def lint(path):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [path, "-o", asm], check=True, cwd=os.path.dirname(path),
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lines = open(asm).read().splitlines()
    hits, kernel, body, in_asm, n_kernels, n_loads = [], None, [], False, 0, 0
    for no, line in enumerate(lines, 1):
        t = line.strip()
        if re.match(r"^_Z\w+:", t):
            kernel, body, in_asm = t.split(":")[0], [], False
            n_kernels += 1
            continue
        if kernel is None or not t:
            continue
        if "#ASMSTART" in t:
            in_asm = True
            continue
        if "#ASMEND" in t:
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            body.append(("label", m.group(1)))
            continue
        if t.startswith(";") or t.startswith("."):
            continue
        t = t.split(";")[0].strip()
        if in_asm and t.startswith("global_load"):
            n_loads += 1
        body.append((no, t, in_asm))
        if t.startswith("s_endpgm"):
            hits += [(kernel,) + h for h in lint_kernel(body)]
            kernel = None
    return n_kernels, n_loads, hits


def main():
    files = sys.argv[1:] or [os.path.join(ROOT, "resampler_amd", "csrc", "fir_split.hip")]
    rc = 0
    for f in files:
        n_k, n_l, hits = lint(os.path.abspath(f))
        print(f"{os.path.basename(f)}: {n_k} kernels, {n_l} inline-asm loads, {len(hits)} reads of a register in flight")
        for k, no, t, regs_ in hits[:40]:
            print(f"  {k[:70]}  line {no}: {t}   <- " + ", ".join(f"v{r} loaded at line {ln}" for r, ln in regs_))
        rc |= 1 if hits else 0
    return rc


if __name__ == "__main__":
    sys.exit(main())
