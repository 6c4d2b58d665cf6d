#!/usr/bin/env python3
"""tools/isa_census.py <file.hip> <kernel substring> [hipcc flags] -- what a kernel's main loop is made of, by instruction
class, from its ISA (device-only compile to assembly; CPU only).

The instructions counted are those of the basic blocks LLVM marks as part of the kernel's outermost loop with the most
instructions (the walk over a run of blocks in the FFT kernels), less the rarely taken paths a source file marks with an
`asm volatile("; ...")` comment (from the comment to the next unconditional branch).  Every loop-body instruction is
executed once per trip in these kernels (their passes are unrolled; trips of a pass that only part of the wave takes
are still issued), so the static count IS the count per stream-block, which `SQ_INSTS_VALU / (stream-blocks incl. halo)`
of a counter pass can be held against.

Classes (VERDICT r05 item 1 asked for this table): packed f32 arithmetic (`v_pk_fma/mul/add_f32`: butterflies, twiddle
and filter products, scaling), scalar f32 arithmetic, address / integer arithmetic, compares + selects, moves
(`v_mov`, `v_accvgpr_*`, `v_pk_mov`), conversions, cross-lane (DPP, readlane, ballot ...), LDS reads / stores, global
loads / stores, scalar ALU, waits + nops, branches."""
import os
import re
import subprocess
import sys
from collections import Counter, OrderedDict

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "resampler_amd", "csrc")

CLASSES = OrderedDict([
    ("packed f32 arithmetic", r"^v_pk_(fma|mul|add)_f32"),
    ("scalar f32 arithmetic", r"^v_(fma|fmac|mul|add|sub|subrev|mac|mad|max|min|rcp|rsq|sqrt|exp|log|ldexp|frexp\w*|fma_mix|fract|trunc|floor|rndne)_(f32|legacy_f32)|^v_fma_mix"),
    ("conversions", r"^v_cvt_"),
    ("compares + selects", r"^v_cmp|^v_cndmask|^v_cmpx"),
    ("moves", r"^v_mov_b32|^v_accvgpr|^v_pk_mov|^v_swap|^v_mov_b64"),
    ("cross-lane", r"dpp|^v_readlane|^v_readfirstlane|^v_writelane|^ds_bpermute|^ds_permute|^ds_swizzle|^v_permlane|^v_bfrev"),
    ("address / integer arithmetic", r"^v_(add|sub|subrev|mul|mad|lshl|lshr|ashr|and|or|xor|bfe|bfi|not|lshlrev|lshrrev|ashrrev|mul_lo|mul_hi|mul_u32|add3|lshl_add|lshl_or|and_or|or3|xad|min_u|max_u|min_i|max_i|alignbit|perm|mbcnt|addc|subb)[a-z0-9_]*"),
    ("LDS reads", r"^ds_read"),
    ("LDS stores", r"^ds_write"),
    ("global loads", r"^global_load|^flat_load|^buffer_load|^scratch_load"),
    ("global stores", r"^global_store|^flat_store|^buffer_store|^scratch_store|^global_atomic"),
    ("waits + nops", r"^s_waitcnt|^s_nop|^s_sleep|^s_setprio|^s_barrier"),
    ("branches", r"^s_cbranch|^s_branch"),
    ("scalar ALU / memory", r"^s_"),
])


def compile_asm(src, flags):
    out = "/tmp/isa_census_%d.s" % os.getpid()
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950",
           "--cuda-device-only", "-S", src, "-o", out] + flags
    subprocess.run(cmd, cwd=HERE, check=True, capture_output=True)
    text = open(out).read().splitlines()
    os.remove(out)
    return text


def kernel_lines(text, needle):
    start = None
    for i, ln in enumerate(text):
        if start is None:
            if ln.startswith("_Z") and ln.rstrip().split(":")[0].find("_Z") == 0 and needle_matches(ln, needle):
                start = i
        elif ln.strip().startswith(".Lfunc_end") or ln.strip().startswith("s_endpgm") and False:
            return text[start:i]
        elif ln.lstrip().startswith(".section") and start is not None and i > start + 5:
            return text[start:i]
    return text[start:] if start is not None else []


def needle_matches(line, needle):
    name = line.split(":")[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout
    return all(part in dem for part in needle.split("&"))


def census(lines):
    # basic blocks with their loop header annotation
    blocks, cur = [], {"label": "entry", "header": None, "ins": []}
    for ln in lines:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", ln)
        if m:
            blocks.append(cur)
            hdr = None
            note = m.group(2) or ""
            mh = re.search(r"Header=(BB\d+_\d+) Depth=1", note)
            if mh:
                hdr = mh.group(1)
            elif "Loop Header: Depth=1" in note:
                hdr = m.group(1)[2:]
            cur = {"label": m.group(1), "header": hdr, "ins": []}
            continue
        s = ln.strip()
        if not s or s.startswith(".") or s.startswith(";;"):
            continue
        if s.startswith(";"):
            cur["ins"].append(("comment", s))
            continue
        if re.match(r"^[a-z_0-9]+(\s|$)", s):
            cur["ins"].append(("ins", s.split(";")[0].strip()))
    blocks.append(cur)
    loops = Counter()
    for b in blocks:
        if b["header"]:
            loops[b["header"]] += sum(1 for k, _ in b["ins"] if k == "ins")
    if not loops:
        raise SystemExit("no loop found")
    main = loops.most_common(1)[0][0]
    hot, rare = Counter(), Counter()
    skipping = False
    for b in blocks:
        if b["header"] != main:
            continue
        for kind, s in b["ins"]:
            if kind == "comment":
                if re.search(r"; a channel of this block", s):
                    skipping = True
                continue
            cls = "other"
            for name, pat in CLASSES.items():
                if re.search(pat, s):
                    cls = name
                    break
            (rare if skipping else hot)[cls] += 1
            if skipping and re.match(r"^s_branch", s):
                skipping = False
    return main, hot, rare


def main():
    src, needle = sys.argv[1], sys.argv[2]
    text = compile_asm(src, sys.argv[3:])
    lines = kernel_lines(text, needle)
    if not lines:
        raise SystemExit("kernel not found: " + needle)
    header, hot, rare = census(lines)
    name = subprocess.run(["c++filt", lines[0].split(":")[0]], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").replace("rsmp::", "")
    print("kernel: %s" % re.sub(r"\(.*\)$", "", name))
    print("main loop: %s; instructions per trip (hot path) / in rarely taken paths" % header)
    vec = 0
    for cls in list(CLASSES.keys()) + ["other"]:
        if hot[cls] or rare[cls]:
            print("  %-30s %6d %6d" % (cls, hot[cls], rare[cls]))
        if cls in ("packed f32 arithmetic", "scalar f32 arithmetic", "conversions", "compares + selects", "moves",
                   "cross-lane", "address / integer arithmetic"):
            vec += hot[cls]
    print("  %-30s %6d" % ("vector ALU total (hot path)", vec))
    print("  %-30s %6d" % ("LDS total (hot path)", hot["LDS reads"] + hot["LDS stores"]))


if __name__ == "__main__":
    main()
