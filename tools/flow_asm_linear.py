#!/usr/bin/env python3
"""Whole-kernel companion of flow_asm_audit.py: the first round and the tail of the hand-counted pipelines, and every block
the compiler lays out of line, not only the steady loop.

A hand-issued load (inline asm `buffer_load_dword* ..., sN offen`: destination registers the compiler allocates but does
not track) is IN FLIGHT from its issue until a `s_waitcnt vmcnt(N)` with N <= the vector memory operations issued behind
it (loads return in order).  Until then no instruction may read or write its destination registers: a compiler short of
registers parks a set in AGPRs or re-uses a register as a temporary, and the data lands on top of it (seen in round 5 on
36-entry rows: wrong values on the GPU, while the steady loop was clean).

The check is a forward dataflow over the kernel's control-flow graph (basic blocks from the labels and s_branch /
s_cbranch instructions of the gfx950 assembly): the state is, per load, the smallest number of younger vector memory
operations over all paths; blocks are re-visited until nothing changes.
usage: python tools/flow_asm_linear.py [source.hip] [name-filter]      exit code 0 = clean; one line per kernel."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from flow_asm_audit import regs_of, all_regs, HIPCC, ROOT  # noqa: E402

VMEM = re.compile(r"(buffer|global|flat|scratch)_(load|store|atomic)")
HAND = re.compile(r"buffer_load_dword\w* .*, s\d+ offen$")
CAP = 64


def blocks_of(lines):
    """-> (list of blocks [(label or None, [instructions])], label -> block index)"""
    blocks, cur, lab = [], [], None
    for l in lines:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", t)
            if m:
                if cur or lab is not None:
                    blocks.append((lab, cur))
                cur, lab = [], m.group(1)
            continue
        if t.endswith(":"):
            continue
        cur.append(t)
        if re.match(r"s_(c?branch|endpgm|setpc)", t):
            blocks.append((lab, cur))
            cur, lab = [], None
    if cur or lab is not None:
        blocks.append((lab, cur))
    index = {lab: i for i, (lab, _) in enumerate(blocks) if lab is not None}
    return blocks, index


def successors(blocks, index, i):
    ins = blocks[i][1]
    last = ins[-1] if ins else ""
    m = re.match(r"s_(c?)branch\w*\s+(\.LBB\d+_\d+)", last)
    out = []
    if last.startswith("s_endpgm"):
        return out
    if m:
        if m.group(2) in index:
            out.append(index[m.group(2)])
        if m.group(1) == "c" and i + 1 < len(blocks):
            out.append(i + 1)
    elif i + 1 < len(blocks):
        out.append(i + 1)
    return out


def scan(lines):
    blocks, index = blocks_of(lines)
    n = len(blocks)
    dst = {}        # (block, pos) -> destination registers of a hand-issued load
    for b, (_, ins) in enumerate(blocks):
        for p, c in enumerate(ins):
            if HAND.match(c):
                dst[(b, p)] = regs_of(re.findall(r"v\[\d+:\d+\]|\bv\d+\b", c)[0])
    state_in = [None] * n       # dict load -> younger count (min over paths)
    state_in[0] = {}
    work = [0]
    bad = {}
    while work:
        b = work.pop()
        st = dict(state_in[b])
        for p, c in enumerate(blocks[b][1]):
            m = re.match(r"s_waitcnt .*vmcnt\((\d+)\)", c)
            if m:
                nwait = int(m.group(1))
                st = {k: v for k, v in st.items() if v < nwait}
                continue
            if st:
                touched = all_regs(c)
                if touched:
                    for k in st:
                        if touched & dst[k] and k != (b, p):
                            bad.setdefault(k, (b, p, c))
            if VMEM.match(c):
                st = {k: min(CAP, v + 1) for k, v in st.items()}
                if (b, p) in dst:
                    st[(b, p)] = 0
        for s in successors(blocks, index, b):
            if state_in[s] is None:
                state_in[s] = dict(st)
                work.append(s)
            else:
                merged = dict(state_in[s])
                changed = False
                for k, v in st.items():
                    if k not in merged or v < merged[k]:
                        merged[k] = v
                        changed = True
                if changed:
                    state_in[s] = merged
                    work.append(s)
    ninstr = sum(len(i) for _, i in blocks)
    return [(blocks[k[0]][1][k[1]], v[2], blocks[v[0]][0]) for k, v in sorted(bad.items())], ninstr, len(dst)


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "flow_inst.hip")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S", "-o", out, src] + os.environ.get("AUDIT_HIP_FLAGS", "").split()
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            print(r.stdout.decode(errors="replace")[-2000:])
            return 2
        text = open(out).read().split("\n")
    kernels, cur = {}, None
    for l in text:
        m = re.match(r"^(_ZN4amgh2bw1[78]gs_bw_(?:flow|relay)_kernel\w+):", l)
        if m:
            cur = m.group(1)
            kernels[cur] = []
        elif cur is not None:
            if l.strip().startswith(".Lfunc_end"):
                cur = None
            else:
                kernels[cur].append(l)
    if not kernels:
        print("no gs_bw_flow_kernel / gs_bw_relay_kernel instantiation found")
        return 2
    rc = 0
    for name, lines in sorted(kernels.items()):
        if flt and flt not in name:
            continue
        bad, n, nl = scan(lines)
        short = re.sub(r"^_ZN4amgh2bw1[78]gs_bw_(flow|relay)_kernelI", r"\1 ", name)[:30]
        if bad:
            rc = 1
            l, c, lab = bad[0]
            print(f"FAIL {short:30s} {len(bad)} of {nl} hand-issued loads touched in flight; first: {l[:56]}  <-  {c[:60]} (block {lab})")
        else:
            print(f"ok   {short:30s} {nl} hand-issued loads, {n} instructions")
    return rc


if __name__ == "__main__":
    sys.exit(main())
