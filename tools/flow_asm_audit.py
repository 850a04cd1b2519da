#!/usr/bin/env python3
"""Audit of the hand-counted memory pipeline of gs_bw_flow_kernel (csrc/hip/gs_flow.hpp).

The walker's loads are inline asm ("=v" destinations the compiler allocates, waited for by hand-counted `s_waitcnt
vmcnt(N)`): what the compiler does not know it cannot protect.  This script compiles the kernel to gfx950 assembly and
checks, for every instantiation it finds:
  1. between a register set's loads and the wait that covers them, no instruction reads or writes a destination register
     of those loads (a compiler copy or a re-used temporary there would see — or clobber — data that has not landed);
  2. the steady-state loop holds exactly the hand-written waits: D of them, all vmcnt((D-1)(L+2)), and no other vmcnt wait
     (a compiler-inserted one would mean it tracks a load of its own inside the pipeline);
  3. every asm buffer load of the loop carries an SGPR offset and the loop has no waterfall (v_readfirstlane) — the
     descriptors stayed wave-uniform;
  4. the kernel does not spill (scratch accesses are memory operations the hand-written counts do not know).
usage: python tools/flow_asm_audit.py [source.hip]     exit code 0 = clean; prints one line per kernel."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def regs_of(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def all_regs(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", line):
        out |= regs_of(tok)
    return out


def is_code(l):
    t = l.strip()
    return bool(t) and not t.startswith(";") and not t.startswith(".") and not t.endswith(":")


def audit_kernel(name, lines):
    """lines: the kernel's assembly.  Returns (ok, message)."""
    code = [l.strip() for l in lines if is_code(l)]
    waits = [(i, int(m.group(1))) for i, l in enumerate(code) for m in [re.match(r"s_waitcnt vmcnt\((\d+)\)$", l)] if m]
    if not waits:
        return False, "no vmcnt waits found"
    # the steady wait follows from the instantiation: L = value chunks + column chunks + b loads per step, S = 2 stores,
    # D register sets (gs_flow.hpp FlowOps / FlowDepth) -> vmcnt((D - 1)(L + S))
    m = re.search(r"gs_bw_(flow|relay)_kernelI([df])Lb[01]ELb[01]ELi(\d+)ELi(\d+)E(?:Lb([01])E)?", name)
    if not m:
        return False, "cannot read the instantiation from the name"
    relay = m.group(1) == "relay"
    rb, maxk = (8 if m.group(2) == "d" else 4), int(m.group(3))
    vpc = 16 // rb
    L = (maxk + 2 + vpc - 1) // vpc + (maxk + 7) // 8 + 1
    dic = m.group(5) == "1"   # (FlowOpsD: the values come out of the block's dictionary in LDS)
    if dic:
        L = (maxk + 7) // 8 + 1
    depth = int(os.environ.get("BW_FLOW_DEPTH", "4"))
    nc = int(m.group(4))   # (relay kernels: the walker waves per block)
    if relay:   # gs_relay.hpp RelayDepth
        D = int(os.environ.get("BW_RELAY_DEPTH_SHORT", "3")) if maxk <= 12 else int(os.environ.get("BW_RELAY_DEPTH_LONG", "2"))
        if dic:
            D = int(os.environ.get("BW_RELAY_DICT_DEPTH", "3"))
    else:
        D = (min(depth, 3) if nc > 1 else depth) if maxk <= 6 else (min(depth, 4) if maxk <= 12 else 3)
        if dic:
            D = int(os.environ.get("BW_FLOW_DICT_DEPTH", "3"))
    steady = (D - 1) * (L + 2)
    idx = [i for i, n in waits if n == steady]
    loads = [i for i, l in enumerate(code) if re.match(r"buffer_load_dword", l)]
    if len(idx) < D + 1:
        return False, f"expected at least {D + 1} waits of vmcnt({steady}), found {len(idx)}"
    loop_waits = idx[-D:]
    lo, hi = loop_waits[0], None
    # the loop ends at the backward branch behind the last issue: the first s_cbranch/s_branch after the last loop load
    loop_loads = [i for i in loads if i > loop_waits[-1]]
    # loads of the last step's issue: contiguous group right after the last wait's step
    grp_end = None
    for a, b in zip(loop_loads, loop_loads[1:] + [10 ** 9]):
        if b - a > 40:
            grp_end = a
            break
    if grp_end is None:
        return False, "cannot delimit the loop"
    hi = grp_end + 1
    while hi < len(code) and not re.match(r"s_(c)?branch", code[hi]):
        hi += 1
    body = code[lo:hi + 1]
    msgs = []
    # 2. only the hand-written waits
    bw = [m.group(1) for l in body for m in [re.match(r"s_waitcnt vmcnt\((\d+)\)", l)] if m]
    if len(bw) != D or any(int(x) != steady for x in bw):
        msgs.append(f"vmcnt waits in the loop: {bw} (expected {D} x {steady})")
    # 3. uniform descriptors
    # (the relay's hand-over polls read their LDS words through v_readfirstlane: not a waterfall — its buffer accesses are checked below)
    if not relay and any("v_readfirstlane" in l for l in body):
        msgs.append("waterfall (v_readfirstlane) inside the loop")
    bl = [l for l in body if l.startswith("buffer_load")]
    if not bl or any(not re.search(r", s\d+ offen$|, s\[\d+:\d+\], s\d+ offen", l) for l in bl):
        msgs.append("a buffer load of the loop without an SGPR offset")
    if len(bl) != L * D:
        msgs.append(f"{len(bl)} loads in the loop, expected {D} steps x {L}")
    # 1. nobody touches a set between its loads and its wait (the loop, wrapped around)
    wl = [i for i, l in enumerate(body) if re.match(r"s_waitcnt vmcnt\(", l)]
    li = [(i, regs_of(re.findall(r"v\[\d+:\d+\]|\bv\d+\b", l)[0])) for i, l in enumerate(body) if l.startswith("buffer_load")]
    groups = []
    for i, r in li:
        if groups and i - groups[-1][-1][0] < 40:
            groups[-1].append((i, r))
        else:
            groups.append([(i, r)])
    bad = 0
    for g in groups:
        dst = set().union(*[r for _, r in g])
        end = g[-1][0]
        w = max([x for x in wl if x < g[0][0]], default=None)
        if w is None:
            continue
        for i in list(range(end + 1, len(body))) + list(range(0, w)):
            touched = all_regs(body[i]) & dst
            if touched:
                bad += 1
                msgs.append(f"line '{body[i][:70]}' touches {sorted(touched)[:4]} of a set in flight")
                break
    ok = not msgs
    return ok, (f"D={D} L={L} vmcnt({steady}) x {D}, {len(body)} instructions in the loop" if ok else "; ".join(msgs))


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "flow_inst.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S", "-o", out, src] + os.environ.get("AUDIT_HIP_FLAGS", "").split()
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            print(r.stdout.decode(errors="replace")[-2000:])
            return 2
        text = open(out).read().split("\n")
    # no kernel may spill: scratch accesses are memory operations the hand-written counts do not know
    scratch = {m.group(1): int(m.group(2)) for m in re.finditer(r"\.amdhsa_kernel (\S+).*?\.amdhsa_private_segment_fixed_size (\d+)", "\n".join(text), re.S)}
    kernels = {}
    cur = None
    for l in text:
        m = re.match(r"^(_ZN4amgh2bw1[78]gs_bw_(?:flow|relay)_kernel\w+):", l)
        if m:
            cur = m.group(1)
            kernels[cur] = []
        elif cur is not None:
            kernels[cur].append(l)
            if "s_endpgm" in l and ".Lfunc_end" in "".join(text[text.index(l):text.index(l) + 3]):
                cur = None
    if not kernels:
        print("no gs_bw_flow_kernel instantiation found")
        return 2
    rc = 0
    for name, lines in sorted(kernels.items()):
        ok, msg = audit_kernel(name, lines)
        if scratch.get(name, 0) != 0 or any("scratch_" in l for l in lines):
            ok, msg = False, f"spills ({scratch.get(name, 0)} bytes of scratch): uncounted memory operations; " + msg
        short = re.sub(r"^_ZN4amgh2bw1[78]gs_bw_(flow|relay)_kernelI", r"\1 ", name)[:30]
        print(f"{'ok  ' if ok else 'FAIL'} {short:30s} {msg}")
        rc |= 0 if ok else 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
