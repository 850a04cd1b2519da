"""The dataflow sweep kernel (csrc/hip/gs_flow.hpp) counts its memory pipeline by hand: loads are inline asm whose
destination registers the compiler allocates but does not track, waited for by `s_waitcnt vmcnt((D - 1)(L + 2))`.  This
test compiles all 80 instantiations (double / float x GS / SOR x forward / backward x rows of 6 / 12 / 18 entries x
1 / 2 / 3 [/ 4 for rows of 6 entries] right-hand-side columns per workgroup) to gfx950
assembly (hipcc cross-compiles without a GPU) and audits the steady loop of each (tools/flow_asm_audit.py): no instruction
touches a register set between its loads and its wait, the loop holds the hand-written waits and no compiler-inserted
one, every load carries a scalar offset, no waterfall.  The 96 relayed kernels (csrc/hip/gs_relay.hpp: the same pipeline per
walker wave, 3 waves per block; on the plain records and on the dictionary layout, whose sets are the column chunks and b;
each with the stored-order row sum and with the dependency-aware one, template flag LATE) are audited with them.  A second, linear scan (tools/flow_asm_linear.py) covers what the
loop audit does not — the first round and the tail of every pipeline: from each hand-issued load to the wait that covers
it nothing reads or writes its destination (a compiler short of registers parks sets in AGPRs before they have landed:
seen on 36-entry rows in round 5, wrong values on the GPU).  A compiler update that changes any of this fails here, not
as a wrong bit on the GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_flow_kernel_pipeline_survives_the_compiler():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "flow_asm_audit.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    lines = [l for l in out.splitlines() if l.startswith(("ok", "FAIL"))]
    assert r.returncode == 0 and len(lines) == 256 and all(l.startswith("ok") for l in lines), out
    assert sum("relay" in l for l in lines) == 96, out


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_no_set_in_flight_is_touched_anywhere_in_the_kernels():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "flow_asm_linear.py"), os.path.join(ROOT, "tools", "flow_inst.hip")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    lines = [l for l in out.splitlines() if l.startswith(("ok", "FAIL"))]
    assert r.returncode == 0 and len(lines) == 256 and all(l.startswith("ok") for l in lines), out
