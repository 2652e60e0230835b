"""CPU: the counted `s_waitcnt vmcnt(N)` of csrc/hvn_conv_chain_x3r.hip against the ISA hipcc generates for gfx950.

The kernel's barriers wait until at most N vector-memory operations are outstanding and rely on N being the number of operations the wave
issues AFTER the weight-chunk LDS-DMAs whose data the next phase reads (memory operations complete in issue order): this chunk's y stores
and the next chunk's residual loads.  If a compiler change moved, merged or dropped one of those, the wait would let a DMA stay in flight
past the barrier -- a race no functional test is guaranteed to catch.  So the count is checked where it is decided: in the assembly.
For every instantiation, in the chunk loop: between the last `buffer_load ... lds` of a DMA block and the next hand-written
`s_waitcnt vmcnt(N) lgkmcnt(0)`, the number of vector-memory instructions must be >= N (more only makes the wait stricter), and it must
be exactly the 4 stores (+ 4 loads with a residual) the source says."""
import os
import re
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hover_net_amd", "csrc")
VMEM = re.compile(r"^\s+(buffer_(load|store)|global_(load|store)|scratch_(load|store)|flat_(load|store))")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc (cross-compiles without a GPU)")
def test_chain_x3r_counted_waits_match_the_isa(tmp_path):
    out = tmp_path / "x3r.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + CSRC, "-S", "--cuda-device-only",
                    "-o", str(out), os.path.join(CSRC, "hvn_conv_chain_x3r.hip")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    text = out.read_text().split("\n")
    starts = [i for i, l in enumerate(text) if re.match(r"^_Z\d+hvn_conv_chain_x3r\w*:", l)]
    assert len(starts) == 6, "six instantiations (cout2 64 | 128, fused shortcut, 6 | 9 terms)"
    checked = 0
    for s in starts:
        name = text[s].split(":")[0]
        end = next(i for i in range(s, len(text)) if "s_endpgm" in text[i])
        body = text[s:end]
        has_residual = "ILi64ELb1E" not in name                      # the fused-shortcut instantiation carries no residual loads
        i = 0
        while i < len(body):
            if "buffer_load" in body[i] and body[i].rstrip().endswith("lds"):
                j = i
                while j + 1 < len(body) and not re.search(r"s_waitcnt vmcnt\(\d+\) lgkmcnt\(0\)", body[j + 1]):
                    j += 1
                    if "buffer_load" in body[j] and body[j].rstrip().endswith("lds"):
                        i = j                                       # still inside the DMA block: restart the window at its last instruction
                m = re.search(r"s_waitcnt vmcnt\((\d+)\) lgkmcnt\(0\)", body[j + 1]) if j + 1 < len(body) else None
                if m is None:
                    break
                window = body[i + 1:j + 1]
                n_wait = int(m.group(1))
                vm = [l for l in window if VMEM.match(l) and not l.rstrip().endswith("lds")]
                stores = [l for l in vm if "store" in l]
                if any("s_barrier" in l for l in window) and n_wait in (4, 8):     # the chunk-end wait of the loop body (not the prologue's)
                    assert len(vm) >= n_wait, (name, n_wait, len(vm))
                    assert len(stores) == 4 and len(vm) == (8 if has_residual else 4), (name, len(stores), len(vm))
                    assert n_wait == (8 if has_residual else 4), (name, n_wait)
                    checked += 1
                i = j + 1
            i += 1
    assert checked >= 6, checked
