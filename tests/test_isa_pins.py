"""CPU only. Pins on the ISA of the shipped library (lib/libnfagg.so, gfx950 code objects) where the C++ source cannot hold them.

The 16-byte write-through store of csrc/nfagg_device.h (ast16) is inline asm: hipcc sees one opaque instruction and neither knows
that it is a 128-bit store nor pads its hazards. Round 4 met the consequence (profiles/r04_ast16_hazard.txt): once register
allocation moved, the store's data registers were rewritten — with the address of the NEXT store — before the store had read them,
and ~3 % of freshly claimed slots carried a key no record has. The pad (`s_nop 1`: the two wait states gfx940+ wants between a
store of more than 64 bits and a write to its data registers) now sits inside the asm string; this test makes sure that it is
still there, after EVERY such store of EVERY kernel, whatever a compiler bump or an edit did to the string."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
LIB = os.path.join(ROOT, "netobserv-ebpf-agent_amd", "lib", "libnfagg.so")


@pytest.fixture(scope="module")
def disassembly(tmp_path_factory):
    if not os.path.exists(OBJDUMP):
        pytest.skip("no llvm-objdump in this image")
    d = tmp_path_factory.mktemp("isa")
    shutil.copy(LIB, d / "libnfagg.so")
    subprocess.check_call([OBJDUMP, "--offloading", "libnfagg.so"], cwd=d, stdout=subprocess.DEVNULL)
    out = []
    for co in sorted(glob.glob(str(d / "libnfagg.so.*gfx950*"))):
        out.append(subprocess.check_output([OBJDUMP, "-d", co], text=True))
    assert out, "no gfx950 code object in lib/libnfagg.so"
    return "\n".join(out)


def _instructions(dis):
    for line in dis.splitlines():
        m = re.match(r"^\s+([a-z_0-9]+)(?:\s+([^/]*?))?\s*//", line)
        if m:
            yield m.group(1), (m.group(2) or "").strip()


def test_every_16_byte_write_through_store_is_followed_by_its_hazard_pad(disassembly):
    ins = list(_instructions(disassembly))
    stores = [k for k, (op, args) in enumerate(ins) if op == "global_store_dwordx4" and args.endswith("off sc1")]
    assert len(stores) > 1000, "ast16 stores not found: has the asm string of nfagg_device.h changed?"
    bad = [k for k in stores if ins[k + 1] != ("s_nop", "1")]
    assert not bad, "%d of %d write-through 16-byte stores without `s_nop 1` behind them, first: %r -> %r" % (
        len(bad), len(stores), ins[bad[0]], ins[bad[0] + 1])


def test_the_cut_walk_scans_with_dpp_and_the_sketch_kernel_reduces_with_it(disassembly):
    """north_star names wave-level DPP for the sketch updates; the cut walk of nfagg_account uses the same row_shr / row_bcast scan.
    If a compiler bump turned the builtins into LDS swizzles the kernels would still be right, and slower: say so here."""
    kernels = {}
    cur = None
    for line in disassembly.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line.strip())
        if m:
            cur = m.group(1)
            kernels[cur] = 0
        elif cur and ("row_bcast:15" in line or "row_bcast:31" in line):
            kernels[cur] += 1
    assert any("k_par_cuts" in k and v >= 2 for k, v in kernels.items()), "k_par_cuts: no DPP row broadcasts in its ISA"
    assert any("k_sketch_update" in k and v >= 2 for k, v in kernels.items()), "k_sketch_update: no DPP row broadcasts in its ISA"


def test_no_record_address_is_formed_from_an_unmasked_sort_key(disassembly):
    """Round 5: hipcc took `(key & 0xFFFFFF) * 144` for a 24-bit multiply, dropped the mask and emitted v_mad_u64_u32 on the unmasked
    low dword of the key (profiles/r05_mul24_miscompile.txt, tools/gpu/mul24_miscompile.hip): k_par_links read records far outside
    the batch. csrc/nfagg_epoch_par.hip key_index() keeps the mask behind an opaque register. The signature of the miscompile — a
    register multiplied into an address and masked with 0xffffff only AFTERWARDS — must not be in the kernels that index by sort key."""
    cur, window, hits = None, [], []
    for line in disassembly.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line.strip())
        if m:
            cur, window = m.group(1), []
            continue
        if not cur or "k_par_" not in cur:
            continue
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        mm = re.match(r"v_and_b32_e32 (v\d+), 0xffffff, \1$", ins)
        if mm:
            reg = mm.group(1)
            for prev in window[-8:]:
                if prev.startswith("v_mad_u64_u32") and re.search(r", %s, s\d+" % reg, prev):
                    hits.append((cur[:60], prev, ins))
        window.append(ins)
    assert not hits, hits[:3]


def test_the_cut_walk_keeps_twelve_loads_in_flight(disassembly):
    """Round 5: the walk was bound by memory round trips, not by its arithmetic — a bounds branch around its loads made the compiler
    close every one of them with s_waitcnt vmcnt(0), and once that was gone it drained all loads before the loop that first reads
    them (profiles/r05_cut_walk_latency.txt). The kernel now waits for the sixteen registers of the block it walks and for nothing
    younger: vmcnt(12), four times (once per register set), and no vmcnt(0) anywhere in the walk. (Round 6: behind the walk the kernel
    copies the cuts it found into the host's list — the fence and the read-back wait with vmcnt(0), AFTER the last vmcnt(12).)"""
    cur, waits, order = None, {}, []
    for line in disassembly.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line.strip())
        if m:
            cur = m.group(1)
            continue
        if cur and "k_par_cuts" in cur:
            m = re.search(r"s_waitcnt\s+(.*?)\s*//", line)
            if m and "vmcnt" in m.group(1):
                n = int(re.search(r"vmcnt\((\d+)\)", m.group(1)).group(1))
                waits[n] = waits.get(n, 0) + 1
                order.append(n)
    assert waits.get(12, 0) == 4, "k_par_cuts waits on its loads as %r (wanted: vmcnt(12) x 4)" % waits
    last12 = max(i for i, n in enumerate(order) if n == 12)
    assert 0 not in order[:last12], "k_par_cuts drains its loads inside the walk: %r" % order
