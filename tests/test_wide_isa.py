"""The premises of csrc/wide.hip that only the generated ISA can show, checked without a GPU (hipcc cross-compiles gfx950):

  * the kernels own the accumulator file by name: the compiler must not have parked values in accumulator registers (no `v_accvgpr_*`
    outside the kernel's inline-asm blocks) nor spilled to scratch;
  * the K loop's exact wait `s_waitcnt vmcnt(15 - j + 3 WS)` (round 5) counts the stream-out's stores: between two consecutive first
    waits of a resident chunk there must be exactly 4 ring loads and WS stores (2 in the forward instantiation <0>: 16-byte piece + sign
    byte; 1 in the backward chain <1>), as single instructions.  Fewer stores than counted would make the wait too lenient.
"""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def isa():
    from scenerf_amd import build as b
    hipcc = b._hipcc()
    td = tempfile.mkdtemp()
    out = os.path.join(td, "wide.s")
    cmd = [hipcc] + b.FLAGS + b.EXTRA.get("wide.hip", []) + ["--cuda-device-only", "-S", os.path.join(b.CSRC, "wide.hip"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


def _kernel(s, mode):
    name = "_Z15mlp_wide_kernelILi%dEEv9FusedArgs" % mode
    a = s.index(name + ":")
    return s[a:s.index(".Lfunc_end", a)], name


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_no_compiler_made_accumulator_traffic_and_no_scratch(isa, mode):
    body, name = _kernel(isa, mode)
    inasm, made = False, 0
    for ln in body.split("\n"):
        if "#ASMSTART" in ln:
            inasm = True
        elif "#ASMEND" in ln:
            inasm = False
        elif "v_accvgpr" in ln and not inasm:
            made += 1
    assert made == 0, "%s: %d compiler-made v_accvgpr_* instructions (the register allocator parked values in the accumulator file)" % (name, made)
    meta = isa[isa.index(".name:", isa.index("amdhsa.kernels")):]
    m = re.search(r"\.name:\s+%s\b.*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)" % re.escape(name), meta, re.S)
    if m is None:       # (field order differs between compiler versions: look the two fields up separately within this kernel's record)
        rec = meta[meta.index(name):]
        rec = rec[:rec.index(".name:", 10)] if ".name:" in rec[10:] else rec
        priv = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", rec).group(1))
        spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", rec).group(1))
    else:
        priv, spill = int(m.group(1)), int(m.group(2))
    assert priv == 0 and spill == 0, "%s: scratch %d bytes, %d spilled VGPRs" % (name, priv, spill)


@pytest.mark.parametrize("mode,ws", [(0, 2), (1, 1)])
def test_exact_wait_counts_what_a_chunk_issues(isa, mode, ws):
    body, name = _kernel(isa, mode)
    lines = body.split("\n")
    first = "s_waitcnt vmcnt(%d)" % (15 + 3 * ws)
    idx = [i for i, ln in enumerate(lines) if first in ln]
    assert len(idx) >= 9, "%s: the exact first wait %r appears %d times" % (name, first, len(idx))
    checked = 0
    for a, b in zip(idx[:-1], idx[1:]):
        if b - a > 400:       # (the next occurrence belongs to another inlined copy of the loop: epilogues in between)
            continue
        ops = [ln for ln in lines[a:b] if re.search(r"\b(global|buffer|flat)_(load|store)", ln)]
        loads = [ln for ln in ops if "global_load_dwordx4" in ln]
        stores = [ln for ln in ops if "_store" in ln]
        other = [ln for ln in ops if ln not in loads and ln not in stores]
        assert len(loads) == 4 and len(stores) == ws and not other, "%s: a resident chunk issues %d loads, %d stores, %d other vector-memory " \
            "operations (the wait counts 4 + %d): %s" % (name, len(loads), len(stores), len(other), ws, [ln.strip()[:50] for ln in ops])
        checked += 1
    assert checked >= 8
    # the instantiations without certain stores keep the load-only count
    for m2 in (2, 3, 4):
        b2, _ = _kernel(isa, m2)
        assert "s_waitcnt vmcnt(15)" in b2 and first not in b2


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_nothing_is_stored_behind_the_last_ring_load_before_everything_has_landed(isa, mode):
    """Round 6: the K loop's weight loads are inline asm with register outputs; the loop is branch-free and requests four chunks past the
    last one into ring registers that die with the loop.  The compiler does not know those writes are pending and reuses the registers for
    the tail's store addresses -- a late load then overwrites a pointer (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION, 1 in 12-46 processes
    on a busy GPU).  In the ISA: between the LAST inline-asm ring load of a kernel and the first vector-memory store behind it there must
    be an `s_waitcnt vmcnt(0)`."""
    body, name = _kernel(isa, mode)
    lines = body.split("\n")
    inasm, last, wait_at = False, -1, -1
    for i, ln in enumerate(lines):
        if "#ASMSTART" in ln:
            inasm = True
        elif "#ASMEND" in ln:
            inasm = False
        elif inasm and re.match(r"\s*global_load_dwordx4\s+v\[\d+:\d+\],\s*v\d+,\s*s\[\d+:\d+\]", ln):
            last = i            # (saddr form with a register destination: H_LW)
        elif inasm and re.match(r"\s*s_waitcnt\s+vmcnt\(0\)\s*$", ln):
            wait_at = i         # (the hand-written one: the compiler's own waits are outside the asm blocks)
    assert last > 0, name + ": no inline-asm ring load found (did H_LW change form?)"
    # (text order: the tail is laid out behind every K-loop body in all five instantiations; the stores between the textually last ring
    #  load and the wait are the K loop's own stream-out and the epilogues, in front of which the ring is still live)
    assert wait_at > last, name + ": no hand-written `s_waitcnt vmcnt(0)` behind the last ring load (the tail's stores would use registers " \
                                  "that pending loads still write)"
    assert any(re.match(r"\s*global_store", ln) for ln in lines[wait_at:]), name + ": no tail store behind the wait (did the tail move?)"


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_the_compiler_copies_no_ring_register_once_the_asm_loads_have_started(isa, mode):
    """The other half of the same premise: the compiler believes an inline-asm load's destination is defined at once.  If it decided to
    COPY a ring register (a phi move at a loop back-edge, a re-assignment between two code regions) the copy would read a register whose
    load may still be in flight.  Today it does not: every register-to-register move of a ring register sits in front of the first asm
    load, where the ring is filled by plain loads the compiler waits for itself."""
    body, name = _kernel(isa, mode)
    lines = body.split("\n")
    inasm, first, ring = False, None, set()
    for i, ln in enumerate(lines):
        if "#ASMSTART" in ln:
            inasm = True
        elif "#ASMEND" in ln:
            inasm = False
        elif inasm:
            m = re.match(r"\s*global_load_dwordx4\s+v\[(\d+):(\d+)\],\s*v\d+,\s*s\[\d+:\d+\]", ln)
            if m:
                ring.update(range(int(m.group(1)), int(m.group(2)) + 1))
                first = i if first is None else first
    assert first is not None and len(ring) >= 64
    # (a ring register is an ordinary register again behind the tail's hand-written `s_waitcnt vmcnt(0)`: only moves in front of it count)
    inasm, tail_wait = False, len(lines)
    for i, ln in enumerate(lines):
        if "#ASMSTART" in ln:
            inasm = True
        elif "#ASMEND" in ln:
            inasm = False
        elif inasm and re.match(r"\s*s_waitcnt\s+vmcnt\(0\)\s*$", ln) and i > first:
            tail_wait = i
    inasm, bad = False, []
    for ln in lines[first:tail_wait]:
        if "#ASMSTART" in ln:
            inasm = True
        elif "#ASMEND" in ln:
            inasm = False
        elif not inasm:
            t = ln.strip()
            # (a fragment is four registers: on gfx950 its copy is a pair of v_mov_b64 / v_pk_mov_b32 -- the prologue's re-assignments of
            #  the plain-loaded ring look exactly like that; single v_mov_b32 of a register that serves the ring in ANOTHER region are
            #  address copies at loop back-edges and not meant here)
            m = re.match(r"v_(?:mov_b64_e32|pk_mov_b32)\s+v\[?(\d+)(?::\d+)?\]?,\s*v\[?(\d+)(?::(\d+))?\]?", t)
            if m:
                src = set(range(int(m.group(2)), int(m.group(3) or m.group(2)) + 1))
                if src & ring:
                    bad.append(t)
    assert not bad, "%s: compiler-made copies of ring registers between the first asm load and the tail: %s" % (name, bad[:5])
