"""Build-time guard for the store-data hazard of DESIGN.md section 5.2b: on gfx950 a buffer store of more than 64
bits reads its data VGPRs late, so a VALU write to one of them in the very next instruction corrupts the stored
value; hipcc 7.2 inserts the wait state itself EXCEPT when the store has an SGPR soffset.  The kernels therefore keep
the whole offset in the VGPR (immediate soffset).  This test disassembles the built library and fails if any kernel
contains a dwordx3 / dwordx4 buffer store with an SGPR soffset directly followed by a VALU write to its data registers
-- i.e. if a code or compiler change brings the pattern back."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _vregs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="no llvm-objdump")
def test_no_wide_store_with_sgpr_soffset_is_followed_by_a_write_to_its_data(tmp_path):
    import __graft_entry__ as g
    g.build()
    lib = os.path.join(ROOT, "radio-mapper_amd", "csrc", "librmx_hip.so")
    work = str(tmp_path)
    shutil.copy(lib, os.path.join(work, "lib.so"))
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=work, check=True, capture_output=True)
    objs = [f for f in os.listdir(work) if "gfx950" in f]
    assert objs, "no gfx950 code object in the library"
    dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", objs[0]], cwd=work, check=True, capture_output=True,
                         text=True).stdout
    lines = [ln.split("//")[0].strip() for ln in dis.splitlines()]
    lines = [ln for ln in lines if ln and not ln.endswith(":")]
    n_wide = n_sgpr = 0
    bad = []
    for k, ln in enumerate(lines[:-1]):
        m = re.match(r"buffer_store_dwordx[34]\s+(v\[\d+:\d+\]),\s*(\S+),\s*(s\[\d+:\d+\]),\s*(\S+)", ln)
        if not m:
            continue
        n_wide += 1
        soff = m.group(4).rstrip(",")
        if not re.fullmatch(r"s\d+|m0|vcc_lo|vcc_hi", soff):
            continue                                   # immediate / 'off': the compiler inserts the wait state itself
        n_sgpr += 1
        nxt = lines[k + 1]
        if not nxt.startswith("v_"):
            continue
        dst = nxt.split()[1].rstrip(",") if len(nxt.split()) > 1 else ""
        if _vregs(dst) & _vregs(m.group(1)):
            bad.append((ln, nxt))
    assert n_wide > 0, "the disassembly shows no wide buffer store at all: the parser is out of date"
    assert not bad, "store-data hazard pattern: %s" % bad[:3]
    print("wide buffer stores: %d, with an SGPR soffset: %d, hazard patterns: 0" % (n_wide, n_sgpr))


READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


@pytest.mark.skipif(not (os.path.exists(OBJDUMP) and os.path.exists(READELF)), reason="no llvm-objdump / llvm-readelf")
def test_fused_kernel_keeps_its_registers_and_uses_no_scratch(tmp_path):
    """Resource guard for the dominant kernel: k_win (both input types) runs two waves per SIMD with every array in
    registers.  A code or compiler change that pushes it into scratch memory (a scratch access waits on vmcnt and drains
    the spectra requested a pair ahead: DESIGN.md section 5.1) or past 256 VGPRs shows up here, not as a slower bench."""
    import __graft_entry__ as g
    g.build()
    lib = os.path.join(ROOT, "radio-mapper_amd", "csrc", "librmx_hip.so")
    work = str(tmp_path)
    shutil.copy(lib, os.path.join(work, "lib.so"))
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=work, check=True, capture_output=True)
    objs = [f for f in os.listdir(work) if "gfx950" in f]
    assert objs
    notes = subprocess.run([READELF, "--notes", objs[0]], cwd=work, check=True, capture_output=True, text=True).stdout
    # the AMDGPU metadata note is YAML: one "- .agpr_count: ..." block per kernel
    blocks = re.split(r"\n\s*- \.agpr_count:", notes)
    seen = {}
    for b in blocks[1:]:
        name = re.search(r"\.name:\s+(\S+)", b)
        if not name or "k_winILb" not in name.group(1):
            continue
        f = {k: int(re.search(r"\.%s:\s+(\d+)" % k, b).group(1)) for k in
             ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "max_flat_workgroup_size")}
        seen[name.group(1)] = f
    assert len(seen) == 2, "expected k_win<false> and k_win<true> in the code object, found %s" % list(seen)
    for n, f in seen.items():
        assert f["private_segment_fixed_size"] == 0, (n, f)
        assert f["vgpr_spill_count"] == 0, (n, f)
        assert f["vgpr_count"] <= 256, (n, f)        # two waves per SIMD of the 512-register file
        assert f["max_flat_workgroup_size"] == 512, (n, f)
    print(seen)
