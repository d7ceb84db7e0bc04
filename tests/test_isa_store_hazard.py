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
