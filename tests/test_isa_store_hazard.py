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


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="no llvm-objdump")
def test_m0_in_the_fused_kernel_is_written_before_every_use(tmp_path):
    """loc_write4 (fft_r16.hpp) writes M0 inside an asm statement -- it now says so in its clobber list.  M0 has
    two users in k_win, and both write it themselves right in front of their reads: the statement's own `s_mov_b32 m0, sN`
    + ds_write_addtid_b32 stores, and the peak search's indexed register read (`s_set_gpr_idx_on sN` ... `s_set_gpr_idx_off`,
    which loads the index into M0: a bracket of at most four instructions with no LDS store inside).  Nothing else may
    depend on M0: no movrel, no LDS-DMA, no GWS, no sendmsg payload, no instruction naming m0 other than that s_mov."""
    import __graft_entry__ as g
    g.build()
    lib = os.path.join(ROOT, "radio-mapper_amd", "csrc", "librmx_hip.so")
    work = str(tmp_path)
    shutil.copy(lib, os.path.join(work, "lib.so"))
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=work, check=True, capture_output=True)
    objs = [f for f in os.listdir(work) if "gfx950" in f]
    assert objs
    dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", objs[0]], cwd=work, check=True, capture_output=True,
                         text=True).stdout
    kernels = {}
    cur = None
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            cur = m.group(1) if "k_winILb" in m.group(1) else None
            if cur:
                kernels[cur] = []
            continue
        if cur:
            t = ln.split("//")[0].strip()
            if t:
                kernels[cur].append(t)
    assert len(kernels) == 2, list(kernels)
    implicit = re.compile(r"^(v_movrel|s_movrel|v_interp|ds_gws|s_sendmsg|ds_ordered|ds_append|ds_consume|buffer_load\S* .*\blds\b|global_load_lds)")
    for name, ins in kernels.items():
        named = [t for t in ins if re.search(r"\bm0\b", t)]
        assert named and all(re.fullmatch(r"s_mov_b32 m0, s\d+", t) for t in named), (name, [t for t in named if not t.startswith("s_mov_b32 m0")][:5])
        assert not [t for t in ins if implicit.match(t)], name
        n_addtid = sum(t.startswith("ds_write_addtid_b32") for t in ins)
        assert n_addtid >= 64 and n_addtid <= 8 * len(named), (name, n_addtid, len(named))
        # every addtid store sits behind an s_mov m0 of its own statement: walking back from it, the s_mov comes before
        # any gpr-index bracket does
        for k, t in enumerate(ins):
            if t.startswith("ds_write_addtid_b32"):
                j = k - 1
                while j >= 0 and not ins[j].startswith(("s_mov_b32 m0", "s_set_gpr_idx")):
                    j -= 1
                assert j >= 0 and ins[j].startswith("s_mov_b32 m0"), (name, k, ins[max(j, 0)])
            if t.startswith("s_set_gpr_idx_on"):
                window = ins[k + 1:k + 5]
                assert any(w.startswith("s_set_gpr_idx_off") for w in window), (name, window)
                inside = window[:next(i for i, w in enumerate(window) if w.startswith("s_set_gpr_idx_off"))]
                assert not any(w.startswith("ds_") for w in inside), (name, inside)
