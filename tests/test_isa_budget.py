"""CPU tier: the register / scratch / instruction budget of the five hot fp32 kernels, read from the device assembly hipcc produces
here (no GPU needed).  What a kernel may hold per SIMD is decided by its VGPR count, and a kernel that starts spilling inside its
loops loses more than any tuning pass ever won (profiles/r04_notes.md: 75 / 120 us against 46) -- neither shows up in a parity test.
The numbers are the build's own (profiles/tools/isa_stats.py on the Makefile's flags); a change that moves them has to move them
here too, on purpose."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))

# kernel (mangled-name fragment) -> (waves per SIMD the registers allow, most scratch bytes, most static vector instructions)
BUDGET = {
    "k_g2p_p2gIfLb0": (4, 0, 3050),              # (round 6: no scratch at all any more -- two compiler-made private arrays removed)
    "k_g2p_gradIfLb0": (4, 0, 1620),
    "k_p2g_gradIf": (2, 128, 11200),           # (the pose-adjoint workgroups share this kernel: fp64 collide adjoints, ~8k of the count)
    "k_grid_opIfLb0": (4, 0, 1600),
    "k_grid_op_gradIf": (2, 0, 4700),
}


def makefile_flags():
    text = open(os.path.join(ROOT, "plasticinelab_amd", "csrc", "Makefile")).read()
    m = re.search(r"^FLAGS\s*:=\s*(.*)$", text, re.M)
    flags = m.group(1).replace("$(ARCH)", "gfx950").replace("$(EXTRA)", "")
    return flags.split()


@pytest.fixture(scope="module")
def listing(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("isa") / "capi.s")
    src = os.path.join(ROOT, "plasticinelab_amd", "csrc", "plmpm_capi.hip")
    subprocess.check_call([hipcc] + makefile_flags() + ["--cuda-device-only", "-S", src, "-o", out], stderr=subprocess.DEVNULL)
    return out


@pytest.mark.timeout(300)
def test_hot_kernels_keep_their_register_and_instruction_budget(listing):
    import isa_stats
    kernels, meta = isa_stats.parse(listing)
    seen = set()
    for name, body in kernels.items():
        for frag, (waves, scratch, valu) in BUDGET.items():
            if frag in name and name in meta:
                seen.add(frag)
                md = meta[name]
                alloc = -(-md["next_free_vgpr"] // 8) * 8
                assert min(8, 512 // alloc) >= waves, (frag, md)
                assert md.get("private_segment_fixed_size", 0) <= scratch, (frag, md)
                assert body["valu"] <= valu, (frag, body["valu"])
    assert seen == set(BUDGET), sorted(set(BUDGET) - seen)


@pytest.mark.timeout(300)
def test_no_scratch_access_inside_a_loop_of_the_particle_kernels(listing):
    """Scratch in straight-line code costs an instruction; inside the 27-node loops it costs the kernel."""
    text = open(listing).read()
    for frag in ("k_g2p_p2gIfLb0", "k_g2p_gradIfLb0"):
        m = re.search(r"^(_ZN3plb\d+%s\w*):\s.*?^\s*s_endpgm" % frag, text, re.M | re.S)
        assert m, frag
        # hipcc annotates every basic block that belongs to a loop ("Loop Header", "in Loop: Header=..."), labelled or not
        in_loop, loops = False, 0
        for line in m.group(0).splitlines():
            if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", line):
                in_loop = "Loop" in line
                loops += in_loop
            elif in_loop and re.match(r"^\s*scratch_", line):
                raise AssertionError(f"{frag}: {line.strip()} inside a loop")
        assert loops > 0, frag
