"""CPU: the compiler placements that round 3's kernel gains hang on, checked in the gfx950 ISA (tools/isa_guard.py).

Parity tests cannot see these: a misplaced `s_waitcnt vmcnt(0)` or 128 hoisted compares change a kernel's time, not its
results.  hipcc cross-compiles without a GPU, so this runs in the CPU suite.  The second test builds the same kernels with
-DET_GUARD_DROP_PINS (ET_PIN expands to nothing) and asserts that the guard turns red where each pin was -- the guard is only
worth its name if it would notice."""
import os
import shutil
import sys

import pytest

from tests.conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_guard  # noqa: E402

pytestmark = pytest.mark.skipif(not (os.path.exists(isa_guard.HIPCC) or shutil.which("hipcc")), reason="no hipcc")


def test_kernel_isa_keeps_the_placements_the_design_describes(tmp_path):
    res = isa_guard.run_checks(isa_guard.compile_isa(str(tmp_path)))
    bad = [f"{name} -- {detail}" for name, ok, detail in res if not ok]
    assert not bad, "\n".join(bad)
    assert len(res) >= 12


def test_the_guard_turns_red_without_the_pins(tmp_path):
    res = {name: ok for name, ok, _ in isa_guard.run_checks(isa_guard.compile_isa(str(tmp_path), ("-DET_GUARD_DROP_PINS",)))}
    red = [name for name, ok in res.items() if not ok]
    # each of the three pinned placements is missed by name: K4's prefetch (et_kernels.hip, k_encode_tiles), D3's
    # (k_dec_write_wave's WV_TAKE) and D1's edge limits (et_treewalk.hip, k_tw_sync's seam loop)
    assert any(n.startswith("K4: no s_waitcnt vmcnt between") for n in red), red
    assert any(n.startswith("K4: the next chunk is waited for directly behind") for n in red), red
    assert any(n.startswith("D3: the next unit's words are taken in front of") for n in red), red
    assert any(n.startswith("D1: k_tw_sync spills no SGPRs") for n in red), red
