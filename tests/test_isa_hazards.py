"""CPU: the gfx950 ISA of the HIP library must not contain the 128-bit-store data hazard (tools/check_store_hazard.py).
hipcc cross-compiles without a GPU; about a minute."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_wide_buffer_stores_are_hazard_free():
    import check_store_hazard as chk
    stores, sgpr_soffset, overwritten = chk.scan(chk.isa_text())
    assert stores > 100                      # the row-pair stores are there at all
    assert not sgpr_soffset, sgpr_soffset[:5]
    assert not overwritten, overwritten[:5]
