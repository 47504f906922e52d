"""CPU: two properties of the gfx950 ISA of the HIP library, proved from the assembly (tools/check_store_hazard.py): no 128-bit store with the data
hazard the compiler does not guard, and the hand-off protocol of k_pipeline -- every workspace load of that kernel is an sc1 load.
hipcc cross-compiles without a GPU; about two minutes (one compilation for both)."""
import functools
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@functools.lru_cache(maxsize=1)
def _isa():
    import check_store_hazard as chk
    return chk.isa_text()


@pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_wide_buffer_stores_are_hazard_free():
    import check_store_hazard as chk
    stores, sgpr_soffset, overwritten = chk.scan(_isa())
    assert stores > 100                      # the row-pair stores are there at all
    assert not sgpr_soffset, sgpr_soffset[:5]
    assert not overwritten, overwritten[:5]


@pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_every_workspace_load_of_the_pipeline_bypasses_the_vector_l1():
    """k_pipeline hands tiles between the compute units of an XCD inside ONE launch: producer = plain stores, `s_waitcnt vmcnt(0)` (acknowledged by
    the XCD's L2), flag; consumer = flag, then loads.  A CU's vector L1 is never refreshed by another CU's stores, and `buffer_inv sc0` does not drop
    its lines (profiles/r05_store_pairing.txt, tools/ubench/stale_l1.hip) -- so the protocol is: EVERY workspace load of the kernel carries sc1 (served
    by the L2).  The cache policy is a literal of the kernel folded into the accessors (DevParams::xcu); this test is what notices an accessor that
    forgets it, or a constant that no longer folds (a run-time select would show as loads of both kinds)."""
    import check_store_hazard as chk
    res = chk.scan_xcu_loads(_isa())
    pipes = {f: v for f, v in res.items() if "k_pipeline" in f}
    assert len(pipes) >= 4                                   # nx = 5 / 6 x variants
    for f, (n_buf, n_sc1, n_plain_global) in pipes.items():
        assert n_buf > 100 and n_sc1 == n_buf, (f, n_buf, n_sc1)
        # (global loads without sc1: the read-only bounds table of the batch, copied to LDS -- at most two fills of [LB | UB] per role)
        assert n_plain_global <= 8, (f, n_plain_global)
    for f, (n_buf, n_sc1, _) in res.items():
        if "k_pipeline" not in f:
            assert n_sc1 == 0, (f, n_sc1)                    # the other kernels read their own CU's rows / rows of earlier launches: plain loads
