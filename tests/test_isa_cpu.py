"""ISA check of the kernels whose vector-memory loads are inline asm with hand-counted waits (csrc/dense_bf16x3.h,
csrc/class_bf16x3.h): along every control-flow path of the compiled kernel no instruction touches a load's destination
registers before the `s_waitcnt vmcnt(N)` that retires it (tools/check_asm_loads.py).  hipcc cannot see these loads: it may
copy their destination registers at a loop's back edge or reuse them, and parity tests on cache-resident inputs do not notice
(round 5: DESIGN.md 4.1, docs/NOTEBOOK_r5.md).  Cross-compiles for gfx950; no GPU needed."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
@pytest.mark.parametrize("src,patterns,min_kernels", [
    ("cuda_gcn_amd/csrc/matmul.hip", ["class_fwd_bf16x3_kernelILi0", "class_bwd_bf16x3_kernelILi"], 5),
    ("cuda_gcn_amd/csrc/spmm.hip", ["dense_fwd_bf16x3", "dense_bwd_bf16x3"], 4),
])
def test_no_access_to_registers_of_loads_in_flight(src, patterns, min_kernels):
    import check_asm_loads as chk
    asm = chk.compile_to_asm(os.path.join(ROOT, src))
    try:
        ks = chk.kernels(asm, patterns)
        assert len(ks) >= min_kernels, sorted(ks)
        for name, lines in ks.items():
            n_loads = sum(1 for l in lines if chk.LOAD.match(l.split(";")[0]))
            assert n_loads >= 20, (name, n_loads)              # the kernels were found whole (not cut at an early s_endpgm)
            bad = chk.check(lines)
            assert not bad, (name, [(pc, text, lines[issued].strip()) for pc, text, issued in bad[:6]])
    finally:
        os.unlink(asm)
