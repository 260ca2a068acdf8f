"""ISA-level check of the built gfx950 code (no GPU needed): no hand-written LDS-DMA reads an SGPR that a VALU instruction wrote
fewer than 5 wait states earlier (tools/check_dma_hazard.py; DESIGN.md 7a, the bf16 fused kernel's LDS-DMA paragraph)."""
import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_dma_hazard as H  # noqa: E402


def test_scanner_sees_the_round2_hazard_and_its_fix():
    stale = [("v_readlane_b32", "s14, v254, 14"), ("s_or_b32", "s9, s38, s11"), ("v_readlane_b32", "s15, v254, 15"),
             ("s_mov_b32", "s10, m0"), ("s_mov_b32", "m0, s9"), ("s_nop", "0"), ("global_load_lds_dwordx4", "v202, s[14:15]")]
    assert len(H.scan(stale)) == 1
    fixed = stale[:5] + [("s_nop", "4")] + stale[6:]
    assert H.scan(fixed) == []
    unrelated = [("v_readlane_b32", "s20, v254, 1"), ("global_load_lds_dwordx4", "v202, s[14:15] offset:1024")]
    assert H.scan(unrelated) == []
    via_buffer = [("v_readfirstlane_b32", "s7, v3"), ("buffer_load_dwordx4", "v1, s[4:7], s12 offen lds")]
    assert len(H.scan(via_buffer)) == 1


def test_built_kernels_have_no_lds_dma_hazard():
    objs = sorted(glob.glob(os.path.join(ROOT, "so-net_amd", "build", "*.o")))
    if not objs or not os.path.exists(H.OBJDUMP):
        pytest.skip("object files of the in-tree build (so-net_amd/build) or llvm-objdump not present")
    total_dma = 0
    for o in objs:
        if os.path.basename(o) not in ("pointmlp_x3.o", "pointmlp_h3p.o", "pointresnet_bf16.o", "pointresnet_fused.o"):
            continue
        ins = H.disassemble(o)
        total_dma += sum(1 for mn, _ in ins if mn.startswith("global_load_lds"))
        assert H.scan(ins) == [], os.path.basename(o)
    assert total_dma > 0, "the kernels that stream weights by LDS-DMA were not found in the build"


def test_third_generation_layer_has_no_compiler_waits_or_spills_in_its_pass_loop():
    """pointmlp_h3p.hip counts its vector-memory requests by hand: hipcc must not have added a vmcnt wait of its own (a spill reload or a
    compiler-visible load inside the loop drains the look-ahead every iteration) -- tools/check_h3p_asm.py compiles the source and checks."""
    import check_h3p_asm
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not present")
    assert check_h3p_asm.main([]) == 0


def test_streaming_bf16_layer_never_touches_an_x_register_in_flight():
    """pointmlp_bf16r_kernel (pointmlp_bf16.hip) issues its X loads as inline asm: hipcc believes the destination registers are valid at
    once, and a move out of one (a copy for a tied asm operand, a phi copy) before the hand-counted wait reads a stale value -- which is
    how the first version of the kernel returned wrong columns.  tools/check_bf16r_asm.py compiles the source and follows every ring
    register from its load to the wait + perm statement; it also counts the stores the wait counts are built on."""
    import check_bf16r_asm
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not present")
    sys_argv = sys.argv
    sys.argv = ["check_bf16r_asm.py"]
    try:
        assert check_bf16r_asm.main() == 0
    finally:
        sys.argv = sys_argv
