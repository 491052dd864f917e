"""Build-time guard of the kernels whose LDS-DMA requests and vmcnt waits are hand-written: the generated gfx950 ISA of
every flash_prefill_kernel instantiation has no spills / scratch traffic, no compiler-visible vector-memory instruction inside the
MFMA blocks of the step loop, and only the hand-written counted waits there.  Needs hipcc (cross-compiles without a GPU)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="hipcc not installed")
def test_flash_prefill_isa_keeps_the_hand_counted_ring_intact():
    import check_kernel_isa
    assert check_kernel_isa.check_flash(verbose=False) == []


@pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="hipcc not installed")
def test_gemm_tiled_rope_position_requests_are_not_touched_before_their_wait():
    """the inline-asm position prefetch of the mid-batch qkv GEMM's RoPE epilogue (kernels/gemm_tiled.hip) — no spills, and the
    destination register pair is neither read nor written before the counted wait that covers the request."""
    import check_kernel_isa
    assert check_kernel_isa.check_gemm_tiled(verbose=False) == []


@pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="hipcc not installed")
def test_attention_merge_arithmetic_has_one_form_in_every_instantiation():
    """r05: the split-KV merge in its three homes (merge launch, last-arriver workgroup, own-partition workgroup) is promised to be bit-identical; hipcc's
    per-instantiation choice between fma and multiply + add broke that once (1 output in 10^5, found at 300 sequences).  The source pins both forms; this
    reads the generated ISA of every attn_rows_kernel / attn_merge_kernel instantiation."""
    import check_kernel_isa
    assert check_kernel_isa.check_attention_merge_forms(verbose=False) == []


@pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="hipcc not installed")
def test_linear_stream_rope_requests_are_not_touched_before_their_wait():
    """the inline-asm position / slot prefetch of the streaming qkv GEMM's RoPE epilogue (kernels/linear_stream.hip, both builds) — no scratch,
    no spilled VGPRs, and the destination registers are neither read nor written before the vmcnt wait that covers the requests."""
    import check_kernel_isa
    assert check_kernel_isa.check_linear_stream(verbose=False) == []


def test_the_request_walker_follows_control_flow():
    """tools/check_kernel_isa.py request_problems on hand-written ISA (no compiler needed): a covering wait on every path passes; a register copy of the
    destination in front of the wait, a path that branches around the wait, and a wait that leaves the request itself in flight are reported."""
    import check_kernel_isa as c
    ok = """;;#ASMSTART
global_load_dwordx4 v[10:13], v[2:3], off nt
;;#ASMEND
global_load_dwordx4 v[20:23], v[4:5], off
s_cbranch_scc1 .LBB0_2
v_add_u32_e32 v1, v2, v3
.LBB0_2:
s_waitcnt vmcnt(1)
v_mov_b32_e32 v30, v10
s_endpgm""".split("\n")
    assert c.request_problems(ok, 1) == []
    copied = list(ok); copied.insert(5, "v_mov_b32_e32 v40, v11")           # on the fall-through path, before the wait
    assert any("touched before its wait" in p for p in c.request_problems(copied, 1))
    around = """;;#ASMSTART
global_load_dwordx4 v[10:13], v[2:3], off nt
;;#ASMEND
s_cbranch_execz .LBB0_3
s_waitcnt vmcnt(0)
.LBB0_3:
v_mov_b32_e32 v30, v10
s_endpgm""".split("\n")
    assert any("touched before its wait" in p for p in c.request_problems(around, 1))     # the branch skips the wait
    loose = """;;#ASMSTART
global_load_dwordx4 v[10:13], v[2:3], off nt
;;#ASMEND
s_waitcnt vmcnt(1)
v_mov_b32_e32 v30, v10
s_endpgm""".split("\n")
    assert any("touched before its wait" in p for p in c.request_problems(loose, 1))      # vmcnt(1) with nothing younger leaves the request in flight
    nowait = """;;#ASMSTART
global_load_dwordx4 v[10:13], v[2:3], off nt
;;#ASMEND
s_endpgm""".split("\n")
    assert any("without a covering wait" in p for p in c.request_problems(nowait, 1))
