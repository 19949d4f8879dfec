"""Static checks on the BUILT library (no GPU needed): the kernels that wait for their own loads with hand-counted
`s_waitcnt vmcnt(N)` must use no scratch memory and spill no register, and in the weight-gradient kernel (all of whose
loads are inline asm) no instruction may touch a load's destination register before the wait that covers it
(tools/kernel_guard.py).  Re-run after any change of toolchain, NERF_WG_DEPTH or register budget."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_guard  # noqa: E402

LIB = os.path.join(ROOT, "nerf-loc_amd", "lib", "libnerf_hip.so")


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built")
def test_hand_counted_kernels_have_no_scratch_and_no_early_register_use():
    report, bad = kernel_guard.run(LIB)
    names = [n for n, _, _, _ in report]
    for want in ("wgrad_kernelILi8", "wgrad16_kernel", "mlp16_bf16_kernel", "mlp_split_kernelILi3"):
        assert any(want in n for n in names), want
    assert bad == 0, [(n, p[:3]) for n, _, _, p in report if p]


def test_stream_checker_sees_a_register_used_before_its_wait():
    """The checker itself: a load whose destination is read before the covering wait is reported, the same code with the
    wait in front of the use is not; vmcnt(N) keeps the N youngest operations outstanding."""
    bad = [(1, "global_load_dwordx2 v[4:5], v[0:1], off"), (2, "global_load_dword v6, v[2:3], off"),
           (3, "s_waitcnt vmcnt(1)"), (4, "v_add_f32_e32 v7, v6, v4")]
    found = kernel_guard.check_stream(bad)
    assert found and all("('v', 6)" in f for f in found) and any("line 4" in f for f in found)
    good = bad[:2] + [(3, "s_waitcnt vmcnt(0)"), (4, "v_add_f32_e32 v7, v6, v4")]
    assert kernel_guard.check_stream(good) == []
    copy_before_wait = [(1, "global_load_dword v8, v[0:1], off"), (2, "v_mov_b32_e32 v9, v8"), (3, "s_waitcnt vmcnt(0)")]
    assert any("line 2" in f for f in kernel_guard.check_stream(copy_before_wait))
    lds_dma = [(1, "global_load_lds_dwordx4 v[0:1], off"), (2, "v_mov_b32_e32 v0, v1"), (3, "s_waitcnt vmcnt(0)")]
    assert kernel_guard.check_stream(lds_dma) == []


def test_scratch_is_tolerated_only_outside_the_layer_loop():
    mfma = "v_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], a[0:3]"
    before = [(0, "scratch_store_dword off, v2, off offset:16"), (1, mfma), (2, "v_add_u32_e32 v0, v1, v2"), (3, mfma),
              (4, "scratch_load_dword v2, off, off offset:16")]
    assert kernel_guard.scratch_in_layer_loop(before) == []
    inside = [(0, mfma), (1, "scratch_load_dword v2, off, off offset:16"), (2, mfma)]
    found = kernel_guard.scratch_in_layer_loop(inside)
    assert len(found) == 1 and "line 1" in found[0]
