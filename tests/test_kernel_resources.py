"""CPU guard on the compiled kernels' resources: hipcc cross-compiles gfx950 here, so a register spill
is visible before any GPU sees it.  Round 4 shipped `gemm_dma_kernel<bf16,128,256,3,true>` with 105 spilled registers and
332 B of scratch per lane (a run-time split-K branch in the shared epilogue); the ConvTranspose input-gradient launches of
the headline step doubled and nothing failed.  This test compiles every .hip of the library to assembly and reads the
code-object metadata (tools/kres.py)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kres  # noqa: E402

# kernels allowed to carry scratch, with the bound they are allowed: (substring of the demangled name) -> bytes per lane
ALLOW_SCRATCH = {
    # fp32-parity window attention forward (VALU softmax beside MFMA, 325 registers): 32 B, outside every timed path
    "winattn_fwd_mfma_kernel": 32,
}


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(kres.HIPCC):
        pytest.skip("no hipcc")
    return kres.collect(jobs=min(8, os.cpu_count() or 4))


def test_no_kernel_spills_vector_registers_or_uses_scratch(kernels):
    assert len(kernels) > 300          # the whole library was seen, not an empty parse
    bad = []
    for k in kernels:
        scratch = k.get("private_segment_fixed_size", 0)
        spills = k.get("vgpr_spill_count", 0)
        allowed = max([v for s, v in ALLOW_SCRATCH.items() if s in k["demangled"]] or [0])
        if scratch > allowed or (spills and not allowed):
            bad.append(f"{k['file']}: {kres.short(k['demangled'])}: {spills} VGPRs spilled, {scratch} B scratch")
    assert not bad, "kernels with spills / scratch:\n" + "\n".join(bad)


def test_the_gemm_that_regressed_in_round_4_is_back_at_its_registers(kernels):
    """`<bf16, 128, 256, 3, BNRED>` (ConvTranspose input gradient + BatchNorm-backward sums): <= 256 registers, no spill;
    the plain form stays well under (203 before the split-K branch, 256 + 15 spilled with it)."""
    by = {k["name"]: k for k in kernels}
    red = [k for n, k in by.items() if "gemm_dma_kernelIDF16bLi128ELi256ELi3ELb1ELb0E" in n]
    plain = [k for n, k in by.items() if "gemm_dma_kernelIDF16bLi128ELi256ELi3ELb0ELb0E" in n]
    split = [k for n, k in by.items() if "gemm_dma_kernelIDF16bLi128ELi256ELi3ELb0ELb1E" in n]
    assert len(red) == 1 and len(plain) == 1 and len(split) == 1
    assert red[0]["vgpr_count"] <= 256 and red[0]["vgpr_spill_count"] == 0
    assert plain[0]["vgpr_count"] <= 216 and plain[0]["vgpr_spill_count"] == 0
    assert split[0]["vgpr_spill_count"] == 0


def test_one_workgroup_per_cu_kernels_fit_the_lds(kernels):
    for k in kernels:
        assert k.get("group_segment_fixed_size", 0) <= 160 * 1024, k["demangled"]


# loads written under a branch are closed by s_waitcnt vmcnt(0) (hipcc 7.2) and run one memory round trip after the
# other.  The kernels below were rewritten with unconditional loads; (substring of the MANGLED name) -> most load -> vmcnt(0) ->
# load points the compiled kernel may show (what is left are prologue / tail loads, not the unrolled groups).
LOAD_WAIT_BOUNDS = {
    "16layernorm_kernelIDF16bLb0ELi3ELi1ELb0E": 2,      # LayerNorm forward, three chunks per lane (swin, all four levels)
    "16layernorm_kernelIDF16bLb1ELi3ELi2ELb0E": 2,      # ... backward
    "14ln_head_kernelIDF16bLb0ELi1ELi3ELi1E": 3,        # LayerNorm + 1x1 head, forward (was 59)
    "14ln_head_kernelIDF16bLb1ELi1ELi3ELi2E": 3,        # ... backward
    "24winattn_fwd_mfma2_kernel": 3,                    # window attention forward: table prologue + prefetch (was 16)
    "20bn_relu_apply_kernelIDF16bLb1ELb0E": 3,          # pooled BatchNorm apply: the four taps (was 13)
    "18bn_relu_bwd_kernelIDF16bLb1ELi1ELb0E": 3,        # pooled BatchNorm backward, reduce pass
    "18bn_relu_bwd_kernelIDF16bLb1ELi2ELb0E": 3,        # ... apply pass
    "18outconv_bwd_kernelIDF16bLi1ELb1ELb1E": 3,        # OutConv backward on the raw tensor (x = NULL)
    "18outconv_bwd_kernelIDF16bLi1ELb1ELb0E": 3,
    "21conv_first_fwd_kernelILi2E": 2,                  # first convolution: halo prefetch + weight prologue (was 33)
    "23conv_first_wgrad_kernelILi2E": 2,
    "15gemm_dma_kernelIDF16bLi128ELi256ELi3ELb1ELb0E": 3,   # ConvTranspose input gradient + BatchNorm-backward sums: epilogue (was 16)
    "21chanattn_probs_kernelIDF16bLi16ELb0E": 3,        # channel-attention probabilities, forward (was 65)
    "19grad_combine_kernelIDF16bLb1E": 2,               # pooled gradient combine (was 12)
}


def test_rewritten_kernels_keep_their_loads_in_flight(kernels):
    by = {k["name"]: k for k in kernels}
    bad, seen = [], 0
    for sub, bound in LOAD_WAIT_BOUNDS.items():
        hit = [k for n, k in by.items() if sub in n]
        assert hit, f"no kernel symbol contains {sub} (renamed? update LOAD_WAIT_BOUNDS)"
        for k in hit:
            seen += 1
            if k["load_wait_points"] > bound:
                bad.append(f"{k['file']}: {kres.short(k['demangled'])}: {k['load_wait_points']} load -> vmcnt(0) -> load points "
                           f"(bound {bound}, {k['loads']} loads)")
    assert seen >= len(LOAD_WAIT_BOUNDS)
    assert not bad, "kernels whose loads are waited for one by one again (DESIGN 3h):\n" + "\n".join(bad)
