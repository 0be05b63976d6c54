"""CPU: bench.py's bookkeeping that only shows on the GPU box when it is wrong -- round 3's driver line carried
`roofline.traffic: null` because the kernel-symbol keys no longer matched the regenerated PMC table."""
import glob
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _newest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    assert files, f"no profiles/{pattern} committed"
    return files[-1]


def test_dominant_kernel_of_the_committed_bench_line_resolves_in_the_committed_pmc_table():
    b = _bench()
    pmc_path = b.newest_pmc_file()
    assert pmc_path == _newest("r[0-9][0-9]_pmc_traffic.json")
    pmc = json.load(open(pmc_path))["kernels"]
    line = json.load(open(_newest("r[0-9][0-9]_default_cmd_bench_line.json")))
    dom = line["roofline"]["kernel"]
    assert dom in b.PMC_KEYS, f"bench.py has no PMC key for the dominant kernel family {dom}"
    rows = b.pmc_rows(dom, pmc)
    assert rows, f"{dom}: no kernel symbol of {os.path.basename(pmc_path)} contains any of {b.PMC_KEYS[dom]}"
    assert all(r["launches"] > 0 and r["hbm_read_mb_per_launch_corrected"] > 0 for r in rows)


def test_every_timed_family_with_a_key_resolves():
    """every family of the committed line's kernel_ms_per_step that bench.py has a key for is found in the PMC table of the
    same round (a key that matches nothing is a stale key)"""
    b = _bench()
    pmc_path = b.newest_pmc_file()
    pmc = json.load(open(pmc_path))["kernels"]
    rnd = os.path.basename(pmc_path)[:3]
    line_path = os.path.join(ROOT, "profiles", f"{rnd}_default_cmd_bench_line.json")
    if not os.path.exists(line_path):
        line_path = _newest("r[0-9][0-9]_default_cmd_bench_line.json")
    fams = json.load(open(line_path))["kernel_ms_per_step"]
    checked = 0
    for fam in fams:
        base = fam[:-len("_bnred")] if fam.endswith("_bnred") else fam
        if base in b.PMC_KEYS:
            assert b.pmc_rows(base, pmc), f"{base}: stale PMC key {b.PMC_KEYS[base]}"
            checked += 1
    assert checked >= 3


def test_pingpong_variants_share_one_key():
    b = _bench()
    fake = {"void (anonymous namespace)::conv3x3_pp_kernel<(anonymous namespace)::PpCfg<16, 32, 4, 2, 1, false>, false, false, false>(x)": 1,
            "void (anonymous namespace)::conv3x3_pp_kernel<(anonymous namespace)::PpCfg<16, 32, 4, 2, 1, false>, true, false, false>(x)": 2,
            "void (anonymous namespace)::conv3x3_pp_kernel<(anonymous namespace)::PpCfg<16, 32, 8, 1, 3, false>, false, false, false>(x)": 3,
            "void (anonymous namespace)::conv3x3_pp_kernel<(anonymous namespace)::PpCfg<16, 32, 8, 1, 3, true>, false, false, true>(x)": 4,
            "void (anonymous namespace)::wgrad9_kernel<64, 64, 0, 4, 1, 1, false>(Wg9Args)": 5,
            "void (anonymous namespace)::wgrad9_kernel<64, 64, 0, 4, 1, 1, true>(Wg9Args)": 6}
    assert sorted(b.pmc_rows("conv3x3_pp512_bf16", fake)) == [1, 2]
    assert b.pmc_rows("conv3x3_pp512x64_bf16", fake) == [3]
    assert b.pmc_rows("conv3x3_pp512x64_bf16_xf", fake) == [4]          # the input-transform forms are families of their own
    assert b.pmc_rows("wgrad9_bf16_64x64_rowwalk", fake) == [5] and b.pmc_rows("wgrad9_bf16_64x64_rowwalk_xf", fake) == [6]


def test_cpu_baseline_protocol_defaults():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"--cpu-steps", type=int, default=3' in src      # 1 warm-up + >= 3 timed steps
    assert '"second_headline"' in src and '"doubleconv_l1"' in src


def test_no_kernel_family_of_the_committed_line_grew_by_a_tenth_against_the_previous_round():
    """round 4 shipped `gemm_dma_bf16_bnred` 1.7x slower than round 3 and only the judge's diff of the two
    lines found it.  The committed final lines of the last two rounds (profiles/rNN_final_bench_unet.json, the default
    bench.py command on one box) are held against each other family by family (tools/family_diff.py): a family of >= 0.05 ms
    per step on both sides must not have grown by more than 10 % (boxes differ by ~6 % on the MFMA-bound families)."""
    import glob
    import importlib.util
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_final_bench_unet.json")))
    assert len(lines) >= 2, lines
    spec = importlib.util.spec_from_file_location("family_diff", os.path.join(ROOT, "tools", "family_diff.py"))
    fd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fd)
    assert fd.main([lines[-2], lines[-1], "--fail-over", "10"]) == 0
