"""bench.py's bookkeeping that needs no GPU: the committed PMC summary (profiles/pmc_summary.json) has the shape bench.py's
rooflines consume -- the headline pair and one section per extra leg -- and, when its fingerprint matches the kernel sources of this
tree, every issue block computes (a stale summary is reported by the bench line itself as STALE, not silently used)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_pmc_summary_shape_and_issue_blocks():
    import bench
    raw = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
    assert set(raw["kernels"]) == {"isx_trace_assist_kernel", "isx_bin_cols_kernel"}
    assert {"brdf", "discs", "lobe", "rough"} <= set(raw.get("sections", {}))
    everything = list(raw["kernels"].items()) + [kv for sec in raw["sections"].values() for kv in sec["kernels"].items()]
    for name, pk in everything:
        for key in ("kernel_ms", "valu_wave_insts_per_ray", "valu_lane_utilization", "hbm_bytes_per_launch", "issue_mix", "kernel_source_sha"):
            assert key in pk, (name, key)
        assert pk["kernel_source_sha"] == raw["kernel_source_sha"]
        mix = pk["issue_mix"]
        assert 0 < mix["cycles_per_ray_spec"] <= mix["cycles_per_ray"] and mix["cycles_spec"]["f64"] == 4.0
        blk = bench.issue_block(name, pk["kernel_ms"], pk, pk["rays_per_launch"], 256)
        # at its own profiled time a kernel's fractions are those of its summary
        assert abs(blk["frac"] - mix["frac_at_2p4GHz"]) < 1e-9 and abs(blk["frac_spec"] - mix["frac_spec_at_2p4GHz"]) < 1e-9
        assert 0.3 < blk["frac_spec"] <= blk["frac"] < 1.05, (name, blk["frac_spec"], blk["frac"])
    pj, note = bench.load_pmc()
    if raw["kernel_source_sha"] == bench.kernel_source_sha():
        assert pj and "STALE" not in note
    else:
        assert pj == {} and "STALE" in note      # kernels changed since the last profile pass: bench.py says so and omits the fractions


def test_extra_leg_keys_are_in_the_line():
    """The keys the N = 1 line promises (VERDICT r04 'next' 2) exist in bench.py's output dict (static check of the source)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ('"configs0"', '"configs2"', '"configs3"', '"perpos_8p1e8"', '"size_sweep"', '"surfaces"', '"chord_mode"', '"extra_legs_error"',
                '"roofline"', '"cpu_baseline"', '"frac_spec"'):
        assert key in src or key.strip('"') + "=" in src, key
