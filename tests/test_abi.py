"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/isx.h declares, agrees with the oracle on host-side tables and fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mod():
    import altair_raytracing_amd as m
    m.load()
    return m


def _declared():
    text = open(os.path.join(ROOT, "include", "isx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(isx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(mod):
    lib = mod.load()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/isx.h but not exported by libisx.so"
    assert sorted(mod.EXPORTS) == names


def test_abi_version_and_struct_layout(mod):
    assert mod.load().isx_abi_version() == 3
    assert mod.load().isx_stream_version() == 4
    # isx_config: 2 u32, 6 dbl, 2 i32, 6 dbl, 2 i32, 3 dbl, 2 i32, 3 dbl, 2 i32 ; isx_stats: 7 u64 + dbl
    assert C.sizeof(mod.Config) == 8 + 8 * 6 + 8 + 8 * 6 + 8 + 8 * 3 + 8 + 8 * 3 + 8
    assert mod.Config.struct_size.offset == 0
    assert mod.default_config().struct_size == C.sizeof(mod.Config)
    assert C.sizeof(mod.Stats) == 64


def test_config_of_another_abi_is_refused(mod):
    # ABI v3: a struct whose struct_size is not the library's is refused before anything reads its fields
    # (no GPU needed: isx_detector_table is host code)
    c = mod.default_config()
    out = np.zeros(c.n_theta * c.n_phi * 6)
    for bad in (0, C.sizeof(mod.Config) - 8, C.sizeof(mod.Config) + 8):
        c.struct_size = bad
        rc = mod.load().isx_detector_table(C.byref(c), out.ctypes.data_as(C.POINTER(C.c_double)))
        assert rc == -2, (bad, rc)


def test_default_config_matches_reference_constants(mod, orc):
    c, o = mod.default_config(), orc.default_config()
    assert bytes(c) == bytes(o)
    # fluxAtObserverOptimize.C:33-41,199,456-461,495 ; sweepSeries :892-896
    assert (c.r_in, c.r_out, c.theta_max_deg, c.reflectance, c.roughness_rad, c.box_half) == (100.1, 101.0, 170.0, 0.99, 0.01, 300.0)
    assert (c.lambertian, c.max_points, c.n_theta, c.n_phi) == (1, 50000, 180, 90)
    assert list(c.src) == [-60.0, 0.0, -75.0] and list(c.dir) == [5.0, 0.0, 0.0]
    assert (c.det_diameter, c.det_distance, c.exit_port_z) == (40.0, 100.0, -100.0)


def test_detector_table_bit_exact_with_oracle(mod, orc):
    for nt, nph, dist, pz in [(180, 90, 100.0, -100.0), (45, 20, 100.0, -100.0), (7, 13, 55.5, -90.0)]:
        c, o = mod.default_config(), orc.default_config()
        for k in (c, o):
            k.n_theta, k.n_phi, k.det_distance, k.exit_port_z = nt, nph, dist, pz
        a, b = mod.detector_table(c), orc.detector_table(o)
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_no_cpu_fallback(mod):
    """Without a device the product refuses to compute (it never routes to a CPU path)."""
    lib = mod.load()
    rc = lib.isx_init(0)
    if rc == 0:
        lib.isx_shutdown()
        pytest.skip("a GPU is present; the no-device behaviour is checked in the CPU container")
    assert rc == mod.abi.ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.isx_strerror(rc)
    with pytest.raises(mod.IsxError) as e:
        mod.fluxmap(mod.default_config(), 10, 1)
    assert e.value.status == mod.abi.ERR_NOT_INIT
    with pytest.raises(mod.IsxError):
        mod.trace_endstates(mod.default_config(), 10, 1)


def test_product_does_not_reference_the_oracle():
    """Nothing under altair-raytracing_amd/ may include, link or import anything from oracle/."""
    pkg = os.path.join(ROOT, "altair-raytracing_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".so", ".o", ".pyc")):
                continue
            text = open(os.path.join(dp, f), errors="ignore").read()
            assert "isx_oracle" not in text and "isxo_" not in text and "libisx_oracle" not in text, os.path.join(dp, f)
            assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(dp, f)
    import subprocess
    out = subprocess.run(["ldd", os.path.join(pkg, "csrc", "libisx.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_shard_partition():
    from altair_raytracing_amd import shard
    for n in (0, 1, 7, 50_000_000, 10 ** 9 + 3):
        for w in (1, 2, 3, 8):
            parts = [shard(n, r, w) for r in range(w)]
            assert parts[0][0] == 0
            assert sum(c for _, c in parts) == n
            for (f0, c0), (f1, _) in zip(parts, parts[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    with pytest.raises(ValueError):
        shard(10, 3, 2)


def test_no_generic_pointer_access_and_no_scratch_in_the_binning_kernels():
    """VERDICT r03 (weak #3): a pointer that reaches a kernel through the LDS copy of a parameter block is a generic pointer to hipcc, and
    an access through it a `flat_*` (out-of-order completion: the wave waits for vmcnt(0) AND lgkmcnt(0)); a value kept alive across the
    inlined consumer of a binning kernel at its 128-VGPR limit goes to scratch and back per batch (0.8 GB of HBM writes per launch in
    round 3).  tools/isa_stats.py reads both off the gfx950 ISA (cross-compiled here, no GPU needed): no `flat_` access in ANY kernel of
    the library, no scratch in any binning kernel, none in the headline trace kernel."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_stats", os.path.join(ROOT, "tools", "isa_stats.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    seen = {}
    for name, body in m.kernels(m.compile_isa([])):
        seen[name] = m.census(body)
    assert len(seen) >= 25 and "isx_bin_cols_kernel" in seen
    for name, c in seen.items():
        assert c["flat_ld"] + c["flat_st"] + c["flat_at"] == 0, (name, c)
    for name in ("isx_bin_cols_kernel", "isx_bin_slots_kernel", "isx_bin_lines_kernel", "isx_bin_discs_kernel", "isx_trace_assist_kernel"):
        assert seen[name]["scratch"] == 0 and seen[name]["scr_ld"] + seen[name]["scr_st"] == 0, (name, seen[name])
