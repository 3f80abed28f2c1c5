"""Replay of a reference-side ROBAST bounce dump through the CPU oracle (the one thing that can pin
SURVEY.md section 8c).  tools/ref_dump/dumpBounces.C writes the file on a machine that has ROOT + ROBAST; nobody has yet, so
the real test is skipped and the committed SYNTHETIC file (the oracle's own rays in the same format) keeps reader and
replay working.  CPU only."""
import os

import pytest

import robast_dump

HERE = os.path.dirname(os.path.abspath(__file__))
REAL = os.path.join(HERE, "golden", "robast_bounces.txt")
SYNTHETIC = os.path.join(HERE, "golden", "robast_bounces_synthetic.txt")


def test_replay_of_the_synthetic_dump(orc):
    out = robast_dump.replay(SYNTHETIC)
    assert out["synthetic"] and out["rays"] == 24 and out["segments"] > 500 and out["emissions"] > 500
    assert out["polar_form"] == "sqrt(1-u)" and out["detector_flags"] == 24 * 5


def test_the_synthetic_dump_is_what_the_oracle_writes_today(orc, tmp_path):
    """(a changed oracle must come with a regenerated file: python -c 'import robast_dump; robast_dump.write_synthetic(...)')"""
    p = tmp_path / "again.txt"
    robast_dump.write_synthetic(str(p))
    assert p.read_text() == open(SYNTHETIC).read()


def test_replay_notices_a_wrong_emission_law(orc, tmp_path):
    """Sanity of the checker: a dump whose directions were not drawn by the cosine law about the geometric normal fails."""
    head, dets, rays = robast_dump.read(SYNTHETIC)
    lines = open(SYNTHETIC).read().split("\n")
    k = [i for i, ln in enumerate(lines) if ln.startswith("U ")][0]
    vals = lines[k].split()[1:]
    vals[0] = repr(float(vals[0]) * 0.5 + 0.1)          # the first polar uniform of ray 0 no longer matches its segment
    lines[k] = "U " + " ".join(vals)
    p = tmp_path / "bad.txt"
    p.write_text("\n".join(lines))
    with pytest.raises(AssertionError):
        robast_dump.replay(str(p))


@pytest.mark.skipif(not os.path.exists(REAL), reason="tests/golden/robast_bounces.txt: no ROBAST-side dump has been produced yet "
                    "(tools/ref_dump/dumpBounces.C, INTEGRATION.md 'Pinning the oracle'); parity stays unpinned")
def test_replay_of_the_robast_dump(orc):
    out = robast_dump.replay(REAL)
    print(out)
    assert not out["synthetic"] and out["rays"] >= 100 and out["emissions"] > 1000
