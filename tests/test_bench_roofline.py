"""bench.py's roofline comes from a PMC profile of THIS build or not at all (r01 quoted numbers copied
from a fixed profile file whatever binary was running)."""
import json
import os

import bench
from renderbaby_amd import _lib


def _write(root, name, **kw):
    os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
    d = {"kernel": "k_trace", "source_fingerprint": _lib.source_fingerprint(),
         "per_segment": {"valu_instr": 27.8, "hbm_bytes": 6.7, "l2_bytes": 17.0, "tcp_accesses": 5.1}, "derived": {}}
    d.update(kw)
    with open(os.path.join(root, "profiles", name), "w") as f:
        json.dump(d, f)


def test_profile_of_this_build_is_used(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    _write(str(tmp_path), "r02_c2_pmc.json")
    pmc, note = bench.load_pmc("c2", "k_trace", _lib.source_fingerprint())
    assert note is None and pmc["per_segment"]["valu_instr"] == 27.8 and pmc["_path"].endswith("r02_c2_pmc.json")


def test_profile_of_another_build_is_refused(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    _write(str(tmp_path), "r01_c2_pmc.json", source_fingerprint="0123456789abcdef")
    pmc, note = bench.load_pmc("c2", "k_trace", _lib.source_fingerprint())
    assert pmc is None and "refused" in note
    # ... and so is one of another kernel or another workload
    _write(str(tmp_path), "r02_c2_pmc.json", kernel="k_trace_bvh")
    assert bench.load_pmc("c2", "k_trace", _lib.source_fingerprint())[0] is None
    _write(str(tmp_path), "r02_c3_pmc.json")
    assert bench.load_pmc("c2", "k_trace", _lib.source_fingerprint())[0] is None


def test_fingerprint_follows_the_sources(tmp_path):
    a = _lib.source_fingerprint()
    assert len(a) == 16 and a == _lib.source_fingerprint()


def test_roofline_peaks_are_the_guides_figures():
    # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, a wave64 VALU instruction issues over 2 cycles, 2.4 GHz; HBM3E 8 TB/s
    assert abs(bench.VALU_PEAK_GINSTR - 1228.8) < 1e-9 and bench.HBM_PEAK_GBPS == 8000.0
