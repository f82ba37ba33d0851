"""The reference-fixture scenes live in the package (bench.py uses them too): renderbaby_amd/refscenes.py."""
from renderbaby_amd.refscenes import ref_cornell, ref_cornell_mesh, ref_lamp  # noqa: F401
