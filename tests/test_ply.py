"""CPU tests of the PLY export / import (SURVEY.md 8(f) "next" row 4): byte format against files written by
the reference's OWN vendored tinyply (tests/golden/ply_*.ply, produced by oracle/_ref/write_ply through
tests/golden/make_golden_ply.py in the build container) -- this row's parity is PINNED to reference-library
output, not to a restatement."""
import os

import numpy as np
import pytest

from gs_livm_amd import ply

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {"ply_P7_M4": (7, 4), "ply_P300_M1": (300, 1), "ply_P33_M16": (33, 16)}


def _model(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")}


@pytest.mark.parametrize("name", sorted(CASES))
def test_writer_is_byte_identical_to_the_references_tinyply(name):
    P, M = CASES[name]
    m = _model(name)
    rows = ply.rows_numpy(**m)
    assert rows.shape == (P, 14 + 3 * M)
    want = open(os.path.join(GOLDEN, name + ".ply"), "rb").read()
    assert ply.ply_bytes(rows, M) == want


def test_attribute_order_is_the_references():
    # construct_list_of_attributes, src/gs/gaussian.cu:474-492
    n = ply.attribute_names(4)
    assert n[:6] == ["x", "y", "z", "nx", "ny", "nz"] and n[6:9] == ["f_dc_0", "f_dc_1", "f_dc_2"]
    assert n[9:18] == ["f_rest_%d" % i for i in range(9)]
    assert n[18:] == ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert len(ply.attribute_names(1)) == 17 and len(ply.attribute_names(16)) == 62


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_files_load_back_to_the_model(name, tmp_path):
    """load_ply on a file the reference's library wrote returns the tensors it was written from (this also checks
    the channel-major f_dc / f_rest column order in the reading direction)."""
    m = _model(name)
    got = ply.load_ply(os.path.join(GOLDEN, name + ".ply"))
    for k, v in m.items():
        assert got[k].shape == v.shape and np.array_equal(got[k], v), k


def test_foreign_layout_is_rejected(tmp_path):
    bad = tmp_path / "bad.ply"
    bad.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 0\nend_header\n")
    with pytest.raises(ValueError):
        ply.load_ply(str(bad))
    m = _model("ply_P7_M4")
    blob = ply.ply_bytes(ply.rows_numpy(**m), 4).replace(b"property float opacity\n", b"property float alpha\n")
    bad.write_bytes(blob)
    with pytest.raises(ValueError, match="property"):
        ply.load_ply(str(bad))
