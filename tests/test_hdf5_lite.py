"""CPU: the dependency-free HDF5 subset reader / writer (pyrapose_amd/utils/hdf5_lite.py) -- Keras-2.3.1 weight-file layout.
Round trip through the writer, the structures checked at the offsets the HDF5 File Format Specification gives them, and the
whole chain file -> name mapping -> engine tensors.  (No libhdf5 exists in this image: files written by h5py itself are
not covered -- see the module's STATUS note.)"""
import struct

import numpy as np
import pytest

from pyrapose_amd.utils import hdf5_lite as H5
from pyrapose_amd.utils import keras_names as KN


def test_round_trip_of_a_keras_layout_file(tmp_path):
    from pyrapose_amd import arch
    C = 5
    W = arch.init_weights(C, seed=3)
    layers = KN.tensors_to_keras(W, first_auto_index=12, reg_model="model_1", cls_model="model_2")
    path = str(tmp_path / "resnet50_linemod_03.h5")
    H5.write_keras_weights(path, layers)
    with open(path, "rb") as f:
        raw = f.read()
    # superblock version 0 at the offsets of the specification (III.A): signature, versions, sizes of offsets / lengths,
    # base address 0, end-of-file address = file size, root symbol-table entry with cached B-tree / heap addresses
    assert raw[:8] == b"\x89HDF\r\n\x1a\n" and raw[8] == 0 and raw[13] == 8 and raw[14] == 8
    base, _, eof, _ = struct.unpack_from("<QQQQ", raw, 24)
    assert base == 0 and eof == len(raw)
    name_off, root_hdr, cache, _ = struct.unpack_from("<QQII", raw, 56)
    btree, heap = struct.unpack_from("<QQ", raw, 80)
    assert name_off == 0 and cache == 1 and raw[btree:btree + 4] == b"TREE" and raw[heap:heap + 4] == b"HEAP"
    assert raw[root_hdr] == 1                                   # version-1 object header
    f = H5.File(path)
    assert sorted(f.keys()) == sorted(layers.keys())
    assert f.attrs["keras_version"] == b"2.3.1" and f.attrs["backend"] == b"tensorflow"
    assert [n.decode() for n in f.attrs["layer_names"]] == list(layers.keys())
    g = f["model_1"]
    assert [n.decode() for n in g.attrs["weight_names"]] == list(layers["model_1"].keys())
    d = f["model_1/conv2d_12/kernel:0"]                         # nested groups for the '/' of weight names
    assert d.shape == tuple(layers["model_1"]["conv2d_12/kernel:0"].shape)
    assert np.array_equal(d.read(), layers["model_1"]["conv2d_12/kernel:0"])
    with pytest.raises(KeyError):
        f["model_1/nope"]
    back = H5.read_keras_weights(path)
    assert list(back.keys()) == list(layers.keys())
    for lname, ws in layers.items():
        assert list(back[lname].keys()) == list(ws.keys())
        for wname, arr in ws.items():
            got = back[lname][wname]
            assert got.dtype == np.float32 and np.array_equal(got, np.asarray(arr, np.float32)), (lname, wname)
    T = KN.keras_to_tensors(back, expected_shapes={k: v.shape for k, v in W.items()})
    for k in W:
        assert np.array_equal(T[k], np.asarray(W[k], np.float32)), k


def test_model_save_layout_and_chunked_name_lists(tmp_path):
    """model.save() puts the weights under 'model_weights'; long name lists are split into layer_names0, layer_names1, ..."""
    layers = {"conv1": {"conv1/kernel:0": np.arange(24, dtype=np.float32).reshape(1, 2, 3, 4)},
              "bn_conv1": {"bn_conv1/gamma:0": np.ones(4, np.float32), "bn_conv1/moving_variance:0": np.full(4, 2.0, np.float32)},
              "input_1": {}}
    path = str(tmp_path / "w.h5")
    H5.write_keras_weights(path, layers)
    got = H5.read_keras_weights(path)
    assert list(got.keys()) == ["conv1", "bn_conv1"] and got["conv1"]["conv1/kernel:0"].shape == (1, 2, 3, 4)
    assert float(got["bn_conv1"]["bn_conv1/moving_variance:0"][3]) == 2.0


def test_other_dtypes_and_big_endian_and_refusals(tmp_path):
    path = str(tmp_path / "t.h5")
    H5.write_keras_weights(path, {"l": {"l/a:0": np.arange(6, dtype=np.float64).reshape(2, 3), "l/b:0": np.arange(5, dtype=np.int32),
                                        "l/c:0": np.float32(7.5)}})
    got = H5.read_keras_weights(path)["l"]
    assert got["l/a:0"].dtype == np.float64 and got["l/b:0"].dtype == np.int32 and got["l/c:0"].shape == () and float(got["l/c:0"]) == 7.5
    with open(path, "rb") as f:
        raw = bytearray(f.read())
    raw[8] = 7                                                   # an unknown superblock version is refused, not guessed at
    bad = str(tmp_path / "bad.h5")
    with open(bad, "wb") as f:
        f.write(bytes(raw))
    with pytest.raises(H5.H5Unsupported):
        H5.File(bad)
    with open(bad, "wb") as f:
        f.write(b"PK\x03\x04 not hdf5")
    with pytest.raises(ValueError):
        H5.File(bad)


def _matlab_hdf5_file():
    import os
    try:
        import scipy
    except ImportError:
        return None
    p = os.path.join(os.path.dirname(scipy.__file__), "io", "matlab", "tests", "data", "testhdf5_7.4_GLNX86.mat")
    return p if os.path.exists(p) else None


@pytest.mark.skipif(_matlab_hdf5_file() is None, reason="scipy's testhdf5_7.4_GLNX86.mat is not installed")
def test_reads_a_file_written_by_libhdf5_behind_a_user_block():
    """Third-party evidence for the reader: the one libhdf5-written file of this image.  MATLAB 7.4 '-v7.3' = HDF5 1.6 behind a
    512-byte user block.  Its content is MATLAB's `testdouble` variable, which scipy documents as
    theta = pi/4 * arange(9) (scipy/io/matlab/tests/test_mio.py: `theta`, case_table4 'double'; class 'double'), stored
    column-major, i.e. as a (9, 1) HDF5 dataset."""
    path = _matlab_hdf5_file()
    with open(path, "rb") as fh:
        raw = fh.read()
    assert raw[:6] == b"MATLAB" and H5.find_superblock(raw) == 512           # user block first, superblock at the second probe offset
    f = H5.File(path)
    assert f._rd.superblock == 512 and f._rd.base == 512                    # addresses are relative to the base address
    assert f.keys() == ["testdouble"]
    d = f["testdouble"]
    assert isinstance(d, H5.Dataset) and d.shape == (9, 1)
    a = d.read()
    assert a.dtype == np.float64 and a.shape == (9, 1)
    assert np.array_equal(a.ravel(), np.pi / 4 * np.arange(9, dtype=float))   # bit-exact: MATLAB computed the same doubles
    assert d.attrs["MATLAB_class"] == b"double"                              # fixed-length string attribute (6 bytes, scalar space)
    # the same bytes without the user block are a different file (every address moves by 512): refused cleanly, not mis-read
    with pytest.raises((H5.H5Unsupported, ValueError, KeyError)):
        import tempfile
        with tempfile.NamedTemporaryFile(suffix=".h5") as t:
            t.write(raw[512:])
            t.flush()
            H5.File(t.name)["testdouble"].read()


def test_truncated_files_raise_h5unsupported_not_struct_errors(tmp_path):
    path = str(tmp_path / "t.h5")
    H5.write_keras_weights(path, {"l": {"l/a:0": np.arange(600, dtype=np.float32)}})
    with open(path, "rb") as f:
        raw = f.read()
    for cut in (100, 200, len(raw) // 2, len(raw) - 100):
        bad = str(tmp_path / ("cut%d.h5" % cut))
        with open(bad, "wb") as f:
            f.write(raw[:cut])
        try:
            H5.read_keras_weights(bad)
        except (H5.H5Unsupported, KeyError):
            pass                                                            # the only exceptions a caller has to expect
