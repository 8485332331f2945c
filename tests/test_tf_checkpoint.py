"""CPU tests of the TensorFlow tensor-bundle reader / writer and the variable-name map.

No TensorFlow-written file exists offline, so these pin the implementation against hand-assembled
bytes of the documented format (varints, protobuf fields, table footer, masked crc32c known answers)
and a write -> read round trip; see the PARITY UNPINNED note in mvsnet_amd/tf_checkpoint.py."""
import os
import struct

import numpy as np
import pytest

from mvsnet_amd import tf_checkpoint as ck


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors for CRC32C (Castagnoli)
    assert ck.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert ck.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert ck.crc32c(bytes(range(32))) == 0x46DD794E
    assert ck.crc32c(b"123456789") == 0xE3069283
    # the masking of leveldb / tensorflow: rotate right by 15 and add a constant
    assert ck.mask_crc(0) == 0xA282EAD8
    assert ck.mask_crc(0xE3069283) == (((0xE3069283 >> 15) | (0xE3069283 << 17)) + 0xA282EAD8) & 0xFFFFFFFF
    # the chunked / vectorised path (long buffers) against the bytewise definition, with and without a seed
    d = np.random.RandomState(5).bytes(300_007)
    assert ck.crc32c(d) == ck._crc_bytes(d)
    assert ck.crc32c(d[1234:], ck._crc_bytes(d[:1234])) == ck._crc_bytes(d)


def test_entry_proto_bytes():
    # dtype=DT_FLOAT(1), shape [3,4], offset 16, size 48, crc 0x01020304 -- assembled by hand
    want = bytes([0x08, 0x01,                          # field 1 varint 1
                  0x12, 0x08,                          # field 2, 8 bytes: two Dim messages
                  0x12, 0x02, 0x08, 0x03, 0x12, 0x02, 0x08, 0x04,
                  0x20, 0x10,                          # field 4 offset 16
                  0x28, 0x30,                          # field 5 size 48
                  0x35, 0x04, 0x03, 0x02, 0x01])       # field 6 fixed32
    got = ck._encode_entry(ck.DT_FLOAT, (3, 4), 0, 16, 48, 0x01020304)
    assert got == want
    e = ck._parse_entry(want)
    assert e["dtype"] == 1 and e["shape"] == [3, 4] and e["offset"] == 16 and e["size"] == 48 and e["crc32c"] == 0x01020304
    # multi-byte varints
    assert ck._put_varint(300) == b"\xac\x02" and ck._get_varint(b"\xac\x02", 0) == (300, 2)


def test_round_trip_many_blocks(tmp_path):
    rs = np.random.RandomState(0)
    tensors = {"layer%03d/kernel" % i: rs.standard_normal((3, 3, i % 5 + 1, 2)).astype(np.float32) for i in range(300)}
    tensors["global_step"] = np.array(400000, np.int64)
    tensors["scalar64"] = np.array(1.5, np.float64)
    prefix = str(tmp_path / "m" / "model.ckpt-7")
    ck.write_checkpoint(prefix, tensors)
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == ck.TABLE_MAGIC and len(raw) > 3 * 4096     # several data blocks
    got = ck.read_checkpoint(prefix)
    assert sorted(got) == sorted(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v)
    names = [n for n, _s, _d in ck.list_variables(prefix)]
    assert names == sorted(tensors, key=lambda s: s.encode())
    assert ck.read_checkpoint(prefix, ["global_step"])["global_step"] == 400000
    with pytest.raises(KeyError):
        ck.read_checkpoint(prefix, ["missing/kernel"])
    # a flipped data byte is caught by the per-tensor checksum
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read()); data[5] ^= 0xFF
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(ValueError):
        ck.read_checkpoint(prefix)


def test_variable_map_and_param_round_trip(tmp_path):
    from mvsnet_amd import synthetic as S
    unet, regnet, gru = S.make_unet_params("normal", 3), S.make_regnet_params("normal", 1, random_affine=True), \
        S.make_gru_params("normal", 2, random_affine=True)
    names = ck.variable_names("normal", "3DCNN")
    assert names[("regnet", "3dconv1_0", "w")] == "3dconv1_0/kernel"
    assert names[("regnet", "3dconv6_0", "gamma")] == "3dconv6_0/bn/gamma"
    assert ("regnet", "3dconv6_2", "gamma") not in names            # output conv has no BN (mvsnetworks.py:158)
    assert names[("unet", "2dconv0_1", "beta")] == "2dconv0_1/gn/beta" and ("unet", "conv10_2", "gamma") not in names
    g = ck.variable_names("normal", "GRU")
    assert g[("gru", "gru1", "gates_w")] == "conv_gru1/Gates/conv/kernel"
    assert g[("gru", "gru3", "update_gamma")] == "conv_gru3/Gates/LayerNorm_1/gamma"
    assert g[("gru", "gru2", "out_b")] == "conv_gru2/Output/output_conv/bias" and g[("gru", None, "prob_w")] == "prob_conv/kernel"
    base = str(tmp_path / "models")
    prefix = ck.model_path(ck.ckpt_path(base, "3DCNN", "normal"), 400000)
    assert prefix == os.path.join(base, "3DCNN", "normal", "model.ckpt-400000")
    written = ck.export_mvsnet_params(prefix, unet=unet, regnet=regnet, gru=gru)
    assert "prob_conv/bias" in written and "3dconv0_1/bn/beta" in written
    p = ck.load_mvsnet_params(prefix, "normal", "3DCNN")
    assert p["gru"] is None and sorted(p["regnet"]) == sorted(regnet)
    for layer, d in regnet.items():
        for f, v in d.items():
            assert np.array_equal(p["regnet"][layer][f], v)
    for layer, d in unet.items():
        for f, v in d.items():
            assert np.array_equal(p["unet"][layer][f], v)
    q = ck.load_mvsnet_params(prefix, "normal", "GRU")
    assert q["regnet"] is None and np.array_equal(q["gru"]["gru2"]["gates_w"], gru["gru2"]["gates_w"])
    assert np.array_equal(q["gru"]["prob_b"], gru["prob_b"])
