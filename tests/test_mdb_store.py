"""The replay store's LMDB-format container (csrc/nbp_mdb.cpp through utility/nbp_utils.py::MdbEnv) against an independent pure-Python
reader of the format (tests/mdb_reader.py) and a dict model.  CPU only: the container is host code.  liblmdb is not in this image, so
what is pinned here is the published format (lmdb.h / mdb.c 0.9), not the library."""
import os
import random
import struct

import numpy as np
import pytest
import torch

from nextbestpath_amd.utility import nbp_utils as nu
from tests import mdb_reader


def _check(path, model):
    meta, items, found = mdb_reader.read_env(path)
    assert dict(items) == model
    assert [k for k, _ in items] == sorted(model)
    return meta, found


def test_empty_environment_is_two_meta_pages(tmp_path):
    p = str(tmp_path / "db")
    env = nu.MdbEnv(p, map_size=1 << 30)
    assert env.entries() == 0 and env.keys() == [] and env.get(b"x") is None
    env.close()
    raw = open(os.path.join(p, "data.mdb"), "rb").read()
    assert len(raw) == 2 * 4096
    for which in (0, 1):                                   # mdb_env_init_meta: both pages, txnid 0, empty databases, last page 1
        pg = raw[which * 4096:(which + 1) * 4096]
        assert struct.unpack_from("<QHH", pg, 0) == (which, 0, 8)
        assert struct.unpack_from("<II", pg, 16) == (0xBEEFC0DE, 1)
        assert struct.unpack_from("<Q", pg, 16 + 16)[0] == 1 << 30                      # mm_mapsize
        assert struct.unpack_from("<IHH", pg, 16 + 24) == (4096, 8, 0)                  # mm_psize, MDB_INTEGERKEY, depth 0
        assert struct.unpack_from("<Q", pg, 16 + 24 + 40)[0] == 2 ** 64 - 1             # FREE_DBI root P_INVALID
        assert struct.unpack_from("<Q", pg, 16 + 72 + 40)[0] == 2 ** 64 - 1             # MAIN_DBI root
        assert struct.unpack_from("<QQ", pg, 16 + 120) == (1, 0)                        # last_pg, txnid
    _check(p, {})


def test_first_record_layout_byte_for_byte(tmp_path):
    """One small record: meta page 1 becomes current (txnid 1), page 2 is a leaf whose single node sits at the page's end."""
    p = str(tmp_path / "db")
    env = nu.MdbEnv(p)
    env.put(b"0000000000123", b"hello")
    env.close()
    raw = open(os.path.join(p, "data.mdb"), "rb").read()
    assert len(raw) == 3 * 4096
    m1 = raw[4096:8192]
    assert struct.unpack_from("<QQ", m1, 16 + 120) == (2, 1)                            # last_pg 2, txnid 1 -> page 1
    assert struct.unpack_from("<IHHQQQQQ", m1, 16 + 72) == (0, 0, 1, 0, 1, 0, 1, 2)     # depth 1, 1 leaf, 1 entry, root = page 2
    leaf = raw[8192:]
    node = 8 + 13 + 5 + 0                                   # header + key + data, even
    assert struct.unpack_from("<QHHHH", leaf, 0) == (2, 0, 2, 18, 4096 - node)
    assert struct.unpack_from("<H", leaf, 16)[0] == 4096 - node
    assert leaf[4096 - node:] == struct.pack("<HHHH", 5, 0, 0, 13) + b"0000000000123" + b"hello"
    _check(p, {b"0000000000123": b"hello"})


def test_random_puts_deletes_replacements_against_a_dict(tmp_path):
    """Keys in timestamp form (mostly ascending, some out of order), values from empty to multi-page; deletes in bursts that empty
    whole leaves; reopen after every phase.  The independent reader validates the tree each time."""
    p = str(tmp_path / "db")
    rng = random.Random(5)
    model = {}
    env = nu.MdbEnv(p)
    t = 1_700_000_000_000
    sizes = [0, 1, 5, 100, 2016, 2017, 2018, 2019, 5000, 4080, 4081, 70_000]
    for phase in range(6):
        for _ in range(400):
            t += rng.randrange(1, 50)
            key = f"{(t if rng.random() < 0.9 else rng.randrange(1_600_000_000_000, t)):012d}".encode()
            n = rng.choice(sizes) if rng.random() < 0.5 else rng.randrange(0, 300)
            val = bytes(rng.getrandbits(8) for _ in range(min(n, 64))) * (n // 64 + 1)
            val = val[:n]
            env.put(key, val)
            model[key] = val
        assert env.entries() == len(model)
        ks = sorted(model)
        # a contiguous burst (drains leaves and, sooner or later, a branch page's children) and scattered single keys
        a = rng.randrange(0, max(1, len(ks) - 300))
        for key in ks[a:a + rng.randrange(100, 300)] + rng.sample(ks, 40):
            assert env.delete(key) == (key in model)
            model.pop(key, None)
        assert not env.delete(b"no such key")
        assert env.keys() == sorted(model)
        for key in rng.sample(sorted(model), 25):
            assert env.get(key) == model[key]
        env.close()
        meta, found = _check(p, model)
        assert meta["dbs"][1]["depth"] >= 2 and found["branch"] >= 1
        env = nu.MdbEnv(p)                                  # reloads the tree from the file
        assert env.entries() == len(model) and env.keys() == sorted(model)
    # drain everything: the database ends empty (root P_INVALID, depth 0)
    for key in sorted(model):
        assert env.delete(key)
    assert env.entries() == 0 and env.stat()["depth"] == 0
    env.put(b"0000000000001", b"again")
    env.close()
    _check(p, {b"0000000000001": b"again"})


def test_many_small_records_grow_to_depth_three(tmp_path):
    p = str(tmp_path / "db")
    env = nu.MdbEnv(p)
    model = {}
    for i in range(30_000):
        key = f"{i * 7:012d}".encode()
        env.put(key, b"v" * (i % 9))
        model[key] = b"v" * (i % 9)
    st = env.stat()
    assert st["depth"] == 3 and st["entries"] == 30_000, st
    env.close()
    _check(p, model)
    env = nu.MdbEnv(p)
    for i in range(0, 30_000, 2):                           # thin every leaf out, then check and reopen
        env.delete(f"{i * 7:012d}".encode())
        model.pop(f"{i * 7:012d}".encode())
    env.close()
    _check(p, model)


def test_replay_store_functions_on_the_lmdb_format(tmp_path):
    """store_experience / store_validation_data / read_combined_data (nbp_utils.py:32-141) on the container; records survive reopening."""
    p = str(tmp_path / "replay")
    env = nu.open_experience_db(p) if not _has_lmdb() else nu.MdbEnv(p)
    assert isinstance(env, nu.MdbEnv)
    rng = np.random.default_rng(0)
    recs = []
    for i in range(12):
        rec = {"current_model_input": torch.from_numpy(rng.random((1, 5, 32, 32), dtype=np.float32)),
               "current_gt_2d_layout": torch.from_numpy((rng.random((1, 1, 32, 32)) > 0.5).astype(np.float32)),
               "target_value_map_pixel": torch.from_numpy(rng.integers(0, 8, (4, 3))),
               "actual_coverage_gain": torch.from_numpy(rng.random(4, dtype=np.float32)), "pose_i": i}
        nu.store_experience(env, rec)
        recs.append(rec)
    assert env.entries() == 12
    env.close()
    env = nu.open_experience_db(p) if not _has_lmdb() else nu.MdbEnv(p)
    got = nu.read_combined_data(env, sample_m=None)
    assert [r["pose_i"] for r in got] == list(range(12))
    assert np.array_equal(got[3]["current_model_input"], recs[3]["current_model_input"].numpy())
    val = nu.store_validation_data(env, num=4)              # every third record moves out
    assert [r["pose_i"] for r in val] == [0, 3, 6, 9] and env.entries() == 8
    env.close()
    meta, items, found = mdb_reader.read_env(p)
    assert len(items) == 8 and found["overflow"] > 0        # 20 KB records live in overflow pages
    assert [nu.unpack_record(v)["pose_i"] for _, v in items] == [1, 2, 4, 5, 7, 8, 10, 11]


def test_a_directory_with_the_old_log_keeps_opening_as_a_log(tmp_path):
    p = str(tmp_path / "old")
    log = nu.LogEnv(p)
    log.put(b"0000000000001", b"x")
    if not _has_lmdb():
        assert isinstance(nu.open_experience_db(p), nu.LogEnv)


def test_files_that_are_not_lmdb_are_refused(tmp_path):
    p = tmp_path / "bad"
    p.mkdir()
    (p / "data.mdb").write_bytes(b"\x00" * 8192)
    with pytest.raises(Exception):
        nu.MdbEnv(str(p))


def _has_lmdb():
    try:
        import lmdb  # noqa: F401
        return True
    except ImportError:
        return False


def test_long_keys_of_unequal_length(tmp_path):
    """Keys from 1 to 511 bytes (LMDB's limit), so that branch pages hold few separators of very different sizes: splits pick a cut with
    both halves inside a page, and a delete that swaps a short separator for a long one may split its parent (erase returns a sibling)."""
    p = str(tmp_path / "db")
    rng = random.Random(11)
    env = nu.MdbEnv(p)
    model = {}
    alphabet = b"abcdefghijklmnopqrstuvwxyz"
    for round_ in range(4):
        for _ in range(1500):
            n = rng.choice([1, 2, 7, 64, 200, 400, 510, 511])
            key = bytes(rng.choice(alphabet) for _ in range(min(n, 6))) + bytes([rng.randrange(97, 123)]) * max(0, n - 6)
            val = b"x" * rng.choice([0, 3, 900, 1500, 3000])
            env.put(key, val)
            model[key] = val
        ks = sorted(model)
        for key in rng.sample(ks, len(ks) // 2):
            assert env.delete(key)
            del model[key]
        assert env.keys() == sorted(model)
        env.close()
        _check(p, model)
        env = nu.MdbEnv(p)
    with pytest.raises(RuntimeError):
        env.put(b"k" * 512, b"too long a key")
    env.close()
