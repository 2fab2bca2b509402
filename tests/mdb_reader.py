"""Independent pure-Python READER of LMDB's on-disk format (test infrastructure; no code shared with csrc/nbp_mdb.cpp).

Restated from LMDB 0.9's lmdb.h / mdb.c: MDB_page (pgno u64, pad u16, flags u16, lower u16 / upper u16 or pages u32), MDB_node (lo u16, hi u16,
flags u16, ksize u16, data), MDB_meta (magic 0xBEEFC0DE, version 1, address, mapsize, MDB_db[2], last_pg, txnid), MDB_db (pad u32, flags u16,
depth u16, branch / leaf / overflow pages, entries, root).  `walk` returns the records in key order and checks what mdb.c asserts or relies
on: page numbers match positions, P_LEAF / P_BRANCH flags, pointer array sorted by key, nodes inside [upper, page end), branch pages with
more than one key and an empty key on node 0, every leaf at depth md_depth, md_entries / md_*_pages equal to what the walk finds, overflow
runs inside the file (last_pg).  liblmdb itself is not in this image: parity against it stays unpinned."""
import os
import struct

PSIZE, PHDR = 4096, 16
P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 1, 2, 4, 8
F_BIGDATA = 1
P_INVALID = 2 ** 64 - 1


def _meta(buf, which):
    pg = buf[which * PSIZE:(which + 1) * PSIZE]
    pgno, pad, flags = struct.unpack_from("<QHH", pg, 0)
    assert pgno == which and flags & P_META, (pgno, flags)
    magic, version, address, mapsize = struct.unpack_from("<IIQQ", pg, PHDR)
    assert magic == 0xBEEFC0DE and version == 1, (hex(magic), version)
    dbs = []
    for d in range(2):
        md_pad, md_flags, md_depth, br, lf, ov, ent, root = struct.unpack_from("<IHHQQQQQ", pg, PHDR + 24 + 48 * d)
        dbs.append(dict(pad=md_pad, flags=md_flags, depth=md_depth, branch_pages=br, leaf_pages=lf, overflow_pages=ov, entries=ent, root=root))
    last_pg, txnid = struct.unpack_from("<QQ", pg, PHDR + 120)
    return dict(mapsize=mapsize, dbs=dbs, last_pg=last_pg, txnid=txnid)


def read_env(path):
    """-> (meta, [(key, value)] in key order, stats) of <path>/data.mdb, with every structural check applied."""
    file = os.path.join(path, "data.mdb")
    with open(file, "rb") as fh:
        buf = fh.read()
    assert len(buf) >= 2 * PSIZE
    metas = [_meta(buf, 0), _meta(buf, 1)]
    meta = metas[1] if metas[1]["txnid"] > metas[0]["txnid"] else metas[0]
    free_db, main = meta["dbs"]
    assert free_db["pad"] == PSIZE and free_db["flags"] & 8, free_db                 # mm_psize, MDB_INTEGERKEY
    assert (meta["last_pg"] + 1) * PSIZE <= len(buf), (meta["last_pg"], len(buf))
    assert meta["mapsize"] >= (meta["last_pg"] + 1) * PSIZE
    found = dict(branch=0, leaf=0, overflow=0, entries=0)
    out = []

    def page(pgno):
        assert 2 <= pgno <= meta["last_pg"], pgno
        return buf[pgno * PSIZE:(pgno + 1) * PSIZE]

    def walk(pgno, depth, lo_key):
        pg = page(pgno)
        no, pad, flags, lower, upper = struct.unpack_from("<QHHHH", pg, 0)
        assert no == pgno and pad == 0, (no, pgno)
        assert flags in (P_BRANCH, P_LEAF), flags
        assert PHDR <= lower <= upper <= PSIZE and (lower - PHDR) % 2 == 0
        nk = (lower - PHDR) // 2
        ptrs = struct.unpack_from("<%dH" % nk, pg, PHDR)
        assert all(upper <= p < PSIZE and p % 2 == 0 for p in ptrs), ptrs
        assert len(set(ptrs)) == nk
        prev = None
        if flags == P_LEAF:
            assert depth == main["depth"], (depth, main["depth"])
            assert nk >= 1
            found["leaf"] += 1
            for p in ptrs:
                lo, hi, nf, ks = struct.unpack_from("<HHHH", pg, p)
                key = bytes(pg[p + 8:p + 8 + ks])
                assert 1 <= ks <= 511
                assert prev is None or prev < key, (prev, key)
                assert lo_key is None or key >= lo_key, (lo_key, key)
                prev = key
                size = lo | (hi << 16)
                assert nf in (0, F_BIGDATA), nf
                if nf & F_BIGDATA:
                    assert 8 + ks + size > 2038                                        # me_nodemax
                    ov = struct.unpack_from("<Q", pg, p + 8 + ks)[0]
                    opg = page(ov)
                    ono, opad, oflags, npages = struct.unpack_from("<QHHI", opg, 0)
                    assert ono == ov and oflags == P_OVERFLOW and npages == (PHDR - 1 + size) // PSIZE + 1, (ono, oflags, npages)
                    assert ov + npages - 1 <= meta["last_pg"]
                    found["overflow"] += npages
                    val = bytes(buf[ov * PSIZE + PHDR:ov * PSIZE + PHDR + size])
                else:
                    assert 8 + ks + size <= 2038 and p + 8 + ks + size <= PSIZE
                    val = bytes(pg[p + 8 + ks:p + 8 + ks + size])
                out.append((key, val))
                found["entries"] += 1
            return
        found["branch"] += 1
        assert nk > 1, "mdb_page_search_root asserts NUMKEYS > 1 on branch pages"
        bound = lo_key
        kids = []
        for i, p in enumerate(ptrs):
            lo, hi, nf, ks = struct.unpack_from("<HHHH", pg, p)
            child = lo | (hi << 16) | (nf << 32)
            key = bytes(pg[p + 8:p + 8 + ks])
            if i == 0:
                assert ks == 0, "node 0 of a branch page carries no key"
            else:
                assert ks >= 1 and (prev is None or prev < key), (prev, key)
                assert lo_key is None or key >= lo_key
                prev = key
                bound = key
            kids.append((child, bound))
        for j, (child, b) in enumerate(kids):
            n0 = len(out)
            walk(child, depth + 1, b)
            # everything under child j sorts below the next separator
            if j + 1 < len(kids):
                assert all(k < kids[j + 1][1] for k, _ in out[n0:]), (j, kids[j + 1][1])

    if main["root"] != P_INVALID:
        walk(main["root"], 1, None)
    else:
        assert main["depth"] == 0 and main["entries"] == 0
    assert found["entries"] == main["entries"], (found, main)
    assert found["branch"] == main["branch_pages"] and found["leaf"] == main["leaf_pages"], (found, main)
    assert found["overflow"] == main["overflow_pages"], (found, main)
    assert [k for k, _ in out] == sorted(k for k, _ in out)
    return meta, out, found
