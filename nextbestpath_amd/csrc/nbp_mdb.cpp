// nbp_mdb.cpp -- the replay store's container in LMDB's on-disk format (host only, no device code).
//
// The reference keeps its experience records in an LMDB environment (next_best_path/trainers/train_nbp_model.py:61-63:
// lmdb.open(path, map_size); next_best_path/utility/nbp_utils.py:32-141: one write transaction per record, ordered cursors, deletes
// of the validation records).  liblmdb / py-lmdb are not in this image and cannot be installed, so this file restates the FILE
// FORMAT of LMDB 0.9.x (the library py-lmdb 1.4 bundles: lmdb.h / mdb.c, MDB_DATA_VERSION 1, 4096-byte pages) for what the
// reference uses of it -- the unnamed main database, plain keys (memcmp order), no duplicates, no sub-databases:
//   <path>/data.mdb = [meta page 0][meta page 1][pages ...]
//   page header (16 B): pgno u64 | pad u16 | flags u16 (P_BRANCH 1, P_LEAF 2, P_OVERFLOW 4, P_META 8) | lower u16, upper u16
//                       (overflow pages: page count u32 in place of lower / upper); node pointers u16[] from byte 16 upwards,
//                       nodes from the page's end downwards; lower / upper are absolute offsets within the page.
//   node (8 B + key + data, even-sized): lo u16 | hi u16 | flags u16 | ksize u16 | key | data.  Leaf: data size = lo | hi << 16;
//                       F_BIGDATA (1): the data is the u64 number of the first overflow page.  Branch: child = lo | hi << 16 |
//                       flags << 32, node 0 carries no key.  A record goes to overflow pages when 8 + ksize + dsize > 2038.
//   meta (behind the page header): magic 0xBEEFC0DE u32 | version 1 u32 | address u64 | mapsize u64 | two 48-byte database records
//                       (FREE_DBI, MAIN_DBI: pad u32 | flags u16 | depth u16 | branch, leaf, overflow pages u64 | entries u64 | root u64;
//                       FREE_DBI's pad is the page size, its flags MDB_INTEGERKEY 8) | last page u64 | txnid u64.  A commit writes the
//                       meta page txnid & 1; the valid one with the larger txnid is current.
// PARITY UNPINNED against liblmdb (absent): tests/test_mdb_store.py checks the files with an independent pure-Python reader of the
// same format and the invariants mdb.c asserts (sorted keys, uniform depth, > 1 key per branch page, page / entry counts).
//
// Writer model: one writer, every put / delete is its own committed transaction (as the reference's), copy-on-write along the
// root-to-leaf path, overflow pages appended.  Pages a commit supersedes are NOT recorded in the free database (it is carried over
// unchanged): they are leaked -- 12 KB per commit beside records of ~1.6 MB -- and a real LMDB that later writes the file is
// unaffected.  The tree's branch / leaf pages are mirrored in memory (31 B per record); record payloads stay on disk.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "nbp_hip.h"

namespace {

constexpr uint32_t MDB_MAGIC = 0xBEEFC0DEu, MDB_VERSION = 1;
constexpr size_t PSIZE = 4096, PHDR = 16, NODEHDR = 8;
constexpr uint16_t P_BRANCH = 1, P_LEAF = 2, P_OVERFLOW = 4, P_META = 8;
constexpr uint16_t F_BIGDATA = 1, F_SUBDATA = 2, F_DUPDATA = 4;
constexpr uint64_t P_INVALID = ~0ull;
constexpr size_t NODEMAX = (((PSIZE - PHDR) / 2) & ~(size_t)1) - 2;      // 2038: larger records go to overflow pages
constexpr size_t MAXKEY = 511;

struct DbRec { uint32_t pad; uint16_t flags, depth; uint64_t branch_pages, leaf_pages, overflow_pages, entries, root; };
static_assert(sizeof(DbRec) == 48, "MDB_db");

inline size_t even(size_t v) { return (v + 1) & ~(size_t)1; }

struct Val { bool big = false; uint64_t ovpg = 0; uint32_t size = 0; std::string inl; };
struct Node {
    bool leaf = true, dirty = true;
    uint64_t pgno = 0;
    std::vector<std::string> keys;                 // branch: keys[i] = separator of kids[i] (keys[0] is not written)
    std::vector<Val> vals;                         // leaf
    std::vector<std::unique_ptr<Node>> kids;       // branch
    size_t bytes() const {
        size_t b = PHDR;
        for (size_t i = 0; i < keys.size(); ++i) {
            if (leaf) b += 2 + even(NODEHDR + keys[i].size() + (vals[i].big ? 8 : vals[i].size));
            else b += 2 + even(NODEHDR + (i ? keys[i].size() : 0));
        }
        return b;
    }
};

struct Env {
    int fd = -1;
    std::string file;
    uint64_t mapsize = 0, last_pg = 1, txnid = 0;
    DbRec free_db{}, main_db{};
    std::unique_ptr<Node> root;
    bool sync_each = false;
    int err = 0;
};

bool pread_all(int fd, void* buf, size_t n, uint64_t off) {
    char* p = (char*)buf;
    while (n) {
        const ssize_t r = pread(fd, p, n, (off_t)off);
        if (r <= 0) { if (r < 0 && errno == EINTR) continue; return false; }
        p += r; n -= (size_t)r; off += (uint64_t)r;
    }
    return true;
}
bool pwrite_all(int fd, const void* buf, size_t n, uint64_t off) {
    const char* p = (const char*)buf;
    while (n) {
        const ssize_t r = pwrite(fd, p, n, (off_t)off);
        if (r <= 0) { if (r < 0 && errno == EINTR) continue; return false; }
        p += r; n -= (size_t)r; off += (uint64_t)r;
    }
    return true;
}

template <class T> T rd(const unsigned char* p) { T v; memcpy(&v, p, sizeof v); return v; }
template <class T> void wr(unsigned char* p, T v) { memcpy(p, &v, sizeof v); }

const std::string& min_key(const Node* n) { while (!n->leaf) n = n->kids[0].get(); return n->keys[0]; }

// ---- loading the tree of an existing file
std::unique_ptr<Node> load_page(Env& e, uint64_t pgno, int depth_left, bool& ok) {
    std::unique_ptr<Node> n(new Node);
    unsigned char pg[PSIZE];
    if (depth_left <= 0 || !pread_all(e.fd, pg, PSIZE, pgno * PSIZE) || rd<uint64_t>(pg) != pgno) { ok = false; return n; }
    const uint16_t flags = rd<uint16_t>(pg + 10), lower = rd<uint16_t>(pg + 12), upper = rd<uint16_t>(pg + 14);
    if (!(flags & (P_BRANCH | P_LEAF)) || lower < PHDR || lower > upper || upper > PSIZE) { ok = false; return n; }
    n->leaf = (flags & P_LEAF) != 0;
    n->dirty = false;
    n->pgno = pgno;
    const int nk = (lower - (int)PHDR) / 2;
    for (int i = 0; i < nk && ok; ++i) {
        const uint16_t off = rd<uint16_t>(pg + PHDR + 2 * i);
        if (off < lower || off + NODEHDR > PSIZE) { ok = false; break; }
        const unsigned char* nd = pg + off;
        const uint16_t lo = rd<uint16_t>(nd), hi = rd<uint16_t>(nd + 2), nf = rd<uint16_t>(nd + 4), ks = rd<uint16_t>(nd + 6);
        if (off + NODEHDR + ks > PSIZE) { ok = false; break; }
        n->keys.emplace_back((const char*)nd + NODEHDR, ks);
        if (n->leaf) {
            if (nf & (F_SUBDATA | F_DUPDATA)) { ok = false; e.err = NBP_E_SHAPE; break; }     // named / dupsort databases: not the reference's
            Val v;
            v.size = (uint32_t)lo | ((uint32_t)hi << 16);
            if (nf & F_BIGDATA) {
                if (off + NODEHDR + ks + 8 > PSIZE) { ok = false; break; }
                v.big = true; v.ovpg = rd<uint64_t>(nd + NODEHDR + ks);
            } else {
                if (off + NODEHDR + ks + v.size > PSIZE) { ok = false; break; }
                v.inl.assign((const char*)nd + NODEHDR + ks, v.size);
            }
            n->vals.push_back(std::move(v));
        } else {
            const uint64_t child = (uint64_t)lo | ((uint64_t)hi << 16) | ((uint64_t)nf << 32);
            n->kids.push_back(load_page(e, child, depth_left - 1, ok));
        }
    }
    if (ok && !n->leaf && n->kids.size() < 2) ok = false;
    if (ok && n->leaf && n->keys.empty()) ok = false;
    // (node 0 of a branch page carries no key on disk: in memory keys[0] is kept as a lower bound of the first child's keys)
    if (ok && !n->leaf) n->keys[0] = min_key(n->kids[0].get());
    return n;
}

bool read_meta(Env& e) {
    unsigned char pg[2][PSIZE];
    int best = -1;
    uint64_t best_txn = 0;
    for (int m = 0; m < 2; ++m) {
        if (!pread_all(e.fd, pg[m], PSIZE, m * PSIZE)) continue;
        const unsigned char* mt = pg[m] + PHDR;
        if (!(rd<uint16_t>(pg[m] + 10) & P_META) || rd<uint32_t>(mt) != MDB_MAGIC || rd<uint32_t>(mt + 4) != MDB_VERSION) continue;
        if (rd<uint32_t>(mt + 24) != PSIZE) continue;                       // FREE_DBI's pad = the page size: only 4096 here
        const uint64_t txn = rd<uint64_t>(mt + 24 + 96 + 8);
        if (best < 0 || txn > best_txn) { best = m; best_txn = txn; }
    }
    if (best < 0) return false;
    const unsigned char* mt = pg[best] + PHDR;
    e.mapsize = std::max(e.mapsize, rd<uint64_t>(mt + 16));
    memcpy(&e.free_db, mt + 24, 48);
    memcpy(&e.main_db, mt + 72, 48);
    e.last_pg = rd<uint64_t>(mt + 120);
    e.txnid = best_txn;
    return true;
}

bool write_meta(Env& e, uint64_t which) {
    unsigned char pg[PSIZE];
    memset(pg, 0, sizeof pg);
    wr<uint64_t>(pg, which);
    wr<uint16_t>(pg + 10, P_META);
    unsigned char* mt = pg + PHDR;
    wr<uint32_t>(mt, MDB_MAGIC); wr<uint32_t>(mt + 4, MDB_VERSION);
    wr<uint64_t>(mt + 8, 0); wr<uint64_t>(mt + 16, e.mapsize);
    memcpy(mt + 24, &e.free_db, 48);
    memcpy(mt + 72, &e.main_db, 48);
    wr<uint64_t>(mt + 120, e.last_pg); wr<uint64_t>(mt + 128, e.txnid);
    return pwrite_all(e.fd, pg, PSIZE, which * PSIZE);
}

// ---- serialising dirty pages (post-order: children get their page numbers first)
bool flush(Env& e, Node* n) {
    if (!n->dirty) return true;
    if (!n->leaf)
        for (auto& k : n->kids) if (!flush(e, k.get())) return false;
    unsigned char pg[PSIZE];
    memset(pg, 0, sizeof pg);
    n->pgno = ++e.last_pg;
    wr<uint64_t>(pg, n->pgno);
    wr<uint16_t>(pg + 10, n->leaf ? P_LEAF : P_BRANCH);
    size_t upper = PSIZE;
    const size_t nk = n->keys.size();
    for (size_t i = 0; i < nk; ++i) {
        const std::string& key = n->keys[i];
        const size_t ks = (n->leaf || i) ? key.size() : 0;
        size_t dsz = 0;
        if (n->leaf) dsz = n->vals[i].big ? 8 : n->vals[i].size;
        const size_t nsz = even(NODEHDR + ks + dsz);
        if (upper < PHDR + 2 * nk + nsz) return false;                     // (cannot happen: splits keep bytes() <= PSIZE)
        upper -= nsz;
        unsigned char* nd = pg + upper;
        if (n->leaf) {
            const Val& v = n->vals[i];
            wr<uint16_t>(nd, (uint16_t)(v.size & 0xffff)); wr<uint16_t>(nd + 2, (uint16_t)(v.size >> 16));
            wr<uint16_t>(nd + 4, v.big ? F_BIGDATA : 0);
            wr<uint16_t>(nd + 6, (uint16_t)ks);
            memcpy(nd + NODEHDR, key.data(), ks);
            if (v.big) wr<uint64_t>(nd + NODEHDR + ks, v.ovpg);
            else memcpy(nd + NODEHDR + ks, v.inl.data(), v.size);
        } else {
            const uint64_t child = n->kids[i]->pgno;
            wr<uint16_t>(nd, (uint16_t)(child & 0xffff)); wr<uint16_t>(nd + 2, (uint16_t)((child >> 16) & 0xffff));
            wr<uint16_t>(nd + 4, (uint16_t)((child >> 32) & 0xffff));
            wr<uint16_t>(nd + 6, (uint16_t)ks);
            memcpy(nd + NODEHDR, key.data(), ks);
        }
        wr<uint16_t>(pg + PHDR + 2 * i, (uint16_t)upper);
    }
    wr<uint16_t>(pg + 12, (uint16_t)(PHDR + 2 * nk));
    wr<uint16_t>(pg + 14, (uint16_t)upper);
    n->dirty = false;
    return pwrite_all(e.fd, pg, PSIZE, n->pgno * PSIZE);
}

void count_pages(const Node* n, uint64_t& br, uint64_t& lf) {
    if (n->leaf) { ++lf; return; }
    ++br;
    for (auto& k : n->kids) count_pages(k.get(), br, lf);
}

bool commit(Env& e) {
    if (e.root) {
        if (!flush(e, e.root.get())) return false;
        e.main_db.root = e.root->pgno;
        uint64_t br = 0, lf = 0;
        count_pages(e.root.get(), br, lf);
        e.main_db.branch_pages = br; e.main_db.leaf_pages = lf;
    } else {
        e.main_db.root = P_INVALID; e.main_db.depth = 0; e.main_db.branch_pages = e.main_db.leaf_pages = 0;
    }
    ++e.txnid;
    e.mapsize = std::max(e.mapsize, (e.last_pg + 1) * PSIZE);
    // data before the meta page that makes it current (the order LMDB keeps; fdatasync between the two only when asked to)
    if (e.sync_each && fdatasync(e.fd) != 0) return false;
    if (!write_meta(e, e.txnid & 1)) return false;
    if (e.sync_each && fdatasync(e.fd) != 0) return false;
    return true;
}

// index of the child of branch n that covers key: the last i with keys[i] <= key (keys[0] = -inf)
size_t child_of(const Node* n, const std::string& key) {
    size_t lo = 1, hi = n->keys.size();          // first i in [1, nk) with keys[i] > key
    while (lo < hi) { const size_t m = (lo + hi) / 2; if (n->keys[m] <= key) lo = m + 1; else hi = m; }
    return lo - 1;
}

// splits an over-full node; returns the new right sibling.  The cut is the most balanced one for which BOTH halves fit a page (one
// exists: a node is at most 2040 bytes with its pointer, so the longest prefix that fits leaves less than a page behind it)
std::unique_ptr<Node> split(Node* n) {
    std::unique_ptr<Node> r(new Node);
    r->leaf = n->leaf;
    const size_t nk = n->keys.size();
    std::vector<size_t> sz(nk);
    size_t total = 0;
    for (size_t i = 0; i < nk; ++i) {
        sz[i] = 2 + (n->leaf ? even(NODEHDR + n->keys[i].size() + (n->vals[i].big ? 8 : n->vals[i].size)) : even(NODEHDR + n->keys[i].size()));
        total += sz[i];
    }
    const size_t min_side = n->leaf ? 1 : 2, cap = PSIZE - PHDR;
    size_t cut = min_side, best = ~(size_t)0, left = 0;
    for (size_t i = 0; i < nk; ++i) {
        left += sz[i];
        const size_t c = i + 1;
        if (c < min_side || c + min_side > nk) continue;
        // (a branch's right half drops the key of its node 0: it can only get smaller than counted here)
        if (left > cap || total - left > cap) continue;
        const size_t d = left > total - left ? left - (total - left) : (total - left) - left;
        if (d < best) { best = d; cut = c; }
    }
    r->keys.assign(std::make_move_iterator(n->keys.begin() + cut), std::make_move_iterator(n->keys.end()));
    n->keys.resize(cut);
    if (n->leaf) {
        r->vals.assign(std::make_move_iterator(n->vals.begin() + cut), std::make_move_iterator(n->vals.end()));
        n->vals.resize(cut);
    } else {
        for (size_t i = cut; i < nk; ++i) r->kids.push_back(std::move(n->kids[i]));
        n->kids.resize(cut);
    }
    n->dirty = r->dirty = true;
    return r;
}

// inserts into the subtree; returns a new right sibling of n when n had to split
std::unique_ptr<Node> insert(Env& e, Node* n, const std::string& key, Val&& v, bool& replaced) {
    n->dirty = true;
    if (n->leaf) {
        const auto it = std::lower_bound(n->keys.begin(), n->keys.end(), key);
        const size_t i = (size_t)(it - n->keys.begin());
        if (it != n->keys.end() && *it == key) { n->vals[i] = std::move(v); replaced = true; }
        else { n->keys.insert(it, key); n->vals.insert(n->vals.begin() + i, std::move(v)); }
    } else {
        const size_t c = child_of(n, key);
        std::unique_ptr<Node> r = insert(e, n->kids[c].get(), key, std::move(v), replaced);
        if (c == 0 && key < n->keys[0]) n->keys[0] = key;       // (kept as the subtree's lower bound; not written for node 0)
        if (r) {
            n->keys.insert(n->keys.begin() + c + 1, min_key(r.get()));
            n->kids.insert(n->kids.begin() + c + 1, std::move(r));
        }
    }
    return n->bytes() > PSIZE ? split(n) : nullptr;
}

// removes key from the subtree; true when found.  Afterwards n may be under-full (leaf: no key; branch: one child): the
// caller repairs it (merge into / borrow from a sibling), so that every branch page keeps > 1 key and the depth stays uniform.
void repair(Node* parent, size_t c) {
    Node* ch = parent->kids[c].get();
    const bool under = ch->leaf ? ch->keys.empty() : ch->kids.size() < 2;
    if (!under) return;
    if (ch->leaf) {                                 // an empty leaf simply leaves its parent
        parent->kids.erase(parent->kids.begin() + c);
        parent->keys.erase(parent->keys.begin() + c);
        return;
    }
    // a branch with one child: hand that child to a sibling branch (or take one of the sibling's when the sibling is full)
    const size_t s = c ? c - 1 : c + 1;             // (the parent has >= 2 children before this repair)
    Node* sib = parent->kids[s].get();
    sib->dirty = true;
    if (s < c) {                                    // sibling on the left: append
        sib->keys.push_back(ch->keys[0]);
        sib->kids.push_back(std::move(ch->kids[0]));
    } else {                                        // sibling on the right: prepend
        sib->keys.insert(sib->keys.begin(), ch->keys[0]);
        sib->kids.insert(sib->kids.begin(), std::move(ch->kids[0]));
        parent->keys[s] = sib->keys[0];
    }
    parent->kids.erase(parent->kids.begin() + c);
    parent->keys.erase(parent->keys.begin() + c);
    const size_t si = s < c ? s : s - 1;
    Node* merged = parent->kids[si].get();
    if (merged->bytes() > PSIZE) {                  // the sibling was full: split it again (both halves have >= 2 children)
        std::unique_ptr<Node> r = split(merged);
        parent->keys.insert(parent->keys.begin() + si + 1, r->keys[0]);
        parent->kids.insert(parent->kids.begin() + si + 1, std::move(r));
    }
}
// (with long keys of unequal length a repair can leave n itself over-full -- a separator replaced by a longer one: n then splits, and
// the new right sibling goes up like insert()'s)
std::unique_ptr<Node> erase(Node* n, const std::string& key, bool& found) {
    if (n->leaf) {
        const auto it = std::lower_bound(n->keys.begin(), n->keys.end(), key);
        if (it == n->keys.end() || *it != key) return nullptr;
        const size_t i = (size_t)(it - n->keys.begin());
        n->keys.erase(it);
        n->vals.erase(n->vals.begin() + i);
        n->dirty = true;
        found = true;
        return nullptr;
    }
    const size_t c = child_of(n, key);
    std::unique_ptr<Node> r = erase(n->kids[c].get(), key, found);
    if (!found) return nullptr;
    n->dirty = true;
    if (r) {
        n->keys.insert(n->keys.begin() + c + 1, r->keys[0]);
        n->kids.insert(n->kids.begin() + c + 1, std::move(r));
    }
    repair(n, c);
    return n->bytes() > PSIZE ? split(n) : nullptr;
}

const Val* find(const Node* n, const std::string& key) {
    while (n && !n->leaf) n = n->kids[child_of(n, key)].get();
    if (!n) return nullptr;
    const auto it = std::lower_bound(n->keys.begin(), n->keys.end(), key);
    if (it == n->keys.end() || *it != key) return nullptr;
    return &n->vals[(size_t)(it - n->keys.begin())];
}

void collect(const Node* n, std::string& out) {
    if (!n) return;
    if (n->leaf) {
        for (const auto& k : n->keys) { const uint16_t l = (uint16_t)k.size(); out.append((const char*)&l, 2); out.append(k); }
        return;
    }
    for (const auto& k : n->kids) collect(k.get(), out);
}

}  // namespace

extern "C" int nbp_mdb_open(const char* dir_path, unsigned long long map_size, int sync_each_commit, void** env_out) {
    if (!dir_path || !env_out) return NBP_E_ARG;
    *env_out = nullptr;
    if (mkdir(dir_path, 0775) != 0 && errno != EEXIST) return NBP_E_ARG;
    std::unique_ptr<Env> e(new Env);
    e->file = std::string(dir_path) + "/data.mdb";
    e->mapsize = map_size;
    e->sync_each = sync_each_commit != 0;
    e->fd = open(e->file.c_str(), O_RDWR | O_CREAT, 0664);
    if (e->fd < 0) return NBP_E_ARG;
    struct stat st;
    if (fstat(e->fd, &st) != 0) { close(e->fd); return NBP_E_ARG; }
    if (st.st_size == 0) {
        // mdb_env_init_meta: both meta pages (txnid 0 and 0), empty databases
        memset(&e->free_db, 0, sizeof(DbRec)); memset(&e->main_db, 0, sizeof(DbRec));
        e->free_db.pad = (uint32_t)PSIZE; e->free_db.flags = 8 /* MDB_INTEGERKEY */; e->free_db.root = P_INVALID;
        e->main_db.root = P_INVALID;
        e->last_pg = 1; e->txnid = 0;
        e->mapsize = std::max<uint64_t>(e->mapsize, 2 * PSIZE);
        if (!(write_meta(*e, 0) && write_meta(*e, 1))) { close(e->fd); return NBP_E_ARG; }      // both with txnid 0; the first commit (1) takes page 1
    } else {
        if (!read_meta(*e)) { close(e->fd); return NBP_E_SHAPE; }
        if (e->main_db.root != P_INVALID) {
            bool ok = true;
            e->root = load_page(*e, e->main_db.root, 32, ok);
            if (!ok) { const int rc = e->err ? e->err : NBP_E_SHAPE; close(e->fd); return rc; }
        }
    }
    *env_out = e.release();
    return 0;
}

extern "C" int nbp_mdb_close(void* env) {
    if (!env) return NBP_E_ARG;
    Env* e = (Env*)env;
    fdatasync(e->fd);
    close(e->fd);
    delete e;
    return 0;
}

extern "C" long long nbp_mdb_entries(void* env) { return env ? (long long)((Env*)env)->main_db.entries : -1; }

extern "C" int nbp_mdb_put(void* env, const void* key, size_t klen, const void* val, size_t vlen) {
    if (!env || !key || klen < 1 || klen > MAXKEY || (!val && vlen) || vlen >= 0xffffffffull) return NBP_E_ARG;
    Env& e = *(Env*)env;
    const std::string k((const char*)key, klen);
    Val v;
    v.size = (uint32_t)vlen;
    if (NODEHDR + klen + vlen > NODEMAX) {
        // overflow pages: header (pgno, flags, page count) then the bytes, contiguous
        const uint64_t np = (PHDR + vlen + PSIZE - 1) / PSIZE;
        const uint64_t pg0 = e.last_pg + 1;
        std::vector<unsigned char> buf(np * PSIZE, 0);
        wr<uint64_t>(buf.data(), pg0);
        wr<uint16_t>(buf.data() + 10, P_OVERFLOW);
        wr<uint32_t>(buf.data() + 12, (uint32_t)np);
        memcpy(buf.data() + PHDR, val, vlen);
        if (!pwrite_all(e.fd, buf.data(), buf.size(), pg0 * PSIZE)) return NBP_E_ARG;
        e.last_pg += np;
        e.main_db.overflow_pages += np;
        v.big = true; v.ovpg = pg0;
    } else {
        v.inl.assign((const char*)val, vlen);
    }
    bool replaced = false;
    if (!e.root) { e.root.reset(new Node); e.main_db.depth = 1; }
    const Val* old = find(e.root.get(), k);
    if (old && old->big) e.main_db.overflow_pages -= (PHDR + old->size + PSIZE - 1) / PSIZE;      // (its pages are leaked, not reused)
    std::unique_ptr<Node> r = insert(e, e.root.get(), k, std::move(v), replaced);
    if (r) {                                        // the root split: one level more
        std::unique_ptr<Node> nr(new Node);
        nr->leaf = false;
        nr->keys.push_back(min_key(e.root.get()));
        nr->keys.push_back(min_key(r.get()));
        nr->kids.push_back(std::move(e.root));
        nr->kids.push_back(std::move(r));
        e.root = std::move(nr);
        ++e.main_db.depth;
    }
    if (!replaced) ++e.main_db.entries;
    return commit(e) ? 0 : NBP_E_ARG;
}

extern "C" int nbp_mdb_del(void* env, const void* key, size_t klen) {
    if (!env || !key || klen < 1 || klen > MAXKEY) return NBP_E_ARG;
    Env& e = *(Env*)env;
    if (!e.root) return 1;
    const std::string k((const char*)key, klen);
    const Val* old = find(e.root.get(), k);
    if (!old) return 1;                              // MDB_NOTFOUND: nothing written
    if (old->big) e.main_db.overflow_pages -= (PHDR + old->size + PSIZE - 1) / PSIZE;
    bool found = false;
    std::unique_ptr<Node> r = erase(e.root.get(), k, found);
    if (r) {
        std::unique_ptr<Node> nr(new Node);
        nr->leaf = false;
        nr->keys.push_back(min_key(e.root.get()));
        nr->keys.push_back(r->keys[0]);
        nr->kids.push_back(std::move(e.root));
        nr->kids.push_back(std::move(r));
        e.root = std::move(nr);
        ++e.main_db.depth;
    }
    --e.main_db.entries;
    // the root: an empty leaf -> an empty database; a branch with one child -> that child (one level less)
    while (e.root && !e.root->leaf && e.root->kids.size() == 1) {
        std::unique_ptr<Node> ch = std::move(e.root->kids[0]);
        e.root = std::move(ch);
        e.root->dirty = true;
        --e.main_db.depth;
    }
    if (e.root && e.root->leaf && e.root->keys.empty()) e.root.reset();
    return commit(e) ? 0 : NBP_E_ARG;
}

// value of `key` into buf (cap bytes); *vlen_out = its size whatever cap is.  1 = not found.
extern "C" int nbp_mdb_get(void* env, const void* key, size_t klen, void* buf, size_t cap, size_t* vlen_out) {
    if (!env || !key || !vlen_out) return NBP_E_ARG;
    Env& e = *(Env*)env;
    const Val* v = e.root ? find(e.root.get(), std::string((const char*)key, klen)) : nullptr;
    if (!v) return 1;
    *vlen_out = v->size;
    if (!buf || cap < v->size) return 0;
    if (!v->big) { memcpy(buf, v->inl.data(), v->size); return 0; }
    return pread_all(e.fd, buf, v->size, v->ovpg * PSIZE + PHDR) ? 0 : NBP_E_SHAPE;
}

// every key in order, packed as [u16 length][bytes]...; *needed_out = the packed size (call with cap = 0 to size the buffer)
extern "C" int nbp_mdb_keys(void* env, void* buf, size_t cap, size_t* needed_out) {
    if (!env || !needed_out) return NBP_E_ARG;
    Env& e = *(Env*)env;
    std::string out;
    collect(e.root.get(), out);
    *needed_out = out.size();
    if (buf && cap >= out.size()) memcpy(buf, out.data(), out.size());
    return 0;
}

// {depth, branch pages, leaf pages, overflow pages, entries, last page, txnid, page size}
extern "C" int nbp_mdb_stat(void* env, unsigned long long* out8) {
    if (!env || !out8) return NBP_E_ARG;
    Env& e = *(Env*)env;
    out8[0] = e.main_db.depth; out8[1] = e.main_db.branch_pages; out8[2] = e.main_db.leaf_pages; out8[3] = e.main_db.overflow_pages;
    out8[4] = e.main_db.entries; out8[5] = e.last_pg; out8[6] = e.txnid; out8[7] = PSIZE;
    return 0;
}
