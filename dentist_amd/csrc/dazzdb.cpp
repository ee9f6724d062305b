// dazzdb.cpp -- DAZZ_DB on-disk databases (.db / .dam stub + hidden .idx / .bps / .hdr), host only.
//
// DENTIST never touches these files itself: it calls fasta2DB / fasta2DAM / DBsplit / DBdump
// (source/dentist/dazzler.d:6233-6330, 6445-6517) and the aligners open them.  A drop-in for the
// aligners therefore has to read (and, for the tests, write) the same files.  The format is
// DAZZ_DB's DB.h (pinned version d22ae58, conda/recipes/dazz_db/meta.yaml:10-14), restated here and
// PINNED by the reference's own checksums: tests/test-commands.sh:54-61 holds the md5 of the
// .idx, .bps and .hdr that `fasta2DAM -i` produces for the embedded 4 097 bp assembly -- this
// writer reproduces all three byte for byte (tests/test_dazzdb.py).  The text stub's md5 could not
// be reproduced (it embeds a name that is not recoverable from the script) and is unpinned; its
// grammar follows DB.h (`files =`, file lines, `blocks =`, `size = .. cutoff = .. all = ..`,
// block table) as parsed by source/dentist/dazzler.d:4383-4479.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <vector>

#include "../../include/dentist_hip.h"

int dh_fail(int code, const std::string &msg);

namespace {

#define DB_QV 0x03ff
#define DB_CCS 0x0400
#define DB_BEST 0x0800
#define DB_ALL 0x1

#pragma pack(push, 1)
struct DazzHeader {  // image of struct DAZZ_DB as fwrite()n by fasta2DB/fasta2DAM, 112 bytes
    int32_t ureads, treads, cutoff, allarr;
    float freq[4];
    int32_t maxlen, pad0;
    int64_t totlen;
    int32_t nreads, trimmed, part, ufirst, tfirst, pad1;
    int64_t path;
    int32_t loaded, pad2;
    int64_t bases, reads, tracks;
};
struct DazzRead {  // struct DAZZ_READ, 40 bytes
    int32_t origin, rlen, fpulse, pad0;
    int64_t boff, coff;
    int32_t flags, pad1;
};
#pragma pack(pop)
static_assert(sizeof(DazzHeader) == 112, "DAZZ_DB image");
static_assert(sizeof(DazzRead) == 40, "DAZZ_READ image");

struct Paths {
    std::string stub, dir, root, ext;
    int block = 0;
    std::string hidden(const char *suffix) const { return dir + "." + root + "." + suffix; }
};

// "dir/root.dam", "dir/root", "dir/root.3", "dir/root.3.dam" ...
bool split_path(const std::string &in, Paths &p, bool must_exist)
{
    std::string s = in;
    std::string ext;
    if (s.size() > 4 && s.compare(s.size() - 4, 4, ".dam") == 0) {
        ext = ".dam";
        s.resize(s.size() - 4);
    } else if (s.size() > 3 && s.compare(s.size() - 3, 3, ".db") == 0) {
        ext = ".db";
        s.resize(s.size() - 3);
    }
    // trailing .<digits> = block
    size_t dot = s.find_last_of('.');
    size_t slash = s.find_last_of('/');
    if (dot != std::string::npos && (slash == std::string::npos || dot > slash) && dot + 1 < s.size()) {
        bool digits = true;
        for (size_t i = dot + 1; i < s.size(); i++) digits = digits && isdigit((unsigned char)s[i]);
        if (digits && s[dot + 1] != '0') {
            p.block = atoi(s.c_str() + dot + 1);
            s.resize(dot);
        }
    }
    slash = s.find_last_of('/');
    p.dir = slash == std::string::npos ? "" : s.substr(0, slash + 1);
    p.root = slash == std::string::npos ? s : s.substr(slash + 1);
    if (ext.empty() && must_exist) {
        for (const char *e : {".db", ".dam"}) {
            FILE *f = fopen((s + e).c_str(), "r");
            if (f) {
                fclose(f);
                ext = e;
                break;
            }
        }
        if (ext.empty()) return false;
    }
    p.ext = ext;
    p.stub = s + ext;
    return true;
}

inline int code_of(char c)
{
    switch (c) {
    case 'a': case 'A': return 0;
    case 'c': case 'C': return 1;
    case 'g': case 'G': return 2;
    case 't': case 'T': return 3;
    default: return -1;
    }
}

struct Rec {
    std::vector<uint8_t> codes;
    int32_t origin = 0, fpulse = 0, flags = DB_BEST;
    int64_t coff = 0;
};

int write_db(const Paths &p, const std::vector<Rec> &recs, const std::string &hdr_bytes, bool is_dam,
             const std::string &srcname)
{
    FILE *idx = fopen(p.hidden("idx").c_str(), "wb");
    FILE *bps = fopen(p.hidden("bps").c_str(), "wb");
    FILE *stub = fopen(p.stub.c_str(), "w");
    FILE *hdr = is_dam ? fopen(p.hidden("hdr").c_str(), "wb") : nullptr;
    if (!idx || !bps || !stub || (is_dam && !hdr)) {
        for (FILE *f : {idx, bps, stub, hdr})
            if (f) fclose(f);
        return dh_fail(DH_EIO, "cannot create DAZZ_DB files for " + p.stub);
    }
    DazzHeader h;
    memset(&h, 0, sizeof(h));
    int64_t count[4] = {0, 0, 0, 0}, totlen = 0;
    int32_t maxlen = 0;
    for (const Rec &r : recs) {
        for (uint8_t c : r.codes) count[c]++;
        totlen += (int64_t)r.codes.size();
        maxlen = std::max<int32_t>(maxlen, (int32_t)r.codes.size());
    }
    h.ureads = h.treads = (int32_t)recs.size();
    h.cutoff = -1;
    h.allarr = 0;
    for (int c = 0; c < 4; c++) h.freq[c] = totlen ? (float)((1. * count[c]) / totlen) : 0.f;
    h.maxlen = maxlen;
    h.totlen = totlen;
    fwrite(&h, sizeof(h), 1, idx);
    int64_t boff = 0;
    std::vector<uint8_t> packed;
    for (const Rec &r : recs) {
        DazzRead d;
        memset(&d, 0, sizeof(d));
        d.origin = r.origin;
        d.rlen = (int32_t)r.codes.size();
        d.fpulse = r.fpulse;
        d.boff = boff;
        d.coff = r.coff;
        d.flags = r.flags;
        fwrite(&d, sizeof(d), 1, idx);
        packed.assign((r.codes.size() + 3) / 4, 0);
        for (size_t i = 0; i < r.codes.size(); i++) packed[i >> 2] |= (uint8_t)(r.codes[i] << (6 - 2 * (i & 3)));
        if (!packed.empty()) fwrite(packed.data(), 1, packed.size(), bps);
        boff += (int64_t)packed.size();
    }
    if (hdr && !hdr_bytes.empty()) fwrite(hdr_bytes.data(), 1, hdr_bytes.size(), hdr);
    fprintf(stub, "files = %9d\n", 1);
    fprintf(stub, "  %9d %s %s\n", (int)recs.size(), srcname.c_str(), srcname.c_str());
    fprintf(stub, "blocks = %9d\n", 0);
    bool ok = true;
    for (FILE *f : {idx, bps, stub, hdr})
        if (f && fclose(f) != 0) ok = false;
    return ok ? DH_OK : dh_fail(DH_EIO, "short write for " + p.stub);
}

}  // namespace

// fasta2DAM -i: scaffolds are cut into contigs at runs of non-ACGT characters; origin = contig
// number inside its scaffold, fpulse = its start in the scaffold, coff = offset of the scaffold's
// header line in .hdr (consumer dazzler.d:4689-4761).
extern "C" int dh_dazz_create_dam(const char *path, const char *fasta, int64_t n)
{
    Paths p;
    if (!path || !fasta || !split_path(std::string(path), p, false)) return dh_fail(DH_EINVAL, "bad DAM path");
    if (p.ext.empty()) {
        p.ext = ".dam";
        p.stub += ".dam";
    }
    std::vector<Rec> recs;
    std::string hdrs;
    int64_t i = 0;
    while (i < n) {
        if (fasta[i] != '>') {
            i++;
            continue;
        }
        int64_t e = i;
        while (e < n && fasta[e] != '\n') e++;
        const int64_t coff = (int64_t)hdrs.size();
        hdrs.append(fasta + i, (size_t)(e - i));
        hdrs.push_back('\n');
        i = e + 1;
        int32_t pos = 0, contig = 0;
        Rec cur;
        bool open = false;
        auto flush = [&]() {
            if (open && !cur.codes.empty()) {
                cur.origin = contig++;
                cur.coff = coff;
                recs.push_back(cur);
            }
            cur = Rec();
            open = false;
        };
        while (i < n && fasta[i] != '>') {
            const char c = fasta[i++];
            if (c == '\n' || c == '\r' || c == ' ') continue;
            const int code = code_of(c);
            if (code < 0)
                flush();
            else {
                if (!open) {
                    open = true;
                    cur.fpulse = pos;
                }
                cur.codes.push_back((uint8_t)code);
            }
            pos++;
        }
        flush();
    }
    return write_db(p, recs, hdrs, true, p.root);
}

// fasta2DB -i: PacBio headers `>name/well/beg_end [RQ=0.xxx]` (dazzler.d:1389-1393); origin = well,
// fpulse = beg; every read is the best of its well.
extern "C" int dh_dazz_create_db(const char *path, const char *fasta, int64_t n)
{
    Paths p;
    if (!path || !fasta || !split_path(std::string(path), p, false)) return dh_fail(DH_EINVAL, "bad DB path");
    if (p.ext.empty()) {
        p.ext = ".db";
        p.stub += ".db";
    }
    std::vector<Rec> recs;
    std::string prolog;
    int64_t i = 0;
    while (i < n) {
        if (fasta[i] != '>') {
            i++;
            continue;
        }
        int64_t e = i;
        while (e < n && fasta[e] != '\n') e++;
        std::string h(fasta + i + 1, (size_t)(e - i - 1));
        Rec r;
        int well = (int)recs.size(), beg = 0, end = 0;
        const size_t s1 = h.find('/');
        if (recs.empty()) prolog = h.substr(0, std::min(s1, h.find(' ')));  // the movie name: DBdump's H line
        if (s1 != std::string::npos) sscanf(h.c_str() + s1 + 1, "%d/%d_%d", &well, &beg, &end);
        float rq = 0;
        const size_t rqp = h.find("RQ=0.");
        if (rqp != std::string::npos) sscanf(h.c_str() + rqp + 3, "%f", &rq);
        r.origin = well;
        r.fpulse = beg;
        r.coff = -1;
        r.flags = DB_BEST | ((int)(rq * 1000.f) & DB_QV);
        i = e + 1;
        while (i < n && fasta[i] != '>') {
            const int code = code_of(fasta[i++]);
            if (code >= 0) r.codes.push_back((uint8_t)code);
        }
        recs.push_back(r);
    }
    return write_db(p, recs, "", false, prolog.empty() ? p.root : prolog);
}

// DBsplit -x<cutoff> [-a] -s<mb>: rewrites the block table of the stub and cutoff/all/treads of
// the .idx header (tests/test-commands.sh:188-189 uses -x20).
extern "C" int dh_dazz_split(const char *path, int32_t cutoff, int32_t all, int64_t size_mb)
{
    Paths p;
    if (!path || !split_path(std::string(path), p, true)) return dh_fail(DH_EIO, "DAZZ_DB not found");
    FILE *idx = fopen(p.hidden("idx").c_str(), "r+b");
    if (!idx) return dh_fail(DH_EIO, "cannot open " + p.hidden("idx"));
    DazzHeader h;
    if (fread(&h, sizeof(h), 1, idx) != 1) {
        fclose(idx);
        return dh_fail(DH_EIO, "short .idx");
    }
    std::vector<DazzRead> reads((size_t)h.ureads);
    if (h.ureads && fread(reads.data(), sizeof(DazzRead), reads.size(), idx) != reads.size()) {
        fclose(idx);
        return dh_fail(DH_EIO, "short .idx");
    }
    const bool is_dam = p.ext == ".dam";
    const int64_t size = size_mb * 1000000ll;
    std::vector<std::pair<int, int>> blocks{{0, 0}};
    int treads = 0;
    int64_t acc = 0;
    for (int i = 0; i < h.ureads; i++) {
        const bool keep = reads[(size_t)i].rlen >= cutoff && (is_dam || all || (reads[(size_t)i].flags & DB_BEST));
        if (!keep) continue;
        if (acc + reads[(size_t)i].rlen > size && acc > 0) {
            blocks.push_back({i, treads});
            acc = 0;
        }
        acc += reads[(size_t)i].rlen;
        treads++;
    }
    blocks.push_back({h.ureads, treads});
    h.cutoff = cutoff;
    h.allarr = (h.allarr & ~DB_ALL) | ((all || is_dam) ? DB_ALL : 0);
    h.treads = treads;
    rewind(idx);
    fwrite(&h, sizeof(h), 1, idx);
    fclose(idx);
    // rewrite the stub keeping the file lines
    FILE *st = fopen(p.stub.c_str(), "r");
    if (!st) return dh_fail(DH_EIO, "cannot open " + p.stub);
    std::string keep;
    char line[4096];
    while (fgets(line, sizeof(line), st)) {
        if (strncmp(line, "blocks =", 8) == 0) break;
        keep += line;
    }
    fclose(st);
    st = fopen(p.stub.c_str(), "w");
    if (!st) return dh_fail(DH_EIO, "cannot rewrite " + p.stub);
    fputs(keep.c_str(), st);
    fprintf(st, "blocks = %9d\n", (int)blocks.size() - 1);
    fprintf(st, "size = %11lld cutoff = %9d all = %1d\n", (long long)size_mb, cutoff, (all || is_dam) ? 1 : 0);
    for (auto &b : blocks) fprintf(st, " %9d %9d\n", b.first, b.second);
    fclose(st);
    return DH_OK;
}

struct dh_dazz {
    std::vector<uint8_t> bases;
    std::vector<int64_t> off;
    std::vector<int32_t> origin, fpulse, uid, flags;
    std::vector<std::string> header;  // DAM only: scaffold header of every contig
    int32_t tfirst = 0, cutoff = 0, is_dam = 0, ureads = 0, treads = 0;
    std::string root;
};

// Open a DB or one block of it ("name.3"): the TRIMMED view the aligners work on -- reads shorter
// than the cutoff (and, for .db without -a, non-best reads) are invisible and ids are trimmed ids.
extern "C" int dh_dazz_open(const char *path, dh_dazz **out)
{
    Paths p;
    if (!path || !out || !split_path(std::string(path), p, true)) return dh_fail(DH_EIO, std::string("DAZZ_DB not found: ") + (path ? path : ""));
    FILE *st = fopen(p.stub.c_str(), "r");
    if (!st) return dh_fail(DH_EIO, "cannot open " + p.stub);
    int nfiles = 0, nblocks = 0, cutoff = 0, all = 0;
    long long size = 0;
    std::vector<std::pair<int, int>> blocks;
    char line[4096];
    if (!fgets(line, sizeof(line), st) || sscanf(line, "files = %d", &nfiles) != 1) {
        fclose(st);
        return dh_fail(DH_EIO, "bad stub " + p.stub);
    }
    std::vector<std::pair<int, std::string>> prologs;  // (cumulative read count, prolog) per input file
    for (int i = 0; i < nfiles; i++) {
        if (!fgets(line, sizeof(line), st)) break;
        int cum = 0;
        char fname[2048], prolog[2048];
        if (sscanf(line, " %d %2047s %2047s", &cum, fname, prolog) == 3) prologs.push_back({cum, std::string(prolog)});
    }
    if (fgets(line, sizeof(line), st)) sscanf(line, "blocks = %d", &nblocks);
    if (nblocks > 0 && fgets(line, sizeof(line), st)) {
        sscanf(line, "size = %lld cutoff = %d all = %d", &size, &cutoff, &all);
        for (int i = 0; i <= nblocks; i++) {
            int u = 0, t = 0;
            if (fgets(line, sizeof(line), st) && sscanf(line, "%d %d", &u, &t) == 2) blocks.push_back({u, t});
        }
    }
    fclose(st);
    if (p.block > 0 && (p.block > nblocks || (int)blocks.size() != nblocks + 1))
        return dh_fail(DH_EIO, "block " + std::to_string(p.block) + " does not exist in " + p.stub);
    FILE *idx = fopen(p.hidden("idx").c_str(), "rb");
    FILE *bps = fopen(p.hidden("bps").c_str(), "rb");
    if (!idx || !bps) {
        if (idx) fclose(idx);
        if (bps) fclose(bps);
        return dh_fail(DH_EIO, "cannot open hidden files of " + p.stub);
    }
    DazzHeader h;
    if (fread(&h, sizeof(h), 1, idx) != 1) {
        fclose(idx);
        fclose(bps);
        return dh_fail(DH_EIO, "short .idx");
    }
    std::vector<DazzRead> reads((size_t)h.ureads);
    if (h.ureads && fread(reads.data(), sizeof(DazzRead), reads.size(), idx) != reads.size()) {
        fclose(idx);
        fclose(bps);
        return dh_fail(DH_EIO, "short .idx");
    }
    fclose(idx);
    const bool is_dam = p.ext == ".dam";
    std::string hdrs;
    if (is_dam) {
        FILE *hf = fopen(p.hidden("hdr").c_str(), "rb");
        if (hf) {
            char buf[65536];
            size_t got;
            while ((got = fread(buf, 1, sizeof(buf), hf)) > 0) hdrs.append(buf, got);
            fclose(hf);
        }
    }
    dh_dazz *db = new dh_dazz();
    db->is_dam = is_dam;
    db->ureads = h.ureads;
    db->treads = h.treads;
    db->cutoff = h.cutoff;
    db->root = p.root;
    const int ulo = p.block > 0 ? blocks[(size_t)p.block - 1].first : 0;
    const int uhi = p.block > 0 ? blocks[(size_t)p.block].first : h.ureads;
    db->tfirst = p.block > 0 ? blocks[(size_t)p.block - 1].second : 0;
    db->off.push_back(0);
    std::vector<uint8_t> packed;
    for (int i = ulo; i < uhi; i++) {
        const DazzRead &r = reads[(size_t)i];
        const bool keep = r.rlen >= h.cutoff && (is_dam || (h.allarr & DB_ALL) || (r.flags & DB_BEST));
        if (!keep) continue;
        packed.resize(((size_t)r.rlen + 3) / 4);
        if (fseek(bps, (long)r.boff, SEEK_SET) != 0 ||
            (!packed.empty() && fread(packed.data(), 1, packed.size(), bps) != packed.size())) {
            fclose(bps);
            delete db;
            return dh_fail(DH_EIO, "short .bps");
        }
        const size_t o = db->bases.size();
        db->bases.resize(o + (size_t)r.rlen);
        for (int32_t x = 0; x < r.rlen; x++) db->bases[o + (size_t)x] = (packed[(size_t)x >> 2] >> (6 - 2 * (x & 3))) & 3;
        db->off.push_back((int64_t)db->bases.size());
        db->origin.push_back(r.origin);
        db->fpulse.push_back(r.fpulse);
        db->uid.push_back(i);
        db->flags.push_back(r.flags);
        if (is_dam) {
            size_t e = (size_t)r.coff;
            while (e < hdrs.size() && hdrs[e] != '\n') e++;
            db->header.push_back((size_t)r.coff < hdrs.size() ? hdrs.substr((size_t)r.coff, e - (size_t)r.coff) : std::string());
        } else {  // .db: the prolog of the file the read came from (the H line of DBdump)
            std::string pl;
            for (auto &f : prologs)
                if (i < f.first) {
                    pl = f.second;
                    break;
                }
            db->header.push_back(pl);
        }
    }
    fclose(bps);
    *out = db;
    return DH_OK;
}

extern "C" void dh_dazz_close(dh_dazz *db) { delete db; }
extern "C" int32_t dh_dazz_nreads(const dh_dazz *db) { return db ? (int32_t)db->off.size() - 1 : 0; }
extern "C" int32_t dh_dazz_first_id(const dh_dazz *db) { return db ? db->tfirst : 0; }
extern "C" const uint8_t *dh_dazz_bases(const dh_dazz *db) { return db ? db->bases.data() : nullptr; }
extern "C" const int64_t *dh_dazz_offsets(const dh_dazz *db) { return db ? db->off.data() : nullptr; }
extern "C" const int32_t *dh_dazz_origin(const dh_dazz *db) { return db ? db->origin.data() : nullptr; }
extern "C" const int32_t *dh_dazz_fpulse(const dh_dazz *db) { return db ? db->fpulse.data() : nullptr; }
extern "C" const int32_t *dh_dazz_flags(const dh_dazz *db) { return db ? db->flags.data() : nullptr; }
extern "C" const char *dh_dazz_header(const dh_dazz *db, int32_t i)
{
    return (db && i >= 0 && (size_t)i < db->header.size()) ? db->header[(size_t)i].c_str() : "";
}

// ---------------------------------------------------------------------------------- mask tracks
// source/dentist/dazzler.d:4870-5170 (getMaskFiles / readMask / writeMask).

extern "C" int dh_dazz_write_mask(const char *db_path, const char *name, int32_t nreads, const int64_t *ptr,
                                  const int32_t *iv)
{
    Paths p;
    if (!db_path || !name || !ptr || !split_path(std::string(db_path), p, true)) return dh_fail(DH_EIO, "DAZZ_DB not found");
    if (strchr(name, '/') || strchr(name, '.')) return dh_fail(DH_EINVAL, "mask name must not contain dots or slashes");
    FILE *an = fopen(p.hidden((std::string(name) + ".anno").c_str()).c_str(), "wb");
    FILE *da = fopen(p.hidden((std::string(name) + ".data").c_str()).c_str(), "wb");
    if (!an || !da) {
        if (an) fclose(an);
        if (da) fclose(da);
        return dh_fail(DH_EIO, "cannot create mask files");
    }
    const int32_t head[2] = {nreads, 0};  // size 0 marks the track as a mask (dazzler.d:5143)
    fwrite(head, 4, 2, an);
    for (int32_t i = 0; i <= nreads; i++) {
        const int64_t off = ptr[i] * 2 * (int64_t)sizeof(int32_t);
        fwrite(&off, 8, 1, an);
    }
    if (ptr[nreads] > 0) fwrite(iv, 4, (size_t)(2 * ptr[nreads]), da);
    const bool ok = fclose(an) == 0;
    return (fclose(da) == 0 && ok) ? DH_OK : dh_fail(DH_EIO, "short write of mask files");
}

extern "C" int64_t dh_dazz_read_mask(const dh_dazz *db, const char *db_path, const char *name, int64_t *ptr,
                                     int32_t *iv, int64_t iv_cap)
{
    Paths p;
    if (!db || !db_path || !name || !ptr || !split_path(std::string(db_path), p, true)) return dh_fail(DH_EIO, "DAZZ_DB not found");
    FILE *an = fopen(p.hidden((std::string(name) + ".anno").c_str()).c_str(), "rb");
    FILE *da = fopen(p.hidden((std::string(name) + ".data").c_str()).c_str(), "rb");
    if (!an || !da) {
        if (an) fclose(an);
        if (da) fclose(da);
        return dh_fail(DH_EIO, std::string("mask track not found: ") + name);
    }
    int32_t head[2] = {0, 0};
    if (fread(head, 4, 2, an) != 2 || head[1] != 0 || head[0] < 0) {
        fclose(an);
        fclose(da);
        return dh_fail(DH_EIO, "corrupted mask: expected 0");
    }
    std::vector<int64_t> offs((size_t)head[0] + 1);
    if (fread(offs.data(), 8, offs.size(), an) != offs.size()) {
        fclose(an);
        fclose(da);
        return dh_fail(DH_EIO, "corrupted mask: unexpected number of data pointers");
    }
    fclose(an);
    std::vector<int32_t> data;
    {
        int32_t buf[4096];
        size_t got;
        while ((got = fread(buf, 4, 4096, da)) > 0) data.insert(data.end(), buf, buf + got);
    }
    fclose(da);
    // the track is stored for the whole DB: trimmed ids if it has as many entries as the trimmed DB,
    // untrimmed ids otherwise (dazzler.d:4960-4969); the opened view knows both
    const int32_t nview = (int32_t)db->off.size() - 1;
    int64_t n = 0;
    ptr[0] = 0;
    for (int32_t i = 0; i < nview; i++) {
        const bool untrimmed_ids = head[0] == db->ureads && db->ureads != db->treads;
        const int64_t id = untrimmed_ids ? db->uid[(size_t)i] : (int64_t)db->tfirst + i;
        if (id < head[0]) {
            const int64_t a = offs[(size_t)id] / 4, b = offs[(size_t)id + 1] / 4;
            if (a < 0 || a > b || b > (int64_t)data.size() || (a % 2) || (b % 2))
                return dh_fail(DH_EIO, "corrupted mask: data pointer out of bounds");
            for (int64_t x = a; x < b; x += 2) {
                if (iv && n < iv_cap) {
                    iv[2 * n] = data[(size_t)x];
                    iv[2 * n + 1] = data[(size_t)x + 1];
                }
                n++;
            }
        }
        ptr[i + 1] = n;
    }
    return n;
}


// ---------------------------------------------------------------------------------- byte tracks
// Per-read byte vectors (`qual` / `inqual`: one intrinsic QV per trace tile, written by DASqv /
// computeintrinsicqv and shown by `DBdump -i`, dazzler.d:2877-2897): .anno = int32 nreads, int32 8,
// int64 byte offsets[nreads + 1]; .data = the bytes.  For the whole trimmed DB.
extern "C" int dh_dazz_write_track(const char *db_path, const char *name, int32_t nreads, const int64_t *ptr,
                                   const uint8_t *data)
{
    Paths p;
    if (!db_path || !name || !ptr || !split_path(std::string(db_path), p, true)) return dh_fail(DH_EIO, "DAZZ_DB not found");
    if (strchr(name, '/') || strchr(name, '.')) return dh_fail(DH_EINVAL, "track name must not contain dots or slashes");
    FILE *an = fopen(p.hidden((std::string(name) + ".anno").c_str()).c_str(), "wb");
    FILE *da = fopen(p.hidden((std::string(name) + ".data").c_str()).c_str(), "wb");
    if (!an || !da) {
        if (an) fclose(an);
        if (da) fclose(da);
        return dh_fail(DH_EIO, "cannot create track files");
    }
    const int32_t head[2] = {nreads, 8};
    fwrite(head, 4, 2, an);
    fwrite(ptr, 8, (size_t)nreads + 1, an);
    if (ptr[nreads] > 0) fwrite(data, 1, (size_t)ptr[nreads], da);
    const bool ok = fclose(an) == 0;
    return (fclose(da) == 0 && ok) ? DH_OK : dh_fail(DH_EIO, "short write of track files");
}

// ptr gets nreads(view) + 1 entries; data may be NULL to size; returns the number of bytes or < 0
extern "C" int64_t dh_dazz_read_track(const dh_dazz *db, const char *db_path, const char *name, int64_t *ptr,
                                      uint8_t *data, int64_t cap)
{
    Paths p;
    if (!db || !db_path || !name || !ptr || !split_path(std::string(db_path), p, true)) return dh_fail(DH_EIO, "DAZZ_DB not found");
    FILE *an = fopen(p.hidden((std::string(name) + ".anno").c_str()).c_str(), "rb");
    FILE *da = fopen(p.hidden((std::string(name) + ".data").c_str()).c_str(), "rb");
    if (!an || !da) {
        if (an) fclose(an);
        if (da) fclose(da);
        return dh_fail(DH_EIO, std::string("track not found: ") + name);
    }
    int32_t head[2] = {0, 0};
    std::vector<int64_t> offs;
    bool ok = fread(head, 4, 2, an) == 2 && head[1] == 8 && head[0] >= 0;
    if (ok) {
        offs.resize((size_t)head[0] + 1);
        ok = fread(offs.data(), 8, offs.size(), an) == offs.size();
    }
    fclose(an);
    std::vector<uint8_t> all;
    if (ok) {
        uint8_t buf[65536];
        size_t got;
        while ((got = fread(buf, 1, sizeof(buf), da)) > 0) all.insert(all.end(), buf, buf + got);
    }
    fclose(da);
    if (!ok) return dh_fail(DH_EIO, std::string("corrupted track: ") + name);
    const int32_t nview = (int32_t)db->off.size() - 1;
    int64_t n = 0;
    ptr[0] = 0;
    for (int32_t i = 0; i < nview; i++) {
        const int64_t id = (int64_t)db->tfirst + i;
        if (id < head[0]) {
            const int64_t a = offs[(size_t)id], b = offs[(size_t)id + 1];
            if (a < 0 || a > b || b > (int64_t)all.size()) return dh_fail(DH_EIO, "corrupted track: pointer out of bounds");
            for (int64_t x = a; x < b; x++) {
                if (data && n < cap) data[n] = all[(size_t)x];
                n++;
            }
        }
        ptr[i + 1] = n;
    }
    return n;
}

// DBrm: the stub and every hidden file of the DB
extern "C" int dh_dazz_remove(const char *db_path)
{
    Paths p;
    if (!db_path || !split_path(std::string(db_path), p, true)) return dh_fail(DH_EIO, std::string("DAZZ_DB not found: ") + (db_path ? db_path : ""));
    const std::string prefix = p.hidden("");
    const size_t slash = prefix.find_last_of('/');
    const std::string dir = slash == std::string::npos ? "." : prefix.substr(0, slash);
    const std::string base = slash == std::string::npos ? prefix : prefix.substr(slash + 1);
    if (DIR *d = opendir(dir.c_str())) {
        while (struct dirent *e = readdir(d))
            if (strncmp(e->d_name, base.c_str(), base.size()) == 0) remove((dir + "/" + e->d_name).c_str());
        closedir(d);
    }
    remove(p.stub.c_str());
    return DH_OK;
}
